#!/bin/bash
# Round 3, first visit: the persistent-loop parity tests + counter evidence for the sign path (sign2_wpi_kernel<5>,
# matvec_shared_kernel<8,7,5,OUT_W1W0,12>).   gpurun --timeout 2400 -- bash scripts/gpu_r03a.sh [tag]
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc > $OUT/${TAG}_host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/${TAG}_host.txt 2>&1; free -g >> $OUT/${TAG}_host.txt
timeout 1500 python -m pytest tests/test_gpu_persistent_parity.py -x -q --durations=30 > $OUT/${TAG}_pytest_persistent.log 2>&1
echo "pytest exit $?" >> $OUT/${TAG}_pytest_persistent.log
tail -45 $OUT/${TAG}_pytest_persistent.log
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_pmc$i -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py sign 3 > $OUT/${TAG}_pmc$i.log 2>&1
  echo "pass $i ($grp) exit $?"
done
for d in $OUT/${TAG}_pmc*/; do python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $d/p_results.db | grep -E "sign2_wpi|matvec_shared|kernel " | grep -v "at::" | cut -c1-190; done > $OUT/${TAG}_sign_pmc.txt 2>&1
cat $OUT/${TAG}_sign_pmc.txt
