#!/bin/bash
# rocprofv3 PMC passes for a memory-side picture of one target (each group its own run; never with sys/hip traces)
# usage: gpurun -- bash scripts/gpu_pmc3.sh <tag> <target>
TAG=${1:-pmc3}; TARGET=${2:-verify_rot}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_pmc$i -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py $TARGET 3 > $OUT/${TAG}_pmc$i.log 2>&1
  echo "pass $i ($grp) exit $?"
done
for d in $OUT/${TAG}_pmc*/; do python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $d/p_results.db | grep -E "verify_wpi|ntt_" | grep -v "at::" | cut -c1-150; done
