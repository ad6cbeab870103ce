#!/bin/bash
# rocprofv3 PMC passes (each counter group in its own run; never combined with sys/hip traces)
# usage: gpurun -- bash scripts/gpu_pmc.sh <tag> <target>
TAG=${1:-pmc}; TARGET=${2:-verify}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_pmc$i -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py $TARGET 3 > $OUT/${TAG}_pmc$i.log 2>&1
  echo "pass $i ($grp) exit $?"
done
for d in $OUT/${TAG}_pmc*/; do python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $d/p_results.db | grep -v "at::native\|rocclr" ; done
