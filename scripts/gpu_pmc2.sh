#!/bin/bash
# second PMC battery: where do the waves of a fused kernel wait?   usage: gpu_pmc2.sh <tag> <target>
TAG=${1:-pmc}; TARGET=${2:-verify_shared}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_q$i -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py $TARGET 3 > $OUT/${TAG}_q$i.log 2>&1
done
for d in $OUT/${TAG}_q*/; do python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $d/p_results.db | grep -E "dil::" | cut -c1-30,61-; done
