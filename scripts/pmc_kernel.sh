#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes) of whatever kernels a command launches:
#   scripts/pmc_kernel.sh <tag> <command...>      -> gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE/, <tag>_pmc.txt
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${TAG}_pmc_$ctr -o p -- "$@" > $OUT/${TAG}_pmc_$ctr.log 2>&1
  echo "pmc $ctr exit $?"
done
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py $OUT/${TAG}_pmc_FETCH_SIZE/p_results.db > $OUT/${TAG}_pmc.txt 2>&1
python scripts/rocpd_stats.py $OUT/${TAG}_pmc_WRITE_SIZE/p_results.db >> $OUT/${TAG}_pmc.txt 2>&1
grep -E "dil::|PMC" $OUT/${TAG}_pmc.txt | cut -c1-200
