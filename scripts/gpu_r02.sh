#!/bin/bash
# Round-2 GPU visit: parity tests (plain), kernel-coverage trace of the pipeline tests, smoke, bench.
#   gpurun --timeout 2400 -- bash scripts/gpu_r02.sh [tag] [what...]     what: tests cover bench prof pmc (default: all but pmc)
TAG=${1:-r02a}; shift
WHAT=${@:-tests cover smoke bench prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/${TAG}_pytest_gpu.log
  tail -5 $OUT/${TAG}_pytest_gpu.log
fi
if has cover; then
  # which __global__ functions do the parity tests actually reach?  kernel trace of the pipeline / scheme tests
  cd /tmp
  timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_cover -o cover -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_dispatch_parity.py \
      $GRAFT_REPO_ROOT/tests/test_gpu_pipelines.py $GRAFT_REPO_ROOT/tests/test_gpu_codecs.py $GRAFT_REPO_ROOT/tests/test_gpu_hash.py \
      $GRAFT_REPO_ROOT/tests/test_gpu_ntt.py $GRAFT_REPO_ROOT/tests/test_gpu_wire.py $GRAFT_REPO_ROOT/tests/test_gpu_msg.py -m gpu -x -q -p no:cacheprovider > $OUT/${TAG}_cover.log 2>&1
  echo "cover exit $?" >> $OUT/${TAG}_cover.log
  cd $GRAFT_REPO_ROOT
  python scripts/kernel_coverage.py $OUT/${TAG}_cover/cover_results.db > $OUT/${TAG}_pytest_kernel_coverage.txt 2>&1
  rm -rf $OUT/${TAG}_cover          # the raw trace database is tens of MiB: only the summary travels back
  tail -3 $OUT/${TAG}_cover.log; tail -12 $OUT/${TAG}_pytest_kernel_coverage.txt
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
  echo "smoke exit $?" >> $OUT/${TAG}_smoke.log; tail -2 $OUT/${TAG}_smoke.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/${TAG}_bench.log 2>&1
  echo "bench exit $?" >> $OUT/${TAG}_bench.log; tail -2 $OUT/${TAG}_bench.log | cut -c1-3000
fi
if has prof; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_full -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify-overlap > $OUT/${TAG}_prof_full.log 2>&1
  echo "prof_full exit $?" >> $OUT/${TAG}_prof_full.log
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_stats.py $OUT/${TAG}_prof_full/${TAG}_results.db $OUT/${TAG}_kernel_stats_full.txt > /dev/null 2>&1
  rm -rf $OUT/${TAG}_prof_full
  head -30 $OUT/${TAG}_kernel_stats_full.txt | cut -c1-160
fi
if has pmc; then
  cd /tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${TAG}_pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py bench 2 > $OUT/${TAG}_pmc_$ctr.log 2>&1
    echo "pmc $ctr exit $?"
  done
  cd $GRAFT_REPO_ROOT
  python scripts/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_FETCH_SIZE/p_results.db $OUT/${TAG}_pmc_WRITE_SIZE/p_results.db > $OUT/${TAG}_pmc_summary.txt 2>&1
  rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
  cat $OUT/${TAG}_pmc_summary.txt
fi
