#!/bin/bash
# Round 3, second visit: exchange / twiddle microbenchmark + the persistent-loop parity tests
TAG=${1:-r03b}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 scripts/bin/tune_xchg > $OUT/${TAG}_tune_xchg.txt 2>&1; echo "tune exit $?"
cat $OUT/${TAG}_tune_xchg.txt
timeout 1500 python -m pytest tests/test_gpu_persistent_parity.py -x -q --durations=15 > $OUT/${TAG}_pytest_persistent.log 2>&1
echo "pytest exit $?" >> $OUT/${TAG}_pytest_persistent.log
tail -30 $OUT/${TAG}_pytest_persistent.log
