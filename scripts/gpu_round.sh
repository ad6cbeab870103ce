#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Run via
#   gpurun --timeout 1800 -- bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6 > $OUT/${TAG}_rocminfo.txt 2>&1
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/${TAG}_lscpu.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
echo "smoke exit $?" >> $OUT/${TAG}_smoke.log
timeout 600 python bench.py > $OUT/${TAG}_bench.log 2>&1
echo "bench exit $?" >> $OUT/${TAG}_bench.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
echo "prof exit $?" >> $OUT/${TAG}_prof.log
tail -5 $OUT/${TAG}_pytest_gpu.log; tail -3 $OUT/${TAG}_smoke.log; tail -2 $OUT/${TAG}_bench.log; ls -R $OUT/${TAG}_prof | head -20
