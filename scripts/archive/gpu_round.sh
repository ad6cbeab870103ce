#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats + HBM-traffic PMC passes.
#   gpurun --timeout 2400 -- bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6 > $OUT/${TAG}_rocminfo.txt 2>&1
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/${TAG}_lscpu.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
echo "smoke exit $?" >> $OUT/${TAG}_smoke.log
timeout 600 python bench.py > $OUT/${TAG}_bench.log 2>&1
echo "bench exit $?" >> $OUT/${TAG}_bench.log
cd /tmp
# headline kernels alone (the averages the bench line's roofline object must agree with) ...
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/${TAG}_prof.log 2>&1
echo "prof exit $?" >> $OUT/${TAG}_prof.log
# ... and the whole bench including the secondary (verify / sign / scheme-level) kernels
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_full -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_prof_full.log 2>&1
echo "prof_full exit $?" >> $OUT/${TAG}_prof.log
# HBM traffic: separate --pmc passes, kernel trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${TAG}_pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py bench 2 > $OUT/${TAG}_pmc_$ctr.log 2>&1
  echo "pmc $ctr exit $?" >> $OUT/${TAG}_prof.log
done
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py $OUT/${TAG}_prof/${TAG}_results.db $OUT/${TAG}_kernel_stats.txt > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/${TAG}_prof_full/${TAG}_results.db $OUT/${TAG}_kernel_stats_full.txt > /dev/null 2>&1
python scripts/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_FETCH_SIZE/p_results.db $OUT/${TAG}_pmc_WRITE_SIZE/p_results.db > $OUT/${TAG}_pmc_summary.txt 2>&1
tail -4 $OUT/${TAG}_pytest_gpu.log; tail -2 $OUT/${TAG}_smoke.log; tail -2 $OUT/${TAG}_bench.log; head -12 $OUT/${TAG}_kernel_stats.txt | cut -c1-150; cat $OUT/${TAG}_pmc_summary.txt
