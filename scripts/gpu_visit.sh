cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
bash scripts/gpu_trace_sign.sh r04f 3 8192 1 > $OUT/r04f_trace.log 2>&1
bash scripts/gpu_trace_sign.sh r04f5 5 8192 1 >> $OUT/r04f_trace.log 2>&1
tail -3 $OUT/r04f_trace.log
timeout 900 python bench.py > $OUT/r04f_bench.log 2>&1; echo "bench exit $?"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r04f_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r04f_pytest_gpu.log; tail -3 $OUT/r04f_pytest_gpu.log
