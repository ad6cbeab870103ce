cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/r04h_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r04h_pytest_gpu.log; tail -4 $OUT/r04h_pytest_gpu.log
timeout 900 python bench.py > $OUT/r04h_bench.log 2>&1; echo "bench exit $?"
timeout 300 python scripts/bench_mailbox.py 20000 > $OUT/r04h_mailbox.txt 2>&1; cat $OUT/r04h_mailbox.txt
bash scripts/gpu_scale.sh 100 > $OUT/r04h_scale.log 2>&1; tail -3 $OUT/r04h_scale.log
