cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mailbox.py -m gpu -x -q -s > $OUT/r04g_mailbox_tests.log 2>&1; echo "mailbox tests exit $?"; tail -15 $OUT/r04g_mailbox_tests.log
timeout 300 python scripts/bench_mailbox.py 20000 > $OUT/r04g_mailbox.txt 2>&1; cat $OUT/r04g_mailbox.txt
( time timeout 300 oracle/_ref/ref_test_ntt_ntt2x2_dropin ) >> $OUT/r04g_mailbox.txt 2>&1; tail -6 $OUT/r04g_mailbox.txt
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_mailbox.py > $OUT/r04g_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r04g_pytest_gpu.log; tail -3 $OUT/r04g_pytest_gpu.log
