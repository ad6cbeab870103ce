cd $GRAFT_REPO_ROOT
OUT=gpurun_out
{ echo "== round-5 library (commit 3709c31, tree under _old/)"; (cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/scripts/check_retained.py $GRAFT_REPO_ROOT/_old 2>&1 | grep -v amdgpu.ids)
  echo "== this tree"; (cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/scripts/check_retained.py $GRAFT_REPO_ROOT 2>&1 | grep -v amdgpu.ids); } > $OUT/r06g_retained_registrations.txt 2>&1
cat $OUT/r06g_retained_registrations.txt
timeout 900 python -m pytest tests/test_gpu_host_paths.py tests/test_gpu_ntt.py tests/test_gpu_wire.py tests/test_gpu_mailbox.py tests/test_ref_dropin.py tests/test_gpu_resources.py -m gpu -q -s -x \
   --deselect tests/test_gpu_mailbox.py::test_reference_unchanged_hw_main_at_its_own_iteration_count > $OUT/r06g_tests.log 2>&1
echo "tests exit $?"; tail -5 $OUT/r06g_tests.log
timeout 300 python scripts/bench_fuse_sib.py > $OUT/r06g_fuse_sib.txt 2>&1; cat $OUT/r06g_fuse_sib.txt | grep -v amdgpu.ids
timeout 200 python scripts/bench_host_batch_sweep.py > $OUT/r06g_host_batch_sweep.txt 2>&1; cat $OUT/r06g_host_batch_sweep.txt | grep -v amdgpu.ids
