cd $GRAFT_REPO_ROOT
B=scripts/bin; OUT=gpurun_out; mkdir -p $OUT
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
{ for lv in 3 5 2; do ab --kind wire --level $lv --rounds 7 $(L base cur ww0 ww0t0); done
  ab --kind matvec --level 2 --batch 4096 --rounds 9 $(L base cur mvw0)
  for lv in 3 5; do ab --kind matvec --level $lv --rounds 7 $(L base cur mvw0); ab --kind sign1 --level $lv --rounds 7 $(L base cur mvw0); done
} > $OUT/r04l_ab.txt 2>&1
cat $OUT/r04l_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_mailbox.py::test_reference_unchanged_hw_main_at_its_own_iteration_count > $OUT/r04l_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r04l_pytest_gpu.log; tail -3 $OUT/r04l_pytest_gpu.log
