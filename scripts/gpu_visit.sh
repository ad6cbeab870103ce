cd $GRAFT_REPO_ROOT
B=scripts/bin; OUT=gpurun_out; mkdir -p $OUT
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
{ for lv in 3 5 2; do ab --kind wire --level $lv --rounds 7 --shared $(L base cur vs0); ab --kind verify --level $lv --rounds 7 --shared $(L base cur vs0); done
} > $OUT/r04m_ab.txt 2>&1
cat $OUT/r04m_ab.txt
timeout 900 python -m pytest tests/test_gpu_wire.py tests/test_gpu_pipelines.py tests/test_gpu_dispatch_parity.py tests/test_gpu_persistent_parity.py tests/test_gpu_codecs.py -m gpu -x -q 2>&1 | tail -2
