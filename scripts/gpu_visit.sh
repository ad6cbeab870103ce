#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
DIL_VERIFY_COOP=1 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_persistent_parity.py tests/test_gpu_dispatch_parity.py -x -q -m gpu -k "verify" 2>&1 | tail -4
for i in 1 2; do
DIL_LIB_PATH=scripts/bin/libdil256_cur.so python scripts/bench_verify_rot.py 235 2>&1 | grep -v amdgpu
DIL_VERIFY_COOP=1 DIL_LIB_PATH=scripts/bin/libdil256_cur.so python scripts/bench_verify_rot.py 235 2>&1 | grep -v amdgpu | sed 's/^/COOP5 /'
DIL_VERIFY_COOP=1 DIL_LIB_PATH=scripts/bin/libdil256_vc4.so python scripts/bench_verify_rot.py 235 2>&1 | grep -v amdgpu | sed 's/^/COOP4 /'
done
