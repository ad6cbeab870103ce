#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
for i in 1 2 3; do
for v in cur vw4; do DIL_LIB_PATH=scripts/bin/libdil256_$v.so python scripts/bench_verify_rot.py 235 2>&1 | grep -v amdgpu; done
done
