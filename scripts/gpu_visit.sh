#!/bin/bash
mkdir -p gpurun_out
python scripts/fuzz_scheme.py 900 406 2>&1 | grep -v amdgpu > gpurun_out/r04end_fuzz_scheme.txt
python scripts/fuzz_parity.py 600 407 2>&1 | grep -v amdgpu > gpurun_out/r04end_fuzz_parity.txt
tail -n 2 gpurun_out/r04end_fuzz_scheme.txt gpurun_out/r04end_fuzz_parity.txt
python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -2
