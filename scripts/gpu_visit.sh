#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
python -m pytest tests/test_gpu_codecs.py tests/test_gpu_wire.py tests/test_gpu_msg.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for n in 8192 65536; do python scripts/bench_scheme.py $n 2>&1 | grep "sign shared"; done
