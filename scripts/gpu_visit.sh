#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
python -m pytest tests/test_gpu_codecs.py tests/test_gpu_persistent_parity.py tests/test_gpu_wire.py tests/test_gpu_msg.py tests/test_gpu_dispatch_parity.py -x -q -m gpu 2>&1 | tail -5
python scripts/ab_sign_skip.py > gpurun_out/r04s_ab_sign_skip.txt 2>&1
