#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_codecs.py tests/test_gpu_wire.py tests/test_gpu_msg.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python scripts/ab_sign_skip.py > gpurun_out/r04t_ab_sign_runahead.txt 2>&1
cut -c1-250 gpurun_out/r04t_ab_sign_runahead.txt | tail -20
