#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
python scripts/fuzz_parity.py 480 404 > gpurun_out/r04zz_fuzz_parity.txt 2>&1
python scripts/fuzz_scheme.py 480 405 > gpurun_out/r04zz_fuzz_scheme.txt 2>&1
tail -2 gpurun_out/r04zz_fuzz_parity.txt gpurun_out/r04zz_fuzz_scheme.txt
