#!/bin/bash
# scratch: one GPU visit
mkdir -p gpurun_out
for i in 1 2 3; do
for v in cur s2epf; do for n in 8192 65536; do echo "$v n=$n"; DIL_LIB_PATH=scripts/bin/libdil256_$v.so python scripts/bench_scheme.py $n 2>&1 | grep "sign shared"; done; done
done
