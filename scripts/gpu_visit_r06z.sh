#!/bin/bash
# Round-6 closing visit: whole-directory runs of the driver's command (2 side by side x 2 rounds, each in its own copy of the tree), then the
# round script: the suite once more on its own (+ persistent-loop step log), smoke, bench, kernel stats, PMC passes, sign PMC, loop PMC, the
# verify leg alone; the host batch sweep and the fuse_sib A/B.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
bash scripts/stress_concurrent.sh r06z_conc 2 ${1:-2}
bash scripts/gpu_r06.sh r06z tests smoke bench prof pmc signpmc looppmc verifyprof
timeout 200 python scripts/bench_host_batch_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/r06z_host_batch_sweep.txt; tail -3 $OUT/r06z_host_batch_sweep.txt | cut -c1-200
timeout 300 python scripts/bench_fuse_sib.py 2>&1 | grep -v amdgpu.ids > $OUT/r06z_fuse_sib.txt; grep "L3" $OUT/r06z_fuse_sib.txt | head -3 | cut -c1-200
