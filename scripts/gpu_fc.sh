#!/bin/bash
# round 3: the one-launch challenge (hash + SampleInBall) -- parity tests, then dil_sign_dev with and without it, interleaved
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hash.py tests/test_gpu_codecs.py tests/test_gpu_dispatch_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > $OUT/r03_fc_tests.txt
cat $OUT/r03_fc_tests.txt
timeout 900 python scripts/bench_sign_opts.py fuse_challenge 0 1 2 3 5 2>&1 | grep -v amdgpu.ids | tee $OUT/r03_fc_sign.txt
