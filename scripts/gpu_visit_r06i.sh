#!/bin/bash
# Round-6 visit i: the signing loop's count posted into mapped host words (option sign_wake) -- parity, A/B against copy + event and against the
# host-free upper bound (scripts/sign_replay), timelines before / after.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_codecs.py tests/test_gpu_persistent_parity.py tests/test_gpu_msg.py "tests/test_gpu_options.py" -m gpu -q -x -k "sign or kat or random or hardest or msg" \
   > $OUT/r06i_tests.log 2>&1
echo "tests exit $?"; tail -3 $OUT/r06i_tests.log
DIL_LIB_PATH=$GRAFT_REPO_ROOT/scripts/bin/libdil256_replay.so timeout 400 python scripts/sign_replay/bench_sign_replay.py 8192 2>&1 | grep -v amdgpu.ids > $OUT/r06i_sign_replay.txt
cat $OUT/r06i_sign_replay.txt
DIL_LIB_PATH=$GRAFT_REPO_ROOT/scripts/bin/libdil256_replay.so timeout 300 python scripts/sign_replay/bench_sign_replay.py 65536 2>&1 | grep -v amdgpu.ids >> $OUT/r06i_sign_replay.txt
tail -6 $OUT/r06i_sign_replay.txt
{ for w in 0 1; do echo "#### DIL_SIGN_WAKE=$w"; DIL_SIGN_WAKE=$w JOBS="sign 8192" bash scripts/sign_timeline.sh r06i_w$w; done; } > $OUT/r06i_sign_timeline.txt 2>&1
grep -E "####|call ms|launches" $OUT/r06i_sign_timeline.txt
