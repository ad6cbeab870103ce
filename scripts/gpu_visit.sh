cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
python scripts/ab_sign.py scripts/bin/libdil256_prev.so scripts/bin/libdil256_cur.so --levels 3 5 2 2>&1 | grep -v amdgpu.ids | tee $OUT/r04p_ab_sign_loop.txt
