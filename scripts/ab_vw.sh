cd $GRAFT_REPO_ROOT
B=scripts/bin
for lv in 2 3 5; do
  python scripts/ab_verify.py --kind verify --level $lv --rounds 5 $B/libdil256_w3r1.so $B/libdil256_w3r2.so $B/libdil256_w3r3.so $B/libdil256_w4r1.so $B/libdil256_w4r2.so 2>&1 | grep -v amdgpu.ids
done
