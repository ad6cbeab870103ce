cd $GRAFT_REPO_ROOT
B=scripts/bin
for lv in 2 3 5; do
for kind in verify matvec sign1; do
  python scripts/ab_verify.py --kind $kind --level $lv --rounds 5 $B/libdil256_x1.so $B/libdil256_x2.so 2>&1 | grep -v amdgpu.ids
done; done
