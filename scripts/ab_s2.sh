cd $GRAFT_REPO_ROOT
B=scripts/bin
for lv in 2 3 5; do
  python scripts/ab_verify.py --kind sign2 --level $lv --rounds 5 --shared $B/libdil256_s2x0.so $B/libdil256_s2x1.so $B/libdil256_s2x2.so 2>&1 | grep -v amdgpu.ids
  python scripts/ab_verify.py --kind sign2 --level $lv --rounds 5 $B/libdil256_s2x0.so $B/libdil256_s2x1.so $B/libdil256_s2x2.so 2>&1 | grep -v amdgpu.ids
done
