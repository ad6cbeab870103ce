cd $GRAFT_REPO_ROOT
B=scripts/bin
python scripts/ab_verify.py --kind matvec --level 2 --batch 4096 --rounds 7 $B/libdil256_mr1.so $B/libdil256_mr2.so $B/libdil256_mr4.so 2>&1 | grep -v amdgpu.ids
for lv in 2 3 5; do
for kind in matvec sign1; do
  python scripts/ab_verify.py --kind $kind --level $lv --rounds 5 $B/libdil256_mr1.so $B/libdil256_mr2.so $B/libdil256_mr4.so 2>&1 | grep -v amdgpu.ids
done; done
