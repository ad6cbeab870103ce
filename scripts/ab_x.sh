cd $GRAFT_REPO_ROOT
B=scripts/bin
for lv in 3 5; do
for kind in matvec sign1 verify; do
  python scripts/ab_verify.py --kind $kind --level $lv --rounds 5 --shared $B/libdil256_base2.so $B/libdil256_mvs1.so $B/libdil256_mvs2.so 2>&1 | grep -v amdgpu.ids
done; done
python scripts/ab_verify.py --kind ntt --rounds 7 $B/libdil256_base2.so $B/libdil256_nttx2.so 2>&1 | grep -v amdgpu.ids
