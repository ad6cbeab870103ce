cd $GRAFT_REPO_ROOT
for lv in 2 3 5; do
for kind in matvec sign1; do
  python scripts/ab_verify.py --kind $kind --level $lv --rounds 5 --shared scripts/bin/libdil256_base.so default 2>&1 | grep -v amdgpu.ids
done; done
