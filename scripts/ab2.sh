cd $GRAFT_REPO_ROOT
L="default scripts/bin/libdil256_mvx0.so scripts/bin/libdil256_r01.so"
for lv in 3 5; do
python scripts/ab_verify.py --kind matvec --level $lv --rounds 5 $L 2>&1 | grep -v amdgpu.ids
python scripts/ab_verify.py --kind sign1 --level $lv --rounds 5 $L 2>&1 | grep -v amdgpu.ids
done
python scripts/ab_verify.py --kind matvec --level 2 --batch 4096 --rounds 5 $L 2>&1 | grep -v amdgpu.ids
L="default scripts/bin/libdil256_r01.so"
for k in matvec sign1 sign2 verify; do python scripts/ab_verify.py --kind $k --level 3 --rounds 5 --shared $L 2>&1 | grep -v amdgpu.ids; done
python scripts/ab_verify.py --kind sign2 --level 5 --rounds 5 $L 2>&1 | grep -v amdgpu.ids
python scripts/ab_verify.py --kind verify --level 3 --rounds 5 $L 2>&1 | grep -v amdgpu.ids
