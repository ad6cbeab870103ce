cd $GRAFT_REPO_ROOT
B=scripts/bin
for lv in 2 3 5; do
  python scripts/ab_verify.py --kind verify --level $lv --rounds 5 --shared $B/libdil256_base2.so default 2>&1 | grep -v amdgpu.ids
  python scripts/ab_verify.py --kind matvec --level $lv --rounds 5 --shared $B/libdil256_base2.so default 2>&1 | grep -v amdgpu.ids
  python scripts/ab_verify.py --kind sign1 --level $lv --rounds 5 --shared $B/libdil256_base2.so default 2>&1 | grep -v amdgpu.ids
  python scripts/ab_verify.py --kind sign2 --level $lv --rounds 5 --shared $B/libdil256_base2.so default 2>&1 | grep -v amdgpu.ids
done
python scripts/bench_wire.py 2>&1 | grep -v amdgpu.ids | tail -12
DIL_LIB_PATH=$B/libdil256_base2.so python scripts/bench_wire.py 2>&1 | grep -v amdgpu.ids | tail -12
DIL_LIB_PATH=$B/libdil256_vws1.so python scripts/bench_wire.py 2>&1 | grep -v amdgpu.ids | tail -12
