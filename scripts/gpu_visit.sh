cd $GRAFT_REPO_ROOT
B=scripts/bin; OUT=gpurun_out; mkdir -p $OUT
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
{ for lv in 5 3; do
  ab --kind sign1 --level $lv --rounds 9 --shared $(L base cur d1 d2pf d1pf x000 x100 x011 bf64)
  ab --kind sign2 --level $lv --rounds 9 --shared $(L base cur x000 x100 x011 bf64)
done
ab --kind verify --level 3 --rounds 7 $(L base cur bf64)
ab --kind ntt --rounds 7 $(L base cur bf64)
} > $OUT/r04c_ab.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/r04c_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r04c_pytest_gpu.log
cat $OUT/r04c_ab.txt; tail -5 $OUT/r04c_pytest_gpu.log
