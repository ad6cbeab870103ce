cd $GRAFT_REPO_ROOT
B=scripts/bin; OUT=gpurun_out; mkdir -p $OUT
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r04d_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r04d_pytest_gpu.log
tail -5 $OUT/r04d_pytest_gpu.log
{ for lv in 5 3 2; do
  ab --kind sign2 --level $lv --rounds 9 --shared $(L base cur nomad)
  ab --kind sign1 --level $lv --rounds 7 --shared $(L base cur nomad)
done
ab --kind sign2 --level 5 --rounds 7 $(L base cur nomad)
ab --kind sign2 --level 3 --rounds 7 $(L base cur nomad)
ab --kind sign2 --level 5 --rounds 5 --shared --generic $(L base cur)
ab --kind verify --level 3 --rounds 7 $(L base cur nomad)
ab --kind verify --level 3 --rounds 7 --shared $(L base cur nomad)
ab --kind matvec --level 2 --batch 4096 --rounds 7 $(L base cur nomad)
} > $OUT/r04d_ab.txt 2>&1
cat $OUT/r04d_ab.txt
