cd $GRAFT_REPO_ROOT
B=scripts/bin; OUT=gpurun_out; mkdir -p $OUT
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
{ for lv in 5 3 2; do
    ab --kind pair --level $lv --rounds 7 --shared $(L pfnone pf0 cur)
    ab --kind pair --level $lv --rounds 5 --shared --reps 300 $(L pfnone pf0 cur)
    ab --kind sign1 --level $lv --rounds 5 --shared $(L pfnone pf0 cur)
  done
  ab --kind pair --level 5 --rounds 5 $(L pfnone cur)
  ab --kind pair --level 3 --rounds 5 $(L pfnone cur)
} > $OUT/r04r_ab_pair.txt 2>&1
cat $OUT/r04r_ab_pair.txt
python scripts/ab_sign.py $(L pfnone cur) --levels 3 2 --batches 8192 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r04r_ab_pair.txt
