cd $GRAFT_REPO_ROOT
B=scripts/bin; OUT=gpurun_out; mkdir -p $OUT
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
{ ab --kind ntt --rounds 9 $(L cur nttw1 nttw2 nttprio nttw1prio)
  for bpc in 4 6 12 16; do echo "DIL_NTT_BPC=$bpc"; DIL_NTT_BPC=$bpc ab --kind ntt --rounds 5 $(L cur nttw1 nttprio); done
} > $OUT/r04o_ab_ntt.txt 2>&1
cat $OUT/r04o_ab_ntt.txt
