#!/bin/bash
# Round-4 interleaved A/B experiments (profiles/r04*_ab_*.txt).  Build the variants first (scripts/build_variant.py <name> -D...,
# scripts/build_ref_commit.sh <commit> base), then   gpurun -- bash scripts/ab_r04.sh <experiment> [tag]
#   sign    shared-key sign phase 1 / phase 2 / mat-vec / verify: base (round 3) vs the tree, with and without the dual transforms,
#           full twiddle tables, two workgroups per CU                                base cur nodual twc0 w2n10
#   mvsabl  the shared-key phase-1 kernel taken apart                                 cur mvs_nontt mvs_noinv mvs_nofwd mvs_noaread mvs_noemit
cd $GRAFT_REPO_ROOT
B=scripts/bin
OUT=gpurun_out; mkdir -p $OUT
TAG=${2:-r04a}
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
L() { for n in "$@"; do echo -n "$B/libdil256_$n.so "; done; }
case "$1" in
sign)
  { for lv in 5 3 2; do
      ab --kind sign1 --level $lv --rounds 5 --shared $(L base cur nodual twc0 w2n10)
      ab --kind sign2 --level $lv --rounds 5 --shared $(L base cur nodual twc0)
    done
    for lv in 5 3 2; do
      ab --kind matvec --level $lv --rounds 5 --shared $(L base cur nodual)
      ab --kind verify --level $lv --rounds 5 --shared $(L base cur)
      ab --kind sign2 --level $lv --rounds 5 $(L base cur nodual)
    done; } | tee $OUT/${TAG}_ab_sign.txt ;;
mvsabl)
  { for lv in 5 3; do
      ab --kind sign1 --level $lv --rounds 5 --shared $(L cur mvs_nontt mvs_noinv mvs_nofwd mvs_noaread mvs_noemit)
    done; } | tee $OUT/${TAG}_ab_mvsabl.txt ;;
*) echo "usage: ab_r04.sh sign|mvsabl [tag]"; exit 1 ;;
esac
