#!/bin/bash
# Round-3 interleaved A/B experiments (profiles/r03*_ab_*.txt).  Each builds nothing: build the variants first with
# scripts/build_variant.py <name> -D...  (or scripts/build_ref_commit.sh <commit> base), then
#   gpurun -- bash scripts/ab_r03.sh <experiment>
# experiments (the variant libraries they expect under scripts/bin/):
#   s2x     sign phase 2 exchange policy            s2x0 s2x1 s2x2      (-DDIL_S2_XPOL=0|1|2 at the commit that had the hook)
#   shared  shared-key kernels vs a base commit     base                (matvec / sign1 / verify / sign2, --shared, levels 2 3 5)
#   vw      verify_wpi waves x rows in flight       w3r1 w3r2 w3r3 w4r1 w4r2
#   mr      matvec_wpi rows in flight               mr1 mr2 mr4         (-DDIL_MV_ROWS(K)=n)
cd $GRAFT_REPO_ROOT
B=scripts/bin
ab() { python scripts/ab_verify.py "$@" 2>&1 | grep -v amdgpu.ids; }
case "$1" in
s2x)
  for lv in 2 3 5; do
    ab --kind sign2 --level $lv --rounds 5 --shared $B/libdil256_s2x0.so $B/libdil256_s2x1.so $B/libdil256_s2x2.so
    ab --kind sign2 --level $lv --rounds 5 $B/libdil256_s2x0.so $B/libdil256_s2x1.so $B/libdil256_s2x2.so
  done ;;
shared)
  for lv in 2 3 5; do for kind in verify matvec sign1 sign2; do
    ab --kind $kind --level $lv --rounds 5 --shared $B/libdil256_base.so default
  done; done ;;
vw)
  for lv in 2 3 5; do
    ab --kind verify --level $lv --rounds 5 $B/libdil256_w3r1.so $B/libdil256_w3r2.so $B/libdil256_w3r3.so $B/libdil256_w4r1.so $B/libdil256_w4r2.so
  done ;;
mr)
  ab --kind matvec --level 2 --batch 4096 --rounds 7 $B/libdil256_mr1.so $B/libdil256_mr2.so $B/libdil256_mr4.so
  for lv in 2 3 5; do for kind in matvec sign1; do
    ab --kind $kind --level $lv --rounds 5 $B/libdil256_mr1.so $B/libdil256_mr2.so $B/libdil256_mr4.so
  done; done ;;
*) echo "usage: ab_r03.sh s2x|shared|vw|mr"; exit 1 ;;
esac
