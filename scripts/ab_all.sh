#!/bin/bash
# interleaved A/B of library builds over every fused-pipeline shape: scripts/ab_all.sh lib1 lib2 ...
cd $GRAFT_REPO_ROOT
for lv in 3 5; do
for kind in verify matvec sign1 sign2; do
  python scripts/ab_verify.py --kind $kind --level $lv --rounds 5 "$@" 2>&1 | grep -v amdgpu.ids
  python scripts/ab_verify.py --kind $kind --level $lv --rounds 5 --shared "$@" 2>&1 | grep -v amdgpu.ids
done; done
python scripts/ab_verify.py --kind matvec --level 2 --batch 4096 --rounds 5 "$@" 2>&1 | grep -v amdgpu.ids
