# 2 x 2 x 2 ablation of verify_wpi_kernel<3> (transforms / matrix loads / time-domain loads), interleaved: build the eight libraries with
# scripts/build_variant.py abl_no_<combo> -DDIL_ABL_NONTT -DDIL_ABL_NOALOAD -DDIL_ABL_NOSMALL (any subset)
cd $GRAFT_REPO_ROOT
B=scripts/bin
L=""; for n in full nontt noa nosm nontt_noa nontt_nosm noa_nosm nontt_noa_nosm; do L="$L $B/libdil256_abl_no_$n.so"; done
python scripts/ab_verify.py --kind verify --level 3 --rounds 5 $L 2>&1 | grep -v amdgpu.ids
python scripts/ab_verify.py --kind verify --level 5 --rounds 3 $L 2>&1 | grep -v amdgpu.ids
