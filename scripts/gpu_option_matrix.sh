#!/bin/bash
# pytest -m gpu under settings of the library's options (read from the environment at first use): the non-default code paths
#   gpurun --timeout 3000 -- bash scripts/gpu_option_matrix.sh [tag]
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
: > $OUT/${TAG}_option_matrix.txt
for setting in "DIL_FUSE_CHALLENGE=0" "DIL_PACKED_Y=0" "DIL_FUSED_MODE=1" "DIL_FUSED_MODE=2" "DIL_AUX_OVERLAP=0" "DIL_ZEROIZE=1" \
               "DIL_A24=0" "DIL_A24=2" "DIL_SIGN_EARLY=0" "DIL_FUSE_WIRE=0" "DIL_FUSE_KEYGEN=0" "DIL_SIGN_CAP=8192" "DIL_SIGN_SKIP=0" "DIL_SIGN_SKIP=1"; do
  # (tests that assert a specific kernel shape / launch record are deselected where the option changes the shape on purpose)
  DESEL="--deselect tests/test_gpu_mailbox.py::test_reference_unchanged_hw_main_at_its_own_iteration_count --deselect tests/test_gpu_fuzz.py"   # (option-independent, 150 s)
  case "$setting" in DIL_FUSED_MODE=1) DESEL="$DESEL --deselect tests/test_gpu_persistent_parity.py";; esac
  res=$(env $setting timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider $DESEL 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "$setting: $res" | tee -a $OUT/${TAG}_option_matrix.txt
done
