#!/bin/bash
# build libdil256.so of another commit for interleaved A/B runs: scripts/build_ref_commit.sh <commit> <name> -> scripts/bin/libdil256_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${1:-HEAD}; NAME=${2:-base}
TMP=$(mktemp -d /tmp/dilbase.XXXX)
git -C $ROOT archive $REF dilithium_amd/csrc include | tar -x -C $TMP
mkdir -p $ROOT/scripts/bin
SRC=""
for f in kernels pipelines hash_kernels codec_kernels wire_kernels gen_kernels capi scheme multi_gpu; do [ -f $TMP/dilithium_amd/csrc/$f.hip ] && SRC="$SRC $TMP/dilithium_amd/csrc/$f.hip"; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -pthread $SRC -o $ROOT/scripts/bin/libdil256_$NAME.so 2>&1 | grep -v "argument unused" || true
rm -rf $TMP
echo $ROOT/scripts/bin/libdil256_$NAME.so
