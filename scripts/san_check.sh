#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side of the repository (no GPU needed):
#   pass 1 (gcc runtime):   oracle/dil_oracle.c, dilithium_amd/csrc/ref_api.cpp      -> the oracle / KAT / drop-in CPU tests
#   pass 2 (clang runtime): the HOST code of libdil256.so (capi.hip, scheme.hip, multi_gpu.hip ... compiled by hipcc with
#                           -fsanitize=address,undefined; device code unchanged)     -> the C-ABI / options / sharding CPU tests
# usage: scripts/san_check.sh [logfile]      (default profiles/r06_sanitizers.txt)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r06_sanitizers.txt}
SAN=$ROOT/oracle/_san
mkdir -p $SAN
cd $ROOT
: > $LOG
say() { echo "$@" | tee -a $LOG; }
SANFLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1      # CPython itself leaks by design
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1

say "== pass 1: gcc $(gcc -dumpversion), -fsanitize=address,undefined: dil_oracle.c + ref_api.cpp"
make -C oracle san >> $LOG 2>&1 || { say "oracle san build FAILED"; exit 1; }
g++ $SANFLAGS -std=c++17 -shared -fPIC -Wall dilithium_amd/csrc/ref_api.cpp -Ldilithium_amd -ldil256 -Wl,-rpath,$ROOT/dilithium_amd \
    -o $SAN/libdil256_ref.so >> $LOG 2>&1 || { say "ref_api san build FAILED"; exit 1; }
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" DIL_ORACLE_PATH=$SAN/liboracle.so DIL_REF_LIB_PATH=$SAN/libdil256_ref.so \
    python -m pytest tests/test_oracle.py tests/test_kat_oracle.py tests/test_ref_dropin.py tests/test_model_and_cabi.py tests/test_bench_cpu_legs.py \
    -q -m "not gpu" -p no:cacheprovider 2>&1 | grep -v "^gold:\|^test:\|^t Error" | tail -15 | tee -a $LOG
P1=${PIPESTATUS[0]}

say "== pass 2: hipcc host code with clang's -fsanitize=address,undefined: libdil256.so (capi / scheme / multi_gpu host paths)"
SRC=""
for f in kernels pipelines hash_kernels coop_kernels codec_kernels wire_kernels capi scheme multi_gpu; do SRC="$SRC dilithium_amd/csrc/$f.hip"; done
hipcc --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wall -pthread $SANFLAGS -fno-sanitize=vptr,function -shared-libsan $SRC -o $SAN/libdil256.so >> $LOG 2>&1 \
    || { say "libdil256 san build FAILED"; exit 1; }
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD="$RT" DIL_LIB_PATH=$SAN/libdil256.so \
    python -m pytest tests/test_model_and_cabi.py tests/test_sharding.py tests/test_multi_gpu.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15 | tee -a $LOG
P2=${PIPESTATUS[0]}
say "== sanitizer passes: pytest exit codes $P1 (gcc runtime) $P2 (clang runtime)"
[ "$P1" = 0 ] && [ "$P2" = 0 ]
