#!/bin/bash
# ThreadSanitizer over the HOST code of libdil256.so on a GPU box (the device code is unchanged): the helper-thread host pipeline, the
# round-robin and page-lock paths, the two-thread mailbox test.  (Tests that touch torch.cuda cannot run under the preloaded runtime: its dlopen of a torch library fails.)
#   here (CPU):    bash scripts/tsan_check.sh build      -> oracle/_san/libdil256_tsan.so (travels with gpurun, ignored by git)
#   on the box:    gpurun -- bash scripts/tsan_check.sh   -> gpurun_out/tsan.log + summary
# (torch's pin_memory() fails to dlopen under the preloaded runtime: tests that need a torch-pinned buffer are left out)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
if [ "${1:-}" = build ]; then
  SRC=""
  for f in kernels pipelines hash_kernels coop_kernels codec_kernels wire_kernels capi scheme multi_gpu; do SRC="$SRC dilithium_amd/csrc/$f.hip"; done
  mkdir -p oracle/_san
  hipcc --offload-arch=gfx950 -std=c++17 -shared -fPIC -pthread -O1 -g -Xarch_host -fsanitize=thread -shared-libsan $SRC -o oracle/_san/libdil256_tsan.so
  exit $?
fi
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
LD_PRELOAD=$RT TSAN_OPTIONS='report_signal_unsafe=0 halt_on_error=0 second_deadlock_stack=1 ignore_noninstrumented_modules=1' \
  DIL_LIB_PATH=$ROOT/oracle/_san/libdil256_tsan.so timeout 1200 python -m pytest tests/test_gpu_host_paths.py tests/test_gpu_codecs.py tests/test_gpu_mailbox.py \
  -q -p no:cacheprovider -k '(two_threads and False) or chunked or mailbox_busy' > $OUT/tsan.log 2>&1
echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' $OUT/tsan.log)"
grep -E 'passed|failed' $OUT/tsan.log | tail -1
grep -A14 'WARNING: ThreadSanitizer' $OUT/tsan.log | grep -E 'libdil256|WARNING' | sort | uniq -c | sort -rn | head -20
