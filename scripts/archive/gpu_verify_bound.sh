#!/bin/bash
# {int32 A, 24-bit A} x {full, no transforms} on verify_wpi_kernel<3> and verify_wire_wpi_kernel<3>: kernel durations from rocprofv3 traces
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
for b in full nontt; do
  LIBP=""; [ $b = nontt ] && LIBP="DIL_LIB_PATH=$GRAFT_REPO_ROOT/scripts/bin/libdil256_nontt.so"
  env $LIBP timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/vb_$b -o t -- python $GRAFT_REPO_ROOT/scripts/bench_verify_bound.py 3 8192 > $OUT/vb_$b.log 2>&1
  echo "== build: $b   (level 3, 8192 items, two rotating input sets)"
  python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $OUT/vb_$b/t_results.db | grep -E "verify_wpi_kernel|verify_wire_wpi_kernel|expand_a_fast" | cut -c1-150
  rm -rf $OUT/vb_$b
done
