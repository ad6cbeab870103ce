#!/bin/bash
# Round-4 GPU visit (round 3: gpu_r03.sh): parity tests (+ the persistent-loop step log), kernel-coverage trace, smoke, bench, rocprofv3 kernel stats of the
# same bench command, HBM-traffic PMC passes (each its own run; kernel trace only).
#   gpurun --timeout 2400 -- bash scripts/gpu_r04.sh [tag] [what...]     what: tests cover smoke bench prof pmc signpmc   (default: all)
TAG=${1:-r04z}; shift
WHAT=${@:-tests cover smoke bench prof pmc signpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  rm -f $OUT/${TAG}_persistent_steps.txt
  DIL_STEPS_LOG=$OUT/${TAG}_persistent_steps.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/${TAG}_pytest_gpu.log
  tail -5 $OUT/${TAG}_pytest_gpu.log
fi
if has cover; then
  cd /tmp
  timeout 1500 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_cover -o cover -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_dispatch_parity.py \
      $GRAFT_REPO_ROOT/tests/test_gpu_persistent_parity.py $GRAFT_REPO_ROOT/tests/test_gpu_pipelines.py $GRAFT_REPO_ROOT/tests/test_gpu_codecs.py \
      $GRAFT_REPO_ROOT/tests/test_gpu_hash.py $GRAFT_REPO_ROOT/tests/test_gpu_ntt.py $GRAFT_REPO_ROOT/tests/test_gpu_wire.py $GRAFT_REPO_ROOT/tests/test_gpu_msg.py \
      $GRAFT_REPO_ROOT/tests/test_gpu_mailbox.py --deselect tests/test_gpu_mailbox.py::test_reference_unchanged_hw_main_at_its_own_iteration_count \
      -m gpu -x -q -p no:cacheprovider > $OUT/${TAG}_cover.log 2>&1
  echo "cover exit $?" >> $OUT/${TAG}_cover.log
  cd $GRAFT_REPO_ROOT
  { echo "# every *_wpi / shared-key kernel family at batches where a wave re-enters its item loop (tests/test_gpu_persistent_parity.py,";
    echo "# through dil_launch_info): grid x items per step < items";
    sort -u $OUT/${TAG}_persistent_steps.txt 2>/dev/null;
    echo; python scripts/kernel_coverage.py $OUT/${TAG}_cover/cover_results.db; } > $OUT/${TAG}_pytest_kernel_coverage.txt 2>&1
  rm -rf $OUT/${TAG}_cover
  tail -3 $OUT/${TAG}_cover.log; tail -4 $OUT/${TAG}_pytest_kernel_coverage.txt
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
  echo "smoke exit $?" >> $OUT/${TAG}_smoke.log; tail -2 $OUT/${TAG}_smoke.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/${TAG}_bench.log 2>&1
  echo "bench exit $?" >> $OUT/${TAG}_bench.log; tail -2 $OUT/${TAG}_bench.log | cut -c1-1500
fi
if has prof; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --streams 1 > $OUT/${TAG}_prof.log 2>&1
  echo "prof (one stream, headline only) exit $?" >> $OUT/${TAG}_prof.log
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_full -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify-overlap > $OUT/${TAG}_prof_full.log 2>&1
  echo "prof_full exit $?" >> $OUT/${TAG}_prof_full.log
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_stats.py $OUT/${TAG}_prof/${TAG}_results.db $OUT/${TAG}_kernel_stats_one_stream.txt > /dev/null 2>&1
  python scripts/rocpd_stats.py $OUT/${TAG}_prof_full/${TAG}_results.db $OUT/${TAG}_kernel_stats_full.txt > /dev/null 2>&1
  rm -rf $OUT/${TAG}_prof $OUT/${TAG}_prof_full
  head -8 $OUT/${TAG}_kernel_stats_one_stream.txt | cut -c1-160; head -24 $OUT/${TAG}_kernel_stats_full.txt | cut -c1-160
fi
if has pmc; then
  cd /tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${TAG}_pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py bench 2 > $OUT/${TAG}_pmc_$ctr.log 2>&1
    echo "pmc $ctr exit $?"
  done
  cd $GRAFT_REPO_ROOT
  python scripts/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_FETCH_SIZE/p_results.db $OUT/${TAG}_pmc_WRITE_SIZE/p_results.db > $OUT/${TAG}_pmc_summary.txt 2>&1
  rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
  cat $OUT/${TAG}_pmc_summary.txt
fi
if has signpmc; then      # counter evidence for the sign path: sign2_wpi_kernel<5>, matvec_shared_kernel<8,7,5,OUT_W1W0,16>
  cd /tmp
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_spmc$i -o p -- python $GRAFT_REPO_ROOT/scripts/prof_target.py sign 3 > $OUT/${TAG}_spmc$i.log 2>&1
    echo "sign pmc pass $i exit $?"
  done
  for d in $OUT/${TAG}_spmc*/; do python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $d/p_results.db | grep -E "sign2_wpi|matvec_shared|kernel " | grep -v "at::" | cut -c1-190; done > $OUT/${TAG}_sign_pmc.txt 2>&1
  rm -rf $OUT/${TAG}_spmc*/
  cat $OUT/${TAG}_sign_pmc.txt | head -50
  python - "$OUT/${TAG}_sign_pmc.txt" "$OUT/${TAG}_pmc_summary.json" <<'PY'     # VALU instructions per attempt -> the summary bench.py reads
import json, re, sys
txt, js = sys.argv[1], sys.argv[2]
v = {}
for line in open(txt):
    m = re.search(r"(sign2_wpi_kernel|matvec_shared_kernel)<.*SQ_INSTS_VALU\s+([0-9.]+)", line)
    if m:
        v["phase2" if m.group(1).startswith("sign2") else "phase1"] = float(m.group(2)) / 8192
try:
    d = json.load(open(js))
except Exception:
    d = {}
if len(v) == 2:
    d["sign_valu_insts_per_attempt"] = dict(v, source="SQ_INSTS_VALU of matvec_shared_kernel<8,7,5,OUT_W1W0,16> / sign2_wpi_kernel<5> over 8192 attempts")
    json.dump(d, open(js, "w"), indent=1)
    print("sign VALU per attempt:", v)
PY
fi
