#!/bin/bash
# The 1 / 2 / 4 / 8-GPU curve of bench.py on whatever node this runs on (north_star): one process per GPU over RCCL, exactly as the
# driver launches it.  Writes gpurun_out/scale_curve.json (copy to profiles/ to commit): per N the whole-job value, the per-GPU value and
# its flatness against N = 1, the sharded configs[4] leg's per-slice kernel time and gather time, and what RCCL reported about the job
# (rccl_nranks, distinct devices).  On a one-GPU box only the N = 1 row exists -- and says so.
#   gpurun --timeout 3000 -- bash scripts/gpu_scale.sh [steps]
STEPS=${1:-200}
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
: > $OUT/scale_lines.jsonl
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  if [ "$N" = 1 ]; then
    timeout 1200 python bench.py --gpus 1 --steps $STEPS --warmup 20 --no-cpu-baseline 2> $OUT/scale_n$N.err | grep '^{' >> $OUT/scale_lines.jsonl
  else
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
        bench.py --gpus $N --steps $STEPS --warmup 20 --no-cpu-baseline 2> $OUT/scale_n$N.err | grep '^{' >> $OUT/scale_lines.jsonl
  fi
  echo "N=$N exit ${PIPESTATUS[0]}"
done
python - "$OUT/scale_lines.jsonl" "$OUT/scale_curve.json" "$NGPU" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r for r in rows if r["n_gpus"] == 1), None)
curve, bad = [], []
for r in rows:
    n = r["n_gpus"]
    c4 = r.get("secondary", {}).get("configs4_sharded", {})
    dist = r.get("distributed") or {}
    # at N = 1 there is no process group: one rank, one device by construction
    nranks = dist.get("rccl_nranks", 1 if n == 1 else None)
    devices = dist.get("distinct_devices", 1 if n == 1 else None)
    per_gpu = r["value"] / n
    curve.append({"n_gpus": n, "value": r["value"], "unit": r["unit"], "per_gpu_value": per_gpu,
                  "per_gpu_vs_n1": per_gpu / base["value"] if base else None,
                  "kernel_ms": r["ms_per_step"],                     # one step (fwd + inv launch) on every GPU's own batch: flat in N if the path shards cleanly
                  "configs4_kernel_ms": c4.get("attempt_ms_per_rank_slice"), "configs4_attempts_per_s": c4.get("attempts_per_s"),
                  "final_gather_ms": c4.get("final_gather_ms"), "final_gather_bytes": c4.get("final_gather_bytes"),
                  "final_gather_GBps_per_gpu_received": c4.get("final_gather_GBps_per_gpu_received"),
                  "xgmi_bound_GBps_per_gpu": c4.get("xgmi_bound_GBps_per_gpu"),
                  "rccl_nranks": nranks, "distinct_devices": devices, "backend": dist.get("backend"), "rccl_version": dist.get("rccl_version"),
                  "librccl": dist.get("librccl")})
    if nranks != n or devices != n or (n > 1 and dist.get("backend") != "nccl"):
        bad.append(f"N={n}: rccl_nranks={nranks} distinct_devices={devices} backend={dist.get('backend')}")
json.dump({"visible_gpus": int(sys.argv[3]), "note": "efficiency is the driver's to compute; per_gpu_vs_n1 is the flatness of the per-GPU rate",
           "valid": not bad, "refused": bad, "curve": curve}, open(sys.argv[2], "w"), indent=1)
print(json.dumps([{k: c[k] for k in ("n_gpus", "value", "per_gpu_vs_n1", "kernel_ms", "final_gather_ms", "rccl_nranks", "distinct_devices")} for c in curve]))
if bad:
    print("REFUSED: a row whose RCCL job is not N ranks on N distinct devices is not a scaling curve:", "; ".join(bad))
    sys.exit(1)
PY
