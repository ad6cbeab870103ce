#!/bin/bash
# driver-style repeatability of the bench headline: three runs with --steps 20 --warmup 5, one with --steps 1000
TAG=${1:-r03d}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/${TAG}_bench_s20_$i.json 2> $OUT/${TAG}_bench_s20_$i.err
done
python bench.py --steps 1000 --warmup 100 --no-secondary --no-cpu-baseline > $OUT/${TAG}_bench_s1000.json 2> $OUT/${TAG}_bench_s1000.err
python - <<'PY'
import json, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(out + "/r03d_bench_s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "min/max %.4g %.4g" % (d["timing"]["value_min"], d["timing"]["value_max"]),
              "regions", d["timing"]["regions"], "steps/region", d["timing"]["steps_per_timed_region"],
              "frac %.3f ovl %.3f achievable %.0f GB/s (%.3f of peak) frac_of_achievable %.3f" % (r["frac"], r["frac_overlapped"], r["achievable"], r["achievable_frac_of_peak"], r["frac_of_achievable"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-1500:])
PY
