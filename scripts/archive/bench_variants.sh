#!/bin/bash
# The bench's SUSTAINED legs under variant libraries (scripts/build_variant.py): interleaved A/B bursts (ab_verify.py) run in the power
# controller's grace period; the numbers that count are the ones bench.py measures after seconds of load, at the clock the chip then holds.
#   gpurun -- bash scripts/bench_variants.sh <tag> name1 name2 ...      (`default` = the in-tree library)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/${TAG}_bench_variants.txt
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = default ]; then unset DIL_LIB_PATH; else export DIL_LIB_PATH=$GRAFT_REPO_ROOT/scripts/bin/libdil256_$n.so; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['secondary']; oc=s['other_configs']
c4=[v for k,v in oc.items() if k.startswith('configs[4]')][0]; c2=[v for k,v in oc.items() if 'distinct A' in k][0]
sc=s['scheme_level3_wire_format']
print('%-10s ntt %.3f G frac %.3f | verify %.1f M frac %.3f (%.0f MHz) shared %.1f M | matvec2 %.1f M | attempt %.1f us p1 %.1f p2 %.1f (%.0f MHz) | sign %.2f M verify_sig %.1f/%.1f M keygen %.1f M' % (
  '$n', d['value']/1e9, d['roofline']['frac'], s['value']/1e6, s['roofline']['frac'], s['roofline'].get('shader_mhz_observed') or 0, s['shared_pk']['value']/1e6,
  c2['matvecs_per_s']/1e6, c4['ms']*1e3, c4['phase1_ms']*1e3, c4['phase2_ms']*1e3, (c4.get('roofline') or {}).get('shader_mhz_observed') or 0,
  sc['sign_shared_key_per_s']/1e6, sc['verify_shared_pk_per_s']/1e6, sc['verify_distinct_pk_per_s']/1e6, sc['keygen_per_s']/1e6))
" | tee -a $OUT/${TAG}_bench_variants.txt
done; done
