#!/bin/bash
# kernel timelines (rocprofv3 --kernel-trace -> scripts/rocpd_timeline.py) of whole calls: usage  sign_timeline.sh [tag]
#   the level-3 signing loop at 8192 messages under one key, the batch-of-one keygen / sign / verify calls, verification (one key: verify,
#   a key per signature: verifyd) and key generation at 8192
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r05s}
cat > /tmp/t.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from dilithium_amd import api
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)
what, n = sys.argv[1], int(sys.argv[2])
pk, sk = api.keygen(u8(1, 32), 3)
mu = u8(n, 64)
sig, _ = api.sign(sk, mu, 3, shared_sk=True)
seed = u8(n, 32)
if what == "verifyd":
    pkd, skd = api.keygen(seed, 3)
    sigd, _ = api.sign(skd, mu, 3)
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    if what == "sign": api.sign(sk, mu, 3, shared_sk=True)
    elif what == "verify": api.verify_sig(pk, sig, mu, 3, shared_pk=True)
    elif what == "verifyd": api.verify_sig(pkd, sigd, mu, 3)
    else: api.keygen(seed, 3)
    torch.cuda.synchronize(); print(what, n, "call ms", (time.perf_counter() - t) * 1e3)
    time.sleep(0.01)
PY
LIST=("sign 8192" "sign 1" "verify 1" "keygen 1" "verify 8192" "verifyd 8192" "keygen 8192")
if [ -n "${JOBS:-}" ]; then IFS=';' read -ra LIST <<< "$JOBS"; fi        # JOBS="verify 8192;keygen 8192"
for job in "${LIST[@]}"; do
  set -- $job
  rm -rf $OUT/${TAG}_tl
  rocprofv3 --kernel-trace -d $OUT/${TAG}_tl -o p -- python /tmp/t.py $1 $2 > $OUT/${TAG}_tl.log 2>&1
  echo "== $1, batch $2 (level 3; last of four calls; kernel times under the profiler)"
  grep "call ms" $OUT/${TAG}_tl.log | tail -1
  python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $(find $OUT/${TAG}_tl -name "*.db" | head -1) 3000
done > $OUT/${TAG}_call_timelines.txt 2>&1
rm -rf $OUT/${TAG}_tl
cat $OUT/${TAG}_call_timelines.txt
