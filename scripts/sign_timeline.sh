cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cat > /tmp/t.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from dilithium_amd import api
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)
pk, sk = api.keygen(u8(1, 32), 3)
mu = u8(8192, 64)
for i in range(4):
    torch.cuda.synchronize(); t=time.perf_counter()
    api.sign(sk, mu, 3, shared_sk=True)
    torch.cuda.synchronize(); print("sign call ms", (time.perf_counter()-t)*1e3)
    time.sleep(0.01)
PY
rocprofv3 --kernel-trace -d $OUT/r05s_tl -o p -- python /tmp/t.py > $OUT/r05s_tl.log 2>&1
tail -5 $OUT/r05s_tl.log
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $(find $OUT/r05s_tl -name "*.db" | head -1) 3000 > $OUT/r05s_sign_timeline.txt 2>&1
tail -80 $OUT/r05s_sign_timeline.txt
