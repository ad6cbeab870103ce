import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
for level, L in ((3, 5), (5, 7), (2, 4)):
    for n in (2048, 8192, 16384, 32768):
        rp = u8(n, 64); kap = torch.zeros(n, dtype=torch.int32, device="cuda")
        r = []
        for mx in (16384, 1 << 30):
            api.set_option("two_lane_max_sponges", mx)
            r.append(min(timeit(lambda: api.expand_mask(rp, kap, level), 10) for _ in range(3)) * 1e3)
        print(f"L{level} expand_mask n={n:6d} ({n*L:7d} sponges): lane-per-sponge {r[0]:7.1f} us   two-lane {r[1]:7.1f} us")
seed, mu = u8(8192, 32), u8(8192, 64)
for level in (2, 3, 5):
    pk, sk = api.keygen(seed, level)
    for mx in (16384, 131072, 1 << 30):
        api.set_option("two_lane_max_sponges", mx)
        t = min(timeit(lambda: api.sign(sk[:1], mu, level, shared_sk=True), 5) for _ in range(3))
        td = min(timeit(lambda: api.sign(sk, mu, level), 3) for _ in range(2))
        print(f"L{level} sign 8192 two_lane_max={mx}: shared {t*1e3:7.1f} us {8192/t/1e3:6.2f} M/s | distinct {td*1e3:7.1f} us {8192/td/1e3:6.2f} M/s")
