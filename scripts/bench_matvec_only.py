import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit, KL, Q
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randint(0, Q, s, dtype=torch.int32, device="cuda", generator=g)
tag = os.path.basename(os.environ.get("DIL_LIB_PATH", "default"))
for level in (3, 5):
    K, L = KL[level]; n = 8192
    A, y = rnd(n, K, L, 256), rnd(n, L, 256)
    w = torch.empty((n, K, 256), dtype=torch.int32, device="cuda")
    d = timeit(lambda: api.matvec(A, y, level, out=w), 30)
    s = timeit(lambda: api.matvec(A[:1], y, level, shared_A=True, out=w), 30)
    print(f"{tag:24s} L{level} matvec distinct {d*1e3:7.1f} us   shared {s*1e3:7.1f} us")
