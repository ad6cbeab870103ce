#!/usr/bin/env python3
"""verify level 3/5 timing only (for A/B of library builds via DIL_LIB_PATH)"""
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
    A, z, c = rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256)
    t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
    h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
    w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        d = timeit(lambda: api.verify_core(A, z, c, t1, h, level, out=w1), 30)
        s = timeit(lambda: api.verify_core(A[:1], z, c, t1[:1], h, level, shared_pk=True, out=w1), 30)
        print(f"{tag:24s} L{level} distinct {d*1e3:7.1f} us ({n/d/1e3:6.1f} M/s)   shared {s*1e3:7.1f} us ({n/s/1e3:6.1f} M/s)")
