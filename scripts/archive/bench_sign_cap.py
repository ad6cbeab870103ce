#!/usr/bin/env python3
"""sweep of the signing loop's speculation width (option sign_cap = entries in flight per round) by level and batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
for n in (1024, 8192, 65536):
    seed, mu = u8(n, 32), u8(n, 64)
    for level in (2, 3, 5):
        pk, sk = api.keygen(seed, level)
        row = []
        for cap in sorted({max(16384, m * n) for m in (1, 2, 3, 4)} | {16384, 24576, 32768}):
            if cap < n:
                continue
            api.set_option("sign_cap", cap)
            reps = 4 if n <= 8192 else 2
            t = min(timeit(lambda: api.sign(sk[:1], mu, level, shared_sk=True), reps) for _ in range(2))
            td = min(timeit(lambda: api.sign(sk, mu, level), reps) for _ in range(2)) if n <= 8192 else float("nan")
            row.append(f"cap {cap:6d}: {t*1e3:7.0f} / {td*1e3:7.0f}")
        print(f"L{level} n={n:6d} shared/distinct us | " + " | ".join(row))
