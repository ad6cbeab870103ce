#!/usr/bin/env python3
"""Throughput of the row-N1 sampler kernels (device-resident, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit

api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
for level, (K, L) in ((3, (6, 5)), (5, (8, 7))):
    n = 8192
    rho, rhop, ct = u8(n, 32), u8(n, 64), u8(n, 32)
    kappa = torch.zeros(n, dtype=torch.int32, device="cuda")
    w1 = torch.randint(0, 16, (n, K, 256), dtype=torch.uint8, device="cuda", generator=g)
    t = timeit(lambda: api.expand_a(rho, level), 10)
    print(f"L{level} expand_a       n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M keys/s  {n*K*L/t/1e3:8.1f} M polys/s  {n*K*L*5/t/1e6:7.2f} G perm/s")
    t = timeit(lambda: api.expand_mask(rhop, kappa, level), 10)
    print(f"L{level} expand_mask    n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M/s  {n*L*5/t/1e6:7.2f} G perm/s")
    t = timeit(lambda: api.sample_in_ball(ct, level), 10)
    print(f"L{level} sample_in_ball n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M/s")
    t = timeit(lambda: api.pack_w1(w1, level), 10)
    print(f"L{level} pack_w1        n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M/s")
    buf = u8(n, 64 + K * 128)
    t = timeit(lambda: api.shake256(buf, 32), 10)
    print(f"L{level} H(mu||w1)      n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M/s  ({(64+K*128)//136+1} perms each)")
n = 1 << 20
buf = u8(n, 128)
t = timeit(lambda: api.shake256(buf, 32), 5)
print(f"shake256 128B->32B   n={n}: {t*1e3:8.1f} us  {n/t/1e6:8.3f} G hashes/s")
