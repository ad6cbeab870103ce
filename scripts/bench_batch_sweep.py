#!/usr/bin/env python3
"""per-item cost vs batch size for the fused kernels (finds fixed costs / quantisation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit, KL, Q

api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randint(0, Q, s, dtype=torch.int32, device="cuda", generator=g)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K, L = KL[level]
N = 32768
A, z, c = rnd(N, K, L, 256), rnd(N, L, 256), rnd(N, 256)
t1 = torch.randint(0, 1024, (N, K, 256), dtype=torch.int32, device="cuda", generator=g)
h = (torch.rand((N, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
w1 = torch.empty((N, K, 256), dtype=torch.uint8, device="cuda")
w = torch.empty((N, K, 256), dtype=torch.int32, device="cuda")
for n in (256, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768):
    vd = timeit(lambda: api.verify_core(A[:n], z[:n], c[:n], t1[:n], h[:n], level, out=w1[:n]), 20)
    vs = timeit(lambda: api.verify_core(A[:1], z[:n], c[:n], t1[:1], h[:n], level, shared_pk=True, out=w1[:n]), 20)
    md = timeit(lambda: api.matvec(A[:n], z[:n], level, out=w[:n]), 20)
    ms = timeit(lambda: api.matvec(A[:1], z[:n], level, shared_A=True, out=w[:n]), 20)
    print(f"L{level} n={n:6d}  verify d {vd*1e3:7.1f} us s {vs*1e3:7.1f} us | matvec d {md*1e3:7.1f} us s {ms*1e3:7.1f} us | verify-d {n/vd/1e3:7.1f} M/s")
