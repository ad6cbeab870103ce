#!/usr/bin/env python3
"""Where does the kernel-shape switch belong?  verify core / mat-vec per batch size with the workgroup-per-item kernels
(fused_mode = 1) and the wave-per-item / shared-key kernels (fused_mode = 2); the library switches at 8 x #CUs = 2048 items."""
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
N = 4096
A, z, c = rnd(N, K, L, 256), rnd(N, L, 256), rnd(N, 256)
t1 = torch.randint(0, 1024, (N, K, 256), dtype=torch.int32, device="cuda", generator=g)
h = (torch.rand((N, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
w1 = torch.empty((N, K, 256), dtype=torch.uint8, device="cuda")
w = torch.empty((N, K, 256), dtype=torch.int32, device="cuda")
for n in (64, 256, 512, 768, 1024, 1536, 2048, 3072, 4096):
    row = []
    for mode in (1, 2):
        api.set_option("fused_mode", mode)
        vd = min(timeit(lambda: api.verify_core(A[:n], z[:n], c[:n], t1[:n], h[:n], level, out=w1[:n]), 20) for _ in range(2))
        vs = min(timeit(lambda: api.verify_core(A[:1], z[:n], c[:n], t1[:1], h[:n], level, shared_pk=True, out=w1[:n]), 20) for _ in range(2))
        md = min(timeit(lambda: api.matvec(A[:n], z[:n], level, out=w[:n]), 20) for _ in range(2))
        ms = min(timeit(lambda: api.matvec(A[:1], z[:n], level, shared_A=True, out=w[:n]), 20) for _ in range(2))
        row.append((vd, vs, md, ms))
    api.set_option("fused_mode", 0)
    a, b = row
    print(f"L{level} n={n:5d}  [wg-per-item | wave-per-item] verify distinct {a[0]*1e3:6.1f} | {b[0]*1e3:6.1f} us, shared {a[1]*1e3:6.1f} | {b[1]*1e3:6.1f}; "
          f"matvec distinct {a[2]*1e3:6.1f} | {b[2]*1e3:6.1f}, shared {a[3]*1e3:6.1f} | {b[3]*1e3:6.1f}", flush=True)
