#!/usr/bin/env python3
"""ExpandA alone (A/B of library builds via DIL_LIB_PATH): levels 3 / 5, n = 8192 and 65536 keys"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
tag = os.path.basename(os.environ.get("DIL_LIB_PATH", "default"))
for level, (K, L) in ((3, (6, 5)), (5, (8, 7))):
    for n in (8192, 32768):
        rho = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        best = min(timeit(lambda: api.expand_a(rho, level), 10) for _ in range(3))
        print(f"{tag:24s} L{level} expand_a n={n}: {best*1e3:8.1f} us  {n*K*L*5/best/1e6:7.2f} G perm/s")
