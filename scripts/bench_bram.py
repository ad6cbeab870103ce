#!/usr/bin/env python3
"""H6 (hardware-model API on `bram` with MAPPING) at batch 65536, HBM-streaming (rotating buffers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit

api.init(0)
n, R = 65536, 4
g = torch.Generator(device="cuda").manual_seed(0)
bufs = [torch.randint(0, 8380417, (n, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]
muls = [torch.randint(0, 8380417, (n, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]
i = [0]


def rot(fn):
    def f():
        k = i[0] % R
        i[0] += 1
        fn(k)
    return f


names = {0: "NATURAL", 1: "AFTER_NTT", 2: "AFTER_INVNTT"}
for m in (0, 1, 2):
    for label, fn, polys in ((f"ntt2x2_fwdntt mapping={names[m]}", lambda k, m=m: api.ntt2x2_fwdntt(bufs[k], m), 2),
                             (f"ntt2x2_invntt mapping={names[m]}", lambda k, m=m: api.ntt2x2_invntt(bufs[k], m), 2),
                             (f"ntt2x2_mul    mapping={names[m]}", lambda k, m=m: api.ntt2x2_mul(bufs[k], muls[k], m), 3)):
        t = timeit(rot(fn), 40)
        print(f"{label:40s} n={n}: {t*1e3:7.1f} us  {n/t/1e6:6.2f} G/s  {n*polys*1024/t/1e6:7.1f} GB/s algorithmic")
