#!/usr/bin/env python3
"""Pointwise kernels (H4: multiply, multiply-accumulate, add, sub) at batch 65536, HBM-streaming (rotating buffers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit

api.init(0)
n = 65536
g = torch.Generator(device="cuda").manual_seed(0)
R = 4
rnd = lambda: [torch.randint(0, 8380417, (n, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]
a, b, c, acc = rnd(), rnd(), rnd(), rnd()
i = [0]


def rot(fn):
    def f():
        k = i[0] % R
        i[0] += 1
        fn(k)
    return f


for name, fn, polys in (("pointwise      c = a o b", lambda k: api.pointwise_barrett(c[k], a[k], b[k]), 3),
                        ("pointwise_acc  c = acc + a o b", lambda k: api.pointwise_acc(c[k], acc[k], a[k], b[k]), 4),
                        ("poly_add       c = a + b", lambda k: api.poly_add(c[k], a[k], b[k]), 3),
                        ("poly_sub       c = a - b", lambda k: api.poly_sub(c[k], a[k], b[k]), 3)):
    t = timeit(rot(fn), 40)
    print(f"{name:34s} n={n}: {t*1e3:7.1f} us  {n/t/1e6:6.2f} G polys/s  {n*polys*1024/t/1e6:7.1f} GB/s algorithmic ({polys} KiB per polynomial)")
