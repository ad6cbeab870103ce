#!/usr/bin/env python3
"""does ExpandA's write stream overlap better when the batch is issued as 2-4 staggered launches on separate streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from dilithium_amd import api, lib as dlib
api.init(0)
L = dlib.load()
g = torch.Generator(device="cuda").manual_seed(0)
n, level, K, Lv = 8192, 3, 6, 5
rho = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
A = torch.empty((n, K, Lv, 256), dtype=torch.int32, device="cuda")
P = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
for parts in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    hs = [C.c_void_p(s.cuda_stream) for s in streams]
    def run():
        for j in range(parts):
            lo, hi = j * n // parts, (j + 1) * n // parts
            rc = L.dil_expand_a_dev(P(A, lo * K * Lv * 1024), P(rho, lo * 32), level, hi - lo, hs[j])
            assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"ExpandA L3 n=8192 as {parts} launch(es) on {parts} stream(s): {best*1e6:7.1f} us")
