#!/usr/bin/env python3
"""How long does the fused verify core take to reach its steady rate?  After an idle second, consecutive 10-ms regions of
back-to-back launches over two alternating input sets (HBM-streaming, as bench.py's secondary metric): us per launch by region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from dilithium_amd import api, lib as dlib
from scripts.bench_fused import KL, Q

api.init(0)
L = dlib.load()
level, n = 3, 8192
K, Lv = KL[level]
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randint(0, Q, s, dtype=torch.int32, device="cuda", generator=g)
sets = []
for _ in range(2):
    t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
    h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
    sets.append((rnd(n, K, Lv, 256), rnd(n, Lv, 256), rnd(n, 256), t1, h))
w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NREG, PER = 30, 160
evs = [C.c_void_p() for _ in range(NREG + 1)]
for e in evs:
    L.dil_event_create(C.byref(e))
for trial in range(3):
    torch.cuda.synchronize()
    time.sleep(1.0)
    k = 0
    L.dil_event_record(evs[0], st)
    for r in range(NREG):
        for _ in range(PER):
            A, z, c, t1, h = sets[k & 1]
            L.dil_verify_core_dev(P(w1), P(A), P(z), P(c), P(t1), P(h), level, n, 0, st)
            k += 1
        L.dil_event_record(evs[r + 1], st)
    torch.cuda.synchronize()
    ms = C.c_float()
    out = []
    for r in range(NREG):
        L.dil_event_elapsed_ms(C.byref(ms), evs[r], evs[r + 1])
        out.append(ms.value / PER * 1e3)
    print(f"trial {trial}: us per launch by {PER}-launch region: " + " ".join(f"{x:.1f}" for x in out), flush=True)
