#!/usr/bin/env python3
"""The standalone NTT's one-launch HBM fraction by batch size (BASELINE configs[1] is 65536): each size rotates over enough buffers to exceed
the 256 MiB Infinity Cache twice; forward and inverse launches alternate on ONE stream, timed with HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402

api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
for n in (16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    nbuf = max(2, (512 << 20) // (n * 1024))
    bufs = [torch.randint(0, 8380417, (n, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(nbuf)]
    reps = max(40, (1 << 23) // n)
    for i in range(2 * nbuf):
        api.ntt(bufs[i % nbuf]); api.invntt(bufs[i % nbuf])
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            api.ntt(bufs[i % nbuf]); api.invntt(bufs[i % nbuf])
        e1.record(); e1.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / (2 * reps))
    us = sorted(best)[1]
    print(f"batch {n:8d} ({nbuf:2d} rotating buffers): {us:8.2f} us per launch  {n * 2048 / us / 1e3:7.1f} GB/s  {n * 2048 / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    del bufs
