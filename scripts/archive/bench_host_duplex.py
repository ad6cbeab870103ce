#!/usr/bin/env python3
"""configs[1] through dil_ntt_host: the two pipelines of round 5 against the round-robin one.
  page-locked caller buffer: option host_duplex (one stream per direction) by chunk size (host_chunk_pinned, KiB) and staging buffers
  pageable caller buffer:    option host_threads (a helper thread downloads) by chunk size (host_chunk) and staging buffers"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dilithium_amd import api  # noqa: E402
from oracle.oracle import splitmix64_polys  # noqa: E402


def med(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


api.init(0)
n = 65536
a = splitmix64_polys(n, seed=3)
pinned = torch.empty((n, 256), dtype=torch.int32).pin_memory()
print(f"dil_ntt_host, {n} polynomials (64 MiB up + 64 MiB down per call); the link alone: 57 GB/s each way")
for kind, x, opt, chunk_opt in (("page-locked", pinned.numpy(), "host_duplex", "host_chunk_pinned"), ("pageable", np.empty((n, 256), np.int32), "host_threads", "host_chunk")):
    for v in ((0, 1) if opt == "host_duplex" else (1, 2)):
        for ns in (2, 3, 4, 8):
            row = []
            for chunk in (1024, 2048, 4096, 8192, 16384):
                api.set_option(opt, v); api.set_option("host_streams", ns); api.set_option(chunk_opt, chunk)
                x[:] = a
                api.ntt(x); api.invntt(x)
                assert (x == a).all()
                t = med(lambda: api.ntt(x))
                row.append(f"{chunk:5d} KiB {t * 1e3:5.2f} ms {n * 1024 / t / 1e9:4.1f} GB/s")
            print(f"  {kind:11s} {opt}={v} buffers={ns}: " + " | ".join(row), flush=True)
for size in (8192, 16384, 32768, 131072, 262144):
    api.set_option("host_streams", 4); api.set_option("host_chunk", 8192); api.set_option("host_chunk_pinned", 8192)
    y = splitmix64_polys(size, seed=4)
    row = []
    for th in (1, 2):
        api.set_option("host_threads", th)
        t = med(lambda: api.ntt(y))
        row.append(f"host_threads={th}: {t * 1e3:6.2f} ms {size / t / 1e6:5.1f} M NTT/s")
    print(f"  pageable, batch {size:6d}: " + "   ".join(row), flush=True)
