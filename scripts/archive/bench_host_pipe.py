"""End-to-end (H2D + kernel + D2H) rates of the host-pointer entry points, swept over the host pipe's options, beside the link's own rate:
configs[1] (65536 polynomials through dil_ntt_host / dil_invntt_host) and configs[3] (8192 level-3 verify cores through
dil_verify_core_host); pageable and page-locked caller buffers.   python scripts/bench_host_pipe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dilithium_amd import api  # noqa: E402
from oracle.oracle import splitmix64_polys  # noqa: E402


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    api.init(0)
    # the link: pinned 256 MiB each way, plain copies
    hbuf = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    def h2d():
        dbuf.copy_(hbuf, non_blocking=True); torch.cuda.synchronize()
    def d2h():
        hbuf.copy_(dbuf, non_blocking=True); torch.cuda.synchronize()
    for _ in range(2):
        h2d(); d2h()
    r_h2d, r_d2h = (256 << 20) / best(h2d) / 1e9, (256 << 20) / best(d2h) / 1e9
    pg = torch.empty(256 << 20, dtype=torch.uint8)
    def h2d_pageable():
        dbuf.copy_(pg); torch.cuda.synchronize()
    r_pg = (256 << 20) / best(h2d_pageable, 3) / 1e9
    print(f"link: pinned H2D {r_h2d:.1f} GB/s, pinned D2H {r_d2h:.1f} GB/s, pageable H2D {r_pg:.1f} GB/s (256 MiB copies)")
    n = 65536
    a = splitmix64_polys(n, seed=3)
    print(f"configs[1]: dil_ntt_host, {n} polynomials (64 MiB up, 64 MiB down per call)")
    for pin in (0, 1):
        for streams in (1, 2, 3, 4, 8):
            for chunk in (1024, 4096, 16384, 65536):
                if chunk == 65536 and streams > 1:
                    continue
                api.set_option("host_pin", pin); api.set_option("host_streams", streams); api.set_option("host_chunk", chunk)
                x = a.copy()
                api.ntt(x); api.invntt(x)
                assert (x == a).all()
                t = best(lambda: api.ntt(x), 5)
                print(f"  pin={pin} streams={streams} chunk={chunk:6d} KiB: {t * 1e3:7.2f} ms  {n / t / 1e6:6.2f} M NTT/s  {n * 1024 / t / 1e9:5.1f} GB/s each way")
    K, L, nv = 6, 5, 8192
    rng = np.random.default_rng(1)
    A = splitmix64_polys(nv * K * L, seed=5).reshape(nv, K, L, 256)
    z = splitmix64_polys(nv * L, seed=6).reshape(nv, L, 256)
    c = np.zeros((nv, 256), np.int32); c[:, ::7] = 1
    t1 = rng.integers(0, 1024, (nv, K, 256)).astype(np.int32)
    h = (rng.random((nv, K * 256)) < 0.03).astype(np.uint8)
    up = nv * (K * L + L + 1 + K) * 1024 + nv * K * 256
    print(f"configs[3]: dil_verify_core_host, {nv} level-3 items, a key per item ({up / 2**20:.0f} MiB up, {nv * K * 256 / 2**20:.0f} MiB down)")
    for pin in (0, 1):
        for streams in (1, 2, 3, 4):
            for chunk in (4096, 16384, 65536):
                api.set_option("host_pin", pin); api.set_option("host_streams", streams); api.set_option("host_chunk", chunk)
                api.verify_core(A, z, c, t1, h, 3)
                t = best(lambda: api.verify_core(A, z, c, t1, h, 3), 3)
                print(f"  pin={pin} streams={streams} chunk={chunk:6d} KiB: {t * 1e3:7.2f} ms  {nv / t / 1e6:6.3f} M verify/s  {up / t / 1e9:5.1f} GB/s up")


if __name__ == "__main__":
    main()
