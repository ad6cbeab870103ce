#!/usr/bin/env python3
"""Whole-scheme throughput from wire bytes, device-resident (rows N2/N3/N4): keygen, sign, verify.
Usage: bench_scheme.py [batch]      prints one line per (level, op)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit

api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)


def wall(fn, reps=9):
    """median wall time of a (synchronising) call"""
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    wall.last_min = min(ts)
    return sorted(ts)[len(ts) // 2]


for level in (2, 3, 5):
    seed, mu = u8(n, 32), u8(n, 64)
    t = timeit(lambda: api.keygen(seed, level), 5)
    print(f"L{level} keygen              n={n}: {t*1e3:9.1f} us  {n/t/1e3:8.2f} M keys/s")
    pk, sk = api.keygen(seed, level)
    t = wall(lambda: api.sign(sk, mu, level))
    sig, att = api.sign(sk, mu, level)
    print(f"L{level} sign distinct keys  n={n}: {t*1e6:9.1f} us  {n/t/1e6:8.3f} M sig/s   (best {wall.last_min*1e6:7.1f} us)  mean attempts {att.float().mean():.2f} max {int(att.max())}")
    t = wall(lambda: api.sign(sk[:1], mu, level, shared_sk=True))
    sig1, att1 = api.sign(sk[:1], mu, level, shared_sk=True)
    print(f"L{level} sign shared key     n={n}: {t*1e6:9.1f} us  {n/t/1e6:8.3f} M sig/s   (best {wall.last_min*1e6:7.1f} us)  mean attempts {att1.float().mean():.2f} max {int(att1.max())}")
    t = timeit(lambda: api.verify_sig(pk, sig, mu, level), 5)
    assert int(api.verify_sig(pk, sig, mu, level).abs().sum()) == 0
    print(f"L{level} verify distinct pk  n={n}: {t*1e3:9.1f} us  {n/t/1e3:8.2f} M ver/s")
    t = timeit(lambda: api.verify_sig(pk[:1], sig1, mu, level, shared_pk=True), 5)
    assert int(api.verify_sig(pk[:1], sig1, mu, level, shared_pk=True).abs().sum()) == 0
    print(f"L{level} verify shared pk    n={n}: {t*1e3:9.1f} us  {n/t/1e3:8.2f} M ver/s")
