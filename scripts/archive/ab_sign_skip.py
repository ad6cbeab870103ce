#!/usr/bin/env python3
"""Signing loop with and without option sign_skip (phase 2 of a speculative round drops the attempts behind an item's first accepted
one): signatures and attempt counts must be identical, the time is printed interleaved.  Optional sweep of sign_cap / sign_waste with
the skip on (arguments: `sweep`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit

api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
sweep = "sweep" in sys.argv[1:]
MODES = (0, 1, 2, 3)       # sign_skip: bit 0 = drop superseded attempts, bit 1 = work queue


def run(sk, mu, level, shared):
    return api.sign(sk[:1] if shared else sk, mu, level, shared_sk=shared)


for n in (8192, 65536, 1024):
    seed, mu = u8(n, 32), u8(n, 64)
    for level in (3, 5, 2):
        pk, sk = api.keygen(seed, level)
        for shared in (True, False):
            if not shared and n > 8192:
                continue
            out = {}
            tm = {m: [] for m in MODES}
            for rep in range(3):
                for skip in MODES:
                    api.set_option("sign_skip", skip)
                    if rep == 0:
                        out[skip] = run(sk, mu, level, shared)
                    tm[skip].append(timeit(lambda: api.sign(sk[:1] if shared else sk, mu, level, shared_sk=shared), 4 if n <= 8192 else 2))
            same = all(bool((out[0][0] == out[m][0]).all()) and bool((out[0][1] == out[m][1]).all()) for m in MODES)
            print(f"L{level} n={n:6d} {'one key ' if shared else 'key/item'} " + "  ".join(f"skip={m}: {min(tm[m])*1e3:7.0f} us {n/min(tm[m])/1e3:5.2f} M/s" for m in MODES) +
                  f"   ({'identical' if same else 'DIFFERENT'} signatures and attempt counts; mean attempts {out[0][1].float().mean().item():.2f})", flush=True)
            assert same
        if sweep:
            api.set_option("sign_skip", 3)
            for waste in (6144, 12288, 24576):
                api.set_option("sign_waste", waste)
                row = []
                for cap in sorted({max(16384, m * n) for m in (1, 2, 3, 4)} | {16384, 24576, 32768, 49152}):
                    if cap < n:
                        continue
                    api.set_option("sign_cap", cap)
                    t = min(timeit(lambda: api.sign(sk[:1], mu, level, shared_sk=True), 4 if n <= 8192 else 2) for _ in range(2))
                    row.append(f"{cap:6d}: {t*1e3:6.0f}")
                print(f"   L{level} n={n:6d} one key, waste {waste:5d}, us by cap | " + " | ".join(row), flush=True)
            api.set_option("sign_cap", 0)
            api.set_option("sign_waste", 6144)
