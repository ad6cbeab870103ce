#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer forms (numpy in / numpy out, pageable memory): keygen, sign, verify."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dilithium_amd import api

api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)


def wall(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


for level in (2, 3, 5):
    seed = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    mu = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    t = wall(lambda: api.keygen_host(seed, level))
    pk, sk = api.keygen_host(seed, level)
    print(f"L{level} keygen_host        n={n}: {t*1e6:9.1f} us  {n/t/1e6:7.3f} M/s  ({(pk.nbytes+sk.nbytes)/t/1e9:5.1f} GB/s D2H)")
    t = wall(lambda: api.sign_host(sk[:1], mu, level, shared_sk=True))
    sig, att = api.sign_host(sk[:1], mu, level, shared_sk=True)
    print(f"L{level} sign_host shared   n={n}: {t*1e6:9.1f} us  {n/t/1e6:7.3f} M/s")
    t = wall(lambda: api.verify_sig_host(pk[:1], sig, mu, level, shared_pk=True))
    assert (api.verify_sig_host(pk[:1], sig, mu, level, shared_pk=True) == 0).all()
    print(f"L{level} verify_host shared n={n}: {t*1e6:9.1f} us  {n/t/1e6:7.3f} M/s  ({(sig.nbytes+mu.nbytes)/t/1e9:5.1f} GB/s H2D)")
