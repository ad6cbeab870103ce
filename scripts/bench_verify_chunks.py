#!/usr/bin/env python3
"""dil_verify_sig_dev with a key per signature: the three-lane chunk pipeline (option verify_chunks) against the one-pass sequence.
usage: bench_verify_chunks.py [level ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402

api.init(0)
g = torch.Generator(device="cuda").manual_seed(3)
u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for level in [int(a) for a in sys.argv[1:]] or [3]:
    for n in (8192, 65536):
        seed, mu = u8(n, 32), u8(n, 64)
        pk, sk = api.keygen(seed, level)
        sig, _ = api.sign(sk, mu, level)
        bad = sig.clone()
        bad[n // 2 + 1, 77] ^= 2
        bad[n - 1, 3] ^= 1
        ref = None
        for ch in (1, 2, 3, 4, 6, 8):
            api.set_option("verify_chunks", ch)
            v = api.verify_sig(pk, bad, mu, level)
            ref = v if ref is None else ref
            same = bool(torch.equal(v, ref)) and int((v != 0).sum()) == 2
            t = min(timeit(lambda: api.verify_sig(pk, sig, mu, level), 20) for _ in range(3))
            print(f"L{level} n={n:6d} chunks={ch}: {t * 1e6:8.1f} us  {n / t / 1e6:7.2f} M verify/s  verdicts identical to one pass: {same}", flush=True)
        api.set_option("verify_chunks", 4)
