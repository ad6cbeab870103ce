#!/usr/bin/env python3
"""dil_sign_dev under library options, interleaved: usage bench_sign_opts.py option v0 v1 [levels...]   (one key for the batch and a key per message)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402

opt, vals = sys.argv[1], [int(sys.argv[2]), int(sys.argv[3])]
levels = [int(a) for a in sys.argv[4:]] or [2, 3, 5]
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


for level in levels:
    for n in (8192, 65536):
        seed, mu = u8(n, 32), u8(n, 64)
        pk, sk = api.keygen(seed, level)
        for shared in (True, False):
            key = sk[:1] if shared else sk
            ref = None
            res = {v: [] for v in vals}
            for rnd in range(3):
                for v in vals:
                    api.set_option(opt, v)
                    sig, att = api.sign(key, mu, level, shared_sk=shared)
                    if ref is None:
                        ref = (sig.clone(), att.clone())
                    assert torch.equal(sig, ref[0]) and torch.equal(att, ref[1]), (opt, v)
                    res[v].append(timeit(lambda: api.sign(key, mu, level, shared_sk=shared), 5))
            line = "  ".join(f"{opt}={v}: {min(res[v]) * 1e3:7.3f} ms {n / min(res[v]) / 1e6:6.2f} M/s" for v in vals)
            print(f"L{level} n={n:6d} {'one key ' if shared else 'key/item'}  {line}   (signatures and attempt counts identical)", flush=True)
    api.set_option(opt, vals[-1])
