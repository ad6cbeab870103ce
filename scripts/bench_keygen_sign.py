#!/usr/bin/env python3
"""keygen / sign (a key per message) / wire verify core rates of whichever library DIL_LIB_PATH selects: usage bench_keygen_sign.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402

api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(3)
u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for level in (2, 3, 5):
    seed, mu = u8(n, 32), u8(n, 64)
    pk, sk = api.keygen(seed, level)
    sig, _ = api.sign(sk, mu, level)
    A = api.expand_a(pk[:, :32].contiguous(), level)
    tk = timeit(lambda: api.keygen(seed, level), 10)
    ts = timeit(lambda: api.sign(sk, mu, level), 5)
    tw = timeit(lambda: api.verify_wire_core(A, pk, sig, level), 20)
    h = int(pk.sum()) ^ int(sk.sum()) ^ int(sig.sum())
    print(f"L{level} n={n}: keygen {tk * 1e6:8.1f} us {n / tk / 1e6:6.2f} M/s | sign key/item {ts * 1e6:8.1f} us {n / ts / 1e6:6.2f} M/s | "
          f"verify_wire_core {tw * 1e6:7.1f} us | checksum {h & 0xffffffff:08x}", flush=True)
