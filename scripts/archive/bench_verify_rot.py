#!/usr/bin/env python3
"""Fused verify core with HBM-streaming inputs (two alternating input sets, as bench.py's secondary metric), per level:
distinct pk (verify_wpi_kernel) and shared pk (verify_shared_kernel).  A/B of library builds via DIL_LIB_PATH.
usage: bench_verify_rot.py [levels e.g. 3 or 235] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402
from scripts.bench_fused import timeit, KL, Q  # noqa: E402

api.init(0)
levels = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "3")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randint(0, Q, s, dtype=torch.int32, device="cuda", generator=g)  # noqa: E731
tag = os.path.basename(os.environ.get("DIL_LIB_PATH", "default"))
BYTES = {2: 16 * 1024 + 4 * 1024 + 1024 + 4 * 1024 + 2 * 4 * 256, 3: 46080, 5: 56 * 1024 + 7 * 1024 + 1024 + 8 * 1024 + 2 * 8 * 256}
for level in levels:
    K, L = KL[level]
    n = 8192
    sets = []
    for _ in range(2):
        t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
        h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
        sets.append((rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256), t1, h))
    w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
    i = [0]

    def dist():
        A, z, c, t1, h = sets[i[0] & 1]
        i[0] += 1
        api.verify_core(A, z, c, t1, h, level, out=w1)

    def shared():
        A, z, c, t1, h = sets[i[0] & 1]
        i[0] += 1
        api.verify_core(A[:1], z, c, t1[:1], h, level, shared_pk=True, out=w1)
    best = (1e9, 1e9)
    for rep in range(3):
        d = timeit(dist, reps)
        s = timeit(shared, reps)
        best = (min(best[0], d), min(best[1], s))
    d, s = best
    print(f"{tag:28s} L{level} distinct {d*1e3:7.1f} us {n/d/1e3:6.1f} M/s {BYTES[level]*n/d/1e6:7.1f} GB/s frac {BYTES[level]*n/d/1e6/8000:5.3f} | shared {s*1e3:7.1f} us {n/s/1e3:6.1f} M/s")
