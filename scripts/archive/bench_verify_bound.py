#!/usr/bin/env python3
"""What bounds the fused verify kernels?  {int32 A, 24-bit packed A} x {this build} on verify_wpi_kernel (int32 only) and
verify_wire_wpi_kernel, meant to be run under rocprofv3 --kernel-trace for a full build and for a -DDIL_ABL_NONTT build (no
transforms): the kernel durations of the four (six) cells come from the traces.   usage: bench_verify_bound.py [level] [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
K, L = {2: (4, 4), 3: (6, 5), 5: (8, 7)}[level]
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
rnd = lambda *s: torch.randint(0, 8380417, s, dtype=torch.int32, device="cuda", generator=g)  # noqa: E731
sets = []
for _ in range(2):                                   # two rotating input sets: HBM-streaming
    seed, mu = u8(n, 32), u8(n, 64)
    pk, sk = api.keygen(seed, level)
    sig, _ = api.sign(sk, mu, level)
    t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
    h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
    sets.append((pk, sig, mu, rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256), t1, h))
w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
for a24 in (0, 2):
    api.set_option("a24", a24)
    for i in range(12):
        pk, sig, mu, A, z, c, t1, h = sets[i & 1]
        api.verify_sig(pk, sig, mu, level)           # ExpandA (int32 | 24-bit) -> verify_wire_wpi_kernel<level, 0 | 1>
        if a24 == 0:
            api.verify_core(A, z, c, t1, h, level, out=w1)   # verify_wpi_kernel<level> (the public int32 form)
torch.cuda.synchronize()
api.set_option("a24", 1)
