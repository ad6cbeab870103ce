#!/usr/bin/env python3
"""Wire-format verification with a key per signature: A sampled inside the verifying kernel (gen_a = 1, gen_kernels.hip)
against ExpandA to HBM + the fused kernel (gen_a = 0); the gen kernel alone.   usage: bench_gen.py [batch ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402
from scripts.bench_fused import timeit  # noqa: E402

api.init(0)
batches = [int(a) for a in sys.argv[1:]] or [8192]
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
for n in batches:
    for level in (2, 3, 5):
        seed, mu = u8(n, 32), u8(n, 64)
        pk, sk = api.keygen(seed, level)
        sig, _ = api.sign(sk, mu, level)
        A = api.expand_a(pk[:, :32].contiguous(), level)
        wa, va = api.verify_wire_core(A, pk, sig, level)
        wg, vg = api.verify_wire_core(None, pk, sig, level)
        same = bool((wa == wg).all()) and bool((va == vg).all())
        tg = timeit(lambda: api.verify_wire_core(None, pk, sig, level), 10)
        ta = timeit(lambda: api.verify_wire_core(A, pk, sig, level), 10)
        te = timeit(lambda: api.expand_a(pk[:, :32].contiguous(), level), 10)
        res = {}
        for mode in (1, 0):
            api.set_option("gen_a", mode)
            res[mode] = timeit(lambda: api.verify_sig(pk, sig, mu, level), 10)
            ok = int(api.verify_sig(pk, sig, mu, level).abs().sum()) == 0
        api.set_option("gen_a", 1)
        print(f"L{level} n={n}: wire core gen {tg*1e6:7.1f} us | A from HBM {ta*1e6:7.1f} us (+ ExpandA {te*1e6:7.1f} us) | same bytes {same} | "
              f"verify_sig gen_a=1 {res[1]*1e6:7.1f} us {n/res[1]/1e6:6.2f} M/s | gen_a=0 {res[0]*1e6:7.1f} us {n/res[0]/1e6:6.2f} M/s | accept {ok}", flush=True)
