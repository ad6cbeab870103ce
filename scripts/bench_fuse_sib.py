#!/usr/bin/env python3
"""A/B of option fuse_sib (round 6): SampleInBall inside verify_wire_wpi_kernel against the sampling launch in front of it.
verify_wire_core over FOUR rotating input sets (HBM-streaming, as bench.py's configs[3] leg), dil_verify_sig_expanded_dev and
dil_verify_sig_dev with a key per signature, levels 2 / 3 / 5.   usage: bench_fuse_sib.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402
from scripts.bench_fused import timeit  # noqa: E402

api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
for level in (3, 2, 5):
    seed, mu = u8(n, 32), u8(n, 64)
    pk, sk = api.keygen(seed, level)
    sig, _ = api.sign(sk, mu, level)
    A0 = api.expand_a(pk[:, :32].contiguous(), level)
    sets = [(A0 if j == 0 else A0.clone(), pk.clone(), sig.clone()) for j in range(4)]
    ref = None
    for mode in (0, 1, 3, 0, 1):
        api.set_option("fuse_sib", mode)
        i = [0]

        def core():
            A, p_, s_ = sets[i[0] % 4]
            i[0] += 1
            return api.verify_wire_core(A, p_, s_, level)
        w1p, v = core()
        if ref is None:
            ref = (w1p.clone(), v.clone())
        same = bool(torch.equal(w1p, ref[0]) and torch.equal(v, ref[1]))
        t = timeit(core, 12)
        te = timeit(lambda: api.verify_sig_expanded(A0, pk, sig, mu, level), 6)
        ts = timeit(lambda: api.verify_sig(pk, sig, mu, level), 6)
        ok = int(api.verify_sig(pk, sig, mu, level).abs().sum()) == 0 and int(api.verify_sig_expanded(A0, pk, sig, mu, level).abs().sum()) == 0
        print(f"L{level} n={n} fuse_sib={mode}: wire core {t*1e3:7.1f} us {n/t/1e3:7.2f} M/s | verify_sig_expanded {te*1e3:7.1f} us {n/te/1e3:7.2f} M/s | "
              f"verify_sig {ts*1e3:7.1f} us {n/ts/1e3:7.2f} M/s | identical w1 / verdicts {same}, all accept {ok}", flush=True)
    api.set_option("fuse_sib", 1)
