#!/usr/bin/env python3
"""Wire-format verification: the fused kernel alone (two rotating input sets: HBM-streaming) against the int32 verify
core, and dil_verify_sig_dev fused (fuse_wire = 1) vs unfused (= 0), levels 2 / 3 / 5.   usage: bench_wire.py [batch]"""
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
KL = {2: (4, 4), 3: (6, 5), 5: (8, 7)}
for level in (2, 3, 5):
    K, L = KL[level]
    seed, mu = u8(n, 32), u8(n, 64)
    pk, sk = api.keygen(seed, level)
    sig, _ = api.sign(sk, mu, level)
    sig1, _ = api.sign(sk[:1], mu, level, shared_sk=True)
    sets = []
    for j in range(2):
        A = api.expand_a(pk[:, :32].contiguous(), level) if j == 0 else sets[0][0].clone()
        sets.append((A, pk.clone(), sig.clone()))
    i = [0]

    def fused():
        A, p_, s_ = sets[i[0] % 2]
        i[0] += 1
        api.verify_wire_core(A, p_, s_, level)
    t = timeit(fused, 10)
    wire_bytes = K * L * 1024 + sig.shape[1] - 32 + K * 320 + 256 + K * (192 if level == 2 else 128)
    print(f"L{level} verify_wire_core distinct n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M/s  {wire_bytes*n/t/1e6:8.1f} GB/s of wire bytes ({wire_bytes} B/verify)")
    A1 = sets[0][0][:1].contiguous()
    t = timeit(lambda: api.verify_wire_core(A1, pk[:1], sig1, level, shared_pk=True), 10)
    print(f"L{level} verify_wire_core shared   n={n}: {t*1e3:8.1f} us  {n/t/1e3:8.2f} M/s")
    for mode in (1, 0):
        api.set_option("fuse_wire", mode)
        t = timeit(lambda: api.verify_sig(pk, sig, mu, level), 5)
        ts = timeit(lambda: api.verify_sig(pk[:1], sig1, mu, level, shared_pk=True), 5)
        ok = int(api.verify_sig(pk, sig, mu, level).abs().sum()) == 0 and int(api.verify_sig(pk[:1], sig1, mu, level, shared_pk=True).abs().sum()) == 0
        print(f"L{level} verify_sig fuse_wire={mode}  n={n}: distinct {t*1e3:8.1f} us {n/t/1e3:7.2f} M/s | shared {ts*1e3:8.1f} us {n/t/1e3 if False else n/ts/1e3:7.2f} M/s | all accept {ok}")
    api.set_option("fuse_wire", 1)
    Aall = sets[0][0]
    t = timeit(lambda: api.verify_sig_expanded(Aall, pk, sig, mu, level), 5)
    ts = timeit(lambda: api.verify_sig_expanded(A1, pk[:1], sig1, mu, level, shared_pk=True), 5)
    print(f"L{level} verify_sig_expanded (A kept across calls) n={n}: distinct {t*1e3:8.1f} us {n/t/1e3:7.2f} M/s | shared {ts*1e3:8.1f} us {n/ts/1e3:7.2f} M/s")
