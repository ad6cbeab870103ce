#!/usr/bin/env python3
"""Small fixed workloads for rocprofv3 (kernel trace or --pmc passes):
   prof_target.py ntt | verify | verify_rot | verify_shared | sign | signloop | hash | scheme | all   [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402

Q = 8380417


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    api.init(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randint(0, Q, s, dtype=torch.int32, device="cuda", generator=g)  # noqa: E731
    bench = what == "bench"       # the workloads bench.py times: configs[1], configs[3] (distinct pk), the wire-format verify
    if bench:
        what = "ntt+verify"
    if what in ("ntt", "all", "ntt+verify"):
        bufs = [rnd(65536, 256) for _ in range(8)]           # 512 MiB rotating: HBM, not LLC
        for i in range(reps * 8):
            api.ntt(bufs[i % 8])
            api.invntt(bufs[i % 8])
    if what in ("verify", "verify_shared", "all", "ntt+verify"):
        n, K, L = 8192, 6, 5
        A, z, c = rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256)
        t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
        h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
        w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
        for _ in range(reps):
            if what != "verify_shared":
                api.verify_core(A, z, c, t1, h, 3, out=w1)
            if what not in ("verify", "ntt+verify"):
                api.verify_core(A[:1], z, c, t1[:1], h, 3, shared_pk=True, out=w1)
    if what == "verify_rot":        # FOUR input sets rotating (1.44 GB): HBM-streaming like bench.py's secondary metric (two are LLC-assisted)
        n, K, L = 8192, 6, 5
        sets = []
        for _ in range(4):
            t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
            h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
            sets.append((rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256), t1, h))
        w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
        for i in range(4 * reps + 4):
            A, z, c, t1, h = sets[i % 4]
            api.verify_core(A, z, c, t1, h, 3, out=w1)
    if what == "wire" or bench:     # the fused wire-format verify kernel, distinct pk (bench.py's verify_wire_core leg)
        n = 8192
        u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
        seed, mu = u8(n, 32), u8(n, 64)
        pk, sk = api.keygen(seed, 3)
        sig, _ = api.sign(sk, mu, 3)
        A = api.expand_a(pk[:, :32].contiguous(), 3)
        for _ in range(reps):
            api.verify_wire_core(A, pk, sig, 3)
            api.verify_sig(pk, sig, mu, 3)
    if what in ("matvec_shared", "matvec"):
        n, K, L = 8192, 6, 5
        A, y = rnd(n if what == "matvec" else 1, K, L, 256), rnd(n, L, 256)
        w = torch.empty((n, K, 256), dtype=torch.int32, device="cuda")
        for _ in range(reps):
            api.matvec(A, y, 3, shared_A=(what == "matvec_shared"), out=w)
    if what in ("sign", "all"):
        n, K, L = 8192, 8, 7
        A, y, c = rnd(1, K, L, 256), rnd(n, L, 256), rnd(n, 256)
        s1h, s2h, t0h = rnd(1, L, 256), rnd(1, K, 256), rnd(1, K, 256)
        for _ in range(reps):
            w1, w0 = api.sign_phase1(A, y, 5, shared_key=True)
            api.sign_phase2(c, y, w0, w1, s1h, s2h, t0h, 5, shared_key=True, small_key=True)
    if what == "signloop":          # the whole signing loop, level 5, 8192 messages, one key (configs[4]'s loop): options from the environment
        u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
        pk, sk = api.keygen(u8(1, 32), 5)
        mu = u8(8192, 64)
        for _ in range(reps):
            api.sign(sk, mu, 5, shared_sk=True)
    if what in ("hash", "scheme"):
        n = 8192
        u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
        seed, mu = u8(n, 32), u8(n, 64)
        kap = torch.zeros(n, dtype=torch.int32, device="cuda")
        for _ in range(reps):
            if what == "hash":
                api.expand_a(u8(n, 32), 3)
                api.expand_mask(mu, kap, 5)
                api.shake256(u8(n, 1088), 32)
            else:
                pk, sk = api.keygen(seed, 3)
                sig, _ = api.sign(sk[:1], mu, 3, shared_sk=True)
                api.verify_sig(pk[:1], sig, mu, 3, shared_pk=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
