#!/usr/bin/env python3
"""Interleaved A/B timing of library builds on the fused verify core (level 3, batch 8192, two rotating input sets =
HBM-streaming): every build is dlopen'ed into THIS process and the builds take turns, round after round, so clock /
power-state drift hits them equally.  usage: ab_verify.py [--level 3] [--shared] [--kind verify|matvec|sign] lib1.so lib2.so ...
(`default` = the in-tree library)"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dilithium_amd import _build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--shared", action="store_true")
ap.add_argument("--kind", default="verify", choices=["verify", "matvec", "sign1", "sign2", "ntt", "wire", "pair"])
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--generic", action="store_true", help="sign2: the entry point for arbitrary residues even where the small-key one exists")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--sets", type=int, default=4, help="rotating input sets (4 x 360 MiB at level 3: past the 256 MiB Infinity Cache for real; "
                "two sets, the default of rounds 2-4, are LLC-assisted -- profiles/r05i_rotating_sets.txt)")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
KL = {2: (4, 4), 3: (6, 5), 5: (8, 7)}
K, L = KL[a.level]
n = a.batch
Q = 8380417
torch.cuda.init()
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randint(0, Q, s, dtype=torch.int32, device="cuda", generator=g)  # noqa: E731
sets = []
for _ in range(a.sets):
    t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
    h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
    sets.append((rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256), t1, h))
w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
w = torch.empty((n, K, 256), dtype=torch.int32, device="cuda")
w0 = torch.empty((n, K, 256), dtype=torch.int32, device="cuda")
zo = torch.empty((n, L, 256), dtype=torch.int32, device="cuda")
ho = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
fl = torch.empty((n,), dtype=torch.int32, device="cuda")
nk = 1 if a.shared else n
s1h, s2h, t0h = rnd(nk, L, 256), rnd(nk, K, 256), rnd(nk, K, 256)
w1in = torch.randint(0, 16, (n, K, 256), dtype=torch.uint8, device="cuda", generator=g)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
PKB, SGB = {2: 1312, 3: 1952, 5: 2592}[a.level], {2: 2420, 3: 3293, 5: 4595}[a.level]
wire_in = []
if a.kind == "wire":        # wire-format fused verify kernel: random key / signature BYTES (timing and bit-equality across builds only)
    for _ in range(2):
        wire_in.append((torch.randint(0, 256, (1 if a.shared else n, PKB), dtype=torch.uint8, device="cuda", generator=g),
                        torch.randint(0, 256, (n, SGB), dtype=torch.uint8, device="cuda", generator=g)))
w1pk = torch.empty((n, K * (192 if a.level == 2 else 128)), dtype=torch.uint8, device="cuda")
vdw = torch.empty((n,), dtype=torch.int32, device="cuda")
nttb = [rnd(65536, 256) for _ in range(8)] if a.kind == "ntt" else []
libs = []
for path in a.libs:
    L_ = C.CDLL(_build.LIB if path == "default" else os.path.abspath(path))
    L_.dil_verify_core_dev.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    L_.dil_ntt_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L_.dil_invntt_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L_.dil_matvec_dev.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    L_.dil_sign_phase1_dev.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    L_.dil_sign_phase2_dev.argtypes = [C.c_void_p] * 10 + [C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    if hasattr(L_, "dil_sign_phase2_skey_dev"):          # round 4: the small-key kernels have an entry point of their own
        L_.dil_sign_phase2_skey_dev.argtypes = [C.c_void_p] * 10 + [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L_.dil_verify_wire_core_dev.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    L_.dil_event_elapsed_ms.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    assert L_.dil_init(0) == 0
    libs.append((os.path.basename(path), L_))
E = libs[0][1]
if a.kind in ("sign2", "pair"):
    # phase 2 needs a REAL challenge and key: c = SampleInBall(c~) (tau coefficients +-1), s1 / s2 with |coefficients| <= eta, t0 below 2^12 --
    # the kernel reads c s1 and c s2 off one transform, which is exact for such inputs only (pipeline_common.hpp SmallPair)
    E.dil_sample_in_ball_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    eta = {2: 2, 3: 4, 5: 2}[a.level]
    small = lambda lim, *sh: (torch.randint(-lim, lim + 1, sh, dtype=torch.int64, device="cuda", generator=g) % Q).to(torch.int32)  # noqa: E731
    s1h, s2h, t0h = small(eta, nk, L, 256), small(eta, nk, K, 256), small(4095, nk, K, 256)
    for t in (s1h, s2h, t0h):
        assert E.dil_ntt_dev(p(t), t.numel() // 256, None) == 0
    for st in sets:
        ct = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        assert E.dil_sample_in_ball_dev(p(st[2]), p(ct), a.level, n, None) == 0
    torch.cuda.synchronize()
e0, e1 = C.c_void_p(), C.c_void_p()
E.dil_event_create(C.byref(e0))
E.dil_event_create(C.byref(e1))


def run(L_, reps):
    for i in range(reps):
        A, z, c, t1, h = sets[i % len(sets)]
        sh = 1 if a.shared else 0
        if a.kind == "ntt":
            b = nttb[i & 7]
            rc = L_.dil_ntt_dev(p(b), 65536, None) | L_.dil_invntt_dev(p(b), 65536, None)
        elif a.kind == "wire":
            rc = L_.dil_verify_wire_core_dev(p(w1pk), p(vdw), p(A), p(wire_in[i & 1][0]), p(wire_in[i & 1][1]), a.level, n, sh, None)
        elif a.kind == "verify":
            rc = L_.dil_verify_core_dev(p(w1), p(A), p(z), p(c), p(t1), p(h), a.level, n, sh, None)
        elif a.kind == "matvec":
            rc = L_.dil_matvec_dev(p(w), p(A), p(z), a.level, n, sh, None)
        elif a.kind == "sign1":
            rc = L_.dil_sign_phase1_dev(p(w1), p(w0), p(A), p(z), a.level, n, sh, None)
        elif a.kind == "pair":           # the attempt as bench.py times it: phase 1, then phase 2 on its outputs (one key: A, y vary per set)
            rc = L_.dil_sign_phase1_dev(p(w1), p(w0), p(A), p(z), a.level, n, sh, None)
            rc |= L_.dil_sign_phase2_skey_dev(p(zo), p(ho), p(fl), p(c), p(z), p(w0), p(w1), p(s1h), p(s2h), p(t0h), a.level, n, sh, 0, None)
        elif hasattr(L_, "dil_sign_phase2_skey_dev") and not a.generic:
            rc = L_.dil_sign_phase2_skey_dev(p(zo), p(ho), p(fl), p(c), p(z), p(A), p(w1in), p(s1h), p(s2h), p(t0h), a.level, n, sh, 0, None)
        else:
            rc = L_.dil_sign_phase2_dev(p(zo), p(ho), p(fl), p(c), p(z), p(A), p(w1in), p(s1h), p(s2h), p(t0h), a.level, n, sh, None)
        assert rc == 0


ref = None
for name, L_ in libs:            # all builds must agree bit for bit
    run(L_, 2)
    torch.cuda.synchronize()
    out = {"verify": w1, "matvec": w, "sign1": w0, "sign2": zo, "ntt": nttb[0] if nttb else w1, "wire": w1pk, "pair": zo}[a.kind].clone()
    if ref is None:
        ref = out
    if "_no" not in name:            # ablation builds (libdil256_no*.so) compute something else on purpose
        assert torch.equal(ref, out), f"{name} differs from {libs[0][0]}"
for _ in range(60):              # clock / power warm-up
    run(libs[0][1], 10)
torch.cuda.synchronize()
res = {name: [] for name, _ in libs}
for r in range(a.rounds):
    for name, L_ in libs:
        run(L_, 3)
        E.dil_event_record(e0, None)
        run(L_, a.reps)
        E.dil_event_record(e1, None)
        ms = C.c_float()
        E.dil_event_elapsed_ms(C.byref(ms), e0, e1)
        res[name].append(ms.value / a.reps * 1e3)
bytes_per = {2: 30720 - 1024 * 0, 3: 46080, 5: 0}[a.level] if not a.shared else 0
for name, _ in libs:
    v = np.array(res[name])
    frac = f" frac(med) {46080 * n / (np.median(v) * 1e-6) / 8e12:5.3f}" if (a.level == 3 and not a.shared) else ""
    print(f"{name:32s} {a.kind} L{a.level} {'shared' if a.shared else 'distinct'} n={n}: min {v.min():7.2f} med {np.median(v):7.2f} max {v.max():7.2f} us{frac}")
