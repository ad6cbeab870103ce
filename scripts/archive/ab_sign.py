#!/usr/bin/env python3
"""dil_sign_dev (the whole signing loop from sk bytes + mu) under several library builds, interleaved in one process; signatures and
attempt counts must be identical.  usage: ab_sign.py [--levels 3 5] [--batches 8192 65536] lib1.so lib2.so ...  (`default` = in-tree)"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import _build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--levels", type=int, nargs="+", default=[3, 5])
ap.add_argument("--batches", type=int, nargs="+", default=[8192, 65536])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
torch.cuda.init()
g = torch.Generator(device="cuda").manual_seed(3)
u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
libs = []
for path in a.libs:
    L = C.CDLL(_build.LIB if path == "default" else os.path.abspath(path))
    L.dil_sign_dev.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L.dil_keygen_dev.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_size_t, C.c_void_p]
    for f in ("dil_pk_bytes", "dil_sk_bytes", "dil_sig_bytes"):
        getattr(L, f).restype = C.c_size_t
    assert L.dil_init(0) == 0
    libs.append((os.path.basename(path), L))
L0 = libs[0][1]
for level in a.levels:
    for n in a.batches:
        seed, mu = u8(n, 32), u8(n, 64)
        pk = torch.empty((n, L0.dil_pk_bytes(level)), dtype=torch.uint8, device="cuda")
        sk = torch.empty((n, L0.dil_sk_bytes(level)), dtype=torch.uint8, device="cuda")
        assert L0.dil_keygen_dev(p(pk), p(sk), p(seed), level, n, None) == 0
        for shared in (1, 0):
            sig = torch.empty((n, L0.dil_sig_bytes(level)), dtype=torch.uint8, device="cuda")
            att = torch.empty((n,), dtype=torch.int32, device="cuda")
            ref = None
            res = {name: [] for name, _ in libs}
            for r in range(a.rounds):
                for name, L in libs:
                    call = lambda: L.dil_sign_dev(p(sig), p(att), p(sk), p(mu), level, n, shared, 512, None)  # noqa: E731
                    assert call() == 0
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = (sig.clone(), att.clone())
                    assert torch.equal(sig, ref[0]) and torch.equal(att, ref[1]), name
                    reps = 5 if n <= 8192 else 3
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        call()
                    torch.cuda.synchronize()
                    res[name].append((time.perf_counter() - t0) / reps)
            print(f"L{level} n={n:6d} {'one key ' if shared else 'key/item'}  " +
                  "  ".join(f"{name}: {min(v) * 1e3:7.3f} ms {n / min(v) / 1e6:6.2f} M/s" for name, v in res.items()), flush=True)
