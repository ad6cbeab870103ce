#!/usr/bin/env python3
"""The signing loop as it is against its host-free UPPER BOUND (scripts/sign_replay/apply.py builds the variant library):
    DIL_LIB_PATH=scripts/bin/libdil256_replay.so python scripts/sign_replay/bench_sign_replay.py [n]
Per level: one ordinary call gives attempts[] and with the exported speculation rule (dil_sign_round_plan) the pending count after every
round; then the same call is timed with the count read back by copy + event (sign_wake 0), posted into mapped words and polled (sign_wake 1, shipped) and with every round queued back to back (DIL_SIGN_REPLAY set: the bound)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dilithium_amd import api, lib as dlib  # noqa: E402

api.init(0)
L = dlib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(3)
u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
print("library:", os.environ.get("DIL_LIB_PATH", "in-tree"))


def timeit(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


def schedule(level, att):
    """pending counts after each round, by the library's own rule"""
    out, pending, done = [], n, 0
    while pending:
        s, e = C.c_int(), C.c_size_t()
        assert L.dil_sign_round_plan(level, C.c_size_t(n), C.c_size_t(pending), done, 512, C.byref(s), C.byref(e)) == 0
        done += s.value
        pending = int((att > done).sum())
        out.append((s.value, e.value, pending))
    return out


for level in (2, 3, 5):
    for shared in (1, 0):
        seed, mu = u8(1 if shared else n, 32), u8(n, 64)
        pk, sk = api.keygen(seed, level)
        os.environ.pop("DIL_SIGN_REPLAY", None)
        sig0, att = api.sign(sk, mu, level, shared_sk=bool(shared))
        sch = schedule(level, att)
        api.set_option("sign_wake", 0)
        t_event = timeit(lambda: api.sign(sk, mu, level, shared_sk=bool(shared)))
        api.set_option("sign_wake", 1)
        t_host = timeit(lambda: api.sign(sk, mu, level, shared_sk=bool(shared)))
        os.environ["DIL_SIGN_REPLAY"] = ",".join(str(p) for _, _, p in sch)
        sig1, att1 = api.sign(sk, mu, level, shared_sk=bool(shared))
        same = bool((sig0 == sig1).all()) and bool((att == att1).all())
        t_free = timeit(lambda: api.sign(sk, mu, level, shared_sk=bool(shared)))
        os.environ.pop("DIL_SIGN_REPLAY", None)
        print(f"L{level} n={n} {'one key' if shared else 'key/item'}: rounds (S, entries, pending after) {sch} | copy + event {t_event * 1e6:8.1f} us {n / t_event / 1e6:5.2f} M/s | "
              f"posted count (shipped) {t_host * 1e6:8.1f} us {n / t_host / 1e6:5.2f} M/s | queued back to back {t_free * 1e6:8.1f} us {n / t_free / 1e6:5.2f} M/s | {100 * (t_host / t_free - 1):+.1f} % | "
              f"identical {same}", flush=True)
