"""Do the signing loop's two Keccak kernels overlap when they are INDEPENDENT?  A wide round's challenge (H(mu || w1) + SampleInBall, 24576
entries, two lanes per sponge: 768 lone waves, latency-bound) and an ExpandMask of the same width (122880 sponges, lane per sponge:
throughput-bound) on two streams, against the same two launches on one stream.  Also phase 1 (matvec_shared) beside the challenge.
    python scripts/bench_overlap_probe.py [level] [entries]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dilithium_amd import api  # noqa: E402
from dilithium_amd import lib as dlib  # noqa: E402


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 24576
    api.init(0)
    L = dlib.load()
    K, Lv = {2: (4, 4), 3: (6, 5), 5: (8, 7)}[level]
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(1)
    u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
    mu, rp = u8(E, 64), u8(E, 64)
    w1p = u8(E, K * (192 if level == 2 else 128))
    kappa = torch.randint(0, 60000, (E,), dtype=torch.int32, device="cuda", generator=g)
    ct = torch.empty((E, 32), dtype=torch.uint8, device="cuda")
    c = torch.empty((E, 256), dtype=torch.int32, device="cuda")
    y = torch.empty((E, Lv, 256), dtype=torch.int32, device="cuda")
    A = torch.randint(0, 8380417, (1, K, Lv, 256), dtype=torch.int32, device="cuda", generator=g)
    w1 = torch.empty((E, K, 256), dtype=torch.uint8, device="cuda")
    w0 = torch.empty((E, K, 256), dtype=torch.int32, device="cuda")
    y2 = torch.randint(0, 8380417, (E, Lv, 256), dtype=torch.int32, device="cuda", generator=g)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h1, h2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)

    chal = lambda st: L.dil_challenge_dev(P(ct), P(c), P(mu), P(w1p), level, E, st)  # noqa: E731
    mask = lambda st: L.dil_expand_mask_dev(P(y), P(rp), P(kappa), level, E, st)  # noqa: E731
    ph1 = lambda st: L.dil_sign_phase1_dev(P(w1), P(w0), P(A), P(y2), level, E, 1, st)  # noqa: E731

    def t_us(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def serial(a, b):
        def f():
            with torch.cuda.stream(s1):
                dlib.check(a(h1) | b(h1))
        return f

    def forked(a, b):
        def f():
            s2.wait_stream(s1)
            dlib.check(a(h1) | b(h2))
            s1.wait_stream(s2)
        return f

    with torch.cuda.stream(s1):
        only = {n: t_us(lambda fn=fn: dlib.check(fn(h1))) for n, fn in (("challenge", chal), ("expand_mask", mask), ("phase1", ph1))}
        print(f"level {level}, {E} entries; alone: " + ", ".join(f"{n} {v:.1f} us" for n, v in only.items()))
        for na, a, nb, b in (("challenge", chal, "expand_mask", mask), ("challenge", chal, "phase1", ph1), ("expand_mask", mask, "phase1", ph1)):
            ts, tf = t_us(serial(a, b)), t_us(forked(a, b))
            print(f"  {na} + {nb}: one stream {ts:.1f} us, two streams (fork / join) {tf:.1f} us   (max of the two alone {max(only[na], only[nb]):.1f})")


if __name__ == "__main__":
    main()
