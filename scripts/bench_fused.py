#!/usr/bin/env python3
"""Time the fused pipelines (configs[2..4] of BASELINE.json) on one GPU with HIP events.
Knobs come from the environment (DIL_FUSED_WGPC, DIL_NTT_BPC)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dilithium_amd import api  # noqa: E402
from dilithium_amd import lib as dlib  # noqa: E402

KL = {2: (4, 4), 3: (6, 5), 5: (8, 7)}
Q = 8380417


def timeit(fn, reps=20):
    L = dlib.load()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.dil_event_create(C.byref(e0)); L.dil_event_create(C.byref(e1))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.dil_event_record(e0, s)
    for _ in range(reps):
        fn()
    L.dil_event_record(e1, s)
    ms = C.c_float()
    L.dil_event_elapsed_ms(C.byref(ms), e0, e1)
    return ms.value / reps


def main():
    api.init(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *shape: torch.randint(0, Q, shape, dtype=torch.int32, device="cuda", generator=g)  # noqa: E731
    tag = f"wgpc={os.environ.get('DIL_FUSED_WGPC', 'default')}"
    for level, n in ((3, 8192), (2, 4096), (5, 8192)):
        K, L = KL[level]
        A, z, c = rnd(n, K, L, 256), rnd(n, L, 256), rnd(n, 256)
        t1 = torch.randint(0, 1024, (n, K, 256), dtype=torch.int32, device="cuda", generator=g)
        h = (torch.rand((n, K, 256), device="cuda", generator=g) < 0.03).to(torch.uint8)
        w1 = torch.empty((n, K, 256), dtype=torch.uint8, device="cuda")
        w = torch.empty((n, K, 256), dtype=torch.int32, device="cuda")
        kib = lambda polys, bytes_u8=0: (polys * 1024 + bytes_u8) * n  # noqa: E731
        ms = timeit(lambda: api.verify_core(A, z, c, t1, h, level, out=w1))
        b = kib(K * L + L + 1 + K, 2 * K * 256)
        print(f"{tag} L{level} verify distinct n={n}: {ms * 1e3:8.1f} us {n / ms / 1e3:8.2f} M/s {b / ms / 1e6:8.1f} GB/s")
        ms = timeit(lambda: api.verify_core(A[:1], z, c, t1[:1], h, level, shared_pk=True, out=w1))
        b = kib(L + 1, 2 * K * 256)
        print(f"{tag} L{level} verify shared   n={n}: {ms * 1e3:8.1f} us {n / ms / 1e3:8.2f} M/s {b / ms / 1e6:8.1f} GB/s")
        ms = timeit(lambda: api.matvec(A, z, level, out=w))
        b = kib(K * L + L + K)
        print(f"{tag} L{level} matvec distinct n={n}: {ms * 1e3:8.1f} us {n / ms / 1e3:8.2f} M/s {b / ms / 1e6:8.1f} GB/s")
        ms = timeit(lambda: api.matvec(A[:1], z, level, shared_A=True, out=w))
        b = kib(L + K)
        print(f"{tag} L{level} matvec shared   n={n}: {ms * 1e3:8.1f} us {n / ms / 1e3:8.2f} M/s {b / ms / 1e6:8.1f} GB/s")
        w1s, w0s = api.sign_phase1(A[:1], z, level, shared_key=True)
        ms = timeit(lambda: api.sign_phase1(A[:1], z, level, shared_key=True))
        print(f"{tag} L{level} sign1 shared    n={n}: {ms * 1e3:8.1f} us {n / ms / 1e3:8.2f} M/s (incl. torch.empty)")
        s1h, s2h, t0h = rnd(1, L, 256), rnd(1, K, 256), rnd(1, K, 256)
        ms = timeit(lambda: api.sign_phase2(c, z, w0s, w1s, s1h, s2h, t0h, level, shared_key=True))
        b = kib(1 + L + K + L, 2 * K * 256)
        print(f"{tag} L{level} sign2 shared    n={n}: {ms * 1e3:8.1f} us {n / ms / 1e3:8.2f} M/s {b / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
