"""The fused level-3 verify core (configs[3], a key per item) over 1 / 2 / 4 / 8 rotating input sets of 360 MiB: how much of the two-set
figure of rounds 1-4 is the 256 MiB Infinity Cache?  Same for the standalone NTT over 1 ... 32 rotating 64 MiB batches.
    python scripts/bench_verify_sets.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dilithium_amd import api  # noqa: E402
from dilithium_amd import lib as dlib  # noqa: E402


def main():
    api.init(0)
    L = dlib.load()
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *sh: torch.randint(0, 8380417, sh, dtype=torch.int32, device="cuda", generator=g)  # noqa: E731
    n = 8192

    def t_us(fn, reps):
        for i in range(reps // 4 + 8):
            fn(i)
        torch.cuda.synchronize()
        best = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / reps * 1e3)
        return sorted(best)[1]

    sets = []
    for j in range(8):
        sets.append([rnd(n, 6, 5, 256), rnd(n, 5, 256), rnd(n, 256), torch.randint(0, 1024, (n, 6, 256), dtype=torch.int32, device="cuda", generator=g),
                     (torch.rand((n, 6, 256), device="cuda", generator=g) < 0.03).to(torch.uint8), torch.empty((n, 6, 256), dtype=torch.uint8, device="cuda")])
    for k in (1, 2, 3, 4, 6, 8):
        us = t_us(lambda i: L.dil_verify_core_dev(P(sets[i % k][5]), P(sets[i % k][0]), P(sets[i % k][1]), P(sets[i % k][2]), P(sets[i % k][3]),
                                                  P(sets[i % k][4]), 3, n, 0, st), 600)
        print(f"verify core, {k} rotating set(s) ({k * 360} MiB): {us:7.2f} us per launch  {n / us:7.2f} M/s  {46080 * n / us / 1e3 / 8000:.3f} of 8 TB/s")
    del sets
    bufs = [rnd(65536, 256) for _ in range(32)]
    for k in (1, 2, 4, 8, 16, 32):
        us = t_us(lambda i: L.dil_ntt_dev(P(bufs[i % k]), 65536, st) | L.dil_invntt_dev(P(bufs[i % k]), 65536, st), 1000) / 2
        print(f"NTT fwd+inv on one stream, {k} rotating batch(es) ({k * 64} MiB): {us:6.2f} us per launch  {2048 * 65536 / us / 1e3 / 8000:.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
