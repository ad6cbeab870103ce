#!/usr/bin/env python3
"""Aggregate signing throughput with T host threads, each running dil_sign_dev on its own stream over its own slice."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api

api.init(0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
seed, mu = u8(1, 32), u8(n, 64)
pk, sk = api.keygen(seed, level)
torch.cuda.synchronize()
for T in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(T)]
    slices = [mu[i * n // T:(i + 1) * n // T].contiguous() for i in range(T)]

    def work(i, reps):
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                api.sign(sk, slices[i], level, shared_sk=True)
            streams[i].synchronize()

    def run(reps):
        ths = [threading.Thread(target=work, args=(i, reps)) for i in range(T)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return time.perf_counter() - t0

    run(2)
    dt = min(run(4) for _ in range(3)) / 4
    print(f"L{level} n={n} threads={T}: {dt*1e6:9.1f} us per {n} signatures  {n/dt/1e6:7.3f} M sig/s")
