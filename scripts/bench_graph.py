#!/usr/bin/env python3
"""Does a captured graph shorten the dispatch gap between the NTT launches?  K steps (fwd + inv over 65536 polynomials,
8 rotating batches) as a plain launch loop vs one torch.cuda.CUDAGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api

api.init(0)
B, R, K = 65536, 8, 200
g = torch.Generator(device="cuda").manual_seed(0)
bufs = [torch.randint(0, 8380417, (B, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]


def steps(k):
    for i in range(k):
        api.ntt(bufs[i % R])
        api.invntt(bufs[i % R])


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


t_plain = timed(lambda: steps(K))
print(f"plain loop : {t_plain / K * 1e6:7.2f} us per step  ({2 * B * K / t_plain / 1e9:.3f} G NTT/s)")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    steps(8)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        steps(K)
torch.cuda.synchronize()
t_graph = timed(graph.replay)
print(f"graph      : {t_graph / K * 1e6:7.2f} us per step  ({2 * B * K / t_graph / 1e9:.3f} G NTT/s)")

# several streams: step i on stream i % S (fwd then inv of one batch stay ordered on their stream); the other
# streams' kernels fill the dispatch gap and the ramp/tail of each launch
for S in (2, 3, 4):
    ss = [torch.cuda.Stream() for _ in range(S)]

    def stepsS(k):
        for i in range(k):
            with torch.cuda.stream(ss[i % S]):
                api.ntt(bufs[i % R])
                api.invntt(bufs[i % R])

    t = timed(lambda: stepsS(K))
    print(f"{S} streams  : {t / K * 1e6:7.2f} us per step  ({2 * B * K / t / 1e9:.3f} G NTT/s, {2 * B * K * 2048 / t / 1e12:.2f} TB/s)")
