#!/usr/bin/env python3
"""Small-batch latency: one dil_verify_sig_dev call (4 launches, helper-stream fork / join) as plain launches vs replayed from
a captured graph (torch.cuda.CUDAGraph around the C-ABI call), batches 1 / 64 / 1024, level 3, a key per signature and one key."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api

api.init(0)
level = 3
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
n = 1024
seed, mu = u8(n, 32), u8(n, 64)
pk, sk = api.keygen(seed, level)
sig, _ = api.sign(sk, mu, level)
sig1, _ = api.sign(sk[:1], mu, level, shared_sk=True)


def wall(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def one_at_a_time(fn, reps=100):
    """latency of ONE call: synchronise after every call"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for aux in (1, 0):
    api.set_option("aux_overlap", aux)
    for nb in (1, 64, 1024):
        for shared in (False, True):
            p_, s_, m_ = (pk[:1], sig1[:nb].contiguous(), mu[:nb].contiguous()) if shared else (pk[:nb].contiguous(), sig[:nb].contiguous(), mu[:nb].contiguous())
            call = lambda: api.verify_sig(p_, s_, m_, level, shared_pk=shared)  # noqa: E731
            plain, plain1 = wall(call), one_at_a_time(call)
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    call()
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=st):
                        v = call()
                torch.cuda.synchronize()
                rep, rep1 = wall(gr.replay), one_at_a_time(gr.replay)
                ok = int(v.abs().sum()) == 0
                msg = f"graph replay {rep:7.1f} us back-to-back, {rep1:7.1f} us one at a time (verdicts ok: {ok})"
            except Exception as e:  # noqa: BLE001
                msg = f"capture failed: {type(e).__name__}: {str(e)[:80]}"
            print(f"aux_overlap={aux} batch {nb:5d} {'one key ' if shared else 'key/item'}: plain {plain:7.1f} us back-to-back, {plain1:7.1f} us one at a time | {msg}", flush=True)
api.set_option("aux_overlap", 1)
