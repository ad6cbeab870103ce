#!/usr/bin/env python3
"""Operations on (key, message) -- mu = SHAKE256(tr || M) on the device: dil_sign_msg_dev / dil_verify_msg_dev, level 3, 64-byte messages,
one key for the batch and a key per message.   usage: bench_msg.py [batch ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
from scripts.bench_fused import timeit

api.init(0)
level = 3
g = torch.Generator(device="cuda").manual_seed(0)
for n in [int(a) for a in sys.argv[1:]] or [1, 64, 8192]:
    seed = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = api.keygen(seed, level)
    blob = torch.randint(0, 256, (n * 64,), dtype=torch.uint8, device="cuda", generator=g)
    offs = (torch.arange(n, device="cuda", dtype=torch.int64) * 64).contiguous()
    lens = torch.full((n,), 64, dtype=torch.int32, device="cuda")
    sig1, _ = api.sign_msg(sk[:1], blob, offs, lens, level, shared_sk=True)
    sigd, _ = api.sign_msg(sk, blob, offs, lens, level)
    ok = int(api.verify_msg(pk[:1], sig1, blob, offs, lens, level, shared_pk=True).abs().sum()) == 0 and \
        int(api.verify_msg(pk, sigd, blob, offs, lens, level).abs().sum()) == 0
    tv1 = min(timeit(lambda: api.verify_msg(pk[:1], sig1, blob, offs, lens, level, shared_pk=True), 10) for _ in range(3))
    tvd = min(timeit(lambda: api.verify_msg(pk, sigd, blob, offs, lens, level), 10) for _ in range(3))
    ts1 = min(timeit(lambda: api.sign_msg(sk[:1], blob, offs, lens, level, shared_sk=True), 5) for _ in range(3))
    print(f"L{level} n={n}: verify_msg one key {tv1*1e3:8.1f} us | key per message {tvd*1e3:8.1f} us | sign_msg one key {ts1*1e3:8.1f} us | accept {ok}", flush=True)
