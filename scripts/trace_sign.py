#!/usr/bin/env python3
"""One warm + one traced dil_sign_dev call (shared key), for rocprofv3 --kernel-trace. usage: trace_sign.py level batch shared"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
level, n, shared = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
seed, mu = u8(n, 32), u8(n, 64)
pk, sk = api.keygen(seed, level)
k = sk[:1] if shared else sk
for _ in range(2):
    sig, att = api.sign(k, mu, level, shared_sk=bool(shared))
torch.cuda.synchronize()
marker = torch.zeros(1, device="cuda")      # a foreign kernel marks the start of the traced call
marker += 1
sig, att = api.sign(k, mu, level, shared_sk=bool(shared))
torch.cuda.synchronize()
