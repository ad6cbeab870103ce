#!/usr/bin/env python3
"""One warm + one traced dil_verify_sig_dev / dil_keygen_dev call, for rocprofv3 --kernel-trace. usage: trace_verify.py level batch shared"""
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
k, p = (sk[:1], pk[:1]) if shared else (sk, pk)
sig, att = api.sign(k, mu, level, shared_sk=bool(shared))
for _ in range(2):
    api.verify_sig(p, sig, mu, level, shared_pk=bool(shared))
torch.cuda.synchronize()
marker = torch.zeros(1, device="cuda")
marker += 1
api.verify_sig(p, sig, mu, level, shared_pk=bool(shared))
marker += 1
api.keygen(seed, level)
torch.cuda.synchronize()
