#!/usr/bin/env python3
"""host-side timing of dil_sign_dev (DIL_SIGN_TRACE=1 prints the breakdown to stderr)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dilithium_amd import api
level, n = int(sys.argv[1]), int(sys.argv[2])
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
seed, mu = u8(n, 32), u8(n, 64)
pk, sk = api.keygen(seed, level)
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    api.sign(sk[:1], mu, level, shared_sk=True)
    torch.cuda.synchronize()
    print(f"call {i}: wall {1e6*(time.perf_counter()-t0):.0f} us", file=sys.stderr)
