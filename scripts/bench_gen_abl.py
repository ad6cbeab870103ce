#!/usr/bin/env python3
"""Ablations of verify_wire_gen_kernel's phase 2 (scripts/bin/libdil256_gen_abl*.so, -DDIL_GEN_ABL=n): time of the gen wire core
per library, each in its own process.   usage: bench_gen_abl.py [level] [batch]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from dilithium_amd import api
    from scripts.bench_fused import timeit
    level, n = int(sys.argv[2]), int(sys.argv[3])
    api.init(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    pk = torch.randint(0, 256, (n, api.pk_bytes(level)), dtype=torch.uint8, device="cuda", generator=g)
    sig = torch.randint(0, 256, (n, api.sig_bytes(level)), dtype=torch.uint8, device="cuda", generator=g)
    A = api.expand_a(pk[:, :32].contiguous(), level)
    ts = [timeit(lambda: api.verify_wire_core(None, pk, sig, level), 10) for _ in range(3)]
    ta = timeit(lambda: api.verify_wire_core(A, pk, sig, level), 10)
    print(f"{os.environ.get('DIL_LIB_PATH', 'default'):40s} L{level} n={n}: gen core {min(ts)*1e3:7.1f} us (A from HBM: {ta*1e3:6.1f} us)", flush=True)
    sys.exit(0)
level = sys.argv[1] if len(sys.argv) > 1 else "3"
n = sys.argv[2] if len(sys.argv) > 2 else "8192"
libs = [None] + sorted(f for f in os.listdir(os.path.join(ROOT, "scripts", "bin")) if f.startswith("libdil256_gen"))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["DIL_LIB_PATH"] = os.path.join(ROOT, "scripts", "bin", lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", level, n], env=env)
