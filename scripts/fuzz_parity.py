#!/usr/bin/env python3
"""Randomised differential soak (not part of the test suite): random level, batch size (1 .. 6000, biased towards the kernel-shape
boundaries), key mode and kernel shape; verify core, mat-vec and both sign phases against the oracle on every output.
usage: fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dilithium_amd import api
from oracle.oracle import Oracle
from tests.test_gpu_pipelines import KL, dev, synth
from tests.test_gpu_dispatch_parity import sign_inputs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
api.init(0)
o = Oracle()
t0 = time.time()
cases = items = 0
specials = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 2047, 2048, 2049, 2559, 2560, 2561, 2815, 2816, 2817, 3071, 3072, 4095, 4096, 4097]
while time.time() - t0 < budget:
    level = int(rng.choice([2, 3, 5]))
    K, L = KL[level]
    n = int(rng.choice(specials)) if rng.random() < 0.5 else int(rng.integers(1, 6000))
    shared = bool(rng.integers(0, 2))
    mode = int(rng.choice([0, 0, 1, 2]))
    api.set_option("fused_mode", mode)
    seed = int(rng.integers(0, 1 << 30))
    A, z, c, t1, h = synth(level, n, seed)
    k = 1 if shared else n
    w1 = api.verify_core(dev(torch, A[:k]), dev(torch, z), dev(torch, c), dev(torch, t1[:k]), dev(torch, h, np.uint8), level, shared_pk=shared).cpu().numpy()
    assert (w1 == o.verify_core(level, A[:k], z, c, t1[:k], h, shared_pk=shared)).all(), ("verify", level, n, shared, mode, seed)
    w = api.matvec(dev(torch, A[:k]), dev(torch, z), level, shared_A=shared).cpu().numpy()
    assert (w == o.matvec(K, L, A[:k], z, shared_A=shared)).all(), ("matvec", level, n, shared, mode, seed)
    if n <= 3000:
        As, ys, cs, s1h, s2h, t0h = sign_inputs(o, level, n, seed + 1, k)
        gw1, gw0 = api.sign_phase1(dev(torch, As), dev(torch, ys), level, shared_key=shared)
        ow1, ow0 = o.sign_phase1(level, As, ys)
        assert (gw1.cpu().numpy() == ow1).all() and (gw0.cpu().numpy() == ow0).all(), ("sign1", level, n, shared, mode, seed)
        zz, hh, fl = api.sign_phase2(dev(torch, cs), dev(torch, ys), dev(torch, ow0), dev(torch, ow1, np.uint8), dev(torch, s1h), dev(torch, s2h),
                                     dev(torch, t0h), level, shared_key=shared, small_key=bool(rng.integers(0, 2)))
        oz, oh, ofl = o.sign_phase2(level, cs, ys, ow0, ow1, s1h, s2h, t0h)
        assert (fl.cpu().numpy() == ofl).all() and (zz.cpu().numpy() == oz).all() and (hh.cpu().numpy() == oh).all(), ("sign2", level, n, shared, mode, seed)
    cases += 1
    items += n
api.set_option("fused_mode", 0)
print(f"fuzz_parity: {cases} random cases, {items} items, all outputs identical to the oracle ({time.time() - t0:.0f} s)")
