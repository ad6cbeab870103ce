#!/usr/bin/env python3
"""Randomised soak of the byte-level operations (not part of the test suite): random level, batch size, key mode and option settings;
device keygen / sign_msg / verify_msg on random seeds and ragged messages; a sample of the items is recomputed by the host KAT harness
(hashlib SHAKE + oracle arithmetic): same pk / sk / signature bytes and attempt counts; every signature verifies; randomly tampered
signatures, keys and messages are rejected and the host harness agrees.   usage: fuzz_scheme.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dilithium_amd import api
from oracle import dilithium_kat as dk
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
api.init(0)
eng = dk.OracleEngine(Oracle())
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
OPTS = {"a24": (0, 1, 2), "fuse_keygen": (0, 1), "fuse_wire": (0, 1), "aux_overlap": (0, 1), "fused_mode": (0, 1, 2), "sign_early": (0, 1),
        "sign_skip": (0, 1, 2, 3), "packed_y": (0, 1), "fuse_challenge": (0, 1), "coop_max": (0, 700, 3072, 1 << 30), "w0w1_plane": (0, 1)}
DEFAULTS = {"a24": 1, "fuse_keygen": 1, "fuse_wire": 1, "aux_overlap": 1, "fused_mode": 0, "sign_early": 1, "sign_skip": 3, "packed_y": 1,
            "fuse_challenge": 1, "coop_max": 3072, "w0w1_plane": 1}
t0 = time.time()
cases = sigs = checked = 0
while time.time() - t0 < budget:
    level = int(rng.choice([2, 3, 5]))
    p = dk.PARAMS[level]
    n = int(rng.choice([1, 2, 3, 17, 64, 100, 547, 1093, 2048, 2305])) if rng.random() < 0.6 else int(rng.integers(1, 2600))
    shared = bool(rng.integers(0, 2))
    opts = {k: int(rng.choice(v)) for k, v in OPTS.items()} if rng.random() < 0.7 else dict(DEFAULTS)
    for k, v in opts.items():
        api.set_option(k, v)
    nk = 1 if shared else n
    seeds = rng.integers(0, 256, (nk, 32), dtype=np.uint8)
    pk, sk = api.keygen(cu(seeds), level)
    msgs = [rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes() for _ in range(n)]
    blob, offs, lens = api.pack_messages(msgs)
    sig, att = api.sign_msg(sk, blob, offs, lens, level, shared_sk=shared)
    v = api.verify_msg(pk, sig, blob, offs, lens, level, shared_pk=shared).cpu().numpy()
    assert (v == 0).all(), ("verify", level, n, shared, opts)
    pkh, skh, sigh, atth = pk.cpu().numpy(), sk.cpu().numpy(), sig.cpu().numpy(), att.cpu().numpy()
    sample = sorted(set([0, n - 1] + [int(x) for x in rng.integers(0, n, 2)]))
    items = []
    for i in sample:
        ki = 0 if shared else i
        kg = dk.keygen(level, seeds[ki].tobytes(), eng)
        s1p, s2p, t0p = dk.pack_eta(p, kg["s1"]), dk.pack_eta(p, kg["s2"]), dk.pack_t0(p, kg["t0"])
        assert pkh[ki].tobytes() == kg["rho"] + kg["t1_packed"], ("pk", level, n, shared, opts, i)
        assert skh[ki].tobytes() == kg["rho"] + kg["key"] + kg["tr"] + s1p + s2p + t0p, ("sk", level, n, shared, opts, i)
        items.append(dict(rho=kg["rho"], key=kg["key"], tr=kg["tr"], s1_packed=s1p, s2_packed=s2p, t0_packed=t0p, msg=msgs[i]))
    want = dk.sign_batch(level, items, eng, max_attempts=512)
    for j, i in enumerate(sample):
        assert sigh[i].tobytes() == want[j][0] + want[j][1] + want[j][2], ("sig", level, n, shared, opts, i)
        assert atth[i] == want[j][3], ("attempts", level, n, shared, opts, i)
    # tamper: signature byte / message / hint count
    bad = sig.clone()
    tam = sorted(set(int(x) for x in rng.integers(0, n, min(n, 5))))
    zb = 32 + p.L * 32 * p.z_bits
    for t in tam:
        kind = int(rng.integers(0, 3))
        if kind == 0:
            bad[t, int(rng.integers(0, zb))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            bad[t, -1] = p.omega + 1
        else:
            bad[t, int(rng.integers(0, 32))] ^= 0x80
    vb = api.verify_msg(pk, bad, blob, offs, lens, level, shared_pk=shared).cpu().numpy()
    assert set(np.nonzero(vb)[0]) == set(tam), ("tamper", level, n, shared, opts, tam, np.nonzero(vb)[0][:10])
    cases += 1
    sigs += n
    checked += len(sample)
for k, v in DEFAULTS.items():
    api.set_option(k, v)
print(f"fuzz_scheme: {cases} random cases, {sigs} key / sign / verify triples, {checked} recomputed by the host harness byte for byte, "
      f"all tampered items rejected ({time.time() - t0:.0f} s)")
