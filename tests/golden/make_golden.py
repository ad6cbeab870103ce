#!/usr/bin/env python3
"""Regenerate tests/golden/* from the REFERENCE (dev container only: needs /root/reference
and oracle/_ref/libref.so built by `make -C oracle ref`).

What it writes (all data -- inputs and expected outputs; no reference source text):
  zetas_rom.txt        the reference's twiddle ROM image (zetas.txt, a data file)
  ntt_golden.npz       inputs + outputs of the COMPILED reference C++ (ntt, invntt,
                       ntt2x2_ref, invntt2x2_ref, pointwise_barrett, ntt2x2_fwdntt /
                       ntt2x2_invntt / ntt2x2_mul under all 3 MAPPINGs), raw (non-canonical)
  kat_{2,3,5}.npz      the reference's 100 KAT vectors per level as byte matrices
                       (rho,key,tr,ctilde,seed,s1,s2,t0,t1,z,h) + derived expectations
                       (sign attempt counts, w1 of the verify core) from the KAT harness
  kat_msgs.npz         the 100 messages (identical for all levels)
  butterfly_golden.npz inputs + raw outputs of the reference's header templates butterfly<> / buttefly_circuit<>
                       (hardware_code/butterfly_unit.h, instantiated by oracle/ref_shim.cpp) in all three OPERATION modes
                       (`python make_golden.py butterfly` writes only this file)
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.oracle import Oracle, Reference, splitmix64_polys, Q, N  # noqa: E402
from oracle import dilithium_kat as dk  # noqa: E402

REF = "/root/reference"


def butterfly_goldens(ref):
    """4-lane rows: canonical lanes and twiddles, signed ones (the C++ unit works on signed data_t), the edge residues"""
    rng = np.random.default_rng(20260929)
    edge = np.array([0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, -(Q - 1), -1], dtype=np.int32)
    rows_in = np.concatenate([rng.integers(0, Q, (1500, 4)), rng.integers(-(Q - 1), Q, (500, 4)), rng.choice(edge, (600, 4))]).astype(np.int32)
    rows_w = np.concatenate([rng.integers(0, Q, (1500, 4)), rng.integers(-(Q - 1), Q, (500, 4)), rng.choice(edge, (600, 4))]).astype(np.int32)
    out = dict(data_in=rows_in, w=rows_w)
    for mode in (0, 1, 2):
        out[f"circuit_{mode}"] = ref.buttefly_circuit(rows_in, rows_w, mode)
        out[f"butterfly_{mode}"] = np.array([ref.butterfly(mode, w[0], x[0], x[1]) for x, w in zip(rows_in, rows_w)], dtype=np.int32)
    np.savez_compressed(f"{HERE}/butterfly_golden.npz", **out)


def main():
    ref = Reference()
    butterfly_goldens(ref)
    if sys.argv[1:] == ["butterfly"]:
        return
    shutil.copyfile(f"{REF}/zetas.txt", f"{HERE}/zetas_rom.txt")

    polys = [np.arange(N), np.zeros(N), np.full(N, Q - 1), np.full(N, -(Q - 1)), np.full(N, 1)]
    for idx in (0, 1, 2, 63, 64, 127, 128, 255):
        e = np.zeros(N)
        e[idx] = 1
        polys.append(e)
    a = np.concatenate([np.array(polys, dtype=np.int32),
                        splitmix64_polys(64, seed=7),
                        splitmix64_polys(16, seed=8, lo=-(Q - 1), hi=Q)])
    b = splitmix64_polys(a.shape[0], seed=9)
    out = dict(a=a, b=b,
               ntt=ref.ntt(a), invntt=ref.invntt(a),
               ntt2x2=ref.ntt2x2_ref(a), invntt2x2=ref.invntt2x2_ref(a),
               pointwise=ref.pointwise_barrett(a, b),
               zetas_barrett=ref.zetas_barrett)
    ram = splitmix64_polys(16, seed=10)
    mul = splitmix64_polys(16, seed=11)
    out["ram"], out["mul_ram"] = ram, mul
    for m in (0, 1, 2):
        out[f"bram_fwd_{m}"] = ref.bram_fwdntt(ram, m)
        out[f"bram_inv_{m}"] = ref.bram_invntt(ram, m)
        out[f"bram_mul_{m}"] = ref.bram_mul(ram, mul, m)
    # the reference's polymul chain (ntt2x2_test.cpp:109-137): fwd,fwd,mul,inv(AFTER_NTT)
    ra = ref.bram_fwdntt(ram, 0)
    rb = ref.bram_fwdntt(mul, 0)
    rab = ref.bram_mul(ra, rb, 0)
    out["bram_polymul"] = ref.bram_invntt(rab, 1)
    np.savez_compressed(f"{HERE}/ntt_golden.npz", **out)

    eng = dk.OracleEngine(Oracle())
    msgs = None
    for level in (2, 3, 5):
        p = dk.PARAMS[level]
        kat = dk.load_kat_reference(level)
        msgs = kat["msg"]
        mat = lambda k: np.frombuffer(b"".join(kat[k]), dtype=np.uint8).reshape(100, -1)  # noqa: E731
        items = [dict(rho=kat["rho"][i], ctilde=kat["ctilde"][i], z_packed=kat["z"][i],
                      t1_packed=kat["t1"][i], h_packed=kat["h"][i], msg=kat["msg"][i]) for i in range(100)]
        ok, w1 = dk.verify_batch(level, items, eng)
        assert all(ok)
        sitems = [dict(rho=kat["rho"][i], key=kat["key"][i], tr=kat["tr"][i], s1_packed=kat["s1"][i],
                       s2_packed=kat["s2"][i], t0_packed=kat["t0"][i], msg=kat["msg"][i]) for i in range(100)]
        sigs = dk.sign_batch(level, sitems, eng)
        assert all(s[0] == kat["ctilde"][i] and s[1] == kat["z"][i] and s[2] == kat["h"][i]
                   for i, s in enumerate(sigs))
        np.savez_compressed(f"{HERE}/kat_{level}.npz",
                            **{k: mat(k) for k in ("seed", "rho", "key", "tr", "ctilde", "s1", "s2", "t0", "t1", "z", "h")},
                            attempts=np.array([s[3] for s in sigs], dtype=np.int32),
                            w1=np.stack(w1).astype(np.uint8))
        print(level, "ok")
    np.savez_compressed(f"{HERE}/kat_msgs.npz",
                        msg=np.frombuffer(b"".join(msgs), dtype=np.uint8),
                        mlen=np.array([len(m) for m in msgs], dtype=np.int32))


if __name__ == "__main__":
    main()
