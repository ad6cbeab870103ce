"""CPU: the reference's 100 KAT vectors x levels 2/3/5 through the oracle + KAT harness.

This is what pins oracle rows H8 (verify core), H9 (mat-vec, via keygen t = A s1 + s2) and
H10 (sign inner loop): the RTL cannot be simulated here, the KATs close over it.
"""
import numpy as np
import pytest

from oracle import dilithium_kat as dk
from tests.conftest import load_kat


def kat_items(level, msgs):
    k = load_kat(level)
    b = lambda name, i: k[name][i].tobytes()  # noqa: E731
    ver = [dict(rho=b("rho", i), ctilde=b("ctilde", i), z_packed=b("z", i), t1_packed=b("t1", i),
                h_packed=b("h", i), msg=msgs[i]) for i in range(100)]
    sig = [dict(rho=b("rho", i), key=b("key", i), tr=b("tr", i), s1_packed=b("s1", i), s2_packed=b("s2", i),
                t0_packed=b("t0", i), msg=msgs[i]) for i in range(100)]
    return k, ver, sig


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_verify_100(level, kat_msgs, oracle):
    k, ver, _ = kat_items(level, kat_msgs)
    ok, w1 = dk.verify_batch(level, ver, dk.OracleEngine(oracle))
    assert all(ok)
    assert (np.stack(w1) == k["w1"]).all()
    # a flipped bit in z / c~ / h must be rejected (tb_verify_top.v:244-246 prints "Rejected")
    for field, byte in (("z_packed", 11), ("ctilde", 3), ("t1_packed", 100)):
        it = dict(ver[7])
        buf = bytearray(it[field])
        buf[byte] ^= 0x10
        it[field] = bytes(buf)
        assert dk.verify_batch(level, [it], dk.OracleEngine(oracle))[0] == [False]


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_sign_100(level, kat_msgs, oracle):
    k, _, sig = kat_items(level, kat_msgs)
    out = dk.sign_batch(level, sig, dk.OracleEngine(oracle))
    for i, (ct, z, h, att) in enumerate(out):
        assert ct == k["ctilde"][i].tobytes()
        assert z == k["z"][i].tobytes()
        assert h == k["h"][i].tobytes()
        assert att == k["attempts"][i]


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_keygen_first_10(level, oracle):
    k = load_kat(level)
    p = dk.PARAMS[level]
    eng = dk.OracleEngine(oracle)
    for i in range(10):
        kg = dk.keygen(level, k["seed"][i].tobytes(), eng)
        assert kg["rho"] == k["rho"][i].tobytes() and kg["key"] == k["key"][i].tobytes()
        assert kg["tr"] == k["tr"][i].tobytes() and kg["t1_packed"] == k["t1"][i].tobytes()
        assert dk.pack_eta(p, kg["s1"]) == k["s1"][i].tobytes()
        assert dk.pack_eta(p, kg["s2"]) == k["s2"][i].tobytes()
        assert dk.pack_t0(p, kg["t0"]) == k["t0"][i].tobytes()
