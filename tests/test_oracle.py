"""CPU: pin the oracle (oracle/dil_oracle.c) against the reference's golden data.

Goldens in tests/golden/ were produced by the COMPILED reference C++ (make_golden.py) and
by the reference's own data files (zetas.txt, KAT/*).  When oracle/_ref/libref.so is present
(dev container) the oracle is also compared live with the reference on fresh random inputs.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.oracle import (AFTER_INVNTT, AFTER_NTT, NATURAL, N, Oracle, Q, Reference, canon,
                           splitmix64_polys)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_zetas_equal_rom_image(oracle):
    """config 1 of BASELINE.json (plumbing): twiddle table == zetas.txt mod q"""
    rom = np.array([int(x, 16) for x in open(os.path.join(GOLDEN, "zetas_rom.txt")).read().split()])
    assert rom.shape == (256,)
    assert (canon(oracle.zetas()) == rom).all()
    assert oracle.zetas()[0] == 0
    assert np.abs(oracle.zetas()).max() <= (Q - 1) // 2


def test_zetas_equal_reference_table(oracle, golden):
    assert (oracle.zetas() == golden["zetas_barrett"]).all()


def test_survey_anchors(oracle):
    """SURVEY 8c anchors measured on the compiled reference"""
    a = np.arange(N, dtype=np.int32)
    f = oracle.ntt(a)[:8].tolist()
    i = oracle.invntt(a)[:8].tolist()
    assert f == [8023823, 4949942, 5503697, 7227518, 4077164, 903461, 2287113, 3389395]
    assert i == [4190336, 4362708, 288943, 7970686, 6724725, 6626041, 5933429, 5888005]

    def polyhash(v):
        s = 0
        for x in v.tolist():
            s = (s * 1000003 + x) % (1 << 64)
        return s
    assert polyhash(oracle.ntt(a)) == 13996679275599232392
    assert polyhash(oracle.invntt(a)) == 15944396359158734074
    assert oracle.ntt(a, canon=False)[:3].tolist() == [-356594, -3430475, -2876720]
    assert (oracle.invntt(oracle.ntt(a)) == a).all()


@pytest.mark.parametrize("name", ["ntt", "invntt", "ntt2x2", "invntt2x2"])
def test_transforms_bit_exact_vs_reference_golden(oracle, golden, name):
    got = getattr(oracle, name)(golden["a"], canon=False)
    assert (got == golden[name]).all()          # raw, not just congruent


def test_pointwise_bit_exact(oracle, golden):
    assert (oracle.pointwise(golden["a"], golden["b"], canon=False) == golden["pointwise"]).all()


def test_2x2_congruent_to_plain(oracle):
    """the reference's own test (ref_test_ntt_ntt2x2.cpp:51-90): congruence mod q"""
    a = splitmix64_polys(3000, seed=123)
    assert (oracle.ntt2x2(a) == oracle.ntt(a)).all()
    assert (oracle.invntt2x2(a) == oracle.invntt(a)).all()


@pytest.mark.parametrize("mapping", [NATURAL, AFTER_NTT, AFTER_INVNTT])
def test_bram_ops_vs_reference_golden(oracle, golden, mapping):
    ram, mul = golden["ram"], golden["mul_ram"]
    assert (oracle.bram_fwdntt(ram, mapping) == canon(golden[f"bram_fwd_{mapping}"])).all()
    assert (oracle.bram_invntt(ram, mapping) == canon(golden[f"bram_inv_{mapping}"])).all()
    assert (oracle.bram_mul(ram, mul, mapping) == canon(golden[f"bram_mul_{mapping}"])).all()


def test_bram_polymul_chain(oracle, golden):
    """ntt2x2_test.cpp:109-137: fwd, fwd, mul, inv(AFTER_NTT) -> NATURAL == plain polymul"""
    ram, mul = golden["ram"], golden["mul_ram"]
    ra = oracle.bram_fwdntt(ram, NATURAL)
    rb = oracle.bram_fwdntt(mul, NATURAL)
    got = oracle.bram_invntt(oracle.bram_mul(ra, rb, NATURAL), AFTER_NTT)
    assert (got == canon(golden["bram_polymul"])).all()
    plain = oracle.invntt(oracle.pointwise(oracle.ntt(ram), oracle.ntt(mul)))
    assert (got == plain).all()


def test_rtl_barrett_and_decompose(oracle):
    L = oracle.lib
    L.orc_check_barrett.restype = C.c_long
    L.orc_check_barrett.argtypes = [C.c_uint64, C.c_long]
    L.orc_check_decompose_full.restype = C.c_long
    assert L.orc_check_barrett(7, 1_000_000) == 0
    for level in (2, 3, 5):
        assert L.orc_check_decompose_full(level) == 0


def test_rtl_butterfly_modes(oracle):
    """butterfly.v op set on canonical residues vs plain modular arithmetic"""
    rng = np.random.default_rng(5)
    bj, bl = C.c_uint32(), C.c_uint32()
    for _ in range(2000):
        a, b, z, acc = (int(x) for x in rng.integers(0, Q, 4))
        oracle.lib.orc_butterfly_rtl(0, a, b, z, 0, C.byref(bj), C.byref(bl))
        assert (bj.value, bl.value) == ((a + b * z) % Q, (a - b * z) % Q)
        oracle.lib.orc_butterfly_rtl(1, a, b, z, 0, C.byref(bj), C.byref(bl))
        inv2 = (Q + 1) // 2
        assert (bj.value, bl.value) == ((a + b) * inv2 % Q, (a - b) * (Q - z) * inv2 % Q)
        oracle.lib.orc_butterfly_rtl(2, a, b, 0, acc, C.byref(bj), C.byref(bl))
        assert bl.value == (acc + a * b) % Q
        oracle.lib.orc_butterfly_rtl(3, a, b, 0, 0, C.byref(bj), C.byref(bl))
        assert bl.value == (a + b) % Q
        oracle.lib.orc_butterfly_rtl(4, a, b, 0, 0, C.byref(bj), C.byref(bl))
        assert bl.value == (a - b) % Q


def test_butterfly_circuit_vs_reference_unit_goldens(oracle):
    """the 2x2 unit: orc_butterfly_circuit (two RTL butterflies, lane exchange, two RTL butterflies; MUL mode: the a / c lanes switched onto
    the second pair of multipliers and back) against raw outputs of the reference's OWN header templates buttefly_circuit<data2_t, data_t> and
    butterfly<data2_t, data_t> (hardware_code/butterfly_unit.h:29-196, instantiated from where they lie by oracle/ref_shim.cpp;
    tests/golden/make_golden.py butterfly) in all three OPERATION modes, on canonical, signed and edge lanes / twiddles.  The C++ unit works on
    signed data_t with C's %, so its raw outputs are compared mod q."""
    g = np.load(os.path.join(GOLDEN, "butterfly_golden.npz"))
    x, w = g["data_in"], g["w"]
    assert x.shape == (2600, 4)
    for mode in (0, 1, 2):
        assert (oracle.butterfly_circuit(x, w, mode) == canon(g[f"circuit_{mode}"])).all(), mode
    # MUL mode is four products under the lane shuffle of butterfly_unit.h:153-187: out = {w2 a, w0 b, w3 c, w1 d}
    prod = lambda a, b: np.mod(a.astype(np.int64) * b, Q).astype(np.int32)  # noqa: E731
    want = np.stack([prod(x[:, 0], w[:, 2]), prod(x[:, 1], w[:, 0]), prod(x[:, 2], w[:, 3]), prod(x[:, 3], w[:, 1])], axis=1)
    assert (canon(g["circuit_2"]) == want).all()
    # one butterfly<> of the C++ unit == the RTL butterfly of the oracle (butterfly.v), forward and inverse, mod q
    bj, bl = C.c_uint32(), C.c_uint32()
    for mode in (0, 1):
        for (a, b, _, _), (z, _, _, _), (rj, rl) in zip(canon(x)[:600], canon(w)[:600], canon(g[f"butterfly_{mode}"])[:600]):
            oracle.lib.orc_butterfly_rtl(mode, int(a), int(b), int(z), 0, C.byref(bj), C.byref(bl))
            assert (bj.value, bl.value) == (int(rj), int(rl)), (mode, a, b, z)


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_butterfly_circuit_vs_reference_unit_live(oracle):
    r = Reference()
    rng = np.random.default_rng(77)
    x = rng.integers(-(Q - 1), Q, (3000, 4)).astype(np.int32)
    w = rng.integers(-(Q - 1), Q, (3000, 4)).astype(np.int32)
    for mode in (0, 1, 2):
        assert (oracle.butterfly_circuit(x, w, mode) == canon(r.buttefly_circuit(x, w, mode))).all(), mode


def test_twiddle_resolver_schedule(oracle):
    """twiddle_resolver.v addresses == the k-indices of ref_ntt2x2.cpp for every step"""
    out = (C.c_uint * 4)()
    for s in range(4):                        # forward: l = 8 - 2s
        l = 8 - 2 * s
        for m, i in enumerate(range(0, 256, 1 << l)):
            oracle.lib.orc_twiddle_addrs(0, s, m, out)
            k1 = (256 + i) >> l
            assert list(out) == [k1, k1, 2 * k1, 2 * k1 + 1]
    for s in range(4):                        # inverse: l = 2s
        l = 2 * s
        for m, i in enumerate(range(0, 256, 1 << (l + 2))):
            oracle.lib.orc_twiddle_addrs(1, s, m, out)
            ka = ((256 - i // 2) >> l) - 1
            kb = ((256 - i // 2) >> (l + 1)) - 1
            assert list(out) == [ka, ka - 1, kb, kb]


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_compiled_reference_live():
    o, r = Oracle(), Reference()
    a = splitmix64_polys(4000, seed=2024)
    s = splitmix64_polys(500, seed=2025, lo=-(Q - 1), hi=Q)
    b = splitmix64_polys(4000, seed=2026)
    assert (o.ntt(a, canon=False) == r.ntt(a)).all()
    assert (o.invntt(a, canon=False) == r.invntt(a)).all()
    assert (o.ntt(s, canon=False) == r.ntt(s)).all()
    assert (o.invntt(s, canon=False) == r.invntt(s)).all()
    assert (o.ntt2x2(a, canon=False) == r.ntt2x2_ref(a)).all()
    assert (o.invntt2x2(a, canon=False) == r.invntt2x2_ref(a)).all()
    assert (o.pointwise(a, b, canon=False) == r.pointwise_barrett(a, b)).all()
    for m in (0, 1, 2):
        assert (o.bram_fwdntt(a[:300], m) == canon(r.bram_fwdntt(a[:300], m))).all()
        assert (o.bram_invntt(a[:300], m) == canon(r.bram_invntt(a[:300], m))).all()
        assert (o.bram_mul(a[:300], b[:300], m) == canon(r.bram_mul(a[:300], b[:300], m))).all()
