"""GPU parity of the wire-format fused verify kernels (wire_kernels.hip): packed z / t1 / hints / SampleInBall(c~) in,
packed w1 + verdict bits out, against the oracle's int32 verify core fed with the host-decoded fields
(oracle/dilithium_kat.py codecs), at small and dispatch-size batches, distinct and shared public keys; and the whole
wire-format verification fused (option fuse_wire = 1) against the unfused kernel sequence (fuse_wire = 0) on tampered
batches.   rtl_src/decoder.v:89-143, encoder.v:96-133, usehint.v:92-114, combined_top.v:1207-1469."""
import numpy as np
import pytest

from oracle import dilithium_kat as dk
from oracle.oracle import N, Q, splitmix64_polys
from tests.test_gpu_codecs import cu, kat_wire, mus

pytestmark = pytest.mark.gpu


def synth_wire(level, n, seed, nkeys=None, zmax_items=()):
    """random (A, pk bytes, sig bytes) + the decoded fields; items in `zmax_items` get one coefficient at the norm bound"""
    p = dk.PARAMS[level]
    nk = n if nkeys is None else nkeys
    rng = np.random.default_rng(seed)
    A = splitmix64_polys(nk * p.K * p.L, seed=seed).reshape(nk, p.K, p.L, N)
    t1 = rng.integers(0, 1 << 10, (nk, p.K, N)).astype(np.int32)
    bound = p.gamma1 - p.beta
    z = rng.integers(-(bound - 1), bound, (n, p.L, N)).astype(np.int64)
    for j, i in enumerate(zmax_items):
        z[i, j % p.L, (17 * j + 5) % N] = bound if j % 2 == 0 else -bound      # exactly at the bound: rejected
    ct = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    h = np.zeros((n, p.K, N), np.uint8)
    for i in range(n):
        cnt = int(rng.integers(0, p.omega + 1))
        pos = rng.choice(p.K * N, cnt, replace=False)
        h[i].reshape(-1)[pos] = 1
    pk = np.zeros((nk, 32 + p.K * 320), np.uint8)
    pk[:, :32] = rng.integers(0, 256, (nk, 32), dtype=np.uint8)
    for i in range(nk):
        pk[i, 32:] = np.frombuffer(dk.pack_t1(p, t1[i]), dtype=np.uint8)
    zb = p.L * 32 * p.z_bits
    sig = np.zeros((n, 32 + zb + p.omega + p.K), np.uint8)
    sig[:, :32] = ct
    c = np.zeros((n, N), np.int64)
    for i in range(n):
        sig[i, 32:32 + zb] = np.frombuffer(dk.pack_z(p, z[i]), dtype=np.uint8)
        sig[i, 32 + zb:] = np.frombuffer(dk.pack_hint(p, h[i]), dtype=np.uint8)
        c[i] = dk.sample_in_ball(p, ct[i].tobytes())
    return A, pk, sig, dict(t1=t1, z=z, c=c, h=h)


def expected(oracle, level, A, f, shared):
    p = dk.PARAMS[level]
    w1 = oracle.verify_core(level, A, dk.canon(f["z"]), dk.canon(f["c"]), f["t1"], f["h"], shared_pk=shared)
    w1p = np.stack([np.frombuffer(dk.pack_w1(p, w1[i]), dtype=np.uint8) for i in range(w1.shape[0])])
    zrej = (np.abs(f["z"]).reshape(w1.shape[0], -1).max(axis=1) >= p.gamma1 - p.beta)
    return w1p, np.where(zrej, 2, 0).astype(np.int32)


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 37, 2304])
def test_verify_wire_core_distinct_vs_oracle(gpu, oracle, level, n):
    """verify_wire_wpi_kernel<LEVEL>: every packed w1 byte and the ||z|| verdict bit of every item"""
    from dilithium_amd import api
    A, pk, sig, f = synth_wire(level, n, 70 + level + n, zmax_items=[i for i in (0, 5, 36) if i < n])
    w1p, v = api.verify_wire_core(cu(gpu, A), cu(gpu, pk), cu(gpu, sig), level)
    ew1p, ev = expected(oracle, level, A, f, False)
    assert (v.cpu().numpy() == ev).all()
    assert (w1p.cpu().numpy() == ew1p).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_verify_wire_core_sample_in_ball_inside_vs_in_front(gpu, oracle, level):
    """option fuse_sib: c = SampleInBall(c~) sampled inside verify_wire_wpi_kernel (one SHAKE256 state per wavefront, the sampler working in the
    wave's z^ area of LDS) against the sampling launch in front -- 4200 items (more than the 3072 resident waves: waves re-enter their item loop), every packed w1 byte
    and verdict bit identical to each other and to the oracle; then the byte-level entry points under bit 1 of the option"""
    from dilithium_amd import api
    n = 4200
    A, pk, sig, f = synth_wire(level, n, 500 + level, zmax_items=[0, 7, 4199])
    dA, dpk, dsig = cu(gpu, A), cu(gpu, pk), cu(gpu, sig)
    saved = api.get_option("fuse_sib")
    try:
        out = {}
        for mode in (0, 1):
            api.set_option("fuse_sib", mode)
            w1p, v = api.verify_wire_core(dA, dpk, dsig, level)
            out[mode] = (w1p.cpu().numpy(), v.cpu().numpy())
        assert (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all()
    finally:
        api.set_option("fuse_sib", saved)
    ew1p, ev = expected(oracle, level, A, f, False)
    assert (out[1][1] == ev).all() and (out[1][0] == ew1p).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 53, 4099])
def test_verify_wire_core_shared_vs_oracle(gpu, oracle, level, n):
    """verify_wire_shared_kernel<LEVEL, NW>: one pk (A, t1^ LDS-resident), ragged batch"""
    from dilithium_amd import api
    A, pk, sig, f = synth_wire(level, n, 170 + level + n, nkeys=1, zmax_items=[i for i in (2, 50) if i < n])
    w1p, v = api.verify_wire_core(cu(gpu, A), cu(gpu, pk), cu(gpu, sig), level, shared_pk=True)
    ew1p, ev = expected(oracle, level, A, f, True)
    assert (v.cpu().numpy() == ev).all()
    assert (w1p.cpu().numpy() == ew1p).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_verify_wire_core_malformed_hints(gpu, level):
    """the five malformed-encoding classes of the reference decoder (usehint.v:92-114) set verdict bit 4; well-formed
    neighbours do not"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    A, pk, sig, f = synth_wire(level, 12, 900 + level, nkeys=1)
    zb = 32 + p.L * 32 * p.z_bits
    hb = np.zeros(p.omega + p.K, np.uint8)
    hb[:4] = [3, 9, 200, 7]
    hb[p.omega:] = [3] + [4] * (p.K - 1)          # rows: {3, 9, 200}, {7}
    good = hb.copy()
    cases = []
    c1 = good.copy(); c1[p.omega + 1] = 2; cases.append(c1)                   # counts not monotone
    c2 = good.copy(); c2[p.omega + p.K - 1] = p.omega + 1; cases.append(c2)   # count > omega
    c3 = good.copy(); c3[1] = 3; cases.append(c3)                             # positions not strictly increasing (equal)
    c4 = good.copy(); c4[0], c4[1] = 9, 3; cases.append(c4)                   # decreasing inside a row
    c5 = good.copy(); c5[10] = 1; cases.append(c5)                            # non-zero padding
    sig[0, zb:] = good
    for i, cse in enumerate(cases):
        sig[1 + i, zb:] = cse
    _, v = api.verify_wire_core(cu(gpu, A), cu(gpu, pk), cu(gpu, sig), level, shared_pk=True)
    v = v.cpu().numpy()
    assert (v[1:6] & 4 == 4).all() and (v[0] & 4) == 0 and (v[6:] & 4 == 0).all()
    for i in range(12):                                                        # agrees with the host decoder
        assert (dk.unpack_hint(p, sig[i, zb:].tobytes()) is None) == bool(v[i] & 4)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_w1_bytes_through_fused_kernel(gpu, level):
    """the 100 KAT signatures of a level: packed w1 from the fused kernel == the fixture's w1 (configs[3] KAT gate),
    with A expanded on the device from rho"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    k, pk, _, sig = kat_wire(level)
    A = api.expand_a(cu(gpu, k["rho"]), level)
    w1p, v = api.verify_wire_core(A, cu(gpu, pk), cu(gpu, sig), level)
    assert int(v.abs().sum()) == 0
    want = np.stack([np.frombuffer(dk.pack_w1(p, k["w1"][i]), dtype=np.uint8) for i in range(100)])
    assert (w1p.cpu().numpy() == want).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("shared", [False, True])
def test_fused_and_unfused_wire_verify_agree(gpu, level, shared, kat_msgs):
    """dil_verify_sig_dev with fuse_wire = 1 (one fused kernel) and = 0 (codec kernels + int32 core): identical verdict
    words on 2600 signatures of which every 7th is tampered (z bit, c~ bit, hint byte, rho bit, mu bit).  (A flipped LOW
    bit of a packed t1 coefficient is not in the list: about a fifth of those leave every HighBits unchanged and verify.)"""
    from dilithium_amd import api
    k, pk, sk, _ = kat_wire(level)
    rng = np.random.default_rng(level)
    n = 2600
    mu = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    if shared:
        pkd = cu(gpu, pk[:1])
        sig, _ = api.sign(cu(gpu, sk[:1]), cu(gpu, mu), level, shared_sk=True)
    else:
        idx = np.arange(n) % 100
        pkd = cu(gpu, pk[idx])
        sig, _ = api.sign(cu(gpu, sk[idx]), cu(gpu, mu), level)
    sg = sig.cpu().numpy().copy()
    pkn = pkd.cpu().numpy().copy()
    tampered = set()
    for i in range(0, n, 7):
        kind = (i // 7) % 5
        if kind == 0:
            sg[i, 32 + int(rng.integers(0, 500))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            sg[i, int(rng.integers(0, 32))] ^= 0x40
        elif kind == 2:
            sg[i, -1] = dk.PARAMS[level].omega + 3
        elif kind == 3 and not shared:
            pkn[i, int(rng.integers(0, 32))] ^= 2
        else:
            mu[i, 3] ^= 0x10
        tampered.add(i)
    out = {}
    try:
        for mode in (1, 0):
            api.set_option("fuse_wire", mode)
            out[mode] = api.verify_sig(cu(gpu, pkn), cu(gpu, sg), cu(gpu, mu), level, shared_pk=shared).cpu().numpy()
    finally:
        api.set_option("fuse_wire", 1)
    assert (out[0] == out[1]).all()
    assert set(np.nonzero(out[1])[0]) == tampered


def test_lane_per_sponge_hash_kernels_at_65536(gpu):
    """batches >= 65536 switch the hash launchers to the lane-per-sponge kernels (shake256_batch_kernel,
    challenge_hash_kernel -- the two-lane forms serve everything smaller): SHAKE256 vs hashlib on sampled items, and
    65536 level-2 signatures under one key signed (digest path) and verified (compare path), one of them tampered"""
    import hashlib
    from dilithium_amd import api
    rng = np.random.default_rng(8)
    n = 65536 + 64
    data = rng.integers(0, 256, (n, 72), dtype=np.uint8)
    out = api.shake256(cu(gpu, data), 64).cpu().numpy()
    for i in list(range(0, n, 4099)) + [n - 1]:
        assert out[i].tobytes() == hashlib.shake_256(data[i].tobytes()).digest(64)
    k, pk, sk, _ = kat_wire(2)
    mu = cu(gpu, rng.integers(0, 256, (65536, 64), dtype=np.uint8))
    sig, att = api.sign(cu(gpu, sk[:1]), mu, 2, shared_sk=True)
    assert int(att.min()) >= 1
    assert int(api.verify_sig(cu(gpu, pk[:1]), sig, mu, 2, shared_pk=True).abs().sum()) == 0
    sig[65535, 3] ^= 1
    v = api.verify_sig(cu(gpu, pk[:1]), sig, mu, 2, shared_pk=True).cpu().numpy()
    assert v[65535] == 1 and int(np.abs(v[:65535]).sum()) == 0


@pytest.mark.parametrize("level", [2, 3, 5])
def test_verify_sig_with_expanded_keys(gpu, level, kat_msgs):
    """dil_verify_sig_expanded_dev (A expanded once by the caller) == dil_verify_sig_dev: the 100 KAT signatures with a key
    per signature, tampered ones rejected with the same verdict words; and one key for a batch"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(level)
    mu = mus(k, kat_msgs)
    sg = sig.copy()
    sg[3, 100] ^= 0x10
    sg[4, 1] ^= 1
    sg[9, -2] = dk.PARAMS[level].omega + 9
    pkd, sgd, mud = cu(gpu, pk), cu(gpu, sg), cu(gpu, mu)
    A = api.expand_a(pkd[:, :32].contiguous(), level)
    v = api.verify_sig_expanded(A, pkd, sgd, mud, level).cpu().numpy()
    assert (v == api.verify_sig(pkd, sgd, mud, level).cpu().numpy()).all()
    assert set(np.nonzero(v)[0]) == {3, 4, 9}
    rng = np.random.default_rng(level)
    m = cu(gpu, rng.integers(0, 256, (2100, 64), dtype=np.uint8))
    s1, _ = api.sign(cu(gpu, sk[:1]), m, level, shared_sk=True)
    assert int(api.verify_sig_expanded(A[:1].contiguous(), pkd[:1], s1, m, level, shared_pk=True).abs().sum()) == 0
    # ... and with t1^ = NTT(t1 2^13) kept too (dil_expand_t1_dev + dil_verify_sig_expanded2_dev): the same verdict words, and t1^ itself
    # against the codec + transform entry points
    th = api.expand_t1(pkd, level)
    t1 = api.unpack(pkd, api.CODEC_T1, level, 32)
    want = (t1 << 13).contiguous()
    api.ntt(want)
    assert gpu.equal(th, want)
    v2 = api.verify_sig_expanded2(A, th, pkd, sgd, mud, level).cpu().numpy()
    assert (v2 == v).all()
    assert int(api.verify_sig_expanded2(A[:1].contiguous(), th[:1].contiguous(), pkd[:1], s1, m, level, shared_pk=True).abs().sum()) == 0
    # one key for the batch: t1^ is documented as not read -- a NULL t1hat is accepted there (and only there)
    assert int(api.verify_sig_expanded2(A[:1].contiguous(), None, pkd[:1], s1, m, level, shared_pk=True).abs().sum()) == 0
    import ctypes as C
    L = api._lib.load()
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    vd = gpu.empty((sgd.shape[0],), dtype=gpu.int32, device="cuda")
    assert L.dil_verify_sig_expanded2_dev(P(vd), P(A), None, P(pkd), P(sgd), P(mud), level, sgd.shape[0], 0, None) != 0
    # dil_expand_t1_dev validates before it computes strides: unknown level, NULL pointers
    assert L.dil_expand_t1_dev(P(th), P(pkd), 4, 1, None) != 0
    assert L.dil_expand_t1_dev(None, P(pkd), level, 1, None) != 0
    assert L.dil_expand_t1_dev(P(th), None, level, 1, None) != 0
    assert L.dil_expand_t1_dev(None, None, level, 0, None) == 0


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expanded2_on_tampered_signatures_at_dispatch_size(gpu, level):
    """dil_verify_sig_expanded2_dev == dil_verify_sig_dev on 2600 signatures under 2600 keys, every fourth one tampered in z, c~ or the hints"""
    from dilithium_amd import api
    n = 2600
    rng = np.random.default_rng(70 + level)
    seed = cu(gpu, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    mu = cu(gpu, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    pk, sk = api.keygen(seed, level)
    sig, _ = api.sign(sk, mu, level)
    bad = sig.clone()
    idx = np.arange(0, n, 4)
    pos = rng.integers(0, sig.shape[1], idx.size)
    for i, p_ in zip(idx, pos):
        bad[int(i), int(p_)] ^= 1 << int(rng.integers(0, 8))
    A, th = api.expand_a(pk[:, :32].contiguous(), level), api.expand_t1(pk, level)
    want = api.verify_sig(pk, bad, mu, level)
    assert gpu.equal(api.verify_sig_expanded2(A, th, pk, bad, mu, level), want)
    assert int((want != 0).sum()) >= idx.size * 9 // 10 and int(want[1::4].abs().sum()) == 0
