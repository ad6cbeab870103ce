"""GPU parity for SURVEY 8(f) rows N2 (bit-packed codecs on the device) and N4 (key generation):
against the KAT harness's host codecs (oracle/dilithium_kat.py) on random data, on the reference's
KAT byte strings, and end to end: seed -> (pk, sk) byte-identical to the KAT files, wire-format
signatures verified from bytes."""
import hashlib

import numpy as np
import pytest

from oracle import dilithium_kat as dk
from tests.conftest import load_kat

pytestmark = pytest.mark.gpu


def cu(torch, a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def kat_wire(level):
    k = load_kat(level)
    pk = np.concatenate([k["rho"], k["t1"]], axis=1)
    sk = np.concatenate([k["rho"], k["key"], k["tr"], k["s1"], k["s2"], k["t0"]], axis=1)
    sig = np.concatenate([k["ctilde"], k["z"], k["h"]], axis=1)
    return k, pk, sk, sig


def mus(k, msgs):
    return np.stack([np.frombuffer(hashlib.shake_256(k["tr"][i].tobytes() + msgs[i]).digest(64), dtype=np.uint8)
                     for i in range(len(msgs))])


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sizes(gpu, level):
    from dilithium_amd import api
    assert (api.pk_bytes(level), api.sk_bytes(level), api.sig_bytes(level)) == \
        {2: (1312, 2528, 2420), 3: (1952, 4000, 3293), 5: (2592, 4864, 4595)}[level]


@pytest.mark.parametrize("level", [2, 3, 5])
def test_unpack_kat_fields(gpu, level):
    """every packed KAT field decodes as the host codec does, at odd offsets inside pk / sk / sig"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    k, pk, sk, sig = kat_wire(level)
    sb = 32 * p.eta_bits
    cases = [(pk, api.CODEC_T1, 32, lambda i: dk.unpack_t1(p, k["t1"][i].tobytes())),
             (sk, api.CODEC_S1, 96, lambda i: dk.unpack_eta(p, k["s1"][i].tobytes(), p.L)),
             (sk, api.CODEC_S2, 96 + p.L * sb, lambda i: dk.unpack_eta(p, k["s2"][i].tobytes(), p.K)),
             (sk, api.CODEC_T0, 96 + (p.L + p.K) * sb, lambda i: dk.unpack_t0(p, k["t0"][i].tobytes())),
             (sig, api.CODEC_Z, 32, lambda i: dk.unpack_z(p, k["z"][i].tobytes()))]
    for buf, kind, off, want in cases:
        got = api.unpack(cu(gpu, buf), kind, level, off).cpu().numpy()
        assert got.min() >= 0 and got.max() < dk.Q
        for i in range(100):
            assert (got[i] == dk.canon(want(i))).all(), (kind, i)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_pack_roundtrip_and_host_codec(gpu, level):
    """random in-range values, canonical AND centred representatives: device pack == host pack; unpack(pack(x)) == x"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(100 + level)
    n = 33
    specs = [(api.CODEC_T1, p.K, 0, 1023, dk.pack_t1, 320),
             (api.CODEC_T0, p.K, -(1 << 12) + 1, 1 << 12, dk.pack_t0, 416),
             (api.CODEC_S1, p.L, -p.eta, p.eta, dk.pack_eta, 32 * p.eta_bits),
             (api.CODEC_S2, p.K, -p.eta, p.eta, dk.pack_eta, 32 * p.eta_bits),
             (api.CODEC_Z, p.L, -p.gamma1 + 1, p.gamma1, dk.pack_z, 32 * p.z_bits)]
    for kind, polys, lo, hi, host_pack, pb in specs:
        x = rng.integers(lo, hi + 1, (n, polys, 256)).astype(np.int32)
        x[0, 0, :4] = [lo, hi, lo, hi]
        stride, off = polys * pb + 11, 5           # odd stride and offset on purpose
        for rep in (x, dk.canon(x).astype(np.int32)):
            buf = gpu.full((n, stride), 0xA5, dtype=gpu.uint8, device="cuda")
            api.pack(cu(gpu, rep), buf, kind, level, off)
            b = buf.cpu().numpy()
            assert (b[:, :off] == 0xA5).all() and (b[:, off + polys * pb:] == 0xA5).all()   # neighbours untouched
            for i in range(n):
                assert b[i, off:off + polys * pb].tobytes() == host_pack(p, x[i]), (kind, i)
            back = api.unpack(buf, kind, level, off).cpu().numpy()
            assert (back == dk.canon(x)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_hint_codec(gpu, level):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    k, _, _, sig = kat_wire(level)
    hoff = 32 + p.L * 32 * p.z_bits
    h, bad = api.hint_unpack(cu(gpu, sig), level, hoff)
    h, bad = h.cpu().numpy(), bad.cpu().numpy()
    assert (bad == 0).all()
    for i in range(100):
        assert (h[i] == dk.unpack_hint(p, k["h"][i].tobytes())).all()
    # pack is the inverse on well-formed hints
    out = gpu.zeros((100, p.omega + p.K + 3), dtype=gpu.uint8, device="cuda")
    api.hint_pack(cu(gpu, h), out, level, 3)
    assert (out.cpu().numpy()[:, 3:] == k["h"]).all()
    # random hint sets incl. empty and exactly-omega
    rng = np.random.default_rng(level)
    hs = np.zeros((64, p.K, 256), dtype=np.uint8)
    for i in range(64):
        cnt = [0, p.omega, 1][i] if i < 3 else int(rng.integers(0, p.omega + 1))
        idx = rng.choice(p.K * 256, cnt, replace=False)
        hs[i].reshape(-1)[idx] = 1
    out = gpu.zeros((64, p.omega + p.K), dtype=gpu.uint8, device="cuda")
    api.hint_pack(cu(gpu, hs), out, level)
    o = out.cpu().numpy()
    for i in range(64):
        assert o[i].tobytes() == dk.pack_hint(p, hs[i])
    h2, bad2 = api.hint_unpack(out, level)
    assert (bad2.cpu().numpy() == 0).all() and (h2.cpu().numpy() == hs).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_hint_malformed(gpu, level):
    """every way the reference's decoder can reject a hint string (decoder.v hint checks; same cases as the host codec)"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    k = load_kat(level)
    base = k["h"][:40].copy()
    muts = []
    for i in range(40):
        b = base[i].copy()
        cnts = b[p.omega:]
        total = int(cnts[-1])
        kind = i % 5
        if kind == 0 and total >= 2:       # positions not increasing inside a row
            row_end = next(int(c) for c in cnts if c >= 2)
            b[row_end - 1], b[row_end - 2] = b[row_end - 2], b[row_end - 1]
            if b[row_end - 1] == b[row_end - 2]:
                kind = -1
        elif kind == 1:                    # counts decreasing
            b[p.omega] = int(cnts[1]) + 1
        elif kind == 2:                    # count > omega
            b[p.omega + p.K - 1] = p.omega + 1
        elif kind == 3 and total < p.omega:  # non-zero padding
            b[total] = 7
        elif kind == 4 and total >= 2:     # duplicate position inside a row
            row_end = next(int(c) for c in cnts if c >= 2)
            b[row_end - 1] = b[row_end - 2]
        muts.append(b)
    muts = np.stack(muts)
    want_bad = np.array([dk.unpack_hint(p, m.tobytes()) is None for m in muts])
    assert want_bad.sum() >= 20
    h, bad = api.hint_unpack(cu(gpu, muts), level)
    assert ((bad.cpu().numpy() != 0) == want_bad).all()
    good = ~want_bad
    hh = h.cpu().numpy()
    for i in np.nonzero(good)[0]:
        assert (hh[i] == dk.unpack_hint(p, muts[i].tobytes())).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expand_s(gpu, level):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(40 + level)
    n = 50
    rp = rng.integers(0, 256, (n, 64 + 9), dtype=np.uint8)      # stride 73: unaligned rows
    s1, s2 = api.expand_s(cu(gpu, rp), level)
    s1, s2 = s1.cpu().numpy(), s2.cpu().numpy()
    for i in range(n):
        seed = rp[i, :64].tobytes()
        for j in range(p.L):
            assert (s1[i, j] == dk.canon(dk.expand_s_poly(p, seed, j))).all()
        for j in range(p.K):
            assert (s2[i, j] == dk.canon(dk.expand_s_poly(p, seed, p.L + j))).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expand_s_large_ragged_batch(gpu, level):
    """expand_s_fast_kernel<eta> (raw nibbles to LDS byte rows, transposed out) on a batch whose last wave is ragged:
    sampled items -- first, wave boundaries, last -- against the host sampler; unaligned rho' rows; every coefficient in
    [-eta, eta]; and the whole batch equal to the same rows expanded in small pieces"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(140 + level)
    n = 16384 // (p.L + p.K) + 131
    rp = rng.integers(0, 256, (n, 64 + 5), dtype=np.uint8)
    s1, s2 = api.expand_s(cu(gpu, rp), level)
    s1, s2 = s1.cpu().numpy(), s2.cpu().numpy()
    assert s1.min() >= 0 and s1.max() < dk.Q and s2.min() >= 0 and s2.max() < dk.Q
    for i in [0, 1, 63, 64, 777, n - 2, n - 1]:
        seed = rp[i, :64].tobytes()
        for j in range(p.L):
            assert (s1[i, j] == dk.canon(dk.expand_s_poly(p, seed, j))).all()
        for j in range(p.K):
            assert (s2[i, j] == dk.canon(dk.expand_s_poly(p, seed, p.L + j))).all()
    c1 = np.where(s1 > dk.Q // 2, s1 - dk.Q, s1)
    assert np.abs(c1).max() <= p.eta
    for lo in range(0, n, 401):                                   # pieces of 401 keys: other wave boundaries, same rows
        a1, a2 = api.expand_s(cu(gpu, rp[lo:lo + 401]), level)
        assert (a1.cpu().numpy() == s1[lo:lo + 401]).all() and (a2.cpu().numpy() == s2[lo:lo + 401]).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_keygen_kat(gpu, level):
    """seed -> pk, sk byte-identical to the reference's KAT files (PQCsignKAT_Dilithium{2,3,5}.rsp fields)"""
    from dilithium_amd import api
    k, pk, sk, _ = kat_wire(level)
    gpk, gsk = api.keygen(cu(gpu, k["seed"]), level)
    assert (gpk.cpu().numpy() == pk).all()
    assert (gsk.cpu().numpy() == sk).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_keygen_large_batch_consistency(gpu, level, oracle):
    """3000 random seeds (keygen's mat-vec runs the wave-per-item kernel at this size): t1*2^13 + t0 == A s1 + s2 with
    the mat-vec recomputed by the ORACLE (not by the kernel under test) for every key"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(level)
    n = 3000
    seed = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = api.keygen(cu(gpu, seed), level)
    sb = 32 * p.eta_bits
    t1 = api.unpack(pk, api.CODEC_T1, level, 32).long()
    t0 = api.unpack(sk, api.CODEC_T0, level, 96 + (p.L + p.K) * sb).long()
    s1 = api.unpack(sk, api.CODEC_S1, level, 96)
    s2 = api.unpack(sk, api.CODEC_S2, level, 96 + p.L * sb).long()
    A = api.expand_a(pk[:, :32].contiguous(), level)
    w = gpu.from_numpy(oracle.matvec(p.K, p.L, A.cpu().numpy(), s1.cpu().numpy())).cuda().long()
    assert bool((((t1 << 13) + t0 - w - s2) % dk.Q == 0).all())
    assert (pk[:, :32] == sk[:, :32]).all()
    # spot check against the host keygen
    from oracle.oracle import Oracle
    eng = dk.OracleEngine(Oracle())
    for i in (0, n - 1):
        kg = dk.keygen(level, seed[i].tobytes(), eng)
        assert pk[i].cpu().numpy().tobytes() == kg["rho"] + kg["t1_packed"]
        assert sk[i, 32:96].cpu().numpy().tobytes() == kg["key"] + kg["tr"]


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("shared", [False, True])
def test_verify_sig_wire_kat(gpu, level, shared, kat_msgs):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    k, pk, _, sig = kat_wire(level)
    mu = mus(k, kat_msgs)
    if shared:
        # one key, many signatures: sign 24 messages under KAT key 0 on the host oracle path? -- use KAT 0 only, replicated
        pkd = cu(gpu, pk[:1])
        sg = np.repeat(sig[:1], 24, axis=0)
        m = np.repeat(mu[:1], 24, axis=0)
        sg[5, 40] ^= 1            # z bit
        sg[6, 2] ^= 0x80          # c~ bit
        m[7, 0] ^= 1              # other message
        sg[8, -1] = p.omega + 1   # malformed hint
        v = api.verify_sig(pkd, cu(gpu, sg), cu(gpu, m), level, shared_pk=True).cpu().numpy()
        bad = {5, 6, 7, 8}
        assert all((v[i] != 0) == (i in bad) for i in range(24))
        assert v[8] & 4
    else:
        v = api.verify_sig(cu(gpu, pk), cu(gpu, sig), cu(gpu, mu), level).cpu().numpy()
        assert (v == 0).all()
        sg = sig.copy()
        sg[3, 100] ^= 0x10
        sg[4, 1] ^= 1
        sg[9, -2] = p.omega + 9
        pk2 = pk.copy()
        pk2[11, 200] ^= 4
        v = api.verify_sig(cu(gpu, pk2), cu(gpu, sg), cu(gpu, mu), level).cpu().numpy()
        assert set(np.nonzero(v)[0]) == {3, 4, 9, 11}
        assert v[9] & 4


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sign_wire_kat(gpu, level, kat_msgs):
    """sk bytes + mu -> signature bytes identical to the KAT files, attempt counts as the host harness"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(level)
    mu = mus(k, kat_msgs)
    got, att = api.sign(cu(gpu, sk), cu(gpu, mu), level)
    assert (att.cpu().numpy() == k["attempts"]).all()
    assert (got.cpu().numpy() == sig).all()
    # and they verify from bytes
    assert (api.verify_sig(cu(gpu, pk), got, cu(gpu, mu), level).cpu().numpy() == 0).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sign_shared_key_many_messages(gpu, level, kat_msgs):
    """one key, 2000 messages (the signing-server shape): every signature verifies; item 0 is the KAT signature;
    the same messages signed as a distinct-key batch give identical bytes"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(level)
    rng = np.random.default_rng(level)
    n = 2000
    mu = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    mu[0] = mus(k, kat_msgs[:1])[0]
    skd, pkd, mud = cu(gpu, sk[:1]), cu(gpu, pk[:1]), cu(gpu, mu)
    got, att = api.sign(skd, mud, level, shared_sk=True)
    a = att.cpu().numpy()
    assert a.min() >= 1 and a[0] == k["attempts"][0]
    assert (got[0].cpu().numpy() == sig[0]).all()
    assert (api.verify_sig(pkd, got, mud, level, shared_pk=True).cpu().numpy() == 0).all()
    # expected number of attempts of the scheme: 4.25 / 5.1 / 3.85 (round-3 spec, table 2); loose statistical check
    assert 3.0 < a.mean() < 6.5
    m = 300
    got2, att2 = api.sign(cu(gpu, np.repeat(sk[:1], m, axis=0)), mud[:m].contiguous(), level, shared_sk=False)
    assert (got2 == got[:m]).all() and (att2 == att[:m]).all()


def test_sign_with_mu_not_16_byte_aligned(gpu, kat_msgs):
    """mu only 8-byte aligned: the signing loop's fused round-setup kernel (16-byte gathers) steps aside for the generic
    gather / kappa kernels -- same signatures"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(3)
    mu = mus(k, kat_msgs)
    n = 2500
    rng = np.random.default_rng(12)
    big = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    big[:100] = mu
    buf = gpu.zeros(n * 64 + 8, dtype=gpu.uint8, device="cuda")
    off = buf[8:].view(n, 64)
    off.copy_(cu(gpu, big))
    assert off.data_ptr() % 16 == 8
    skd = cu(gpu, sk[:1])
    got, att = api.sign(skd, off, 3, shared_sk=True)
    ref, att2 = api.sign(skd, cu(gpu, big), 3, shared_sk=True)
    assert gpu.equal(got, ref) and gpu.equal(att, att2)
    assert got[0].cpu().numpy().tobytes() == sig[0].tobytes()


def test_sign_unfinished(gpu, kat_msgs):
    from dilithium_amd import api, lib
    k, pk, sk, sig = kat_wire(3)
    mu = mus(k, kat_msgs)
    with pytest.raises(lib.DilError):
        api.sign(cu(gpu, sk), cu(gpu, mu), 3, max_attempts=1)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sign_single_and_small_batches(gpu, level, kat_msgs):
    """batch 1 / 3 / 17: the speculative (wide) rounds must pick the same first-accepted attempt as the sequential loop"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(level)
    mu = mus(k, kat_msgs)
    worst = int(np.argmax(k["attempts"]))
    for lo, hi in ((0, 1), (worst, worst + 1), (5, 8), (20, 37)):
        got, att = api.sign(cu(gpu, sk[lo:hi]), cu(gpu, mu[lo:hi]), level)
        assert (att.cpu().numpy() == k["attempts"][lo:hi]).all()
        assert (got.cpu().numpy() == sig[lo:hi]).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_host_buffer_scheme_kat(gpu, level, kat_msgs):
    """the host-pointer forms (numpy in, numpy out): the three operations of the reference's test benches on KAT bytes"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(level)
    mu = mus(k, kat_msgs)
    gpk, gsk = api.keygen_host(np.ascontiguousarray(k["seed"]), level)
    assert (gpk == pk).all() and (gsk == sk).all()
    gsig, att = api.sign_host(np.ascontiguousarray(sk), mu, level)
    assert (gsig == sig).all() and (att == k["attempts"]).all()
    v = api.verify_sig_host(np.ascontiguousarray(pk), gsig, mu, level)
    assert (v == 0).all()
    bad = gsig.copy()
    bad[17, 50] ^= 2
    assert list(np.nonzero(api.verify_sig_host(np.ascontiguousarray(pk), bad, mu, level))[0]) == [17]
    # one key, several messages
    s1, a1 = api.sign_host(np.ascontiguousarray(sk[:1]), mu[:9], level, shared_sk=True)
    assert (api.verify_sig_host(np.ascontiguousarray(pk[:1]), s1, mu[:9], level, shared_pk=True) == 0).all()
    assert (s1[0] == sig[0]).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_random_keys_and_messages_vs_host_harness(gpu, level):
    """beyond the KAT files: 24 fresh seeds / messages, device keygen + sign byte-identical to the host KAT harness
    (hashlib SHAKE + oracle polynomial arithmetic), and the signatures verify"""
    from dilithium_amd import api
    from oracle.oracle import Oracle
    p = dk.PARAMS[level]
    eng = dk.OracleEngine(Oracle())
    rng = np.random.default_rng(1000 + level)
    n = 24
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = [rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8).tobytes() for _ in range(n)]
    pk, sk = api.keygen(cu(gpu, seeds), level)
    pkh, skh = pk.cpu().numpy(), sk.cpu().numpy()
    items, mu = [], []
    for i in range(n):
        kg = dk.keygen(level, seeds[i].tobytes(), eng)
        s1p, s2p, t0p = dk.pack_eta(p, kg["s1"]), dk.pack_eta(p, kg["s2"]), dk.pack_t0(p, kg["t0"])
        assert pkh[i].tobytes() == kg["rho"] + kg["t1_packed"]
        assert skh[i].tobytes() == kg["rho"] + kg["key"] + kg["tr"] + s1p + s2p + t0p
        items.append(dict(rho=kg["rho"], key=kg["key"], tr=kg["tr"], s1_packed=s1p, s2_packed=s2p, t0_packed=t0p, msg=msgs[i]))
        mu.append(np.frombuffer(hashlib.shake_256(kg["tr"] + msgs[i]).digest(64), dtype=np.uint8))
    mu = np.stack(mu)
    want = dk.sign_batch(level, items, eng)
    sig, att = api.sign(sk, cu(gpu, mu), level)
    sigh, atth = sig.cpu().numpy(), att.cpu().numpy()
    for i in range(n):
        assert sigh[i].tobytes() == want[i][0] + want[i][1] + want[i][2], i
        assert atth[i] == want[i][3]
    assert (api.verify_sig(pk, sig, cu(gpu, mu), level).cpu().numpy() == 0).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_hardest_items_of_a_dispatch_size_batch_vs_host_harness(gpu, level):
    """2500 messages under one key go through every shape of the signing loop (wide first rounds on the lane-per-sponge
    samplers, narrow last rounds with 64 speculative attempts per item on the two-lane ones).  A signature that merely
    VERIFIES proves little -- any y gives one -- so the items that needed the most attempts (the ones the late rounds
    produced) and a few others are re-signed by the host harness: same bytes, same attempt count; and a second device run
    is identical to the first"""
    from dilithium_amd import api
    from oracle.oracle import Oracle
    p = dk.PARAMS[level]
    eng = dk.OracleEngine(Oracle())
    k, pk, sk, _ = kat_wire(level)
    rng = np.random.default_rng(7000 + level)
    n = 2500
    msgs = [rng.integers(0, 256, 48, dtype=np.uint8).tobytes() for _ in range(n)]
    tr = k["tr"][0].tobytes()
    mu = np.stack([np.frombuffer(hashlib.shake_256(tr + m).digest(64), dtype=np.uint8) for m in msgs])
    skd, mud = cu(gpu, sk[:1]), cu(gpu, mu)
    sig, att = api.sign(skd, mud, level, shared_sk=True)
    sig2, att2 = api.sign(skd, mud, level, shared_sk=True)
    assert gpu.equal(sig, sig2) and gpu.equal(att, att2)
    sigh, atth = sig.cpu().numpy(), att.cpu().numpy()
    hard = list(np.argsort(atth)[-5:]) + [0, 1, n - 1]
    assert atth[hard[4]] >= 20
    key = dict(rho=k["rho"][0].tobytes(), key=k["key"][0].tobytes(), tr=tr, s1_packed=k["s1"][0].tobytes(),
               s2_packed=k["s2"][0].tobytes(), t0_packed=k["t0"][0].tobytes())
    want = dk.sign_batch(level, [dict(key, msg=msgs[i]) for i in hard], eng, max_attempts=256)
    for j, i in enumerate(hard):
        assert atth[i] == want[j][3], (i, atth[i], want[j][3])
        assert sigh[i].tobytes() == want[j][0] + want[j][1] + want[j][2], i


def test_scheme_entry_points_edge_cases(gpu):
    from dilithium_amd import api, lib
    L = lib.load()
    # empty batches are no-ops
    e8 = lambda c: gpu.empty((0, c), dtype=gpu.uint8, device="cuda")   # noqa: E731
    pk, sk = api.keygen(e8(32), 3)
    assert pk.shape == (0, 1952) and sk.shape == (0, 4000)
    sig, att = api.sign(e8(4000), e8(64), 3)
    assert sig.shape == (0, 3293)
    assert api.verify_sig(e8(1952), e8(3293), e8(64), 3).numel() == 0
    # unknown level -> error code, not a crash
    seed = gpu.zeros((2, 32), dtype=gpu.uint8, device="cuda")
    assert L.dil_pk_bytes(4) == 0 and L.dil_sk_bytes(1) == 0 and L.dil_sig_bytes(7) == 0
    out = gpu.zeros((2, 8000), dtype=gpu.uint8, device="cuda")
    assert L.dil_keygen_dev(out.data_ptr(), out.data_ptr(), seed.data_ptr(), 4, 2, None) != 0
    assert L.dil_unpack_dev(out.data_ptr(), out.data_ptr(), 100, 0, 9, 3, 1, None) != 0      # unknown codec kind
    # a public key that is not 8-byte aligned is refused (rho is read as 64-bit words)
    k = load_kat(3)
    pkb = np.concatenate([k["rho"], k["t1"]], axis=1)[:1]
    buf = gpu.zeros(pkb.shape[1] + 8, dtype=gpu.uint8, device="cuda")
    buf[1:1 + pkb.shape[1]] = cu(gpu, pkb[0])
    sg = cu(gpu, np.concatenate([k["ctilde"], k["z"], k["h"]], axis=1)[:1])
    mu0 = gpu.zeros((1, 64), dtype=gpu.uint8, device="cuda")
    verdict = gpu.zeros(1, dtype=gpu.int32, device="cuda")
    assert L.dil_verify_sig_dev(verdict.data_ptr(), buf.data_ptr() + 1, sg.data_ptr(), mu0.data_ptr(), 3, 1, 0, None) != 0
    # the one-launch challenge: empty batch is a no-op, unknown level and a misaligned mu are refused
    ct = gpu.zeros((2, 32), dtype=gpu.uint8, device="cuda")
    cc = gpu.zeros((2, 256), dtype=gpu.int32, device="cuda")
    w1p = gpu.zeros((2, 6 * 128 + 8), dtype=gpu.uint8, device="cuda")
    mu2 = gpu.zeros((2 * 64 + 8,), dtype=gpu.uint8, device="cuda")
    assert L.dil_challenge_dev(ct.data_ptr(), cc.data_ptr(), mu2.data_ptr(), w1p.data_ptr(), 3, 0, None) == 0
    assert L.dil_challenge_dev(ct.data_ptr(), cc.data_ptr(), mu2.data_ptr(), w1p.data_ptr(), 4, 2, None) != 0
    assert L.dil_challenge_dev(ct.data_ptr(), cc.data_ptr(), mu2.data_ptr() + 4, w1p.data_ptr(), 3, 2, None) != 0


def test_concurrent_calls_on_two_streams(gpu, kat_msgs):
    """two host threads, each on its own stream, sign and verify at the same time: per-stream scratch arenas and the
    shared helper stream must not let the calls disturb each other"""
    import threading
    from dilithium_amd import api
    level = 3
    k, pk, sk, sig = kat_wire(level)
    mu = mus(k, kat_msgs)
    skd, pkd, mud = cu(gpu, sk), cu(gpu, pk), cu(gpu, mu)
    gpu.cuda.synchronize()
    results, errors = {}, []

    def worker(name, lo, hi):
        try:
            st = gpu.cuda.Stream()
            with gpu.cuda.stream(st):
                for _ in range(6):
                    s, a = api.sign(skd[lo:hi].contiguous(), mud[lo:hi].contiguous(), level)
                    v = api.verify_sig(pkd[lo:hi].contiguous(), s, mud[lo:hi].contiguous(), level)
                    st.synchronize()
                    assert (s.cpu().numpy() == sig[lo:hi]).all() and int(v.abs().sum()) == 0
                results[name] = True
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e)))

    ths = [threading.Thread(target=worker, args=("a", 0, 50)), threading.Thread(target=worker, args=("b", 50, 100)),
           threading.Thread(target=worker, args=("c", 20, 80))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    assert results == {"a": True, "b": True, "c": True}


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [700, 3000, 9000])
def test_sign_options_give_identical_signatures(gpu, level, n):
    """the signing loop's alternative code paths against each other on the same messages: y as int32 vs ExpandMask's raw stream (the
    one- and the two-lanes-per-sponge writers: wide and narrow rounds), the challenge as one launch vs two, early-exit phase 2 vs full,
    and phase 2 of a speculative round with / without dropping the attempts behind an accepted one and with / without its work queues
    (option sign_skip: the default 3 = both, then 0, 1, 2)"""
    from dilithium_amd import api
    g = gpu.Generator(device="cuda").manual_seed(31 * level + n)
    seed = gpu.randint(0, 256, (1, 32), dtype=gpu.uint8, device="cuda", generator=g)
    mu = gpu.randint(0, 256, (n, 64), dtype=gpu.uint8, device="cuda", generator=g)
    pk, sk = api.keygen(seed, level)
    ref, ref_att = api.sign(sk, mu, level, shared_sk=True)
    assert int(api.verify_sig(pk, ref, mu, level, shared_pk=True).abs().sum()) == 0
    try:
        for opt in ("packed_y", "fuse_challenge", "sign_early"):
            api.set_option(opt, 0)
            sig, att = api.sign(sk, mu, level, shared_sk=True)
            api.set_option(opt, 1)
            assert gpu.equal(sig, ref) and gpu.equal(att, ref_att), opt
        for mode in (0, 1, 2):
            api.set_option("sign_skip", mode)
            sig, att = api.sign(sk, mu, level, shared_sk=True)
            assert gpu.equal(sig, ref) and gpu.equal(att, ref_att), ("sign_skip", mode)
            if n == 3000:                                  # a key per item: the per-item form of the early-exit kernel
                sigd, attd = api.sign(sk.repeat(n, 1), mu, level)
                assert gpu.equal(sigd, ref) and gpu.equal(attd, ref_att), ("sign_skip", mode, "key per item")
    finally:
        for opt in ("packed_y", "fuse_challenge", "sign_early"):
            api.set_option(opt, 1)
        api.set_option("sign_skip", 3)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_full_size_roundtrip_keygen_sign_verify(gpu, level):
    """BASELINE config sizes (8192 per GPU), size-independent property: fresh keys -> signatures -> all verify; a
    signature never verifies under its neighbour's key; signing is deterministic"""
    from dilithium_amd import api
    n = 8192
    g = gpu.Generator(device="cuda").manual_seed(7 + level)
    seed = gpu.randint(0, 256, (n, 32), dtype=gpu.uint8, device="cuda", generator=g)
    mu = gpu.randint(0, 256, (n, 64), dtype=gpu.uint8, device="cuda", generator=g)
    pk, sk = api.keygen(seed, level)
    sig, att = api.sign(sk, mu, level)
    assert int(att.min()) >= 1
    assert int(api.verify_sig(pk, sig, mu, level).abs().sum()) == 0
    wrong = api.verify_sig(pk.roll(1, 0).contiguous(), sig, mu, level)
    assert int((wrong == 0).sum()) == 0
    sig2, att2 = api.sign(sk, mu, level)
    assert gpu.equal(sig, sig2) and gpu.equal(att, att2)
    # the shared-key path gives the same bytes for the items signed under key 0
    sig0, _ = api.sign(sk[:1], mu[:64].contiguous(), level, shared_sk=True)
    sigd, _ = api.sign(sk[:1].repeat(64, 1), mu[:64].contiguous(), level)
    assert gpu.equal(sig0, sigd)
