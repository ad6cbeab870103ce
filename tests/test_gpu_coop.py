"""GPU parity of the one-sponge-per-wavefront Keccak forms (csrc/keccak_coop.hpp, coop_bodies.hpp, coop_kernels.hip; the reference's
counterpart is one state per Keccak core, rtl_src/keccak_datapath.vhd:97,116-117, gen_c.v:163-196,318-339).  The launchers pick them by sponge
count (option `coop_max`); here every SHAKE-bound entry point runs with the option forced BOTH ways -- `coop_max` = 2^30 (cooperative at every
size) and 0 (the lane-per-sponge / two-lane forms of rounds 1-4) -- at 1 / 63 / 5504 / 49152 sponges: the two must agree bit for bit, and
sampled items must equal hashlib / the KAT harness's host samplers (oracle/dilithium_kat.py)."""
import hashlib

import numpy as np
import pytest

from oracle import dilithium_kat as dk

pytestmark = pytest.mark.gpu

SIZES = [1, 63, 5504, 49152]
ALWAYS, NEVER = 1 << 30, 0


def cu(torch, a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


@pytest.fixture()
def coop(gpu):
    """run(f) -> (cooperative result, lane-per-sponge result); the option is restored afterwards"""
    from dilithium_amd import api
    saved = api.get_option("coop_max")

    def run(f):
        api.set_option("coop_max", ALWAYS)
        a = f()
        api.set_option("coop_max", NEVER)
        b = f()
        return a, b
    yield run
    api.set_option("coop_max", saved)


def sample(n, k=7):
    return sorted(set([0, n - 1, n // 2] + list(np.random.default_rng(n).integers(0, n, k))))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("in_bytes,out_bytes", [(8, 32), (32, 128), (128, 136), (136, 32), (832, 32), (1952, 32), (64, 640)])
def test_shake256(gpu, coop, n, in_bytes, out_bytes):
    from dilithium_amd import api
    if n == 49152 and in_bytes > 200:
        n = 8192                                   # (long inputs: keep the buffers small)
    data = np.random.default_rng(in_bytes + n).integers(0, 256, (n, in_bytes), dtype=np.uint8)
    d = cu(gpu, data)
    a, b = coop(lambda: api.shake256(d, out_bytes))
    assert gpu.equal(a, b)
    ah = a.cpu().numpy()
    for i in sample(n):
        assert ah[i].tobytes() == hashlib.shake_256(data[i].tobytes()).digest(out_bytes), i


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", SIZES)
def test_challenge_hash_and_sample(gpu, coop, level, n):
    """c~ = H(mu || w1) + c = SampleInBall(c~) in one launch (the signing loop), and SampleInBall alone"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    wb = p.K * (192 if level == 2 else 128)
    rng = np.random.default_rng(level * 1000 + n)
    mu = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    w1p = rng.integers(0, 256, (n, wb), dtype=np.uint8)
    dm, dw = cu(gpu, mu), cu(gpu, w1p)
    (ct_a, c_a), (ct_b, c_b) = coop(lambda: api.challenge(dm, dw, level))
    assert gpu.equal(ct_a, ct_b) and gpu.equal(c_a, c_b)
    cth, ch = ct_a.cpu().numpy(), c_a.cpu().numpy()
    for i in sample(n):
        want = hashlib.shake_256(mu[i].tobytes() + w1p[i].tobytes()).digest(32)
        assert cth[i].tobytes() == want, i
        assert (ch[i] == dk.canon(dk.sample_in_ball(p, want))).all(), i
    assert ((ch != 0).sum(axis=1) == p.tau).all()
    s_a, s_b = coop(lambda: api.sample_in_ball(ct_a, level))
    assert gpu.equal(s_a, s_b) and gpu.equal(s_a, c_a)


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 63, 1101])
def test_expand_mask(gpu, coop, level, n):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(77 * level + n)
    rhop = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    kappa = rng.integers(0, 65536, n).astype(np.int32)
    kappa[0] = 65535
    dr, dk_ = cu(gpu, rhop), cu(gpu, kappa)
    a, b = coop(lambda: api.expand_mask(dr, dk_, level))
    assert gpu.equal(a, b)
    ah = a.cpu().numpy()
    for i in sample(n, 3):
        want = np.stack([dk.expand_mask_poly(p, rhop[i].tobytes(), (int(kappa[i]) + l) & 0xFFFF) for l in range(p.L)])
        assert (ah[i] == dk.canon(want)).all(), i


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 9, 200])
def test_expand_a_and_s(gpu, coop, level, n):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(5 * level + n)
    rho = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    dr = cu(gpu, rho)
    a, b = coop(lambda: api.expand_a(dr, level))
    assert gpu.equal(a, b)
    for i in sample(n, 2):
        assert (a[i].cpu().numpy() == dk.expand_a(p, rho[i].tobytes())).all(), i
    rhop = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    dp = cu(gpu, rhop)
    (s1a, s2a), (s1b, s2b) = coop(lambda: api.expand_s(dp, level))
    assert gpu.equal(s1a, s1b) and gpu.equal(s2a, s2b)
    for i in sample(n, 2):
        s1 = np.stack([dk.expand_s_poly(p, rhop[i].tobytes(), j) for j in range(p.L)])
        s2 = np.stack([dk.expand_s_poly(p, rhop[i].tobytes(), p.L + j) for j in range(p.K)])
        assert (s1a[i].cpu().numpy() == dk.canon(s1)).all() and (s2a[i].cpu().numpy() == dk.canon(s2)).all(), i


@pytest.mark.parametrize("n", [1, 63, 700])
def test_mu_ragged_messages(gpu, coop, n):
    """mu = SHAKE256(tr || M): every message length 0 .. 2 rate blocks + the KAT range, odd offsets into the blob"""
    from dilithium_amd import api
    rng = np.random.default_rng(n)
    lens = [int(x) for x in rng.integers(0, 3400, n)]
    for j, ln in enumerate([0, 1, 3, 4, 100, 103, 104, 105, 135, 136, 137, 239, 240, 241, 272, 3300][:n]):
        lens[j] = ln
    msgs = [rng.integers(0, 256, ln, dtype=np.uint8).tobytes() for ln in lens]
    blob, off, ln = api.pack_messages(msgs)
    tr = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    dt = cu(gpu, tr)
    a, b = coop(lambda: api.mu(dt, blob, off, ln))
    assert gpu.equal(a, b)
    ah = a.cpu().numpy()
    for i in range(min(n, 40)):
        assert ah[i].tobytes() == hashlib.shake_256(tr[i].tobytes() + msgs[i]).digest(64), (i, lens[i])


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 5])
def test_scheme_calls_agree_and_match_the_kats(gpu, coop, kat_msgs, level, n):
    """keygen / sign / verify of the first KAT vectors under both settings: identical bytes, and the reference's KAT bytes"""
    from dilithium_amd import api
    from tests.conftest import load_kat
    k = load_kat(level)
    seed = cu(gpu, k["seed"][:n])
    (pk_a, sk_a), (pk_b, sk_b) = coop(lambda: api.keygen(seed, level))
    assert gpu.equal(pk_a, pk_b) and gpu.equal(sk_a, sk_b)
    mus = []
    for i in range(n):
        m = kat_msgs[i]
        assert pk_a[i].cpu().numpy().tobytes() == k["rho"][i].tobytes() + k["t1"][i].tobytes()
        mus.append(np.frombuffer(hashlib.shake_256(k["tr"][i].tobytes() + m).digest(64), dtype=np.uint8))
    mu = cu(gpu, np.stack(mus))
    (sig_a, att_a), (sig_b, att_b) = coop(lambda: api.sign(sk_a, mu, level))
    assert gpu.equal(sig_a, sig_b) and gpu.equal(att_a, att_b)
    for i in range(n):
        assert sig_a[i].cpu().numpy().tobytes() == k["ctilde"][i].tobytes() + k["z"][i].tobytes() + k["h"][i].tobytes(), i
    v_a, v_b = coop(lambda: api.verify_sig(pk_a, sig_a, mu, level))
    assert gpu.equal(v_a, v_b) and int(v_a.abs().sum()) == 0
    bad = sig_a.clone()
    bad[:, 40] ^= 1
    w_a, w_b = coop(lambda: api.verify_sig(pk_a, bad, mu, level))
    assert gpu.equal(w_a, w_b) and bool((w_a != 0).all())
