"""GPU parity: fused Dilithium pipelines (mat-vec, verify core, sign phases) through the C-ABI
vs the oracle on seeded inputs, and the reference's 100 KATs x levels 2/3/5 pushed through
the HIP kernels (host-side hashing / codecs from the KAT harness)."""
import numpy as np
import pytest

from oracle import dilithium_kat as dk
from oracle.oracle import N, Q, splitmix64_polys
from tests.conftest import load_kat
from tests.test_kat_oracle import kat_items

pytestmark = pytest.mark.gpu

KL = {2: (4, 4), 3: (6, 5), 5: (8, 7)}


def dev(torch, a, dtype=np.int32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


class HipEngine:
    """KAT-harness engine whose polynomial work runs on the HIP kernels"""

    def __init__(self, torch):
        self.t = torch
        from dilithium_amd import api
        self.api = api

    def ntt(self, a):
        t = dev(self.t, dk.canon(a))
        self.api.ntt(t)
        return t.cpu().numpy()

    def matvec(self, p, A, y):
        return self.api.matvec(dev(self.t, A), dev(self.t, dk.canon(y)), p.level, shared_A=False).cpu().numpy()

    def verify_core(self, p, A, z, c, t1, h):
        return self.api.verify_core(dev(self.t, A), dev(self.t, dk.canon(z)), dev(self.t, dk.canon(c)),
                                    dev(self.t, t1), dev(self.t, h, np.uint8), p.level).cpu().numpy()

    def sign_phase1(self, p, A, y):
        w1, w0 = self.api.sign_phase1(dev(self.t, A), dev(self.t, dk.canon(y)), p.level)
        return w1.cpu().numpy(), w0.cpu().numpy()

    def sign_phase2(self, p, c, y, w0, w1, s1h, s2h, t0h):
        z, h, fl = self.api.sign_phase2(dev(self.t, dk.canon(c)), dev(self.t, dk.canon(y)), dev(self.t, w0),
                                        dev(self.t, w1, np.uint8), dev(self.t, s1h), dev(self.t, s2h),
                                        dev(self.t, t0h), p.level)
        return z.cpu().numpy(), h.cpu().numpy(), fl.cpu().numpy()


def synth(level, n, seed):
    K, L = KL[level]
    p = dk.PARAMS[level]
    rng = np.random.default_rng(seed)
    A = splitmix64_polys(n * K * L, seed=seed).reshape(n, K, L, N)
    z = np.mod(rng.integers(-(p.gamma1 - 1), p.gamma1 + 1, (n, L, N)), Q).astype(np.int32)
    c = np.zeros((n, N), np.int32)
    for i in range(n):
        pos = rng.choice(N, p.tau, replace=False)
        c[i, pos] = np.where(rng.integers(0, 2, p.tau) == 1, 1, Q - 1)
    t1 = rng.integers(0, 1 << 10, (n, K, N)).astype(np.int32)
    h = (rng.random((n, K, N)) < 0.03).astype(np.uint8)
    return A, z, c, t1, h


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 37])
def test_matvec_vs_oracle(gpu, oracle, level, n):
    from dilithium_amd import api
    K, L = KL[level]
    A, z, *_ = synth(level, n, 11 * level + n)
    w = api.matvec(dev(gpu, A), dev(gpu, z), level).cpu().numpy()
    assert (w == oracle.matvec(K, L, A, z)).all()
    ws = api.matvec(dev(gpu, A[:1]), dev(gpu, z), level, shared_A=True).cpu().numpy()
    assert (ws == oracle.matvec(K, L, A[:1], z, shared_A=True)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_verify_core_vs_oracle(gpu, oracle, level):
    from dilithium_amd import api
    A, z, c, t1, h = synth(level, 53, 7 + level)
    w1 = api.verify_core(dev(gpu, A), dev(gpu, z), dev(gpu, c), dev(gpu, t1), dev(gpu, h, np.uint8), level)
    assert (w1.cpu().numpy() == oracle.verify_core(level, A, z, c, t1, h)).all()
    w1s = api.verify_core(dev(gpu, A[:1]), dev(gpu, z), dev(gpu, c), dev(gpu, t1[:1]), dev(gpu, h, np.uint8), level,
                          shared_pk=True)
    assert (w1s.cpu().numpy() == oracle.verify_core(level, A[:1], z, c, t1[:1], h, shared_pk=True)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sign_phases_vs_oracle(gpu, oracle, level):
    from dilithium_amd import api
    K, L = KL[level]
    p = dk.PARAMS[level]
    n = 41
    rng = np.random.default_rng(level)
    A, _, c, _, _ = synth(level, n, 99 + level)
    y = np.mod(rng.integers(-(p.gamma1 - 1), p.gamma1 + 1, (n, L, N)), Q).astype(np.int32)
    w1, w0 = api.sign_phase1(dev(gpu, A), dev(gpu, y), level)
    ow1, ow0 = oracle.sign_phase1(level, A, y)
    assert (w1.cpu().numpy() == ow1).all() and (w0.cpu().numpy() == ow0).all()
    s1h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (n, L, N)), Q).astype(np.int32))
    s2h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (n, K, N)), Q).astype(np.int32))
    t0h = oracle.ntt(np.mod(rng.integers(-(1 << 12) + 1, (1 << 12) + 1, (n, K, N)), Q).astype(np.int32))
    z, h, fl = api.sign_phase2(dev(gpu, c), dev(gpu, y), dev(gpu, ow0), dev(gpu, ow1, np.uint8), dev(gpu, s1h),
                               dev(gpu, s2h), dev(gpu, t0h), level)
    oz, oh, ofl = oracle.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
    assert (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all() and (fl.cpu().numpy() == ofl).all()
    assert len(set(ofl.tolist())) > 1      # both accepts and rejects exercised


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_verify_100_through_hip(gpu, level, kat_msgs):
    """configs[3]: KAT bit-exact -- w1 bytes == fixture and host hash == c~"""
    k, ver, _ = kat_items(level, kat_msgs)
    ok, w1 = dk.verify_batch(level, ver, HipEngine(gpu))
    assert all(ok)
    assert (np.stack(w1) == k["w1"]).all()
    it = dict(ver[3])
    buf = bytearray(it["z_packed"])
    buf[17] ^= 4
    it["z_packed"] = bytes(buf)
    assert dk.verify_batch(level, [it], HipEngine(gpu))[0] == [False]


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_sign_100_through_hip(gpu, level, kat_msgs):
    """configs[4] path: the sign inner loop on the HIP kernels reproduces (c~, z, h) of every KAT"""
    k, _, sig = kat_items(level, kat_msgs)
    out = dk.sign_batch(level, sig, HipEngine(gpu))
    for i, (ct, z, h, att) in enumerate(out):
        assert ct == k["ctilde"][i].tobytes() and z == k["z"][i].tobytes() and h == k["h"][i].tobytes()
        assert att == k["attempts"][i]


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_keygen_matvec_through_hip(gpu, level):
    """t = A s1 + s2 through the HIP mat-vec reproduces the KAT public/secret key material"""
    k = load_kat(level)
    p = dk.PARAMS[level]
    eng = HipEngine(gpu)
    for i in range(5):
        kg = dk.keygen(level, k["seed"][i].tobytes(), eng)
        assert kg["t1_packed"] == k["t1"][i].tobytes() and dk.pack_t0(p, kg["t0"]) == k["t0"][i].tobytes()


def test_full_config4_batch_properties(gpu, oracle):
    """configs[3] at full size (level 3, batch 8192): sample parity + determinism"""
    from dilithium_amd import api
    level, n = 3, 8192
    A, z, c, t1, h = synth(level, n, 4242)
    dA, dz, dc, dt, dh = dev(gpu, A), dev(gpu, z), dev(gpu, c), dev(gpu, t1), dev(gpu, h, np.uint8)
    w1 = api.verify_core(dA, dz, dc, dt, dh, level).cpu().numpy()
    w1b = api.verify_core(dA, dz, dc, dt, dh, level).cpu().numpy()
    assert (w1 == w1b).all()
    idx = np.r_[0:16, n - 16:n, 4000:4016]
    assert (w1[idx] == oracle.verify_core(level, A[idx], z[idx], c[idx], t1[idx], h[idx])).all()
    assert w1.max() < 16
