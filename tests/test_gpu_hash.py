"""GPU parity for SURVEY 8(f) row N1: Keccak / SHAKE and the SHAKE-bound samplers on the device,
against hashlib and the KAT harness's host-side samplers (oracle/dilithium_kat.py), and end to end
on the reference's KAT vectors: verify with on-device SampleInBall / w1 hashing, sign with on-device
ExpandMask / challenge hashing."""
import hashlib

import numpy as np
import pytest

from oracle import dilithium_kat as dk
from tests.conftest import load_kat
from tests.test_kat_oracle import kat_items

pytestmark = pytest.mark.gpu


def cu(torch, a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


@pytest.mark.parametrize("in_bytes,out_bytes", [(8, 32), (64, 64), (128, 136), (136, 32), (832, 32), (1600, 272)])
def test_shake256_batch_vs_hashlib(gpu, in_bytes, out_bytes):
    from dilithium_amd import api
    rng = np.random.default_rng(in_bytes)
    n = 131
    data = rng.integers(0, 256, (n, in_bytes), dtype=np.uint8)
    out = api.shake256(cu(gpu, data), out_bytes).cpu().numpy()
    for i in range(n):
        assert out[i].tobytes() == hashlib.shake_256(data[i].tobytes()).digest(out_bytes)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expand_a_vs_host_sampler(gpu, level):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(level)
    rho = rng.integers(0, 256, (9, 32), dtype=np.uint8)
    A = api.expand_a(cu(gpu, rho), level).cpu().numpy()
    for i in range(rho.shape[0]):
        assert (A[i] == dk.expand_a(p, rho[i].tobytes())).all()
    assert A.min() >= 0 and A.max() < dk.Q


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expand_a_throughput_kernel_vs_host_sampler(gpu, level):
    """more than 32768 polynomials switch ExpandA to the lane-per-sponge kernel with the wave-synchronous transposed
    flush (expand_a_fast_kernel): ragged key count (the last wave is partly empty), sampled keys vs the host sampler, and
    ALL keys vs the two-lane kernel (the same rho expanded 8 keys at a time)"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(100 + level)
    n = 32768 // (p.K * p.L) + 37
    rho = cu(gpu, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    A = api.expand_a(rho, level)
    for i in (0, 1, n // 2, n - 2, n - 1):
        assert (A[i].cpu().numpy() == dk.expand_a(p, rho[i].cpu().numpy().tobytes())).all()
    small = gpu.cat([api.expand_a(rho[i:i + 8].contiguous(), level) for i in range(0, n, 8)])
    assert gpu.equal(A, small)
    assert int(A.min()) >= 0 and int(A.max()) < dk.Q


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expand_mask_vs_host_sampler(gpu, level):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(10 + level)
    n = 70
    rhop = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    kappa = rng.integers(0, 60000, n).astype(np.int32)
    y = api.expand_mask(cu(gpu, rhop), cu(gpu, kappa), level).cpu().numpy()
    for i in range(n):
        want = np.stack([dk.expand_mask_poly(p, rhop[i].tobytes(), int(kappa[i]) + l) for l in range(p.L)])
        assert (y[i] == dk.canon(want)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_expand_mask_throughput_kernel(gpu, level):
    """expand_mask_kernel<18|20, false> (lane per sponge, above `two_lane_max_sponges` = 16384 polynomials; wave-synchronous
    transposed stores): ragged last wave, kappa near the 16-bit wrap, sampled items vs the host sampler, and the whole
    batch against the two-lane kernel run in small pieces"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(310 + level)
    n = 16384 // p.L + 77
    rhop = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    kappa = rng.integers(0, 65536, n).astype(np.int32)
    kappa[:4] = [65535, 65530, 0, 65536 - p.L]
    y = api.expand_mask(cu(gpu, rhop), cu(gpu, kappa), level)
    yh = y.cpu().numpy()
    assert yh.min() >= 0 and yh.max() < dk.Q
    for i in (0, 1, 2, 3, 63, 64, n // 2, n - 2, n - 1):
        want = np.stack([dk.expand_mask_poly(p, rhop[i].tobytes(), (int(kappa[i]) + l) & 0xFFFF) for l in range(p.L)])
        assert (yh[i] == dk.canon(want)).all(), i
    small = gpu.cat([api.expand_mask(cu(gpu, rhop[i:i + 512]), cu(gpu, kappa[i:i + 512]), level) for i in range(0, n, 512)])
    assert gpu.equal(y, small)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sample_in_ball_vs_host_sampler(gpu, level):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(20 + level)
    n = 200
    ct = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    c = api.sample_in_ball(cu(gpu, ct), level).cpu().numpy()
    for i in range(n):
        assert (c[i] == dk.canon(dk.sample_in_ball(p, ct[i].tobytes()))).all()
    assert ((c != 0).sum(axis=1) == p.tau).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [1, 31, 33, 200, 70001])
def test_challenge_one_launch_vs_hashlib_and_host_sampler(gpu, level, n):
    """c~ = H(mu || w1_packed) and c = SampleInBall(c~) from ONE launch (gen_c.v:163-196,318-339 is one module): against hashlib
    + the KAT harness's sampler, and against the library's own two-launch form.  n = 31 / 33: a partial workgroup of the
    two-lanes-per-sponge form (32 signatures per workgroup); n = 70001: the lane-per-sponge form (64 per workgroup, ragged tail)"""
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(40 + level + n)
    wb = p.K * (192 if level == 2 else 128)
    mu = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    w1p = rng.integers(0, 256, (n, wb), dtype=np.uint8)
    ct, c = api.challenge(cu(gpu, mu), cu(gpu, w1p), level)
    pick = range(n) if n <= 200 else [0, 1, 63, 64, 65, 4095, 4096, n // 2, n - 66, n - 65, n - 64, n - 2, n - 1]
    cth, ch = ct.cpu().numpy(), c.cpu().numpy()
    for i in pick:
        want = hashlib.shake_256(mu[i].tobytes() + w1p[i].tobytes()).digest(32)
        assert cth[i].tobytes() == want, i
        assert (ch[i] == dk.canon(dk.sample_in_ball(p, want))).all(), i
    assert ((ch != 0).sum(axis=1) == p.tau).all()
    # every entry against the two-launch form (the challenge hash is reached through shake256 on the concatenated input)
    cat = gpu.cat([cu(gpu, mu), cu(gpu, w1p)], dim=1).contiguous()
    ct2 = api.shake256(cat, 32)
    assert gpu.equal(ct, ct2)
    assert gpu.equal(c, api.sample_in_ball(ct2, level))


@pytest.mark.parametrize("level", [2, 3, 5])
def test_pack_w1_vs_host_codec(gpu, level):
    from dilithium_amd import api
    p = dk.PARAMS[level]
    rng = np.random.default_rng(30 + level)
    n = 37
    w1 = rng.integers(0, 44 if level == 2 else 16, (n, p.K, 256)).astype(np.uint8)
    out = api.pack_w1(cu(gpu, w1), level).cpu().numpy()
    for i in range(n):
        assert out[i].tobytes() == dk.pack_w1(p, w1[i])


class DeviceHashEngine:
    """KAT-harness engine with the polynomial work AND the SHAKE-bound steps on the GPU"""

    def __init__(self, torch):
        from tests.test_gpu_pipelines import HipEngine
        from dilithium_amd import api
        self.t, self.api, self.poly = torch, api, HipEngine(torch)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_verify_with_device_hashing(gpu, level, kat_msgs):
    """100 KATs: c = SampleInBall(c~) on device, verify core on device, c~' = H(mu || pack(w1)) on device"""
    from dilithium_amd import api
    torch = gpu
    p = dk.PARAMS[level]
    k, ver, _ = kat_items(level, kat_msgs)
    n = len(ver)
    rho = np.stack([np.frombuffer(it["rho"], np.uint8) for it in ver])
    A = api.expand_a(cu(torch, rho), level)                                     # ExpandA on device
    ct = np.stack([np.frombuffer(it["ctilde"], np.uint8) for it in ver])
    c = api.sample_in_ball(cu(torch, ct), level)
    z = np.stack([dk.canon(dk.unpack_z(p, it["z_packed"])) for it in ver])
    t1 = np.stack([dk.unpack_t1(p, it["t1_packed"]) for it in ver])
    h = np.stack([dk.unpack_hint(p, it["h_packed"]) for it in ver])
    w1 = api.verify_core(A, cu(torch, z), c, cu(torch, t1), cu(torch, h, np.uint8), level)
    assert (w1.cpu().numpy() == k["w1"]).all()
    w1p = api.pack_w1(w1, level)
    mu = np.stack([np.frombuffer(dk.shake256(dk.shake256(it["rho"] + it["t1_packed"], 32) + it["msg"], 64), np.uint8)
                   for it in ver])
    buf = torch.cat([cu(torch, mu), w1p], dim=1).contiguous()
    assert buf.shape[1] % 8 == 0
    got = api.shake256(buf, 32).cpu().numpy()
    assert (got == ct).all()                      # every KAT accepted: H(mu || w1) == c~


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_sign_with_device_hashing(gpu, level, kat_msgs):
    """100 KATs, deterministic signing with y = ExpandMask, c~ = H(mu || w1), c = SampleInBall all on device"""
    from dilithium_amd import api
    torch = gpu
    p = dk.PARAMS[level]
    k, _, sig = kat_items(level, kat_msgs)
    n = len(sig)
    eng = DeviceHashEngine(torch).poly
    rho = np.stack([np.frombuffer(it["rho"], np.uint8) for it in sig])
    A = api.expand_a(cu(torch, rho), level)
    s1h = cu(torch, np.stack([eng.ntt(dk.canon(dk.unpack_eta(p, it["s1_packed"], p.L))) for it in sig]))
    s2h = cu(torch, np.stack([eng.ntt(dk.canon(dk.unpack_eta(p, it["s2_packed"], p.K))) for it in sig]))
    t0h = cu(torch, np.stack([eng.ntt(dk.canon(dk.unpack_t0(p, it["t0_packed"]))) for it in sig]))
    mu = [dk.shake256(it["tr"] + it["msg"], 64) for it in sig]
    rhop = np.stack([np.frombuffer(dk.shake256(it["key"] + m, 64), np.uint8) for it, m in zip(sig, mu)])
    d_mu = cu(torch, np.stack([np.frombuffer(m, np.uint8) for m in mu]))
    d_rhop = cu(torch, rhop)
    live = torch.arange(n, device="cuda")
    kappa = torch.zeros(n, dtype=torch.int32, device="cuda")
    out_c = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    out_z = torch.zeros((n, p.L, 256), dtype=torch.int32, device="cuda")
    out_h = torch.zeros((n, p.K, 256), dtype=torch.uint8, device="cuda")
    attempts = torch.zeros(n, dtype=torch.int32, device="cuda")
    rounds = 0
    while live.numel() and rounds < 64:
        rounds += 1
        y = api.expand_mask(d_rhop[live].contiguous(), kappa[live].contiguous(), level)
        w1, w0 = api.sign_phase1(A[live].contiguous(), y, level)
        ct = api.shake256(torch.cat([d_mu[live], api.pack_w1(w1, level)], dim=1).contiguous(), 32)
        c = api.sample_in_ball(ct, level)
        z, h, fl = api.sign_phase2(c, y, w0, w1, s1h[live].contiguous(), s2h[live].contiguous(), t0h[live].contiguous(), level)
        attempts[live] += 1
        kappa[live] += p.L
        okm = fl == 0
        done = live[okm]
        out_c[done], out_z[done], out_h[done] = ct[okm], z[okm], h[okm]
        live = live[~okm]
    assert live.numel() == 0
    oc, oz, oh, at = out_c.cpu().numpy(), out_z.cpu().numpy(), out_h.cpu().numpy(), attempts.cpu().numpy()
    for i in range(n):
        assert oc[i].tobytes() == k["ctilde"][i].tobytes()
        assert dk.pack_z(p, oz[i]) == k["z"][i].tobytes()
        assert dk.pack_hint(p, oh[i]) == k["h"][i].tobytes()
    assert (at == k["attempts"]).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_verify_single_call(gpu, level, kat_msgs):
    """dil_verify_dev: SampleInBall + verify core + w1 packing + challenge hash + compare in ONE call"""
    from dilithium_amd import api
    torch = gpu
    p = dk.PARAMS[level]
    k, ver, _ = kat_items(level, kat_msgs)
    rho = np.stack([np.frombuffer(it["rho"], np.uint8) for it in ver])
    A = api.expand_a(cu(torch, rho), level)
    ct = np.stack([np.frombuffer(it["ctilde"], np.uint8) for it in ver])
    z = np.stack([dk.canon(dk.unpack_z(p, it["z_packed"])) for it in ver])
    t1 = np.stack([dk.unpack_t1(p, it["t1_packed"]) for it in ver])
    h = np.stack([dk.unpack_hint(p, it["h_packed"]) for it in ver])
    mu = np.stack([np.frombuffer(dk.shake256(dk.shake256(it["rho"] + it["t1_packed"], 32) + it["msg"], 64), np.uint8)
                   for it in ver])
    args = (A, cu(torch, ct), cu(torch, z), cu(torch, t1), cu(torch, h, np.uint8), cu(torch, mu))
    assert (api.verify(*args, level).cpu().numpy() == 0).all()
    # tamper: flip one coefficient of z in item 5, the challenge in item 9, push one z over the norm bound in item 11
    z2, ct2 = z.copy(), ct.copy()
    z2[5, 0, 17] = (z2[5, 0, 17] + 1) % dk.Q
    ct2[9, 0] ^= 1
    z2[11, 1, 3] = p.gamma1 - p.beta
    v = api.verify(A, cu(torch, ct2), cu(torch, z2), cu(torch, t1), cu(torch, h, np.uint8), cu(torch, mu), level).cpu().numpy()
    assert v[5] & 1 and v[9] & 1 and v[11] & 2
    good = np.ones(len(ver), bool)
    good[[5, 9, 11]] = False
    assert (v[good] == 0).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_sign_single_call_attempts(gpu, level, kat_msgs):
    """dil_sign_attempt_dev in a rejection loop reproduces all 100 KAT signatures byte for byte"""
    from dilithium_amd import api
    from tests.test_gpu_pipelines import HipEngine
    torch = gpu
    p = dk.PARAMS[level]
    k, _, sig = kat_items(level, kat_msgs)
    n = len(sig)
    eng = HipEngine(torch)
    rho = np.stack([np.frombuffer(it["rho"], np.uint8) for it in sig])
    A = api.expand_a(cu(torch, rho), level)
    s1h = cu(torch, np.stack([eng.ntt(dk.canon(dk.unpack_eta(p, it["s1_packed"], p.L))) for it in sig]))
    s2h = cu(torch, np.stack([eng.ntt(dk.canon(dk.unpack_eta(p, it["s2_packed"], p.K))) for it in sig]))
    t0h = cu(torch, np.stack([eng.ntt(dk.canon(dk.unpack_t0(p, it["t0_packed"]))) for it in sig]))
    mu = [dk.shake256(it["tr"] + it["msg"], 64) for it in sig]
    d_mu = cu(torch, np.stack([np.frombuffer(m, np.uint8) for m in mu]))
    d_rhop = cu(torch, np.stack([np.frombuffer(dk.shake256(it["key"] + m, 64), np.uint8) for it, m in zip(sig, mu)]))
    live = torch.arange(n, device="cuda")
    kappa = torch.zeros(n, dtype=torch.int32, device="cuda")
    res = {}
    rounds = 0
    while live.numel() and rounds < 64:
        rounds += 1
        sel = lambda t: t[live].contiguous()  # noqa: E731
        ct, z, h, fl = api.sign_attempt(sel(A), sel(d_mu), sel(d_rhop), sel(kappa), sel(s1h), sel(s2h), sel(t0h), level)
        kappa[live] += p.L
        ok = (fl == 0).cpu().numpy()
        idx = live.cpu().numpy()
        for j in np.nonzero(ok)[0]:
            res[int(idx[j])] = (ct[j].cpu().numpy().tobytes(), z[j].cpu().numpy(), h[j].cpu().numpy(), rounds)
        live = live[torch.from_numpy(~ok).cuda()]
    assert len(res) == n
    for i in range(n):
        c_, z_, h_, att = res[i]
        assert c_ == k["ctilde"][i].tobytes() and dk.pack_z(p, z_) == k["z"][i].tobytes()
        assert dk.pack_hint(p, h_) == k["h"][i].tobytes() and att == k["attempts"][i]
