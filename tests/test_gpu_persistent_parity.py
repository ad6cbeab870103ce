"""GPU parity where the PERSISTENT item loops of the wave-per-item kernels are re-entered.

The `*_wpi` kernels launch grid = min(work, resident workgroups) and every wave walks items it, it + nwaves, ... ; the
code after a wave's first item -- the software prefetch of the NEXT item's inputs, the loop-carried row registers --
runs only when batch / 4 exceeds the resident workgroups (768-2048).  The dispatch-size tests (n = 2125 / 2304) never
get there.  Here every output of every such kernel is compared with the oracle at batches of 8192 / 9216 (the lightest kernels
keep 2048 workgroups = 8192 waves resident, so 8192 items would be ONE step for them) and 20000-40000, a key
per item and one key for the batch, and each test ASSERTS through the library's launch record (dil_launch_info) that the
loop was re-entered (steps >= 2 at 8192, >= 3 at the large size).
rtl_src/combined_top.v:1207-1469 (verify), :1850-1933 (mat-vec / FSM1), :1981-2229 (FSM2), :921-1079 (keygen)."""
import numpy as np
import pytest

from oracle import dilithium_kat as dk
from oracle.oracle import N, Q
from tests.test_gpu_pipelines import KL, dev

pytestmark = pytest.mark.gpu


def big_inputs(level, n, seed, nkeys):
    """verify-core style inputs without the per-item python loops of synth(): A, z (also used as y), c, t1, h"""
    K, L = KL[level]
    p = dk.PARAMS[level]
    rng = np.random.default_rng(seed)
    A = rng.integers(0, Q, (nkeys, K, L, N), dtype=np.int32)
    z = np.mod(rng.integers(-(p.gamma1 - 1), p.gamma1 + 1, (n, L, N), dtype=np.int32), Q).astype(np.int32)
    # c: tau coefficients +-1 at distinct positions (argpartition of random keys = a random tau-subset per item)
    pos = np.argpartition(rng.random((n, N), dtype=np.float32), p.tau, axis=1)[:, :p.tau]
    c = np.zeros((n, N), np.int32)
    np.put_along_axis(c, pos, np.where(rng.integers(0, 2, (n, p.tau)) == 1, 1, Q - 1).astype(np.int32), axis=1)
    t1 = rng.integers(0, 1 << 10, (nkeys, K, N), dtype=np.int32)
    h = (rng.random((n, K, N), dtype=np.float32) < 0.03).astype(np.uint8)
    return A, z, c, t1, h


def key_material(oracle, level, nkeys, seed):
    K, L = KL[level]
    p = dk.PARAMS[level]
    rng = np.random.default_rng(seed)
    s1h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (nkeys, L, N)), Q).astype(np.int32))
    s2h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (nkeys, K, N)), Q).astype(np.int32))
    t0h = oracle.ntt(np.mod(rng.integers(-(1 << 12) + 1, (1 << 12) + 1, (nkeys, K, N)), Q).astype(np.int32))
    return s1h, s2h, t0h


def steps(family, n, at_least):
    """the family's last launch covered n items with a grid smaller than the work: every wave re-entered its item loop"""
    import os
    from dilithium_amd import api
    info = api.launch_info(family)
    assert info["items"] == n, (family, info)
    assert info["grid"] * info["items_per_block"] < n and info["steps"] >= at_least, (family, info)
    log = os.environ.get("DIL_STEPS_LOG")          # scripts/gpu_r03.sh: profiles/r03_pytest_kernel_coverage.txt
    if log:
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
        with open(log, "a") as f:
            f.write(f"{family:20s} items {n:6d}  grid {info['grid']:5d} x {info['items_per_block']:2d} items/step = "
                    f"{info['grid'] * info['items_per_block']:6d} < {n:6d}  -> {info['steps']} steps per wave   [{test}]\n")
    return info


@pytest.mark.parametrize("level", [2, 5])
@pytest.mark.parametrize("n,min_steps", [(8192, 2), (20000, 3)])
def test_verify_wpi_persistent_loop_vs_oracle(gpu, oracle, level, n, min_steps):
    """verify_wpi_kernel<2|5>, a public key per item: all n x K x 256 outputs (level 3: test_full_config4_batch_all_items,
    and at 20000 below)"""
    from dilithium_amd import api
    A, z, c, t1, h = big_inputs(level, n, 100 * level + n, n)
    w1 = api.verify_core(dev(gpu, A), dev(gpu, z), dev(gpu, c), dev(gpu, t1), dev(gpu, h, np.uint8), level).cpu().numpy()
    steps("verify_wpi", n, min_steps)
    assert (w1 == oracle.verify_core(level, A, z, c, t1, h)).all()


def test_verify_wpi_level3_20000_vs_oracle(gpu, oracle):
    from dilithium_amd import api
    n = 20000
    A, z, c, t1, h = big_inputs(3, n, 333, n)
    w1 = api.verify_core(dev(gpu, A), dev(gpu, z), dev(gpu, c), dev(gpu, t1), dev(gpu, h, np.uint8), 3).cpu().numpy()
    steps("verify_wpi", n, 3)
    assert (w1 == oracle.verify_core(3, A, z, c, t1, h)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("small", [True, False])
@pytest.mark.parametrize("shared,n,min_steps", [(True, 9216, 2), (True, 40000, 3), (False, 9216, 2), (False, 20000, 3)])
def test_sign_phases_persistent_loop_vs_oracle(gpu, oracle, level, shared, n, min_steps, small):
    """phase 1 (matvec_shared / matvec_wpi <OUT_W1W0>: w1, w0), phase 2 (sign2_wpi_kernel: z, h, flags) and the signing
    loop's early-exit phase 2 (sign2_early_wpi_kernel through its in/out w0 scratch: first failed check, z / h of the
    accepted attempts), one key and a key per item, every attempt of the batch.  small: the entry point that takes the caller's
    word for a key decoded from key bytes (dil_sign_phase2_skey_dev: paired rows, exact small-integer tails -- the signing loop's
    kernels) / the one for arbitrary residues (dil_sign_phase2_dev, dil_sign_phase2_early_dev: one transform per product)"""
    from dilithium_amd import api
    nk = 1 if shared else n
    A, y, c, _, _ = big_inputs(level, n, 7000 + 10 * level + n + shared, nk)
    s1h, s2h, t0h = key_material(oracle, level, nk, 11 * level + n)
    # phase 1
    w1, w0 = api.sign_phase1(dev(gpu, A), dev(gpu, y), level, shared_key=shared)
    steps("sign1_shared" if shared else "sign1_wpi", n, min_steps)
    ow1, ow0 = oracle.sign_phase1(level, A, y)
    assert (w1.cpu().numpy() == ow1).all() and (w0.cpu().numpy() == ow0).all()
    del A
    # phase 2, every check of every attempt
    dc, dy, dw0, dw1 = dev(gpu, c), dev(gpu, y), dev(gpu, ow0), dev(gpu, ow1, np.uint8)
    ds1, ds2, dt0 = dev(gpu, s1h), dev(gpu, s2h), dev(gpu, t0h)
    z, h, fl = api.sign_phase2(dc, dy, dw0, dw1, ds1, ds2, dt0, level, shared_key=shared, small_key=small)
    steps("sign2_wpi", n, min_steps)
    oz, oh, ofl = oracle.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
    assert (fl.cpu().numpy() == ofl).all()
    assert (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all()
    acc = ofl == 0
    assert acc.any() and (~acc).any()
    # phase 2 as the signing loop runs it: stop at the FIRST failed check (include/dil256.h: one key -- rows in turn, r0[k] (-> 2) then
    # z[k] (-> 1); a key per item -- all r0 rows, then all z rows); then the c t0 rows (-> 4 [| 8]).  The expectation is rebuilt row
    # by row from the oracle's z and c s2.
    w0s = dw0.clone()
    ze, he, fle = api.sign_phase2_early(dc, dy, w0s, dw1, ds1, ds2, dt0, level, shared_key=shared, small_key=small)
    steps("sign2_early_wpi", n, min_steps)
    fle = fle.cpu().numpy()
    K, L = KL[level]
    p = dk.PARAMS[level]
    cen = lambda a: np.where(a > Q // 2, a - Q, a)  # noqa: E731
    first_r0 = np.full(n, 99)
    first_z = np.full(n, 99)
    for lo in range(0, n, 4096):
        hi = min(n, lo + 4096)
        chat = oracle.ntt(c[lo:hi])[:, None, :].repeat(K, axis=1)
        s2 = np.broadcast_to(s2h[0], (hi - lo, K, N)) if shared else s2h[lo:hi]
        cs2 = oracle.invntt(oracle.pointwise(np.ascontiguousarray(chat), np.ascontiguousarray(s2)))
        r0 = np.mod(ow0[lo:hi].astype(np.int64) - cs2, Q)
        bad_r0 = np.abs(cen(r0)).max(axis=2) >= p.gamma2 - p.beta                      # [items, K]
        bad_z = np.abs(cen(oz[lo:hi].astype(np.int64))).max(axis=2) >= p.gamma1 - p.beta    # [items, L]
        first_r0[lo:hi] = np.where(bad_r0.any(axis=1), bad_r0.argmax(axis=1), 99)
        first_z[lo:hi] = np.where(bad_z.any(axis=1), bad_z.argmax(axis=1), 99)
    r0f, zf, ctf = (ofl & 2) != 0, (ofl & 1) != 0, (ofl & 4) != 0
    assert ((first_r0 < 99) == r0f).all() and ((first_z < 99) == zf).all()             # the rebuilt rows agree with the oracle's flags
    if shared and small:
        want = np.where(first_r0 <= first_z, 2, 1)         # one key, paired rows: rows in turn, r0[k] before z[k] (both from one transform)
    else:
        want = np.where(first_r0 < 99, 2, 1)               # a key per item: all r0 rows, then all z rows
    early = r0f | zf
    assert (fle[early] == want[early]).all()
    m4 = ~early & ctf
    assert ((fle[m4] & ~8) == 4).all()
    rest = ~early & ~ctf
    assert (fle[rest] == ofl[rest]).all()                  # 0, or 8 (all checks passed, too many hints)
    assert (ze.cpu().numpy()[acc] == oz[acc]).all() and (he.cpu().numpy()[acc] == oh[acc]).all()
    # the scratch holds r0 = w0 - c s2 for every row that was evaluated: all K rows of an attempt that reached stage (C)
    if (~early).any():
        i = int(np.flatnonzero(~early)[-1])
        cs2 = oracle.invntt(oracle.pointwise(np.broadcast_to(oracle.ntt(c[i]), (K, N)).copy(), np.ascontiguousarray(s2h[0 if shared else i])))
        assert (w0s[i].cpu().numpy() == np.mod(ow0[i].astype(np.int64) - cs2, Q)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("shared", [True, False])
def test_sign_phase2_paired_rows_at_the_edge_of_what_key_bytes_decode_to(gpu, oracle, level, shared):
    """Phase 2 reads c s1[k] and c s2[k] off ONE inverse transform (pipeline_common.hpp SmallPair), exact while both stay within
    +-1023.  A secret key's eta-bit fields decode to eta - v for ANY field value v -- [-11, 4] at level 3, [-5, 2] at levels 2 / 5 --
    so malformed key bytes give larger |s| than a well-formed key (|s| <= eta) and are the worst case the scheme can present:
    tau * 11 = 539 at level 3.  Keys drawn from that full decoder range, with the extremes forced into every row; every z, h
    and flag vs the oracle, both key forms, at a batch where the persistent loop re-enters."""
    from dilithium_amd import api
    n = 9216
    nk = 1 if shared else n
    K, L = KL[level]
    lo, hi = (-11, 4) if level == 3 else (-5, 2)
    A, y, c, _, _ = big_inputs(level, n, 9100 + level + shared, nk)
    rng = np.random.default_rng(77 + level)
    s1 = rng.integers(lo, hi + 1, (nk, L, N))
    s2 = rng.integers(lo, hi + 1, (nk, K, N))
    s1[:, :, :128] = lo                                  # a run of the extreme value in every row, and challenges whose tau
    s2[:, :, 100:228] = lo                               # coefficients are +1 (-1) side by side: |c s| = tau |lo|, as large as it can get
    tau = dk.PARAMS[level].tau
    for i in range(8):
        c[i] = 0
        c[i, 7 * i:7 * i + tau] = 1 if i % 2 == 0 else Q - 1
    s1h = oracle.ntt(np.mod(s1, Q).astype(np.int32))
    s2h = oracle.ntt(np.mod(s2, Q).astype(np.int32))
    t0 = rng.integers(-(1 << 12) + 1, (1 << 12) + 1, (nk, K, N))
    t0[:, :, 60:188] = 1 << 12                           # the 13-bit field decodes to 2^12 - v: (-2^12, 2^12]; |c t0| reaches tau 2^12 < 2^18,
    t0[:, 0, 60:188] = -(1 << 12) + 1                    # the bound the exact small-integer row tails (Phase2Coef) rest on
    t0h = oracle.ntt(np.mod(t0, Q).astype(np.int32))
    ow1, ow0 = oracle.sign_phase1(level, A, y)
    del A
    z, h, fl = api.sign_phase2(dev(gpu, c), dev(gpu, y), dev(gpu, ow0), dev(gpu, ow1, np.uint8), dev(gpu, s1h), dev(gpu, s2h), dev(gpu, t0h),
                               level, shared_key=shared, small_key=True)
    steps("sign2_wpi", n, 2)
    oz, oh, ofl = oracle.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
    assert (fl.cpu().numpy() == ofl).all()
    assert (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all()
    # the products really are larger than a well-formed key's beta (so this is not the ordinary case again) and inside the bound
    cs = oracle.invntt(oracle.pointwise(np.broadcast_to(oracle.ntt(c[:1]), (L, N)).copy(), np.ascontiguousarray(s1h[0])))
    big = np.abs(np.where(cs > Q // 2, cs.astype(np.int64) - Q, cs)).max()
    assert big == tau * -lo and dk.PARAMS[level].beta < big <= 1023


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("shared", [True, False])
def test_sign_phase2_generic_entry_takes_any_residues(gpu, oracle, level, shared):
    """dil_sign_phase2_dev / dil_sign_phase2_early_dev promise nothing about their inputs' size (round-3 advisor finding: the paired-row
    kernels silently depended on |c s| <= 1023): UNIFORM residues for c, s1^, s2^, t0^ and w0 -- products of full size -- at a batch the
    wave-per-item kernels serve, every z, h and flag vs the oracle; the early-exit form's flags where they are determined."""
    from dilithium_amd import api
    n = 9216
    nk = 1 if shared else n
    K, L = KL[level]
    rng = np.random.default_rng(4100 + level + shared)
    u = lambda *sh: rng.integers(0, Q, sh, dtype=np.int64).astype(np.int32)  # noqa: E731
    c, y, w0, s1h, s2h, t0h = u(n, N), u(n, L, N), u(n, K, N), u(nk, L, N), u(nk, K, N), u(nk, K, N)
    w1 = rng.integers(0, 44 if level == 2 else 16, (n, K, N)).astype(np.uint8)
    w1[:, :, ::3] = 0                                     # (w1 == 0 is the special case of MakeHint)
    z, h, fl = api.sign_phase2(dev(gpu, c), dev(gpu, y), dev(gpu, w0), dev(gpu, w1, np.uint8), dev(gpu, s1h), dev(gpu, s2h), dev(gpu, t0h),
                               level, shared_key=shared)
    steps("sign2_wpi", n, 2)
    oz, oh, ofl = oracle.sign_phase2(level, c, y, w0, w1, s1h, s2h, t0h)
    assert (fl.cpu().numpy() == ofl).all()
    assert (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all()
    w0s = dev(gpu, w0)
    _, _, fle = api.sign_phase2_early(dev(gpu, c), dev(gpu, y), w0s, dev(gpu, w1, np.uint8), dev(gpu, s1h), dev(gpu, s2h), dev(gpu, t0h),
                                      level, shared_key=shared)
    fle = fle.cpu().numpy()
    assert ((fle != 0) == (ofl != 0)).all()               # uniform residues: every attempt is rejected, by an r0 row
    assert (fle[(ofl & 2) != 0] == 2).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n,min_steps", [(8192, 2), (20000, 3)])
def test_matvec_wpi_persistent_loop_vs_oracle(gpu, oracle, level, n, min_steps):
    """matvec_wpi_kernel<K, L, LEVEL, OUT_W> (a matrix per item), every output"""
    from dilithium_amd import api
    K, L = KL[level]
    A, y, *_ = big_inputs(level, n, 5000 + level + n, n)
    w = api.matvec(dev(gpu, A), dev(gpu, y), level).cpu().numpy()
    steps("matvec_wpi", n, min_steps)
    assert (w == oracle.matvec(K, L, A, y)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_keygen_wpi_persistent_loop(gpu, oracle, level):
    """keygen_wpi_kernel at 20000 keys: t1 2^13 + t0 == A s1 + s2 with the mat-vec recomputed by the ORACLE for EVERY key,
    and (pk, sk) byte-identical to the host KAT harness on 1024 keys spread over every step of the persistent loop"""
    from dilithium_amd import api
    from tests.test_gpu_codecs import cu
    p = dk.PARAMS[level]
    rng = np.random.default_rng(60 + level)
    n = 20000
    seed = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = api.keygen(cu(gpu, seed), level)
    info = steps("keygen_wpi", n, 3)
    sb = 32 * p.eta_bits
    t1 = api.unpack(pk, api.CODEC_T1, level, 32).long()
    t0 = api.unpack(sk, api.CODEC_T0, level, 96 + (p.L + p.K) * sb).long()
    s1 = api.unpack(sk, api.CODEC_S1, level, 96)
    s2 = api.unpack(sk, api.CODEC_S2, level, 96 + p.L * sb).long()
    A = api.expand_a(pk[:, :32].contiguous(), level)
    w = gpu.from_numpy(oracle.matvec(p.K, p.L, A.cpu().numpy(), s1.cpu().numpy())).cuda().long()
    assert bool((((t1 << 13) + t0 - w - s2) % dk.Q == 0).all())
    del A, w
    eng = dk.OracleEngine(oracle)
    pkh, skh = pk.cpu().numpy(), sk.cpu().numpy()
    per_step = info["grid"] * info["items_per_block"]
    sample = set(rng.choice(n, 1000, replace=False).tolist()) | {0, n - 1, per_step - 1, per_step, 2 * per_step - 1, 2 * per_step}
    sample |= set(range(n - 18, n))
    for i in sorted(sample):
        kg = dk.keygen(level, seed[i].tobytes(), eng)
        assert pkh[i].tobytes() == kg["rho"] + kg["t1_packed"], i
        assert skh[i].tobytes() == kg["rho"] + kg["key"] + kg["tr"] + dk.pack_eta(p, kg["s1"]) + dk.pack_eta(p, kg["s2"]) + \
            dk.pack_t0(p, kg["t0"]), i


@pytest.mark.parametrize("level", [2, 3, 5])
def test_verify_wire_wpi_persistent_loop(gpu, oracle, level):
    """verify_wire_wpi_kernel (packed z / t1 / hints in, packed w1 out) at 20000 (key, signature) pairs: keys and signatures
    from the device keygen / signing loop, the kernel's packed w1 against the ORACLE's verify core fed with host-decoded
    fields, for every item"""
    from dilithium_amd import api
    from tests.test_gpu_codecs import cu
    p = dk.PARAMS[level]
    K, L = KL[level]
    rng = np.random.default_rng(80 + level)
    n = 20000
    seed = cu(gpu, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    mu = cu(gpu, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    pk, sk = api.keygen(seed, level)
    sig, _ = api.sign(sk, mu, level)
    A = api.expand_a(pk[:, :32].contiguous(), level)
    w1p, verdict = api.verify_wire_core(A, pk, sig, level)
    steps("verify_wire_wpi", n, 3)
    assert int(verdict.abs().sum()) == 0
    zb = L * 32 * p.z_bits
    z = api.unpack(sig, api.CODEC_Z, level, 32)
    t1 = api.unpack(pk, api.CODEC_T1, level, 32)
    h, bad = api.hint_unpack(sig, level, 32 + zb)
    assert int(bad.sum()) == 0
    c = api.sample_in_ball(sig[:, :32].contiguous(), level)
    ow1 = oracle.verify_core(level, A.cpu().numpy(), z.cpu().numpy(), c.cpu().numpy(), t1.cpu().numpy(), h.cpu().numpy())
    want = api.pack_w1(cu(gpu, ow1), level)
    assert gpu.equal(w1p.view(-1), want.view(-1))
    assert (api.verify_sig(pk, sig, mu, level) == 0).all()
