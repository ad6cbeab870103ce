"""GPU parity at DISPATCH sizes: the fused pipelines pick a kernel shape by batch size (pipelines.hip `use_wpi`:
wave-per-item / shared-key kernels from batch >= 8 x #CUs = 2048, workgroup-per-item kernels below), so parity at
n = 37 says nothing about the kernels bench.py times.  Here every shape is compared with the oracle at batches
>= 2048 (ALL items), and both shapes are run on the same small input through the `fused_mode` option and compared
in full with each other and with the oracle.   rtl_src/combined_top.v:1207-1469 (verify), :1850-1933 (mat-vec),
:1981-2229 (sign phase 2)."""
import numpy as np
import pytest

from oracle import dilithium_kat as dk
from oracle.oracle import N, Q
from tests.test_gpu_pipelines import KL, dev, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def fused_mode(gpu):
    from dilithium_amd import api
    yield lambda m: api.set_option("fused_mode", m)
    api.set_option("fused_mode", 0)


def sign_inputs(oracle, level, n, seed, nkeys):
    K, L = KL[level]
    p = dk.PARAMS[level]
    rng = np.random.default_rng(seed)
    A, _, c, _, _ = synth(level, max(n, 1), seed)
    A = A[:nkeys]
    y = np.mod(rng.integers(-(p.gamma1 - 1), p.gamma1 + 1, (n, L, N)), Q).astype(np.int32)
    s1h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (nkeys, L, N)), Q).astype(np.int32))
    s2h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (nkeys, K, N)), Q).astype(np.int32))
    t0h = oracle.ntt(np.mod(rng.integers(-(1 << 12) + 1, (1 << 12) + 1, (nkeys, K, N)), Q).astype(np.int32))
    return A, y, c, s1h, s2h, t0h


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [2048, 8192])
def test_verify_shared_kernel_vs_oracle(gpu, oracle, level, n):
    """verify_shared_kernel<LEVEL, NW> (one pk for the batch, A and t1^ LDS-resident): every item vs the oracle"""
    from dilithium_amd import api
    A, z, c, t1, h = synth(level, n, 1000 + level + n)
    w1 = api.verify_core(dev(gpu, A[:1]), dev(gpu, z), dev(gpu, c), dev(gpu, t1[:1]), dev(gpu, h, np.uint8), level,
                         shared_pk=True).cpu().numpy()
    assert (w1 == oracle.verify_core(level, A[:1], z, c, t1[:1], h, shared_pk=True)).all()


@pytest.mark.parametrize("level", [2, 5])
def test_verify_wpi_kernel_vs_oracle_all_items(gpu, oracle, level):
    """verify_wpi_kernel<2|5> at dispatch size, every item (level 3 at 8192: test_full_config4_batch_all_items)"""
    from dilithium_amd import api
    n = 2304
    A, z, c, t1, h = synth(level, n, 2000 + level)
    w1 = api.verify_core(dev(gpu, A), dev(gpu, z), dev(gpu, c), dev(gpu, t1), dev(gpu, h, np.uint8), level).cpu().numpy()
    assert (w1 == oracle.verify_core(level, A, z, c, t1, h)).all()


@pytest.mark.parametrize("n", [2047, 2048, 2049, 2063, 2065, 3071])
def test_dispatch_boundary_sizes(gpu, oracle, n):
    """around the kernel-shape switch (2048 = 8 x #CUs) and around whole workgroups of the shared-key kernels (16 waves):
    one item below, on, and above -- verify core and mat-vec, a key per item and one key, every output vs the oracle"""
    from dilithium_amd import api
    level, (K, L) = 3, KL[3]
    A, z, c, t1, h = synth(level, n, 9000 + n)
    for shared in (False, True):
        k = 1 if shared else n
        w1 = api.verify_core(dev(gpu, A[:k]), dev(gpu, z), dev(gpu, c), dev(gpu, t1[:k]), dev(gpu, h, np.uint8), level,
                             shared_pk=shared).cpu().numpy()
        assert (w1 == oracle.verify_core(level, A[:k], z, c, t1[:k], h, shared_pk=shared)).all(), ("verify", shared)
        w = api.matvec(dev(gpu, A[:k]), dev(gpu, z), level, shared_A=shared).cpu().numpy()
        assert (w == oracle.matvec(K, L, A[:k], z, shared_A=shared)).all(), ("matvec", shared)


def test_full_config4_batch_all_items(gpu, oracle):
    """BASELINE configs[3] exactly: level 3, batch 8192, distinct pk -- ALL 8192 x 6 x 256 outputs vs the oracle"""
    from dilithium_amd import api
    A, z, c, t1, h = synth(3, 8192, 31337)
    w1 = api.verify_core(dev(gpu, A), dev(gpu, z), dev(gpu, c), dev(gpu, t1), dev(gpu, h, np.uint8), 3).cpu().numpy()
    assert (w1 == oracle.verify_core(3, A, z, c, t1, h)).all()


def test_matvec_wpi_config2_all_outputs(gpu, oracle):
    """BASELINE configs[2] exactly: level 2 (K = L = 4), batch 4096, distinct A -> matvec_wpi_kernel<4,4,2,OUT_W>,
    all 4096 x 4 x 256 outputs vs the oracle"""
    from dilithium_amd import api
    A, y, *_ = synth(2, 4096, 424242)
    w = api.matvec(dev(gpu, A), dev(gpu, y), 2).cpu().numpy()
    assert (w == oracle.matvec(4, 4, A, y)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("n", [2048, 4099])
def test_matvec_shared_and_wpi_vs_oracle(gpu, oracle, level, n):
    """matvec_shared_kernel<..., OUT_W> (one A, LDS-resident) and matvec_wpi_kernel<..., OUT_W> (A per item), ragged n"""
    from dilithium_amd import api
    K, L = KL[level]
    A, y, *_ = synth(level, n, 3000 + level + n)
    ws = api.matvec(dev(gpu, A[:1]), dev(gpu, y), level, shared_A=True).cpu().numpy()
    assert (ws == oracle.matvec(K, L, A[:1], y, shared_A=True)).all()
    wd = api.matvec(dev(gpu, A), dev(gpu, y), level).cpu().numpy()
    assert (wd == oracle.matvec(K, L, A, y)).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("shared", [True, False])
def test_sign_phases_at_dispatch_size_vs_oracle(gpu, oracle, level, shared):
    """phase 1: matvec_shared / matvec_wpi <OUT_W1W0>;  phase 2: the NON-early sign2_wpi_kernel<LEVEL> (the kernel
    bench.py times for configs[4]) with one key and with a key per item: z, h AND flags of every attempt vs the oracle"""
    from dilithium_amd import api
    n = 2048 + 77
    nk = 1 if shared else n
    A, y, c, s1h, s2h, t0h = sign_inputs(oracle, level, n, 4000 + level, nk)
    w1, w0 = api.sign_phase1(dev(gpu, A), dev(gpu, y), level, shared_key=shared)
    ow1, ow0 = oracle.sign_phase1(level, A, y)
    assert (w1.cpu().numpy() == ow1).all() and (w0.cpu().numpy() == ow0).all()
    z, h, fl = api.sign_phase2(dev(gpu, c), dev(gpu, y), dev(gpu, ow0), dev(gpu, ow1, np.uint8), dev(gpu, s1h), dev(gpu, s2h),
                               dev(gpu, t0h), level, shared_key=shared)
    oz, oh, ofl = oracle.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
    assert (fl.cpu().numpy() == ofl).all()
    assert (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all()
    assert (ofl == 0).any() and (ofl != 0).any()          # accepted and rejected attempts both present


@pytest.mark.parametrize("level", [2, 3, 5])
def test_both_kernel_shapes_on_the_same_input(gpu, oracle, level, fused_mode):
    """`fused_mode` 1 (workgroup-per-item) and 2 (wave-per-item / shared-key) on ONE small input: identical to each
    other and to the oracle for mat-vec, verify core and both sign phases, distinct and shared keys"""
    from dilithium_amd import api
    K, L = KL[level]
    n = 61
    A, z, c, t1, h = synth(level, n, 5000 + level)
    As, ys, cs, s1h, s2h, t0h = sign_inputs(oracle, level, n, 6000 + level, n)
    want = {}
    for shared in (False, True):
        k = 1 if shared else n
        want["mv", shared] = oracle.matvec(K, L, A[:k], z, shared_A=shared)
        want["vy", shared] = oracle.verify_core(level, A[:k], z, c, t1[:k], h, shared_pk=shared)
        want["s1", shared] = oracle.sign_phase1(level, As[:k], ys)
        want["s2", shared] = oracle.sign_phase2(level, cs, ys, want["s1", shared][1], want["s1", shared][0], s1h[:k], s2h[:k], t0h[:k])
    for mode in (1, 2):
        fused_mode(mode)
        assert api.get_option("fused_mode") == mode
        for shared in (False, True):
            k = 1 if shared else n
            got = api.matvec(dev(gpu, A[:k]), dev(gpu, z), level, shared_A=shared).cpu().numpy()
            assert (got == want["mv", shared]).all(), (mode, shared, "matvec")
            got = api.verify_core(dev(gpu, A[:k]), dev(gpu, z), dev(gpu, c), dev(gpu, t1[:k]), dev(gpu, h, np.uint8), level,
                                  shared_pk=shared).cpu().numpy()
            assert (got == want["vy", shared]).all(), (mode, shared, "verify")
            w1, w0 = api.sign_phase1(dev(gpu, As[:k]), dev(gpu, ys), level, shared_key=shared)
            ow1, ow0 = want["s1", shared]
            assert (w1.cpu().numpy() == ow1).all() and (w0.cpu().numpy() == ow0).all(), (mode, shared, "sign1")
            zz, hh, fl = api.sign_phase2(dev(gpu, cs), dev(gpu, ys), dev(gpu, ow0), dev(gpu, ow1, np.uint8), dev(gpu, s1h[:k]),
                                         dev(gpu, s2h[:k]), dev(gpu, t0h[:k]), level, shared_key=shared)
            oz, oh, ofl = want["s2", shared]
            assert (fl.cpu().numpy() == ofl).all() and (zz.cpu().numpy() == oz).all() and (hh.cpu().numpy() == oh).all(), \
                (mode, shared, "sign2")


@pytest.mark.parametrize("level", [2, 3, 5])
def test_packed_matrix_format_same_bytes(gpu, level, fused_mode):
    """option a24: inside keygen / sign / verify a matrix per key crosses HBM as 24-bit packed coefficients
    (expand_a_fast_kernel<true> -> matvec_wpi_kernel / matvec_kernel / verify_wire_wpi_kernel with ARow<L, A_P24>) or as
    int32: byte-identical keys, signatures, attempt counts and verdicts at a dispatch-size batch, in both kernel shapes;
    likewise keygen with its output stage fused (keygen_wpi_kernel) or not (option fuse_keygen)"""
    from dilithium_amd import api
    rng = np.random.default_rng(40 + level)
    n = 2304
    seed = dev(gpu, rng.integers(0, 256, (n, 32), dtype=np.uint8), np.uint8)
    mu = dev(gpu, rng.integers(0, 256, (n, 64), dtype=np.uint8), np.uint8)
    out = {}
    try:
        for a24, mode in ((0, 0), (1, 0), (1, 1), (2, 0), (3, 0)):       # 2: the packed matrix in verification too
            api.set_option("fuse_keygen", 0 if a24 == 3 else 1)                # 3: packed matrix, keygen's output stage unfused
            a24 = 1 if a24 == 3 else a24
            api.set_option("a24", a24)
            fused_mode(mode)
            pk, sk = api.keygen(seed, level)
            sig, att = api.sign(sk, mu, level)
            bad = sig.clone()
            bad[5, 40] ^= 4
            v = api.verify_sig(pk, bad, mu, level)
            out[a24, mode, api.get_option("fuse_keygen")] = [t.cpu().numpy() for t in (pk, sk, sig, att, v)]
    finally:
        api.set_option("a24", 1)
        api.set_option("fuse_keygen", 1)
    for key in ((1, 0, 1), (1, 1, 1), (2, 0, 1), (1, 0, 0)):
        for a, b in zip(out[0, 0, 1], out[key]):
            assert (a == b).all(), key
    v = out[1, 0, 1][4]
    assert v[5] != 0 and int(np.abs(np.delete(v, 5)).sum()) == 0


def test_keygen_into_unaligned_secret_key_buffer(gpu):
    """the fused keygen output stage stores whole dwords: a secret-key buffer that is not 4-byte aligned takes the unfused
    kernels instead -- same bytes"""
    import ctypes as C
    from dilithium_amd import api, lib as dlib
    level, n = 3, 2304
    rng = np.random.default_rng(77)
    seed = dev(gpu, rng.integers(0, 256, (n, 32), dtype=np.uint8), np.uint8)
    pk, sk = api.keygen(seed, level)
    skb = api.sk_bytes(level)
    raw = gpu.zeros(n * skb + 8, dtype=gpu.uint8, device="cuda")
    pk2 = gpu.empty_like(pk)
    dlib.check(dlib.load().dil_keygen_dev(C.c_void_p(pk2.data_ptr()), C.c_void_p(raw.data_ptr() + 1), C.c_void_p(seed.data_ptr()), level, n, None))
    gpu.cuda.synchronize()
    assert (pk2 == pk).all() and (raw[1:1 + n * skb].view(n, skb) == sk).all() and int(raw[0]) == 0 and int(raw[1 + n * skb:].sum()) == 0


def test_options_api(gpu):
    from dilithium_amd import api, DilError
    assert api.get_option("fused_mode") == 0
    with pytest.raises(DilError):
        api.set_option("no_such_option", 1)
    api.set_option("zeroize", 1)
    assert api.get_option("zeroize") == 1
    api.set_option("zeroize", 0)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sign_phase2_early_small_batch_runs_the_full_phase2(gpu, oracle, level):
    """dil_sign_phase2_early_dev below the wave-per-item threshold: the workgroup-per-item kernel evaluates every check (all flag
    bits, z and h of every attempt) and leaves w0 alone"""
    from dilithium_amd import api
    n = 97
    A, y, c, s1h, s2h, t0h = sign_inputs(oracle, level, n, 8800 + level, 1)
    ow1, ow0 = oracle.sign_phase1(level, A, y)
    w0 = dev(gpu, ow0)
    z, h, fl = api.sign_phase2_early(dev(gpu, c), dev(gpu, y), w0, dev(gpu, ow1, np.uint8), dev(gpu, s1h), dev(gpu, s2h), dev(gpu, t0h),
                                     level, shared_key=True)
    oz, oh, ofl = oracle.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
    assert (fl.cpu().numpy() == ofl).all() and (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all()
    assert (w0.cpu().numpy() == ow0).all()
