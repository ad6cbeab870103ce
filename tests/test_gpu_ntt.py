"""GPU parity: the HIP transforms / element-wise ops / bram API through the C-ABI vs the oracle
and the committed reference goldens.  Bit-exact (canonical residues) -- integer work."""
import numpy as np
import pytest

from oracle.oracle import AFTER_INVNTT, AFTER_NTT, NATURAL, N, Q, canon, splitmix64_polys

pytestmark = pytest.mark.gpu


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()


def host(t):
    return t.cpu().numpy()


def test_native_library_is_loaded(gpu):
    """the tests below must run the in-tree HIP library, not a fallback"""
    import dilithium_amd
    lib = dilithium_amd.load()
    assert lib.dil_num_cus() > 0
    maps = open("/proc/self/maps").read()
    assert "dilithium_amd/libdil256.so" in maps


@pytest.mark.parametrize("op", ["ntt", "invntt", "ntt2x2_ref", "invntt2x2_ref"])
def test_golden_vectors(gpu, golden, op):
    from dilithium_amd import api
    t = dev(gpu, golden["a"])
    getattr(api, op)(t)
    key = {"ntt": "ntt", "invntt": "invntt", "ntt2x2_ref": "ntt2x2", "invntt2x2_ref": "invntt2x2"}[op]
    assert (host(t) == canon(golden[key])).all()


@pytest.mark.parametrize("batch", [0, 1, 2, 3, 5, 63, 64, 257, 1001, 8191])
def test_ragged_batches_vs_oracle(gpu, oracle, batch):
    from dilithium_amd import api
    a = splitmix64_polys(max(batch, 1), seed=100 + batch)[:batch]
    f, i = dev(gpu, a), dev(gpu, a)
    api.ntt(f)
    api.invntt(i)
    if batch:
        assert (host(f) == oracle.ntt(a)).all()
        assert (host(i) == oracle.invntt(a)).all()


def test_edge_values(gpu, oracle):
    from dilithium_amd import api
    rows = [np.full(N, Q - 1), np.full(N, -(Q - 1)), np.zeros(N), np.full(N, 1), np.full(N, -1),
            np.tile([0, Q - 1], 128), np.tile([Q - 1, -(Q - 1)], 128), np.arange(N), -np.arange(N)]
    for idx in (0, 1, 63, 64, 127, 128, 255):
        e = np.zeros(N)
        e[idx] = Q - 1
        rows.append(e)
    a = np.array(rows, dtype=np.int32)
    f, i = dev(gpu, a), dev(gpu, a)
    api.ntt(f)
    api.invntt(i)
    assert (host(f) == oracle.ntt(a)).all()
    assert (host(i) == oracle.invntt(a)).all()


def test_forward_value_domain_edge(gpu, oracle):
    """include/dil256.h: the forward transforms accept any int32 with |x| < 2^31 - 7q (lazy butterflies widen a value
    by < 6q in total, the final reduction needs 2^22 of headroom).  Probe the edge itself, both signs, all-equal /
    alternating / random-near-the-edge rows; expected = the oracle on the canonical residues (the map is linear mod q)."""
    from dilithium_amd import api
    edge = (1 << 31) - 7 * Q
    rng = np.random.default_rng(3)
    rows = [np.full(N, edge - 1), np.full(N, -(edge - 1)), np.tile([edge - 1, -(edge - 1)], 128),
            np.tile([-(edge - 1), edge - 1, edge - 1, -(edge - 1)], 64)]
    rows += [(edge - 1 - rng.integers(0, 1 << 20, N)) * rng.choice([-1, 1], N) for _ in range(28)]
    a = np.array(rows, dtype=np.int64)
    assert np.abs(a).max() < edge
    f = dev(gpu, a.astype(np.int32))
    api.ntt(f)
    want = oracle.ntt(np.mod(a, Q).astype(np.int32))
    assert (host(f) == want).all()
    b = dev(gpu, a.astype(np.int32))
    api.ntt2x2_fwdntt(b.view(-1, 64, 4), 0)       # the bram flavour shares the butterfly core: same domain
    from oracle.oracle import canon
    assert (host(b).reshape(len(rows), -1) == canon(oracle.bram_fwdntt(np.mod(a, Q).astype(np.int32), 0))).all()


def test_signed_inputs(gpu, oracle):
    """the reference is correct for any |x| < q (SURVEY 8b value domain)"""
    from dilithium_amd import api
    a = splitmix64_polys(512, seed=77, lo=-(Q - 1), hi=Q)
    f, i = dev(gpu, a), dev(gpu, a)
    api.ntt(f)
    api.invntt(i)
    assert (host(f) == oracle.ntt(a)).all()
    assert (host(i) == oracle.invntt(a)).all()


def test_full_config2_batch_parity_and_roundtrip(gpu, oracle):
    """BASELINE.json configs[1]: batch = 65536.  All outputs vs the oracle, round trip == identity"""
    from dilithium_amd import api
    a = splitmix64_polys(65536, seed=0)
    t = dev(gpu, a)
    api.ntt(t)
    assert (host(t) == oracle.ntt(a)).all()
    api.invntt(t)
    assert (host(t) == a).all()
    api.invntt(t)
    assert (host(t) == oracle.invntt(a)).all()


def test_linearity_property(gpu):
    """size-independent property: NTT(a + b) == NTT(a) + NTT(b) (mod q), 1M polynomials"""
    from dilithium_amd import api
    torch = gpu
    n = 1 << 18
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randint(0, Q, (n, N), dtype=torch.int32, device="cuda", generator=g)
    b = torch.randint(0, Q, (n, N), dtype=torch.int32, device="cuda", generator=g)
    s = (a + b) % Q
    api.ntt(a), api.ntt(b), api.ntt(s)
    assert torch.equal((a + b) % Q, s)
    assert int(s.min()) >= 0 and int(s.max()) < Q


def test_pointwise_and_mac_at_the_product_extremes(gpu):
    """(+-(q-1)) x (+-(q-1)) and acc = +-(q-1) on EVERY lane (ref_ntt.cpp:49-57, butterfly.v:144-150): the largest 64-bit products
    mont_mul's bound argument (modarith.hpp) has to cover, with the accumulator at both ends; against independent integer arithmetic"""
    from dilithium_amd import api
    torch = gpu
    ext = [Q - 1, -(Q - 1), 1, -1, 0, (Q - 1) // 2, -((Q - 1) // 2), (Q + 1) // 2]
    rows_a, rows_b, rows_c = [], [], []
    for x in ext:
        for y in ext:
            for z in (Q - 1, -(Q - 1), 0):
                rows_a.append(np.full(256, x)), rows_b.append(np.full(256, y)), rows_c.append(np.full(256, z))
    # and every lane its own mix of the extremes
    rng = np.random.default_rng(3)
    for _ in range(16):
        rows_a.append(rng.choice(ext, 256)), rows_b.append(rng.choice(ext, 256)), rows_c.append(rng.choice([Q - 1, -(Q - 1)], 256))
    a, b, acc = (np.stack(r).astype(np.int32) for r in (rows_a, rows_b, rows_c))
    ta, tb, tacc = dev(torch, a), dev(torch, b), dev(torch, acc)
    tc = torch.empty_like(ta)
    prod = a.astype(np.int64) * b.astype(np.int64)
    api.pointwise_barrett(tc, ta, tb)
    assert (host(tc) == np.mod(prod, Q)).all()
    api.pointwise_acc(tc, tacc, ta, tb)
    assert (host(tc) == np.mod(acc.astype(np.int64) + prod, Q)).all()
    api.poly_add(tc, ta, tb)
    assert (host(tc) == np.mod(a.astype(np.int64) + b, Q)).all()
    api.poly_sub(tc, ta, tb)
    assert (host(tc) == np.mod(a.astype(np.int64) - b, Q)).all()
    for mapping in (0, 1, 2):                       # the hardware model's MUL on `bram` under every MAPPING, same extremes
        ram = ta.clone()
        api.ntt2x2_mul(ram, tb, mapping)
        r = np.arange(64)
        row = r if mapping == 0 else ((r % 4) * 16 + r // 4 if mapping == 1 else (r % 16) * 4 + r // 16)
        want = a.reshape(-1, 64, 4).astype(np.int64).copy()
        want[:, row, :] = np.mod(want[:, row, :] * b.reshape(-1, 64, 4).astype(np.int64), Q)
        assert (host(ram).reshape(-1, 64, 4) == np.mod(want, Q)).all(), mapping


def test_pointwise_mac_add_sub(gpu, oracle, golden):
    from dilithium_amd import api
    torch = gpu
    a, b = golden["a"], golden["b"]
    ta, tb = dev(torch, a), dev(torch, b)
    tc = torch.empty_like(ta)
    api.pointwise_barrett(tc, ta, tb)
    assert (host(tc) == canon(golden["pointwise"])).all()
    # aliasing c == a (ntt2x2_test.cpp:102)
    t2 = ta.clone()
    api.pointwise_barrett(t2, t2, tb)
    assert (host(t2) == canon(golden["pointwise"])).all()
    acc = splitmix64_polys(a.shape[0], seed=5, lo=-(Q - 1), hi=Q)
    tacc = dev(torch, acc)
    api.pointwise_acc(tc, tacc, ta, tb)
    want = np.mod(acc.astype(np.int64) + a.astype(np.int64) * b.astype(np.int64), Q)
    assert (host(tc) == want).all()
    api.poly_add(tc, ta, tb)
    assert (host(tc) == np.mod(a.astype(np.int64) + b, Q)).all()
    api.poly_sub(tc, ta, tb)
    assert (host(tc) == np.mod(a.astype(np.int64) - b, Q)).all()
    # large random batch vs oracle
    x, y = splitmix64_polys(4099, seed=8), splitmix64_polys(4099, seed=9, lo=-(Q - 1), hi=Q)
    tx, ty = dev(torch, x), dev(torch, y)
    api.pointwise_barrett(tx, tx, ty)
    assert (host(tx) == oracle.pointwise(x, y)).all()


@pytest.mark.parametrize("mapping", [NATURAL, AFTER_NTT, AFTER_INVNTT])
def test_bram_api_vs_reference_golden(gpu, golden, mapping):
    from dilithium_amd import api
    ram, mul = golden["ram"], golden["mul_ram"]
    t = dev(gpu, ram)
    api.ntt2x2_fwdntt(t, mapping)
    assert (host(t) == canon(golden[f"bram_fwd_{mapping}"])).all()
    t = dev(gpu, ram)
    api.ntt2x2_invntt(t, mapping)
    assert (host(t) == canon(golden[f"bram_inv_{mapping}"])).all()
    t = dev(gpu, ram)
    api.ntt2x2_mul(t, dev(gpu, mul), mapping)
    assert (host(t) == canon(golden[f"bram_mul_{mapping}"])).all()


@pytest.mark.parametrize("mapping", [NATURAL, AFTER_NTT, AFTER_INVNTT])
def test_bram_all_ops_4096_random_and_edge_rows_vs_oracle(gpu, oracle, mapping):
    """all 9 (op x mapping) combinations of the hardware-model API (hardware_code/ntt2x2.h:30-34) well beyond the 16 polynomials of the
    goldens: 4096 seeded-random `bram`s plus the 21 edge rows of the golden set (ramp, zeros, all q-1, all -(q-1), ones, unit impulses,
    signed rows), every output row, against the oracle's bram model -- which tests/test_oracle.py pins live against the compiled
    reference (oracle/_ref/libref.so) on every mapping"""
    from dilithium_amd import api
    edge = [np.arange(N), np.zeros(N), np.full(N, Q - 1), np.full(N, -(Q - 1)), np.full(N, 1)]
    for idx in (0, 1, 2, 63, 64, 127, 128, 255):
        e = np.zeros(N)
        e[idx] = 1
        edge.append(e)
    ram = np.concatenate([np.array(edge, dtype=np.int32), splitmix64_polys(8, seed=31 + mapping, lo=-(Q - 1), hi=Q),
                          splitmix64_polys(4096, seed=41 + mapping)])
    mul = splitmix64_polys(ram.shape[0], seed=51 + mapping)
    assert ram.shape[0] == 4096 + 21
    t = dev(gpu, ram)
    api.ntt2x2_fwdntt(t.view(-1, 64, 4), mapping)
    assert (host(t) == canon(oracle.bram_fwdntt(ram, mapping))).all()
    t = dev(gpu, ram)
    api.ntt2x2_invntt(t.view(-1, 64, 4), mapping)
    assert (host(t) == canon(oracle.bram_invntt(ram, mapping))).all()
    t = dev(gpu, ram)
    api.ntt2x2_mul(t.view(-1, 64, 4), dev(gpu, mul).view(-1, 64, 4), mapping)
    assert (host(t) == canon(oracle.bram_mul(ram, mul, mapping))).all()


def test_bram_polymul_chain_like_reference_test(gpu, oracle, golden):
    """ntt2x2_test.cpp:109-137 polymul(): fwd, fwd, mul, inv under AFTER_NTT == plain product;
    b = 31 a as in the reference's main (:171-172)"""
    from dilithium_amd import api
    a = splitmix64_polys(2000, seed=31)
    b = np.mod(a.astype(np.int64) * 31, Q).astype(np.int32)
    ta, tb = dev(gpu, a), dev(gpu, b)
    api.ntt2x2_fwdntt(ta, NATURAL)
    api.ntt2x2_fwdntt(tb, NATURAL)
    assert (host(ta) == oracle.bram_fwdntt(a, NATURAL)).all()
    api.ntt2x2_mul(ta, tb, NATURAL)
    api.ntt2x2_invntt(ta, AFTER_NTT)
    plain = oracle.invntt(oracle.pointwise(oracle.ntt(a), oracle.ntt(b)))
    assert (host(ta) == plain).all()
    g = dev(gpu, golden["ram"])
    gm = dev(gpu, golden["mul_ram"])
    api.ntt2x2_fwdntt(g, NATURAL)
    api.ntt2x2_fwdntt(gm, NATURAL)
    api.ntt2x2_mul(g, gm, NATURAL)
    api.ntt2x2_invntt(g, AFTER_NTT)
    assert (host(g) == canon(golden["bram_polymul"])).all()


def test_host_pointer_entry_points(gpu, oracle):
    """the *_host entry points (what the reference-signature wrappers call)"""
    from dilithium_amd import api
    a = splitmix64_polys(33, seed=12)
    x = a.copy()
    api.ntt(x)
    assert (x == oracle.ntt(a)).all()
    api.invntt(x)
    assert (x == a).all()
    b = splitmix64_polys(33, seed=13)
    c = np.empty_like(a)
    api.pointwise_barrett(c, a, b)
    assert (c == oracle.pointwise(a, b)).all()
    r = a.copy()
    api.ntt2x2_fwdntt(r, NATURAL)
    assert (r == oracle.bram_fwdntt(a, NATURAL)).all()


def _negacyclic_schoolbook(a, b):
    """a * b mod (x^256 + 1, q) by the definition, for a handful of rows (int64 exact: 256 products < 2^46 each)"""
    a, b = a.astype(np.int64) % Q, b.astype(np.int64) % Q
    out = np.zeros_like(a)
    for i in range(N):
        prod = (a[:, i:i + 1] * b) % Q                       # x^i a_i * b
        out[:, i:] += prod[:, :N - i]
        out[:, :i] -= prod[:, N - i:]
    return np.mod(out, Q).astype(np.int32)


def test_polymul_fused_vs_oracle_chain_and_the_definition(gpu, oracle):
    """dil_polymul_dev == the reference's chain ntt, ntt, pointwise_barrett, invntt (ntt2x2_test.cpp:109-137) on 20000 + edge pairs, every
    coefficient; == the negacyclic product by its definition on 40 rows; aliasing c = a and c = b; operands in (-q, q) as the reference's"""
    from dilithium_amd import api
    edge = [np.zeros(N), np.full(N, Q - 1), np.full(N, -(Q - 1)), np.full(N, 1), np.arange(N), np.full(N, (Q - 1) // 2)]
    for idx in (0, 1, 255):
        e = np.zeros(N)
        e[idx] = 1
        edge.append(e)
    a = np.concatenate([np.array(edge, dtype=np.int32), splitmix64_polys(20000, seed=61), splitmix64_polys(500, seed=62, lo=-(Q - 1), hi=Q)])
    b = np.concatenate([splitmix64_polys(len(edge), seed=63), splitmix64_polys(20000, seed=64), splitmix64_polys(500, seed=65, lo=-(Q - 1), hi=Q)])
    b[:len(edge)][3] = Q - 1
    want = oracle.invntt(oracle.pointwise(oracle.ntt(a), oracle.ntt(b)))
    ta, tb = dev(gpu, a), dev(gpu, b)
    tc = gpu.empty_like(ta)
    api.polymul(tc, ta, tb)
    got = host(tc)
    assert (got == want).all()
    assert (host(ta) == a).all() and (host(tb) == b).all()                 # operands untouched
    rows = np.r_[0:len(edge), 1000:1020, 20300:20311]
    assert (got[rows] == _negacyclic_schoolbook(a[rows], b[rows])).all()
    api.polymul(ta, ta, tb)                                                # c aliases a
    assert (host(ta) == want).all()
    ta = dev(gpu, a)
    api.polymul(tb, ta, tb)                                                # c aliases b
    assert (host(tb) == want).all()
    # the unfused chain through the separate entry points agrees too
    xa, xb = dev(gpu, a), dev(gpu, b)
    api.ntt(xa)
    api.ntt(xb)
    api.pointwise_barrett(xa, xa, xb)
    api.invntt(xa)
    assert (host(xa) == want).all()


@pytest.mark.parametrize("n", [1, 2, 77, 2048, 2049, 9000, 20011])
def test_polymul_host_one_round_trip(gpu, oracle, n):
    """dil_polymul_host: host arrays in, host array out (pageable arrays in slices of 2048 pairs through the ring of page-locked slots;
    batch 1 through the mailbox when it is on) == the oracle chain"""
    from dilithium_amd import api
    a, b = splitmix64_polys(n, seed=70 + n), splitmix64_polys(n, seed=71 + n, lo=-(Q - 1), hi=Q)
    want = oracle.invntt(oracle.pointwise(oracle.ntt(a), oracle.ntt(b)))
    c = np.empty_like(a)
    api.polymul(c, a, b)
    assert (c == want).all()
    x = a.copy()
    api.polymul(x, x, b)                                                   # in place
    assert (x == want).all()
    if n == 1:
        saved = api.get_option("host_mailbox")
        try:
            api.set_option("host_mailbox", 1)
            for rep in range(50):                                          # the resident wave serves the whole chain as one request
                c2 = np.empty_like(a)
                api.polymul(c2, a, b)
                assert (c2 == want).all()
        finally:
            api.set_option("host_mailbox", saved)


def test_polymul_host_chunk_and_stream_options(gpu, oracle):
    from dilithium_amd import api
    saved = {k: api.get_option(k) for k in ("host_chunk", "host_streams", "host_copy_threads")}
    try:
        a, b = splitmix64_polys(5000, seed=81), splitmix64_polys(5000, seed=82)
        want = oracle.invntt(oracle.pointwise(oracle.ntt(a), oracle.ntt(b)))
        for chunk, streams, threads in ((128, 1, 1), (600, 3, 3), (1000, 8, 2), (8192, 4, 3)):
            for k, v in (("host_chunk", chunk), ("host_streams", streams), ("host_copy_threads", threads)):
                api.set_option(k, v)
            c = np.empty_like(a)
            api.polymul(c, a, b)
            assert (c == want).all(), (chunk, streams, threads)
    finally:
        for k, v in saved.items():
            api.set_option(k, v)
