"""CPU: (1) the numpy wave model of the kernels' dataflow == oracle, with the twiddle tables
taken from the C++ host library; (2) the C-ABI library loads, exports every symbol that
include/dil256.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle.oracle import Q, splitmix64_polys
from tests.model import wave_model as wm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import dilithium_amd
    return dilithium_amd.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dil256.h")).read()
    names = set(re.findall(r"\b(dil_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    from dilithium_amd.lib import SIGNATURES
    assert names == set(SIGNATURES), names ^ set(SIGNATURES)
    for n in names:
        assert hasattr(lib, n), f"libdil256.so does not export {n}"


def test_host_tables_match_model(lib):
    """the C++ host library's twiddle tables (Montgomery form) == the model's, bit for bit"""
    u32p = C.POINTER(C.c_uint32)
    f, i, ip = (np.zeros(2048, np.uint32) for _ in range(3))
    lib.dil_host_twiddle_tables(f.ctypes.data_as(u32p), i.ctypes.data_as(u32p), ip.ctypes.data_as(u32p))
    # device layout is [pass][half][lane][4]; the model keeps [pass][lane][8]
    dev = lambda t: t.reshape(4, 2, 64, 4).transpose(0, 2, 1, 3).reshape(4, 64, 8)  # noqa: E731
    assert (dev(f) == wm.table_u32(wm.FWD)).all()
    assert (dev(i) == wm.table_u32(wm.INV)).all()
    assert (dev(ip) == wm.table_u32(wm.INV_PIPE)).all()


def test_montgomery_constants():
    """constants hard-coded in modarith.hpp"""
    src = open(os.path.join(ROOT, "dilithium_amd/csrc/modarith.hpp")).read()
    wt, wq = wm.mont_const((1 << 32) % Q)          # multiplying by 2^32 cancels one Montgomery 2^-32
    assert f"R2_WT = {wt};" in src and f"R2_WQ = {wq}u;" in src
    assert f"QINV = {wm.QINV}u;" in src and (wm.QINV * Q) % (1 << 32) == 1
    assert (wm.F256 * 256) % Q == 1 and f"F256 = {wm.F256};" in src


def test_host_zetas_match_rom_and_oracle(lib, oracle):
    z = np.zeros(256, np.int32)
    lib.dil_host_zetas(z.ctypes.data_as(C.POINTER(C.c_int32)))
    assert (z == oracle.zetas()).all()
    rom = np.array([int(x, 16) for x in open(os.path.join(ROOT, "tests/golden/zetas_rom.txt")).read().split()])
    assert (np.mod(z.astype(np.int64), Q) == rom).all()


def test_wave_model_matches_oracle(oracle):
    polys = np.concatenate([
        splitmix64_polys(24, seed=3), splitmix64_polys(8, seed=4, lo=-(Q - 1), hi=Q),
        np.array([np.full(256, Q - 1), np.full(256, -(Q - 1)), np.zeros(256), np.arange(256),
                  np.tile([0, Q - 1], 128), np.tile([Q - 1, -(Q - 1)], 128)], dtype=np.int32)])
    good, goodi = oracle.ntt(polys), oracle.invntt(polys)
    for k, a in enumerate(polys):
        assert (wm.ntt_wave(a)[1] == good[k]).all()
        assert (wm.invntt_wave(a)[1] == goodi[k]).all()
    # widest forward input domain, and the fused mat-vec dataflow incl. its worst-case bounds
    for v in (2**31 - 6 * Q - 1, -(2**31 - 6 * Q - 1), 20 * Q + 5):
        a = np.full(256, v)
        assert (wm.ntt_wave(a)[1] == oracle.ntt(np.mod(a, Q).astype(np.int32))).all()
    A = splitmix64_polys(30, seed=9).reshape(6, 5, 256)
    y = splitmix64_polys(5, seed=10)
    assert (wm.matvec_wave(A, y) == oracle.matvec(6, 5, A, y)[0]).all()
    A2 = np.full((8, 7, 256), Q - 1, dtype=np.int32)
    y2 = np.full((7, 256), Q - 1, dtype=np.int32)
    assert (wm.matvec_wave(A2, y2) == oracle.matvec(8, 7, A2, y2)[0]).all()


def test_no_cpu_fallback_without_gpu(lib):
    """on a box without a GPU every compute entry point must return a HIP error, never a result"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dilithium_amd import api, DilError
    a = np.arange(256, dtype=np.int32)
    before = a.copy()
    with pytest.raises(DilError):
        api.ntt(a)
    assert (a == before).all()
    with pytest.raises(DilError):
        api.init(0)


def test_wire_sizes_and_scheme_entry_points_without_gpu(lib):
    """size queries are pure host arithmetic (round-3 v3.1 wire formats); the whole-operation entry points fail
    loudly -- never produce bytes -- when there is no GPU"""
    sizes = {2: (1312, 2528, 2420), 3: (1952, 4000, 3293), 5: (2592, 4864, 4595)}
    for level, want in sizes.items():
        assert (lib.dil_pk_bytes(level), lib.dil_sk_bytes(level), lib.dil_sig_bytes(level)) == want
    assert lib.dil_pk_bytes(4) == 0
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dilithium_amd import api, DilError
    seed = np.zeros((1, 32), dtype=np.uint8)
    with pytest.raises(DilError):
        api.keygen_host(seed, 3)
    with pytest.raises(DilError):
        api.sign_host(np.zeros((1, 4000), dtype=np.uint8), np.zeros((1, 64), dtype=np.uint8), 3)
    with pytest.raises(DilError):
        api.verify_sig_host(np.zeros((1, 1952), dtype=np.uint8), np.zeros((1, 3293), dtype=np.uint8), np.zeros((1, 64), dtype=np.uint8), 3)


def test_use_hint_half_bucket_form_over_the_whole_field():
    """pipeline_common.hpp use_hint<LEVEL>: v = ceil(a / gamma2) - 1 by one multiply, a1 = (v + 1) >> 1, hinted value
    a1 + 1 - 2 (v & 1) -- the integer operations of the kernel restated in numpy -- against the round-3 Decompose / UseHint
    formulas (the oracle's orc_decompose is pinned to the RTL threshold map by tests/test_oracle.py) for EVERY
    a in [0, q), both hint values, all three levels"""
    import numpy as np
    q = 8380417
    a = np.arange(q, dtype=np.int64)
    for level in (2, 3, 5):
        g2 = (q - 1) // 88 if level == 2 else (q - 1) // 32
        t = (a + 127) >> 7
        if level == 2:
            t = (t * 11275 + (1 << 23)) >> 24
            t = np.where(t > 43, 0, t)
            m = 44
        else:
            t = ((t * 1025 + (1 << 21)) >> 22) & 15
            m = 16
        a0 = a - t * 2 * g2
        a0 = np.where(a0 > (q - 1) // 2, a0 - q, a0)
        want0, want1 = t, np.where(a0 > 0, (t + 1) % m, (t - 1) % m)
        # the kernel's arithmetic, 32-bit
        x = ((a - 1).astype(np.int32)) >> (9 if level == 2 else 8)
        prod = x.astype(np.int64) * (22551 if level == 2 else 32801)
        assert np.abs(prod).max() < 2 ** 31                       # the product fits the 32-bit multiply
        v = (prod.astype(np.int32)) >> (22 if level == 2 else 25)
        odd = v & 1
        for hint, want in ((0, want0), (1, want1)):
            n = ((v + 1) >> 1) + (((odd ^ 1) - odd) & (-hint))
            if level == 2:
                n = n + ((n >> 31) & 44)
                n = n - (((43 - n) >> 31) & 44)
            else:
                n = n & 15
            assert (n == want).all(), (level, hint)


def test_decompose_w0res_over_the_whole_field():
    """pipeline_common.hpp decompose_w0res<LEVEL> (sign phase 1's output stage; phase 2's exact small-integer paths rely on it): w1 =
    HighBits(a) and w0 = LowBits(a) AS A RESIDUE, computed without the centring step of Decompose -- the kernel's 32-bit operations
    restated in numpy -- against Decompose + canonicalisation (the oracle's orc_decompose formulas, pinned to the RTL threshold map by
    tests/test_oracle.py) for EVERY a in [0, q), both gamma2 values, including the wrap where a1 = 44 | 16 becomes 0"""
    import numpy as np
    q = 8380417
    a = np.arange(q, dtype=np.int64)
    for level in (2, 3):                 # (level 5 shares level 3's gamma2)
        g2 = (q - 1) // 88 if level == 2 else (q - 1) // 32
        # reference: Decompose, then LowBits as a residue
        t = (a + 127) >> 7
        if level == 2:
            t = (t * 11275 + (1 << 23)) >> 24
            t = t ^ (((43 - t) >> 63) & t)
        else:
            t = ((t * 1025 + (1 << 21)) >> 22) & 15
        a0 = a - t * 2 * g2
        a0 = a0 - ((((q - 1) // 2 - a0) >> 63) & q)                # centred into (-gamma2, gamma2] (and a - q for the wrapped bucket)
        want_w0 = np.where(a0 < 0, a0 + q, a0)
        # the kernel: uint32 / int32 arithmetic
        au = a.astype(np.uint32)
        tk = (au + np.uint32(127)) >> np.uint32(7)
        if level == 2:
            tk = (tk * np.uint32(11275) + np.uint32(1 << 23)) >> np.uint32(24)
            sg = ((np.int32(43) - tk.astype(np.int32)) >> 31).astype(np.uint32)           # sgn(43 - t): all-ones where t > 43
            tk = tk ^ (sg & tk)
        else:
            tk = ((tk * np.uint32(1025) + np.uint32(1 << 21)) >> np.uint32(22)) & np.uint32(15)
        r = au.astype(np.int32) - tk.astype(np.int32) * np.int32(2 * g2)
        w0 = (r + ((r >> 31) & np.int32(q))).astype(np.uint32)
        assert (tk.astype(np.int64) == t).all(), level
        assert (w0.astype(np.int64) == want_w0).all(), level
        assert int(w0.max()) < q
        wrapped = a >= q - g2                     # the last half bucket: a1 = 44 | 16 -> 0, LowBits = a - q, residue = a itself
        assert (t[wrapped] == 0).all() and (want_w0[wrapped] == a[wrapped]).all()


def test_every_option_is_documented_and_round_trips_without_gpu(lib):
    """dil_set_option / dil_get_option (include/dil256.h): every name in the library's option table is described in the header or
    in INTEGRATION.md, reads back what was set, and an unknown name is refused -- all without a GPU"""
    import ctypes as C
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = open(os.path.join(root, "dilithium_amd", "csrc", "capi.hip")).read()
    names = re.findall(r'\{"([a-z0-9_]+)", "DIL_[A-Z0-9_]+", &', table)
    assert len(names) >= 15 and "sign_skip" in names
    docs = open(os.path.join(root, "include", "dil256.h")).read() + open(os.path.join(root, "INTEGRATION.md")).read()
    for name in names:
        assert name in docs, f"option {name} is not documented"
        v = C.c_int(-1)
        assert lib.dil_get_option(name.encode(), C.byref(v)) == 0, name
        old = v.value
        assert lib.dil_set_option(name.encode(), old + 1) == 0
        assert lib.dil_get_option(name.encode(), C.byref(v)) == 0 and v.value == old + 1
        assert lib.dil_set_option(name.encode(), old) == 0
    v = C.c_int(0)
    assert lib.dil_get_option(b"no_such_option", C.byref(v)) != 0
    assert lib.dil_set_option(b"no_such_option", 1) != 0


def test_host_pipeline_plan_without_gpu(lib):
    """dil_host_plan: which form a dil_ntt_host-style call takes and in which chunks, by batch size, kind of caller memory and options
    (csrc/capi.hip host_plan; thresholds of the page-locked pipelines from profiles/r05t_host_batch_sweep.txt) -- pure host logic"""
    import ctypes as C
    ONE_SHOT, ROUND_ROBIN, DUPLEX, STAGED_RING = 0, 1, 2, 3
    names = ("host_chunk", "host_streams", "host_copy_threads", "host_duplex")
    saved = {}
    for n in names:
        v = C.c_int(0)
        assert lib.dil_get_option(n.encode(), C.byref(v)) == 0
        saved[n] = v.value

    def plan(batch, locked):
        p, c = C.c_int(-1), C.c_size_t(0)
        assert lib.dil_host_plan(C.c_size_t(batch), locked, C.byref(p), C.byref(c)) == 0
        return p.value, c.value

    try:
        for n, v in (("host_chunk", 8192), ("host_duplex", 1)):
            assert lib.dil_set_option(n.encode(), v) == 0
        # pageable: always through the library's own page-locked slots -- one slice up to 4 MiB, a ring of 4-MiB slices above
        assert plan(1, 0) == (ONE_SHOT, 1) and plan(4096, 0) == (ONE_SHOT, 4096)
        assert plan(4097, 0) == (STAGED_RING, 4096) and plan(65536, 0) == (STAGED_RING, 4096) and plan(1 << 20, 0) == (STAGED_RING, 4096)
        # page-locked by the caller: DMA in place -- one shot below 8 MiB, round-robin in 1-MiB chunks, one stream per direction from 64 MiB
        assert plan(4096, 1) == (ONE_SHOT, 4096) and plan(8191, 1) == (ONE_SHOT, 8191)
        assert plan(8192, 1) == (ROUND_ROBIN, 1024) and plan(65535, 1) == (ROUND_ROBIN, 1024)
        assert plan(65536, 1) == (DUPLEX, 8192) and plan(1 << 20, 1) == (DUPLEX, 8192)
        # the options take the pipelines away / move the chunk
        assert lib.dil_set_option(b"host_duplex", 0) == 0 and plan(65536, 1) == (ROUND_ROBIN, 8192)
        assert lib.dil_set_option(b"host_duplex", 1) == 0
        # small chunks (what the GPU tests use to wrap the rings)
        assert lib.dil_set_option(b"host_chunk", 64) == 0
        assert plan(64, 0) == (ONE_SHOT, 64) and plan(65, 0) == (STAGED_RING, 64) and plan(593, 0) == (STAGED_RING, 64)
        assert lib.dil_set_option(b"host_chunk_pinned", 600) == 0        # (the older name of the same option)
        v = C.c_int(0)
        assert lib.dil_get_option(b"host_chunk", C.byref(v)) == 0 and v.value == 600
        assert plan(5417, 1) == (DUPLEX, 600) and plan(4799, 1) == (ROUND_ROBIN, 600) and plan(4799, 0) == (STAGED_RING, 600)
        # gone with round 6: the helper-thread pipeline on pageable memory, page-locking the caller's buffer for a call
        assert lib.dil_set_option(b"host_threads", 2) != 0 and lib.dil_set_option(b"host_pin", 1) != 0
        for batch in (100, 1000, 5000, 20011, 70000, 300000):
            for locked in (0, 1):
                p, c = plan(batch, locked)
                assert 1 <= c <= batch and (p == ONE_SHOT) == (c == batch)
    finally:
        for n, v in saved.items():
            lib.dil_set_option(n.encode(), v)


def test_sign_round_plan_without_gpu(lib):
    """dil_sign_round_plan: the signing loop's speculation rule (csrc/scheme.hip sign_round_cap / sign_round_width) as the host sees it:
    S attempts per pending message so that a round keeps about `cap` entries in flight, never more than 64 per message (phase 2 reads an
    item's earlier attempts one per lane) nor than max_attempts allows, and only while the expected waste stays under option sign_waste"""
    import ctypes as C

    def plan(level, batch, pending, done=0, max_attempts=512):
        s, e = C.c_int(-1), C.c_size_t(0)
        rc = lib.dil_sign_round_plan(level, C.c_size_t(batch), C.c_size_t(pending), done, max_attempts, C.byref(s), C.byref(e))
        assert rc == 0, (level, batch, pending, rc)
        assert e.value == pending * s.value and 1 <= s.value <= 64
        return s.value

    saved = {}
    for n in ("sign_cap", "sign_waste"):
        v = C.c_int(0)
        assert lib.dil_get_option(n.encode(), C.byref(v)) == 0
        saved[n] = v.value
    try:
        assert lib.dil_set_option(b"sign_cap", 0) == 0 and lib.dil_set_option(b"sign_waste", 6144) == 0
        # the level-3 call of 8192 messages in profiles/r05s_sign_timeline.txt: 24576 / 21695 / 24288 / 2752 entries
        assert plan(3, 8192, 8192) == 3 and plan(3, 8192, 4339, 3) == 5 and plan(3, 8192, 1518, 8) == 16 and plan(3, 8192, 43, 24) == 64
        # first rounds: 4 / 3 / 2 attempts per message at levels 2 / 3 / 5 (about one expected signature's worth), capped at 32768 entries
        assert plan(2, 8192, 8192) == 4 and plan(5, 8192, 8192) == 2 and plan(3, 16384, 16384) == 2 and plan(3, 65536, 65536) == 1
        # small batches speculate for free: 16384 entries in flight, 64 attempts per message at most
        assert plan(3, 1, 1) == 64 and plan(3, 100, 100) == 64 and plan(3, 256, 256) == 64 and plan(3, 1024, 1024) == 16
        # max_attempts bounds the width of the last rounds
        assert plan(3, 1, 1, 0, 5) == 5 and plan(3, 8192, 43, 500, 512) == 12
        # the waste rule: a huge pending set stops widening once the attempts expected to be thrown away pass sign_waste
        assert lib.dil_set_option(b"sign_cap", 1 << 20) == 0
        assert plan(3, 65536, 65536) == 1 and plan(3, 65536, 20000) == 2 and plan(3, 65536, 6000) >= 8
        assert lib.dil_set_option(b"sign_waste", 1 << 30) == 0 and plan(3, 65536, 65536) == 16
        # an explicit cap
        assert lib.dil_set_option(b"sign_cap", 16384) == 0 and plan(3, 8192, 8192) == 2
        # refused: nothing pending, more pending than messages, no attempts left
        s, e = C.c_int(0), C.c_size_t(0)
        for args in ((3, 8192, 0, 0, 512), (3, 10, 11, 0, 512), (3, 10, 5, 7, 7), (4, 10, 5, 0, 512)):
            assert lib.dil_sign_round_plan(args[0], C.c_size_t(args[1]), C.c_size_t(args[2]), args[3], args[4], C.byref(s), C.byref(e)) != 0
    finally:
        for n, v in saved.items():
            lib.dil_set_option(n.encode(), v)
