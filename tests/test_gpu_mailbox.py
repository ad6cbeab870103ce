"""The host mailbox (include/dil256.h "HOST MAILBOX"): batch-of-one *_host calls served by one resident wave instead of a launch
per call.  Parity with the oracle / the compiled reference for every operation and `bram` mapping, the retirement / relaunch
protocol, a busy mailbox falling back to the launch path, and the reference's UNCHANGED hardware_code/ntt2x2_test.cpp at its own
10^6 iterations ("4 x 10^6 transforms incl. polymul") through libdil256_ref.so."""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

from oracle.oracle import splitmix64_polys

Q = 8380417
pytestmark = pytest.mark.gpu
i32p = C.POINTER(C.c_int32)


def P(a):
    return a.ctypes.data_as(i32p)


@pytest.fixture()
def mbox(gpu):
    import dilithium_amd
    from dilithium_amd import lib as dlib
    L = dilithium_amd.load()
    dlib.check(L.dil_set_option(b"host_mailbox", 1), "set_option")
    yield L
    dlib.check(L.dil_set_option(b"host_mailbox", 0), "set_option")
    dlib.check(L.dil_set_option(b"mailbox_idle_us", 200), "set_option")


def stats(L):
    c, n, a = C.c_uint64(), C.c_uint64(), C.c_int()
    assert L.dil_mailbox_stats(C.byref(c), C.byref(n), C.byref(a)) == 0
    return c.value, n.value, a.value


def test_mailbox_transforms_and_pointwise_vs_oracle(mbox, oracle):
    """ntt / invntt / pointwise through the mailbox == the oracle (== the compiled reference, tests/test_oracle.py) on edge and
    random polynomials; the calls were really served by the resident wave (stats), which was launched a handful of times"""
    L = mbox
    polys = np.concatenate([splitmix64_polys(40, seed=11), splitmix64_polys(8, seed=12, lo=-(Q - 1), hi=Q),
                            np.array([np.zeros(256), np.full(256, Q - 1), np.full(256, -(Q - 1)), np.arange(256)], dtype=np.int32)])
    c0, n0, _ = stats(L)
    want_f, want_i = np.mod(oracle.ntt(polys), Q), np.mod(oracle.invntt(polys), Q)
    for k, a in enumerate(polys):
        x = a.copy()
        assert L.dil_ntt_host(P(x), 1) == 0
        assert (x == want_f[k]).all(), k
        y = a.copy()
        assert L.dil_invntt_host(P(y), 1) == 0
        assert (y == want_i[k]).all(), k
    b = splitmix64_polys(len(polys), seed=13, lo=-(Q - 1), hi=Q)
    want_p = np.mod(oracle.pointwise(polys, b), Q)
    for k in range(len(polys)):
        c = np.empty(256, np.int32)
        assert L.dil_pointwise_host(P(c), P(polys[k].copy()), P(b[k].copy()), 1) == 0
        assert (c == want_p[k]).all(), k
        a = polys[k].copy()                      # c may alias a (ntt2x2_test.cpp:102)
        assert L.dil_pointwise_host(P(a), P(a), P(b[k].copy()), 1) == 0
        assert (a == want_p[k]).all(), k
    c1, n1, _ = stats(L)
    assert c1 - c0 == 4 * len(polys)
    assert 1 <= n1 - n0 <= 8, (n0, n1)           # resident across back-to-back calls (Python-paced: a few idle retirements at most)


@pytest.mark.parametrize("mapping", [0, 1, 2])
def test_mailbox_bram_ops_equal_the_launch_path(mbox, mapping):
    """the hardware-model API on one `bram` under every MAPPING: mailbox == launch path (itself == the compiled reference's
    goldens, tests/test_gpu_ntt.py) bit for bit"""
    from dilithium_amd import lib as dlib
    L = mbox
    rams = splitmix64_polys(12, seed=20 + mapping)
    muls = splitmix64_polys(12, seed=30 + mapping)
    for fn, two in ((L.dil_bram_fwdntt_host, False), (L.dil_bram_invntt_host, False), (L.dil_bram_mul_host, True)):
        got = {}
        for mode in (1, 0):
            dlib.check(L.dil_set_option(b"host_mailbox", mode), "set_option")
            out = []
            for k in range(len(rams)):
                r = rams[k].copy()
                rc = fn(P(r), P(muls[k].copy()), 1, mapping) if two else fn(P(r), 1, mapping)
                assert rc == 0
                out.append(r)
            got[mode] = np.stack(out)
        assert (got[0] == got[1]).all(), (fn.__name__, mapping)
    dlib.check(L.dil_set_option(b"host_mailbox", 1), "set_option")


def test_mailbox_retires_and_relaunches(mbox, oracle):
    """the resident wave leaves after the idle time (so a device-wide synchronisation is never held for long), and the next call
    brings it back; a burst of calls in between is served by ONE launch"""
    import torch
    from dilithium_amd import lib as dlib
    L = mbox
    dlib.check(L.dil_set_option(b"mailbox_idle_us", 20000), "set_option")      # 20 ms: a Python-paced burst stays resident
    a = splitmix64_polys(1, seed=5)[0]
    want = np.mod(oracle.ntt(a[None])[0], Q)
    x = a.copy()
    assert L.dil_ntt_host(P(x), 1) == 0 and (x == want).all()
    _, n0, alive = stats(L)
    assert alive == 1
    for _ in range(200):
        x = a.copy()
        assert L.dil_ntt_host(P(x), 1) == 0
    assert (x == want).all()
    assert stats(L)[1] == n0                       # no relaunch during the burst
    t0 = time.time()
    torch.cuda.synchronize()                       # waits for the wave to retire: bounded by the idle time
    assert time.time() - t0 < 1.0
    assert stats(L)[2] == 0
    x = a.copy()
    assert L.dil_ntt_host(P(x), 1) == 0 and (x == want).all()
    assert stats(L)[1] == n0 + 1                   # relaunched by the first call after retirement
    dlib.check(L.dil_set_option(b"mailbox_idle_us", 50), "set_option")         # 50 us: every Python-paced call finds it retired
    for _ in range(20):
        x = a.copy()
        assert L.dil_ntt_host(P(x), 1) == 0 and (x == want).all()
        time.sleep(0.002)


def test_mailbox_busy_falls_back_to_launch_path(mbox, oracle):
    """two host threads call at once: one owns the mailbox, the other takes the launch path; both get the right answer"""
    L = mbox
    polys = splitmix64_polys(2, seed=8)
    want = np.mod(oracle.ntt(polys), Q)
    errs = []

    def work(k):
        for _ in range(300):
            x = polys[k].copy()
            if L.dil_ntt_host(P(x), 1) != 0 or not (x == want[k]).all():
                errs.append(k)
                return

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs


def test_reference_unchanged_hw_main_at_its_own_iteration_count(gpu):
    """hardware_code/ntt2x2_test.cpp:139-197, UNCHANGED, linked against the drop-in alone (oracle/Makefile dropin_mains): 10^6
    iterations x {MUL, NTT, INVNTT, polymul}, ~1.4 x 10^7 batch-of-one calls, each compared by the main itself with the reference's
    software model -- which in this binary is the drop-in too, so the test also compares the `bram` path with the plain path on
    every iteration.  Needs the mailbox: at a launch per call this run takes ~10 minutes."""
    from oracle import oracle as orc
    exe = os.path.join(os.path.dirname(orc.__file__), "_ref", "ntt2x2_test_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ntt2x2_test_dropin not built (needs /root/reference)")
    t0 = time.time()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=175)
    dt = time.time() - t0
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.strip().endswith("OK") and "ERROR" not in out.stdout, out.stdout[-2000:]
    print(f"ntt2x2_test_dropin: {dt:.1f} s\n{out.stdout[-400:]}")
