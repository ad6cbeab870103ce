"""The host-pointer entry points (csrc/capi.hip host_inplace / host_binary / host_verify_core) under every setting of their options --
host_chunk (KiB per chunk), host_streams (1 .. 8 staging buffers), host_pin (1: a pageable caller buffer is page-locked for the call;
0: it goes through the library's own page-locked staging buffer in slices), host_duplex (one stream per direction from 8 chunks up) --
against the oracle and the device-pointer entry points; and the round-6 rule itself: NOTHING of a caller's buffer stays registered
with the driver after a call has returned (the runtime's cached page-locks of pageable copies killed the round-5 suite).  The reference's calling convention for the path is caller-owned HOST arrays
(reference_code/ref_ntt.h:30-36, hardware_code/ntt2x2.h:30-34); bench.py's `end_to_end` block times these calls."""
import numpy as np
import pytest

from oracle.oracle import Q, N, splitmix64_polys

pytestmark = pytest.mark.gpu


@pytest.fixture()
def host_opts(gpu):
    from dilithium_amd import api
    saved = {k: api.get_option(k) for k in ("host_chunk", "host_streams", "host_pin", "host_duplex")}
    yield lambda **kw: [api.set_option(k, v) for k, v in kw.items()]
    for k, v in saved.items():
        api.set_option(k, v)


@pytest.mark.parametrize("chunk,streams,pin", [(64, 1, 0), (64, 3, 1), (100, 8, 0), (16384, 3, 0), (1024, 2, 1)])
def test_ntt_host_chunked(gpu, oracle, host_opts, chunk, streams, pin):
    from dilithium_amd import api
    host_opts(host_chunk=chunk, host_streams=streams, host_pin=pin)
    n = 3 * chunk + 17 if chunk < 4096 else 20000
    a = splitmix64_polys(n, seed=chunk + streams)
    x = a.copy()
    api.ntt(x)
    idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 97)), [n - 1, chunk - 1 if chunk < n else 0, min(chunk, n - 1)]]))
    assert (x[idx] == oracle.ntt(a[idx])).all()
    api.invntt(x)
    assert (x == a).all()


@pytest.mark.parametrize("duplex,pin,locked", [(1, 1, False), (0, 1, False), (1, 0, False), (1, 0, True), (0, 0, True)])
@pytest.mark.parametrize("chunk,bufs", [(64, 1), (64, 2), (100, 4), (600, 2), (8192, 3)])
def test_ntt_host_pipelines_ring_wraps_and_ragged_tails(gpu, oracle, host_opts, duplex, pin, locked, chunk, bufs):
    """one stream per direction / round-robin over the streams on a buffer page-locked for the call (pin) or by the caller (locked: torch
    pin_memory), and the staged slices of a buffer that is neither (pin 0) -- with the staging ring wrapping several times (9+ chunks over
    1 .. 4 buffers) and a ragged last chunk, against the oracle and the round trip"""
    from dilithium_amd import api
    host_opts(host_chunk=chunk, host_streams=bufs, host_pin=pin, host_duplex=duplex)
    n = 9 * chunk + 17 if chunk < 4096 else 20011         # (chunk 600: 5417 polynomials, past the 4096 below which a page-locked buffer
    a = splitmix64_polys(n, seed=chunk + bufs)            #  is not treated as one, and >= 8 chunks: the one-stream-per-direction pipeline)
    if locked:
        keep = gpu.empty((n, N), dtype=gpu.int32).pin_memory()
        x = keep.numpy()
        x[:] = a
    else:
        x = a.copy()
    api.ntt(x)
    idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 97)), [n - 1, chunk - 1, min(chunk, n - 1), n - 17, n - 18]]))
    assert (x[idx] == oracle.ntt(a[idx])).all()
    api.invntt(x)
    assert (x == a).all()


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("chunk,streams,pin", [(128, 3, 0), (300, 1, 1), (16384, 3, 0)])
def test_verify_core_host_vs_device_and_oracle(gpu, oracle, host_opts, level, shared, chunk, streams, pin):
    from dilithium_amd import api
    K, L = {2: (4, 4), 3: (6, 5), 5: (8, 7)}[level]
    n = 77
    nk = 1 if shared else n
    rng = np.random.default_rng(level * 10 + chunk)
    A = splitmix64_polys(nk * K * L, seed=level).reshape(nk, K, L, N)
    g1 = 1 << (17 if level == 2 else 19)
    z = np.mod(rng.integers(-g1 + 1, g1 + 1, (n, L, N)), Q).astype(np.int32)
    c = np.zeros((n, N), np.int32)
    for i in range(n):
        pos = rng.choice(N, 39, replace=False)
        c[i, pos] = np.where(rng.random(39) < 0.5, 1, Q - 1)
    t1 = rng.integers(0, 1024, (nk, K, N)).astype(np.int32)
    h = (rng.random((n, K, N)) < 0.03).astype(np.uint8)
    host_opts(host_chunk=chunk, host_streams=streams, host_pin=pin)
    w1 = api.verify_core(A, z, c, t1, h.reshape(n, K * N), level, shared_pk=shared)
    cu = lambda x: gpu.from_numpy(x).cuda()  # noqa: E731
    dev = api.verify_core(cu(A), cu(z), cu(c), cu(t1), cu(h), level, shared_pk=shared).cpu().numpy()
    assert (w1.reshape(dev.shape) == dev).all()
    Ab, tb = (np.broadcast_to(A, (n, K, L, N)), np.broadcast_to(t1, (n, K, N))) if shared else (A, t1)
    assert (dev == oracle.verify_core(level, np.ascontiguousarray(Ab), z, c, np.ascontiguousarray(tb), h)).all()


def test_host_pipelines_from_concurrent_threads(gpu, oracle):
    """three host threads in the *_host transforms at once (pageable buffers large enough to be page-locked for the call, a page-locked one,
    small ones through the staging buffer) beside a fourth on the device-pointer scheme calls: the host entry points of a device serialise on
    their own lock, every result is the oracle's"""
    import threading
    from dilithium_amd import api
    torch = gpu
    errors = []
    n = 20000
    srcs = [splitmix64_polys(n, seed=40 + i) for i in range(3)]
    keep = torch.empty((n, N), dtype=torch.int32).pin_memory()
    idx = np.arange(0, n, 211)
    want = [oracle.ntt(s[idx]) for s in srcs]

    def host_worker(i):
        try:
            x = keep.numpy() if i == 2 else srcs[i].copy()
            for rep in range(5):
                x[:] = srcs[i]
                api.ntt(x)
                assert (x[idx] == want[i]).all(), ("forward", i, rep)
                api.invntt(x)
                assert (x == srcs[i]).all(), ("round trip", i, rep)
                y = srcs[i][:50 + rep].copy()
                api.ntt(y)
                api.invntt(y)
                assert (y == srcs[i][:50 + rep]).all(), ("small", i, rep)
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    def dev_worker():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                g = torch.Generator(device="cuda").manual_seed(1)
                u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
                seed, mu = u8(300, 32), u8(300, 64)
                for _ in range(5):
                    pk, sk = api.keygen(seed, 3)
                    sig, _ = api.sign(sk, mu, 3)
                    assert int(api.verify_sig(pk, sig, mu, 3).abs().sum()) == 0
                st.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(("dev", repr(e)))

    ths = [threading.Thread(target=host_worker, args=(i,)) for i in range(3)] + [threading.Thread(target=dev_worker)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors


def _registered_with_the_driver(arr):
    """VMAs overlapping the array that carry the `dc` (VM_DONTCOPY) flag in /proc/self/smaps: the thunk marks every host range it registers
    with the driver MADV_DONTFORK -- a page-lock the runtime made for a pageable copy, or hipHostRegister -- and clears the mark when the
    registration goes"""
    lo, hi = arr.ctypes.data, arr.ctypes.data + arr.nbytes
    hits, cur = [], None
    for ln in open("/proc/self/smaps"):
        head = ln.split(" ", 1)[0]
        if "-" in head and ln[:1] in "0123456789abcdef":
            a, b = (int(x, 16) for x in head.split("-"))
            cur = (a, b)
        elif ln.startswith("VmFlags:") and cur and cur[0] < hi and cur[1] > lo and " dc" in ln:
            hits.append(cur)
    return hits


def test_nothing_of_the_callers_buffer_stays_registered_after_a_call(gpu, oracle, host_opts):
    """The reference's contract -- the caller owns every buffer, the callee retains nothing (reference_code/ref_ntt.h:30-36) -- down to the
    driver: after dil_ntt_host / dil_invntt_host / dil_pointwise_host / dil_polymul_host / dil_verify_core_host have returned, no page of the
    caller's heap arrays is still registered (VM_DONTCOPY in /proc/self/smaps), whatever the size class and option.  Rounds 4-5 handed the
    caller's pageable pointer to hipMemcpyAsync on the library's private streams: the runtime page-locked the range and kept the lock in
    that stream's cache for the life of the process (this test fails on that library) -- the state in which a later copy of the application
    to the same heap addresses faulted on the GPU (profiles/r06_suite_crash_rootcause.txt)."""
    import ctypes
    from dilithium_amd import api
    libc = ctypes.CDLL("libc.so.6")
    libc.mallopt(-3, 1 << 30)                       # M_MMAP_THRESHOLD: the arrays below come from the heap proper, like the suite's
    for pin, duplex in ((1, 1), (1, 0), (0, 1)):
        host_opts(host_pin=pin, host_duplex=duplex, host_chunk=8192, host_streams=4)
        for n in (3, 300, 5000, 20000, 70000):
            a = splitmix64_polys(n, seed=n)
            x, b = a.copy(), splitmix64_polys(n, seed=n + 1)
            api.ntt(x)
            api.invntt(x)
            assert (x == a).all()
            c = np.empty_like(a)
            api.pointwise_barrett(c, a, b)
            api.polymul(c, a, b)
            left = [r for arr in (x, a, b, c) for r in _registered_with_the_driver(arr)]
            assert not left, (pin, duplex, n, [(hex(lo), hex(hi)) for lo, hi in left])
    K, L, n = 6, 5, 700
    rng = np.random.default_rng(1)
    A = splitmix64_polys(n * K * L, seed=5).reshape(n, K, L, N)
    z = np.mod(rng.integers(-(1 << 19) + 1, 1 << 19, (n, L, N)), Q).astype(np.int32)
    cc = np.zeros((n, N), np.int32)
    cc[:, ::7] = 1
    t1 = rng.integers(0, 1024, (n, K, N)).astype(np.int32)
    h = (rng.random((n, K * N)) < 0.03).astype(np.uint8)
    for pin in (1, 0):
        host_opts(host_pin=pin)
        w1 = api.verify_core(A, z, cc, t1, h, 3)
        left = [r for arr in (A, z, cc, t1, h, w1) for r in _registered_with_the_driver(arr)]
        assert not left, (pin, [(hex(lo), hex(hi)) for lo, hi in left])
