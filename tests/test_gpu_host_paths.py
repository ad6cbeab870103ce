"""The host-pointer entry points (csrc/capi.hip host_inplace / host_binary / host_verify_core) under every setting of their options --
host_chunk (KiB per chunk / slice), host_streams (1 .. 8 staging buffers of the pipelines on caller-page-locked memory), host_duplex (one
stream per direction from 8 chunks up), host_copy_threads (threads of the memcpy between a pageable caller buffer and the library's
page-locked slots) -- against the oracle and the device-pointer entry points.  Round 6: a pageable caller buffer only ever meets memcpy
(the library never page-locks caller memory, and never lets the runtime do it: csrc/capi.hip, profiles/r06_suite_crash_rootcause.txt).  The reference's calling convention for the path is caller-owned HOST arrays
(reference_code/ref_ntt.h:30-36, hardware_code/ntt2x2.h:30-34); bench.py's `end_to_end` block times these calls."""
import numpy as np
import pytest

from oracle.oracle import Q, N, splitmix64_polys

pytestmark = pytest.mark.gpu


@pytest.fixture()
def host_opts(gpu):
    from dilithium_amd import api
    saved = {k: api.get_option(k) for k in ("host_chunk", "host_streams", "host_copy_threads", "host_duplex")}
    yield lambda **kw: [api.set_option(k, v) for k, v in kw.items()]
    for k, v in saved.items():
        api.set_option(k, v)


@pytest.mark.parametrize("chunk,streams,threads", [(64, 1, 1), (64, 3, 3), (100, 8, 2), (16384, 3, 3), (1024, 2, 8)])
def test_ntt_host_chunked(gpu, oracle, host_opts, chunk, streams, threads):
    from dilithium_amd import api
    host_opts(host_chunk=chunk, host_streams=streams, host_copy_threads=threads)
    n = 3 * chunk + 17 if chunk < 4096 else 20000
    a = splitmix64_polys(n, seed=chunk + streams)
    x = a.copy()
    api.ntt(x)
    idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 97)), [n - 1, chunk - 1 if chunk < n else 0, min(chunk, n - 1)]]))
    assert (x[idx] == oracle.ntt(a[idx])).all()
    api.invntt(x)
    assert (x == a).all()


@pytest.mark.parametrize("duplex,threads,locked", [(1, 3, False), (1, 1, False), (1, 3, True), (0, 3, True)])
@pytest.mark.parametrize("chunk,bufs", [(64, 1), (64, 2), (100, 4), (600, 2), (8192, 3)])
def test_ntt_host_pipelines_ring_wraps_and_ragged_tails(gpu, oracle, host_opts, duplex, threads, locked, chunk, bufs):
    """a pageable buffer through the ring of page-locked slots (memcpy by 1 or 3 threads), and one stream per direction / round-robin over
    the streams on a buffer the caller page-locked (locked: torch pin_memory) -- with the rings wrapping several times (9+ chunks over
    1 .. 4 buffers) and a ragged last chunk, against the oracle and the round trip"""
    from dilithium_amd import api
    host_opts(host_chunk=chunk, host_streams=bufs, host_copy_threads=threads, host_duplex=duplex)
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
@pytest.mark.parametrize("chunk,streams,locked", [(128, 3, 0), (300, 1, 1), (16384, 3, 0), (1024, 2, 1)])
def test_verify_core_host_vs_device_and_oracle(gpu, oracle, host_opts, level, shared, chunk, streams, locked):
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
    host_opts(host_chunk=chunk, host_streams=streams)
    if locked:          # every operand page-locked by the caller (torch pin_memory): DMA in place instead of the staged ring
        keep = [gpu.from_numpy(x).pin_memory() for x in (A, z, c, t1, h.reshape(n, K * N))]
        w1 = api.verify_core(*[k.numpy() for k in keep], level, shared_pk=shared)
    else:
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


def test_pageable_caller_buffers_only_meet_memcpy(gpu, oracle, host_opts):
    """The round-6 rule, observed from outside: during host-pointer calls on pageable arrays of every size class the library issues no
    hipHostRegister / hipHostUnregister, and every hipMemcpy* it issues has a page-locked host end (its own slots) -- checked by asking the
    runtime about both ends of the caller's arrays before and after (still unknown to it: pageable, unregistered), and by the results"""
    import ctypes as C
    from dilithium_amd import api
    hip = C.CDLL("libamdhip64.so")

    class Attr(C.Structure):
        _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p), ("isManaged", C.c_int),
                    ("allocationFlags", C.c_uint), ("pad", C.c_char * 64)]

    def known_to_runtime(arr):
        at = Attr()
        ends = (arr.ctypes.data, arr.ctypes.data + arr.nbytes - 1)
        rcs = [hip.hipPointerGetAttributes(C.byref(at), C.c_void_p(p)) for p in ends]
        hip.hipGetLastError()
        return any(rc == 0 and at.type == 1 for rc in rcs)          # hipMemoryTypeHost: registered / page-locked host memory

    for threads in (1, 3):
        host_opts(host_copy_threads=threads, host_chunk=8192, host_streams=4)
        for n in (3, 300, 5000, 20000, 70000):
            a = splitmix64_polys(n, seed=n)
            x, b = a.copy(), splitmix64_polys(n, seed=n + 1)
            assert not any(known_to_runtime(v) for v in (x, a, b))
            api.ntt(x)
            idx = np.arange(0, n, max(1, n // 50))
            assert (x[idx] == oracle.ntt(a[idx])).all()
            api.invntt(x)
            assert (x == a).all()
            c = np.empty_like(a)
            api.pointwise_barrett(c, a, b)
            api.polymul(c, a, b)
            assert (c[idx] == oracle.invntt(oracle.pointwise(oracle.ntt(a[idx]), oracle.ntt(b[idx])))).all()
            assert not any(known_to_runtime(v) for v in (x, a, b, c)), (threads, n)
