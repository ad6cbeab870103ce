"""Nothing accumulates: device memory, host memory, threads and open file descriptors of the process after hundreds of calls of every
kind of entry point -- transforms on device and host pointers (the ring of page-locked slots a pageable buffer goes through, both in-place pipelines of a page-locked one),
the fused cores, the byte-level scheme (whose signing loop reads a count back every round), the batch-of-one mailbox calls.  The
reference is stateless and allocation-free (ref_ntt.h:30-36: caller-owned buffers, nothing retained); the drop-in keeps grow-only scratch
per stream and device, so a steady workload must reach a steady state."""
import os

import numpy as np
import pytest

from oracle.oracle import splitmix64_polys

pytestmark = pytest.mark.gpu
psutil = pytest.importorskip("psutil")


def _snapshot(torch):
    torch.cuda.synchronize()
    p = psutil.Process()
    free, _ = torch.cuda.mem_get_info()
    return {"dev_used": -free, "rss": p.memory_info().rss, "threads": p.num_threads(), "fds": p.num_fds()}


def _steady(torch, body, warm=3, reps=40, dev_slack=8 << 20, rss_slack=48 << 20):
    for _ in range(warm):
        body()
    a = _snapshot(torch)
    for _ in range(reps):
        body()
    b = _snapshot(torch)
    assert b["dev_used"] - a["dev_used"] <= dev_slack, ("device memory grew", a, b)
    assert b["rss"] - a["rss"] <= rss_slack, ("host memory grew", a, b)
    assert b["threads"] <= a["threads"], ("threads left behind", a, b)
    assert b["fds"] <= a["fds"] + 2, ("file descriptors left behind", a, b)


def test_transforms_and_host_pipelines_reach_a_steady_state(gpu):
    from dilithium_amd import api
    torch = gpu
    n = 40000                                             # pageable: the ring of page-locked slots; page-locked: round-robin below 64 MiB
    a = splitmix64_polys(n, seed=9)
    pageable = a.copy()
    keep = torch.empty((70000, 256), dtype=torch.int32).pin_memory()     # >= 64 MiB: one stream per direction
    locked = keep.numpy()
    locked[:] = splitmix64_polys(70000, seed=10)
    small = a[:300].copy()
    dev = torch.from_numpy(a[:8192]).cuda()

    def body():
        api.ntt(pageable), api.invntt(pageable)
        api.ntt(locked), api.invntt(locked)
        api.ntt(small), api.invntt(small)
        api.ntt(dev), api.invntt(dev)
    _steady(torch, body, reps=25)
    assert (pageable == a).all() and (small == a[:300]).all()


@pytest.mark.parametrize("level", [2, 3, 5])
def test_scheme_calls_reach_a_steady_state(gpu, level):
    from dilithium_amd import api
    torch = gpu
    g = torch.Generator(device="cuda").manual_seed(level)
    u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
    seed, mu, mu1 = u8(600, 32), u8(600, 64), u8(1, 64)
    pk, sk = api.keygen(seed, level)

    def body():
        p2, s2 = api.keygen(seed, level)
        sig, _ = api.sign(s2, mu, level)
        assert int(api.verify_sig(p2, sig, mu, level).sum()) == 0
        sig1, _ = api.sign(sk[:1], mu1, level, shared_sk=True)          # batch of one: a single speculative round
        assert int(api.verify_sig(pk[:1], sig1, mu1, level, shared_pk=True).sum()) == 0
    _steady(torch, body, reps=30)


def test_mailbox_calls_reach_a_steady_state(gpu):
    """batch-of-one host calls through the resident mailbox wave (what libdil256_ref.so's ntt() / invntt() are), with the wave retiring
    and being relaunched in between (idle time-out 200 us)"""
    import time
    from dilithium_amd import api
    torch = gpu
    saved = api.get_option("host_mailbox")
    api.set_option("host_mailbox", 1)
    try:
        x = splitmix64_polys(1, seed=5)
        x0 = x.copy()

        def body():
            for _ in range(50):
                api.ntt(x), api.invntt(x)
            time.sleep(0.002)                              # the wave retires; the next call relaunches it
        _steady(torch, body, reps=30)
        assert (x == x0).all()
    finally:
        api.set_option("host_mailbox", saved)
