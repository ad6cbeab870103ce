"""CPU, world_size = 2, gloo: the N>1 path (shard -> compute -> final gather) reproduces the
unsharded result.  The per-shard compute here is the oracle (tests may call it); on the GPU
box the same sharding code feeds the HIP kernels (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from dilithium_amd import sharding


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 100, 65536, 8191):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.oracle import Oracle, splitmix64_polys
    o = Oracle()
    r, w, _ = sharding.init_distributed("gloo")
    assert (r, w) == (rank, world)
    a = torch.from_numpy(splitmix64_polys(n_items, seed=5))
    level, K, L = 3, 6, 5
    A = torch.from_numpy(splitmix64_polys(K * L, seed=6).reshape(1, K, L, 256))
    y = torch.from_numpy(splitmix64_polys(n_items * L, seed=7).reshape(n_items, L, 256))

    def ntt_fn(x):
        return torch.from_numpy(o.ntt(x.numpy()))

    def mv_fn(yy):
        return torch.from_numpy(o.matvec(K, L, A.numpy(), yy.numpy(), shared_A=True))

    got = sharding.run_sharded(ntt_fn, n_items, a)
    got_mv = sharding.run_sharded(mv_fn, n_items, y)
    t = sharding.max_over_ranks(float(rank + 1))
    sharding.barrier()
    if rank == 0:
        q.put((got.numpy(), got_mv.numpy(), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_items", [64, 37])      # even and ragged
def test_two_rank_gloo_shard_and_gather(n_items, oracle):
    from oracle.oracle import splitmix64_polys
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, got_mv, t = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a = splitmix64_polys(n_items, seed=5)
    assert (got == oracle.ntt(a)).all()
    A = splitmix64_polys(30, seed=6).reshape(1, 6, 5, 256)
    y = splitmix64_polys(n_items * 5, seed=7).reshape(n_items, 5, 256)
    assert (got_mv == oracle.matvec(6, 5, A, y, shared_A=True)).all()
    assert t == 2.0


def _worker_sign_slabs(rank, world, port, n_items, q):
    """configs[4] shape: the sign inner loop's (z, h, flag) slabs, tuple-valued fn, ragged batch, gathered on gloo"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.oracle import Oracle
    o = Oracle()
    sharding.init_distributed("gloo")
    c, y, w0, w1, s1h, s2h, t0h = _sign_inputs(n_items)

    def fn(cc, yy, ww0, ww1):
        z, h, f = o.sign_phase2(2, cc.numpy(), yy.numpy(), ww0.numpy(), ww1.numpy(), s1h, s2h, t0h)
        return torch.from_numpy(z), torch.from_numpy(h), torch.from_numpy(f)

    z, h, f = sharding.run_sharded(fn, n_items, *(torch.from_numpy(x) for x in (c, y, w0, w1)))
    if rank == 0:
        q.put((z.numpy(), h.numpy(), f.numpy()))
    sharding.barrier()
    torch.distributed.destroy_process_group()


def _sign_inputs(n):
    from oracle.oracle import Oracle, splitmix64_polys, Q
    o = Oracle()
    K = L = 4
    rng = np.random.default_rng(11)
    g1 = 1 << 17
    y = np.mod(rng.integers(-(g1 - 1), g1 + 1, (n, L, 256)), Q).astype(np.int32)
    A = splitmix64_polys(K * L, seed=3).reshape(1, K, L, 256)
    w1, w0 = o.sign_phase1(2, A, y)
    c = np.zeros((n, 256), np.int32)
    c[:, ::7] = 1
    c[:, 3::11] = Q - 1
    s1h = o.ntt(np.mod(rng.integers(-2, 3, (1, L, 256)), Q).astype(np.int32))
    s2h = o.ntt(np.mod(rng.integers(-2, 3, (1, K, 256)), Q).astype(np.int32))
    t0h = o.ntt(np.mod(rng.integers(-4095, 4097, (1, K, 256)), Q).astype(np.int32))
    return c, y, w0, w1, s1h, s2h, t0h


def test_two_rank_gloo_sign_slabs(oracle):
    n_items = 21
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sign_slabs, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    z, h, f = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c, y, w0, w1, s1h, s2h, t0h = _sign_inputs(n_items)
    oz, oh, of = oracle.sign_phase2(2, c, y, w0, w1, s1h, s2h, t0h)
    assert (z == oz).all() and (h == oh).all() and (f == of).all()


@pytest.mark.gpu
def test_run_sharded_with_hip_compute_single_rank(gpu, oracle):
    """the sharding harness around HIP compute (what bench.py --gpus N runs on every rank), world size 1: slices, the
    level-5 sign phases on the GPU, (z, h, flag) slabs 'gathered' -- against the oracle"""
    from dilithium_amd import api
    from oracle import dilithium_kat as dk
    from oracle.oracle import splitmix64_polys, Q
    level, K, L, n = 5, 8, 7, 2304
    p = dk.PARAMS[level]
    rng = np.random.default_rng(9)
    A = splitmix64_polys(K * L, seed=8).reshape(1, K, L, 256)
    y = np.mod(rng.integers(-(p.gamma1 - 1), p.gamma1 + 1, (n, L, 256)), Q).astype(np.int32)
    c = np.zeros((n, 256), np.int32)
    c[:, ::5] = 1
    c[:, 1::9] = Q - 1
    s1h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (1, L, 256)), Q).astype(np.int32))
    s2h = oracle.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (1, K, 256)), Q).astype(np.int32))
    t0h = oracle.ntt(np.mod(rng.integers(-4095, 4097, (1, K, 256)), Q).astype(np.int32))
    d = lambda a: gpu.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    dA, ds1, ds2, dt0 = d(A), d(s1h), d(s2h), d(t0h)

    def fn(yy, cc):
        w1, w0 = api.sign_phase1(dA, yy.contiguous(), level, shared_key=True)
        return api.sign_phase2(cc.contiguous(), yy.contiguous(), w0, w1, ds1, ds2, dt0, level, shared_key=True)

    z, h, f = sharding.run_sharded(fn, n, d(y), d(c))
    ow1, ow0 = oracle.sign_phase1(level, A, y)
    oz, oh, of = oracle.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
    assert (f.cpu().numpy() == of).all() and (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all()


def test_c_abi_shard_range_matches_python():
    """dil_shard_range (the C++ multi-GPU host layer, csrc/multi_gpu.hip) cuts the same slices as sharding.shard_range"""
    import ctypes as C
    from dilithium_amd import lib as dlib
    L = dlib.load()
    for n in (0, 1, 7, 8, 100, 65536, 8191):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lo, hi = C.c_size_t(), C.c_size_t()
                L.dil_shard_range(n, r, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == sharding.shard_range(n, r, world)


@pytest.mark.gpu
def test_multi_gpu_host_layer(gpu, oracle, kat_msgs):
    """the C++ host layer over every visible device (one here): NTT of a ragged host batch vs the oracle, and KAT
    sign / verify through dil_sign_multi_host / dil_verify_sig_multi_host"""
    import ctypes as C
    import hashlib
    from dilithium_amd import lib as dlib
    from oracle.oracle import splitmix64_polys
    from tests.test_gpu_codecs import kat_wire
    L = dlib.load()
    a = splitmix64_polys(1001, seed=3)
    b = a.copy()
    dlib.check(L.dil_ntt_multi_host(b.ctypes.data_as(C.POINTER(C.c_int32)), 1001, 0, 0))
    assert (b == oracle.ntt(a)).all()
    dlib.check(L.dil_ntt_multi_host(b.ctypes.data_as(C.POINTER(C.c_int32)), 1001, 1, 1))
    assert (b == a).all()
    k, pk, sk, sig = kat_wire(3)
    mu = np.stack([np.frombuffer(hashlib.shake_256(k["tr"][i].tobytes() + kat_msgs[i]).digest(64), dtype=np.uint8) for i in range(100)])
    pk, sk = np.ascontiguousarray(pk), np.ascontiguousarray(sk)
    got = np.zeros_like(sig)
    att = np.zeros(100, np.int32)
    vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    dlib.check(L.dil_sign_multi_host(vp(got), vp(att), vp(sk), vp(mu), 3, 100, 0, 512, 0))
    assert (got == sig).all() and (att == k["attempts"]).all()
    v = np.ones(100, np.int32)
    dlib.check(L.dil_verify_sig_multi_host(vp(v), vp(pk), vp(got), vp(mu), 3, 100, 0, 0))
    assert (v == 0).all()


# ---- the schedule of the final gather (csrc/multi_gpu.hip gather_plan, exported as dil_multi_gather_plan), executed by a model of the
# NCCL API's semantics: the Send / Recv pairing of the gather-to-a-root form has never run on hardware (no multi-GPU node in any round) --
# here every call the library would issue inside its one group is issued against recording "communicators" over numpy buffers ----------
class FakeRccl:
    """G ranks, one byte buffer each; the group's calls in issue order; NCCL's rules: inside a group point-to-point calls match by (source,
    destination) in FIFO order with equal counts; a collective must be called by EVERY rank with the same count (and root); nothing moves
    before the group ends"""

    def __init__(self, bufs):
        self.bufs = bufs
        self.G = len(bufs)
        self.calls = []

    def run(self, ops):
        from collections import defaultdict, deque
        sends, recvs = defaultdict(deque), defaultdict(deque)
        bcasts, gathers = defaultdict(list), []
        for op in ops:
            self.calls.append((("allgather", "broadcast", "recv", "send")[op.kind], op.rank, op.peer, op.offset, op.bytes))
            assert 0 <= op.rank < self.G and op.bytes > 0 and op.offset + op.bytes <= len(self.bufs[op.rank])
            if op.kind == 3:
                assert 0 <= op.peer < self.G and op.peer != op.rank
                sends[(op.rank, op.peer)].append(op)
            elif op.kind == 2:
                assert 0 <= op.peer < self.G and op.peer != op.rank
                recvs[(op.peer, op.rank)].append(op)
            elif op.kind == 1:
                bcasts[(op.peer, op.offset, op.bytes)].append(op.rank)
            else:
                gathers.append(op)
        snapshot = [b.copy() for b in self.bufs]                       # group semantics: every transfer reads pre-group data
        assert set(sends) == set(recvs), ("unmatched point-to-point calls: the group would hang", set(sends) ^ set(recvs))
        for key in sends:
            assert len(sends[key]) == len(recvs[key]), ("send / recv counts differ", key)
            for s_, r_ in zip(sends[key], recvs[key]):
                assert s_.bytes == r_.bytes, ("send / recv sizes differ: NCCL would corrupt or hang", key)
                self.bufs[key[1]][r_.offset:r_.offset + r_.bytes] = snapshot[key[0]][s_.offset:s_.offset + s_.bytes]
        for (root, off, n), ranks in bcasts.items():
            assert sorted(ranks) == list(range(self.G)), ("a broadcast not entered by every rank hangs", root, off, ranks)
            for g in range(self.G):
                self.bufs[g][off:off + n] = snapshot[root][off:off + n]
        if gathers:
            assert sorted(o.rank for o in gathers) == list(range(self.G)) and len({o.bytes for o in gathers}) == 1
            n = gathers[0].bytes
            for o in gathers:
                assert o.offset == o.rank * n                             # in place: the send slab at its own offset of the receive array
                for g in range(self.G):
                    self.bufs[g][o.rank * n:(o.rank + 1) * n] = snapshot[o.rank][o.offset:o.offset + n]


def _gather_plan(batch, item_bytes, root, G, ragged=0):
    import ctypes as C
    from dilithium_amd import lib as dlib
    L = dlib.load()
    n = C.c_size_t()
    assert L.dil_multi_gather_plan(batch, item_bytes, root, G, ragged, None, 0, C.byref(n)) == 0
    ops = (dlib.GatherOp * max(1, n.value))()
    assert L.dil_multi_gather_plan(batch, item_bytes, root, G, ragged, C.cast(ops, C.c_void_p), n.value, C.byref(n)) == 0
    return list(ops)[:n.value]


@pytest.mark.parametrize("G", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("batch", [1, 5, 8, 64, 8195])
@pytest.mark.parametrize("root", [-1, 0, "last"])
def test_final_gather_schedule_against_a_model_of_the_nccl_api(G, batch, root):
    """every device's slab reaches every device (root < 0) or the root -- and only through calls NCCL can match: each non-root device sends
    exactly its own shard once, the root posts exactly one receive per non-empty foreign shard at that shard's offset, sizes agree pairwise"""
    root = G - 1 if root == "last" else root
    item = 7                                                             # bytes per item, odd on purpose
    rng = np.random.default_rng(G * 1000 + batch)
    full = rng.integers(0, 256, batch * item, dtype=np.uint8)
    for ragged in (0, 1):
        bufs = []
        for g in range(G):                                                # device g holds its own slab, junk elsewhere
            lo, hi = sharding.shard_range(batch, g, G)
            b = rng.integers(0, 256, batch * item, dtype=np.uint8)
            b[lo * item:hi * item] = full[lo * item:hi * item]
            bufs.append(b)
        before = [b.copy() for b in bufs]
        ops = _gather_plan(batch, item, root, G, ragged)
        fake = FakeRccl(bufs)
        fake.run(ops)
        kinds = {c[0] for c in fake.calls}
        if root < 0:
            assert kinds <= ({"allgather"} if (batch % G == 0 and not ragged) else {"broadcast"})
            for g in range(G):
                assert (bufs[g] == full).all(), (g, "all-gather incomplete")
        else:
            assert kinds <= {"send", "recv"}
            assert (bufs[root] == full).all(), "the root does not hold every item"
            for g in range(G):
                if g != root:
                    assert (bufs[g] == before[g]).all(), "a non-root buffer was written"
                    lo, hi = sharding.shard_range(batch, g, G)
                    mine = [c for c in fake.calls if c[1] == g]
                    assert mine == ([("send", g, root, lo * item, (hi - lo) * item)] if hi > lo else []), mine
            assert sum(c[0] == "recv" for c in fake.calls) == sum(1 for g in range(G) if g != root and sharding.shard_range(batch, g, G)[1] > sharding.shard_range(batch, g, G)[0])


def test_final_gather_schedule_rejects_bad_arguments():
    import ctypes as C
    from dilithium_amd import lib as dlib
    L = dlib.load()
    n = C.c_size_t()
    assert L.dil_multi_gather_plan(8, 4, 2, 2, 0, None, 0, C.byref(n)) != 0          # root outside the job
    assert L.dil_multi_gather_plan(8, 4, 0, 0, 0, None, 0, C.byref(n)) != 0
    assert L.dil_multi_gather_plan(0, 4, -1, 4, 0, None, 0, C.byref(n)) == 0 and n.value == 0
