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
