"""Multi-GPU layer: independent polynomials / signatures shard embarrassingly.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" on
CPU for tests).  The data path has NO collective: rank g owns the contiguous item slice
[g*B/G, (g+1)*B/G) and per-key constants are replicated.  The only exchange is the final
gather of fixed-size result slabs -- one all_gather_into_tensor.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous slice of rank `rank`; sizes differ by at most one item (ragged batches)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; no-op single process otherwise"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:                      # DIL_DIST_BACKEND=gloo: a multi-rank rehearsal on a box with fewer GPUs than ranks
            backend = os.environ.get("DIL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def local_device(local_rank: int) -> int:
    """the GPU of this rank: its local rank -- except in a gloo rehearsal on a box with fewer GPUs than ranks, where ranks share"""
    n = torch.cuda.device_count()
    if n and local_rank >= n and dist.is_initialized() and dist.get_backend() == "gloo":
        return local_rank % n
    return local_rank


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_slabs(local: torch.Tensor, n_items: int | None = None) -> torch.Tensor:
    """Final gather: every rank's result slab, concatenated along dim 0 in rank order, on every rank.

    Equal slabs: ONE all_gather_into_tensor straight into the result (no staging copy).  Slabs ragged by one item
    (shard_range): each rank places its slab at its offset of the result and the slabs are exchanged in place as one
    broadcast per rank (all-gather-v; the same scheme as csrc/multi_gpu.hip) -- no padding, no torch.cat.
    Single process: returns `local`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_items is None:
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        n_items = int(cnt.item())
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    lo, hi = sizes[rank]
    assert local.shape[0] == hi - lo, "slab size does not match this rank's slice"
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if all(h - l == hi - lo for l, h in sizes):
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    out[lo:hi] = local
    works = [dist.broadcast(out[l:h], src=r, async_op=True) for r, (l, h) in enumerate(sizes) if h > l]
    for w in works:
        w.wait()
    return out


def run_sharded(fn, n_items: int, *batched: torch.Tensor, gather: bool = True):
    """Apply fn(*slices) to this rank's slice of every batched tensor; optionally gather.

    fn returns a tensor (or tuple of tensors) whose dim 0 is the item dimension."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(n_items, rank, world)
    res = fn(*[t[lo:hi] for t in batched])
    if not gather:
        return res
    if isinstance(res, tuple):
        return tuple(gather_slabs(r, n_items) for r in res)
    return gather_slabs(res, n_items)
