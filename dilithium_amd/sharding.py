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
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


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
    """Final gather: concatenate every rank's result slab along dim 0, in rank order.

    Slabs may be ragged by one item (shard_range); they are padded to the largest slab for the
    fixed-size collective and trimmed afterwards.  Single process: returns `local`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if n_items is None:
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        n_items = int(cnt.item())
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    big = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < big:
        pad = torch.zeros((big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    if all(hi - lo == big for lo, hi in sizes):
        return out
    return torch.cat([out[r * big: r * big + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


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
