"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  The path shards trivially — images are independent, weights are replicated — so there is NO collective on the
data path; the only communication is the optional gather of the finished result maps."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE).  No-op for world size 1."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local: torch.Tensor, n_total: int, dst: Optional[int] = 0) -> Optional[torch.Tensor]:
    """Gather per-rank result maps [n_local, C, H, W] back into batch order [n_total, C, H, W].

    dst = None -> all ranks get the result (all_gather); dst = r -> only rank r (others return None).  Shards may be
    ragged; they are padded to the largest shard for the collective (RCCL wants equal counts) and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    buf = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf = torch.cat([local, pad], dim=0)
    buf = buf.contiguous()
    if dst is None:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, parts, dst=dst)
        if rank != dst:
            return None
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
