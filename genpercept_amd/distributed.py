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


class ResultGatherer:
    """The optional result gather of BASELINE.json configs[4] with every buffer allocated ONCE (the timed multi-GPU step used
    to build its receive list with `empty_like` and `torch.cat` the parts on every call).

    Per (shape, dtype, device, n_total, dst): one receive buffer [world * mx, C, H, W] whose chunks are the collective's receive list (rank r's
    padded shard lands at rows [r * mx, (r + 1) * mx)), a padded send buffer only when this rank's shard is shorter than the largest, and --
    only for ragged shardings -- one compact output buffer.  With equal shards (64 images over 8 GPUs) the receive buffer already IS the batch
    in order and `gather()` returns a view of it: no allocation, no copy after the collective.  The returned tensor is overwritten by the next
    call (the bench, and a serving loop that consumes a batch before the next one finishes, never hold two)."""

    def __init__(self, local: torch.Tensor, n_total: int, dst: Optional[int]):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.n_total, self.dst = n_total, dst
        self.sizes = [shard_range(n_total, r, self.world)[1] - shard_range(n_total, r, self.world)[0] for r in range(self.world)]
        self.mx = max(self.sizes)
        tail = tuple(local.shape[1:])
        kw = dict(dtype=local.dtype, device=local.device)
        self.send = torch.zeros((self.mx,) + tail, **kw) if self.sizes[self.rank] < self.mx else None  # (padding rows stay zero)
        self.receives = dst is None or self.rank == dst
        self.recv = torch.empty((self.world * self.mx,) + tail, **kw) if self.receives else None
        self.parts = list(self.recv.split(self.mx, dim=0)) if self.receives else None  # views: the collective writes straight into `recv`
        self.ragged = min(self.sizes) != self.mx
        self.out = torch.empty((n_total,) + tail, **kw) if (self.receives and self.ragged) else None

    def gather(self, local: torch.Tensor) -> Optional[torch.Tensor]:
        buf = local if local.is_contiguous() else local.contiguous()
        if self.send is not None:
            self.send[: local.shape[0]].copy_(local)
            buf = self.send
        if self.dst is None:
            dist.all_gather(self.parts, buf)
        else:
            dist.gather(buf, self.parts, dst=self.dst)
            if not self.receives:
                return None
        if not self.ragged:
            return self.recv[: self.n_total]
        o = 0
        for p, n in zip(self.parts, self.sizes):
            self.out[o:o + n].copy_(p[:n])
            o += n
        return self.out


_gatherers: dict = {}


def gather_results(local: torch.Tensor, n_total: int, dst: Optional[int] = 0, clone: bool = False) -> Optional[torch.Tensor]:
    """Gather per-rank result maps [n_local, C, H, W] back into batch order [n_total, C, H, W].

    dst = None -> all ranks get the result (all_gather); dst = r -> only rank r (others return None).  Shards may be ragged; they are padded
    to the largest shard for the collective (RCCL wants equal counts) and trimmed afterwards.  Buffers are allocated on the first call with a
    given (shape, dtype, device, n_total, dst, world size, rank) and reused (ResultGatherer): the result is a VIEW of that buffer, valid until the
    next call with the same key -- pass clone=True to keep it across calls.  A new process group (other world size / rank remap after
    destroy_process_group + init_process_group) gets its own buffers: the key carries both, and gatherers of another group size are dropped."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local.clone() if clone else local
    world, rank = dist.get_world_size(), dist.get_rank()
    key = (tuple(local.shape[1:]), local.dtype, str(local.device), n_total, dst, world, rank)
    g = _gatherers.get(key)
    if g is None:
        for k in [k for k in _gatherers if k[5:] != (world, rank)]:  # stale: built for a process group that no longer exists
            del _gatherers[k]
        g = _gatherers[key] = ResultGatherer(local, n_total, dst)
    out = g.gather(local)
    return out.clone() if (clone and out is not None) else out


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
