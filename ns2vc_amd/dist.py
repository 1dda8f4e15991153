"""Data-parallel sharding of utterance batches across the GPUs of one node.

The denoiser has no cross-sample reduction (GroupNorm per (sample, group), LayerNorm
per token, attention per (sample, head)), so utterances are independent
work items: rank r takes a contiguous slice of the batch, runs its own captured
sampling loop with no communication, and the finished latents are all-gathered
once at the end (RCCL over xGMI; ``backend='nccl'`` IS RCCL on ROCm).  The reference
has no counterpart (its inference is single-GPU batch 1, inference/infer_tool.py:184-201).

torch.distributed is plumbing here; CPU tests run the same code over gloo.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of ``n_items`` for ``rank`` (first ranks get the remainder)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: Optional[str] = None):
    """One process per GPU, rendezvous from the torchrun env (MASTER_ADDR should be 127.0.0.1 on one node)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    if world == 1:
        return rank, world, local
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_latents(local, n_total: int):
    """All-gather the finished latents of every rank's shard into the global batch order.

    ``local``: (b_r, C, T) tensor of this rank's slice (shard_range order).  Returns (n_total, C, T)
    on every rank.  Uneven shards are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    pad = pad.contiguous()
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
