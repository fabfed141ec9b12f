"""Data-parallel sampling: one process per GPU, the image batch sharded by rank, one gather at the end.

Trajectories of different images never interact (GroupNorm and attention are per sample), so ranks run
the whole sampling loop without any per-step communication; the only collective is the final
all-gather of the latents (RCCL over xGMI through ``torch.distributed`` backend "nccl"; "gloo" on CPU
in the tests).  The reference has no distributed code; this is the build's own multi-GPU mode.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # DM_DIST_BACKEND=gloo: rehearsal of a multi-rank job on a box with fewer GPUs than ranks
            backend = os.environ.get("DM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of this rank's images; the first ``total % world`` ranks take one extra."""
    if total < 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(t: torch.Tensor, world: int = None, rank: int = None) -> torch.Tensor:
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    a, b = shard_bounds(t.shape[0], world, rank)
    return t[a:b]


def gather_latents(local: torch.Tensor) -> torch.Tensor:
    """All ranks' latents concatenated along the batch in rank order (shards may differ in size)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    sizes = [torch.zeros(1, dtype=torch.long, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.long, device=local.device))
    counts: List[int] = [int(s.item()) for s in sizes]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    bufs = [torch.empty((c,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) for c in counts]
    pad = max(counts)
    padded = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    recv = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(recv, padded)
    for i, c in enumerate(counts):
        bufs[i] = recv[i][:c]
    return torch.cat(bufs, 0)
