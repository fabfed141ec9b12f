"""Data-parallel sampling: one process per GPU, the image batch sharded by rank, one gather at the end.

Trajectories of different images never interact (GroupNorm and attention are per sample), so ranks run
the whole sampling loop without any per-step communication; the only collective is the final
all-gather of the latents (RCCL over xGMI through ``torch.distributed`` backend "nccl"; "gloo" on CPU
in the tests).  The reference has no distributed code; this is the build's own multi-GPU mode.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

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


def gather_latents(local: torch.Tensor, total: Optional[int] = None) -> torch.Tensor:
    """All ranks' latents concatenated along the batch in rank order: ONE collective, no host synchronisation.

    ``total`` is the job's image count when the shards came from ``shard_bounds(total, world, rank)`` and may differ
    in size by one; every rank derives every shard's size from that same function, so nothing has to be exchanged
    first.  ``None`` means equal shards (``world * local.shape[0]`` images).  A ragged job pads each shard to the
    largest one for the collective and drops the padding afterwards.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local.contiguous()
    n = local.shape[0]
    if total is None:
        total = world * n
    counts: List[int] = [b - a for a, b in (shard_bounds(total, world, r) for r in range(world))]
    if counts[rank] != n:
        raise ValueError(f"rank {rank} holds {n} images but shard_bounds({total}, {world}, {rank}) gives {counts[rank]}")
    tail = tuple(local.shape[1:])
    pad = max(counts)
    if pad == min(counts):
        out = torch.empty((total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    send = local
    if n < pad:
        send = torch.zeros((pad,) + tail, dtype=local.dtype, device=local.device)
        send[:n] = local
    recv = torch.empty((world * pad,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.view((world, pad) + tail)
    return torch.cat([recv[r, :c] for r, c in enumerate(counts)], 0)
