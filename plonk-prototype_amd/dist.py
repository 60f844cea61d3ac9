"""Multi-GPU MSM: one process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm, "gloo" in the CPU tests).

The MSM shards by POINTS (mode P): rank g owns bases/scalars
[lo_g, hi_g), computes its partial sum on its own GPU with no data-path collective, then the
ranks exchange one 144-byte projective point each (all_gather) and every rank folds the
partials with the group law -- an all-reduce under point addition (RCCL has no EC reduce op).
NTTs of a prover round are independent polynomials: whole vectors are assigned to ranks and
nothing is exchanged.
"""
from __future__ import annotations

import numpy as np

from .host import g1_fold


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous near-equal split of [0, n): the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_fold(partial_xyz: np.ndarray, device=None) -> np.ndarray:
    """All-gather each rank's partial G1 point and fold them; identical result on every rank.

    partial_xyz: [18] uint64 projective point (pm_g1_msm_dev output).  Uses the default process
    group; `device` is the tensor device the backend needs ("cuda:<local_rank>" for RCCL)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(partial_xyz, dtype=np.uint64).reshape(18)
    world = dist.get_world_size()
    # int64 view: uint64 tensors are not universally supported by the collectives
    mine = torch.from_numpy(np.ascontiguousarray(partial_xyz, dtype=np.uint64).reshape(18).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    stacked = torch.stack(parts).cpu().numpy().view(np.uint64)
    return g1_fold(stacked)
