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

from .host import Bases, Context, Error, g1_fold, g1_to_affine
from . import _lib


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


def allgather_fold_many(partials_xyz: np.ndarray, device=None) -> np.ndarray:
    """[k, 18] partial points per rank -> [k, 18] folded sums, one all_gather for all k."""
    import torch
    import torch.distributed as dist

    p = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 18)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return p
    world = dist.get_world_size()
    mine = torch.from_numpy(p.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    stacked = torch.stack(parts).cpu().numpy().view(np.uint64)          # [world, k, 18]
    return np.stack([g1_fold(stacked[:, j]) for j in range(p.shape[0])])


class ShardedCommitKey:
    """``CommitKey`` whose ``powers_of_g`` are split by index range over the ranks of the default
    process group (BASELINE.json configs[4]): this rank holds powers [lo, lo + len(powers_slice)) of
    `total`.  ``commit_batch_dev`` runs the local slice of every MSM, then one all_gather + fold."""

    def __init__(self, powers_slice, lo: int, total: int, ctx: Context, device=None, precompute: bool = False):
        self.ctx, self.lo, self.total, self.device = ctx, lo, total, device
        self._bases = Bases(ctx, powers_slice)
        if precompute and self._bases.n:
            self._bases.precompute()
        one = np.array([0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745,
                        0x5C071A97A256EC6D, 0x15F65EC3FA80E493], dtype=np.uint64)   # Fp Montgomery 1
        self._identity = np.zeros(18, np.uint64)
        self._identity[6:12] = one

    def max_degree(self) -> int:
        return self.total - 1

    def commit_batch_dev(self, d_ptr: int, n: int, batch: int, stride: int | None = None) -> list:
        if n > self.total:
            raise Error(_lib.PM_ERR_LENGTH, "PolynomialDegreeTooLarge")
        cnt = min(self.lo + self._bases.n, n) - self.lo
        if cnt > 0:
            part = self._bases.msm_batch_dev(d_ptr + 32 * self.lo, cnt, batch, stride=stride if stride is not None else n)
        else:
            part = np.tile(self._identity, (batch, 1))
        return [g1_to_affine(p)[0] for p in allgather_fold_many(part, self.device)]
