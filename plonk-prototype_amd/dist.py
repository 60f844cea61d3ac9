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


_MAX_PARTIALS = 16


def allgather_fold_many(partials_xyz, device=None):
    """[k, 18] partial points per rank -> [k, 18] folded sums, one all_gather for all k (k <= 16).

    ``partials_xyz=None`` is the abort marker of a rank whose local work failed: it still takes part in
    the collective (fixed-size message: count + 16 points), and every rank -- the failed one included --
    gets ``None`` back instead of blocking in a collective its peer never enters."""
    import torch
    import torch.distributed as dist

    if partials_xyz is None:
        p = np.zeros((0, 18), np.uint64)
    else:
        p = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 18)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None if partials_xyz is None else p
    if p.shape[0] > _MAX_PARTIALS:
        raise ValueError("at most 16 partial points per exchange")
    world = dist.get_world_size()
    msg = np.zeros(1 + 18 * _MAX_PARTIALS, np.uint64)
    msg[0] = 0 if partials_xyz is None else p.shape[0]
    msg[1:1 + p.size] = p.reshape(-1)
    mine = torch.from_numpy(msg.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    stacked = torch.stack(parts).cpu().numpy().view(np.uint64)          # [world, 1 + 288]
    counts = stacked[:, 0]
    if partials_xyz is None or (counts == 0).any() or (counts != counts[0]).any():
        return None
    k = int(counts[0])
    pts = stacked[:, 1:1 + 18 * k].reshape(world, k, 18)
    return np.stack([g1_fold(np.ascontiguousarray(pts[:, j])) for j in range(k)]) if k else p


class ShardedCommitKey:
    """``CommitKey`` whose ``powers_of_g`` are split by index range over the ranks of the default
    process group (BASELINE.json configs[4]): this rank holds powers [lo, lo + len(powers_slice)) of
    `total`.  ``commit_batch_dev`` runs the local slice of every MSM, then one all_gather + fold."""

    def __init__(self, powers_slice, lo: int, total: int, ctx: Context, device=None, precompute: bool = False,
                 native: bool = False):
        """native: exchange through the library's own RCCL communicator (``Context.comm_init`` first) instead
        of torch.distributed -- what a host without PyTorch does."""
        self.ctx, self.lo, self.total, self.device, self.native = ctx, lo, total, device, native
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
        folded = self.ctx.g1_allgather_fold(part) if self.native else allgather_fold_many(part, self.device)
        if folded is None:
            raise Error(_lib.PM_ERR_EXCHANGE, "a peer rank gave up")
        return [g1_to_affine(p)[0] for p in folded]


# ---------------------------------------------------------------------------------------------
# One NTT split over the GPUs of a node: the four-step decomposition with
# the matrix distributed by rows and all-to-all transposes between the steps, NATIVE in the library
# (pm_fr_ntt_fourstep_dev, csrc/ntt.hip: pack / unpack kernels, the library's batched passes, RCCL
# ncclSend / ncclRecv).  This class only owns the staging tensor and, for process groups the library cannot
# drive itself (gloo rehearsals), the all-to-all callback over torch.distributed.
class FourStepNTT:
    """A 2^log_n-point NTT whose vector is block-distributed in natural order over the ranks of the
    default process group: rank r holds x[r N/W, (r+1) N/W) on its GPU, and receives the same block
    of the result (bit-identical to the single-GPU plan).  N = N1 N2 as an N1 x N2 row-major matrix:

        transpose (all-to-all) -> N2/W batched NTTs of size N1 -> twiddles w_N^(j k1)
        -> transpose -> N1/W batched NTTs of size N2 -> transpose (natural order)

    Each all-to-all moves (W-1)/W of the rank's 32 N / W bytes; over RCCL that is one direct xGMI
    transfer per peer.  Needs W | N1 and W | N2.  ``native_comm=True`` uses the context's own RCCL
    communicator (``Context.comm_init``); otherwise the exchange runs through torch.distributed."""

    def __init__(self, ctx: Context, log_n: int, native_comm: bool = False):
        import torch.distributed as dist
        self.ctx, self.log_n, self.native_comm = ctx, log_n, native_comm
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.l1 = log_n // 2
        self.l2 = log_n - self.l1
        self.n1, self.n2 = 1 << self.l1, 1 << self.l2
        if self.n1 % self.world or self.n2 % self.world:
            raise Error(_lib.PM_ERR_BAD_ARG, "the number of ranks must divide both factors of the transform")
        self._stage = None
        self._cb = None

    def _exchange_callback(self, stage):
        """ALLTOALL_FN over torch.distributed on the two halves of ``stage`` (the library passes their addresses)."""
        import torch
        import torch.distributed as dist
        blk = stage.shape[0] // 2
        send, recv = stage[:blk], stage[blk:]

        def cb(_user, d_send, d_recv, _bytes_per_peer):
            try:
                if d_send != send.data_ptr() or d_recv != recv.data_ptr():
                    return 1
                if dist.get_backend() != "nccl":                      # gloo rehearsal: exchange on the host
                    s, r = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
                    dist.all_to_all_single(r, s)
                    recv.copy_(r)
                else:
                    dist.all_to_all_single(recv, send)
                torch.cuda.synchronize(stage.device)
                return 0
            except Exception:                                         # never raise through the C frame
                return 1
        return _lib.ALLTOALL_FN(cb)

    def __call__(self, x_local, flags: int = 0):
        """x_local: torch int64 [N/W, 4] on this rank's GPU (Fr Montgomery limbs).  Returns the rank's
        block of the transform in natural order (a new tensor)."""
        import torch
        W = self.world
        if x_local.shape[0] * W != self.n1 * self.n2:
            raise Error(_lib.PM_ERR_LENGTH, "x_local must hold N / world elements")
        x = x_local.contiguous().clone()
        blk = x.shape[0]
        if self._stage is None or self._stage.device != x.device:
            self._stage = torch.empty((2 * blk if W > 1 else blk, 4), dtype=torch.int64, device=x.device)
            self._cb = None if (W == 1 or self.native_comm) else self._exchange_callback(self._stage)
        # torch's copy runs on torch's stream, the library's kernels on the context's
        torch.cuda.synchronize(x.device)
        self.ctx.fr_ntt_fourstep_dev(x.data_ptr(), self._stage.data_ptr(), self.log_n, W, self.rank, flags, self._cb)
        self.ctx.sync()
        return x
