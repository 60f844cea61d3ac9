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
# One NTT split over the GPUs of a node: the four-step
# decomposition with the matrix distributed by rows and all-to-all transposes between the steps.
def _transpose_exchange(t, rows: int, cols: int, world: int):
    """t: this rank's [rows/world, cols, 4] slice of a row-distributed [rows, cols] matrix of Fr.
    Returns this rank's [cols/world, rows, 4] slice of the transposed matrix (one all_to_all)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return t.transpose(0, 1).contiguous()
    rl, cl = rows // world, cols // world
    send = t.view(rl, world, cl, 4).permute(1, 0, 2, 3).contiguous()        # [dest][row_local][col_local]
    recv = torch.empty_like(send)
    if send.is_cuda and dist.get_backend() != "nccl":                        # gloo rehearsal: exchange on the host
        s, r = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
        dist.all_to_all_single(r, s)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send)
    return recv.view(rows, cl, 4).transpose(0, 1).contiguous()               # [col_local][row]


class FourStepNTT:
    """A 2^log_n-point NTT whose vector is block-distributed in natural order over the ranks of the
    default process group: rank r holds x[r N/W, (r+1) N/W) on its GPU, and receives the same block
    of the result.  N = N1 N2 as an N1 x N2 row-major matrix:

        transpose (all-to-all) -> N2/W batched NTTs of size N1 -> twiddles w_N^(j k1)
        -> transpose -> N1/W batched NTTs of size N2 -> transpose (natural order)

    Each all-to-all moves (W-1)/W of the rank's 32 N / W bytes; over RCCL that is one direct xGMI
    transfer per peer.  Local transforms are the library's batched NTT kernels; transposes inside a
    rank are strided copies.  Needs W | N1 and W | N2.  Twiddle rows are cached per (log_n, direction)."""

    def __init__(self, ctx: Context, log_n: int):
        import torch.distributed as dist
        from .host import domain_info
        self.ctx, self.log_n = ctx, log_n
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.l1 = log_n // 2
        self.l2 = log_n - self.l1
        self.n1, self.n2 = 1 << self.l1, 1 << self.l2
        if self.n1 % self.world or self.n2 % self.world:
            raise Error(_lib.PM_ERR_BAD_ARG, "the number of ranks must divide both factors of the transform")
        self.omega, self.omega_inv, self.size_inv = domain_info(log_n)
        self._tw = {}

    def _twiddles(self, inverse: bool, like):
        """[N2/W, N1, 4]: row j_local holds w^((j0 + j_local) k1), k1 < N1 (w^-1 for the inverse)."""
        import torch
        from .field import R_MOD, fr_from_limbs, fr_to_limbs
        if inverse not in self._tw:
            cl = self.n2 // self.world
            t = torch.empty((cl, self.n1, 4), dtype=torch.int64, device=like.device)
            w = fr_from_limbs(self.omega_inv if inverse else self.omega)
            one = fr_to_limbs(1)
            for jl in range(cl):
                base = fr_to_limbs(pow(w, self.rank * cl + jl, R_MOD))
                self.ctx.fr_powers(base, one, self.n1, t[jl].data_ptr())
            self.ctx.sync()
            self._tw[inverse] = t
        return self._tw[inverse]

    def _coset(self, x, inverse: bool):
        """x[n] *= 7^n (forward, before) or X[k] *= 7^-k (inverse, after) on this rank's block."""
        import torch
        from .field import GENERATOR, R_MOD, fr_to_limbs
        blk = x.shape[0]
        g = pow(GENERATOR, -1, R_MOD) if inverse else GENERATOR
        p = torch.empty_like(x)
        self.ctx.fr_powers(fr_to_limbs(g), fr_to_limbs(pow(g, self.rank * blk, R_MOD)), blk, p.data_ptr())
        self.ctx.fr_vec_op(2, x.data_ptr(), p.data_ptr(), blk, x.data_ptr(), blk)
        self.ctx.sync()
        return x

    def __call__(self, x_local, flags: int = 0):
        """x_local: torch int64 [N/W, 4] on this rank's GPU (Fr Montgomery limbs).  Returns the rank's
        block of the transform in natural order (a new tensor)."""
        import torch
        inverse, coset = bool(flags & _lib.NTT_INVERSE), bool(flags & _lib.NTT_COSET)
        ctx, W = self.ctx, self.world
        n1, n2 = self.n1, self.n2
        x = x_local.contiguous().clone()
        if x.shape[0] * W != n1 * n2:
            raise Error(_lib.PM_ERR_LENGTH, "x_local must hold N / world elements")
        # torch's copies / transposes run on torch's stream, the library's kernels on the context's:
        # order the hand-overs explicitly (torch -> library: synchronize torch; library -> torch: ctx.sync)
        torch.cuda.synchronize(x.device)
        if coset and not inverse:
            x = self._coset(x, False)
        sub = _lib.NTT_INVERSE if inverse else 0
        a = _transpose_exchange(x.view(n1 // W, n2, 4), n1, n2, W)            # [N2/W][N1]: columns, i contiguous
        torch.cuda.synchronize(x.device)
        ctx.fr_ntt_dev(a.data_ptr(), n1, a.data_ptr(), self.l1, sub, batch=n2 // W)
        tw = self._twiddles(inverse, a)
        cnt = a.shape[0] * n1
        ctx.fr_vec_op(2, a.data_ptr(), tw.data_ptr(), cnt, a.data_ptr(), cnt)
        ctx.sync()
        b = _transpose_exchange(a, n2, n1, W)                                 # [N1/W][N2]: rows k1, j contiguous
        torch.cuda.synchronize(x.device)
        ctx.fr_ntt_dev(b.data_ptr(), n2, b.data_ptr(), self.l2, sub, batch=n1 // W)
        ctx.sync()
        out = _transpose_exchange(b, n1, n2, W).view(-1, 4)                   # [N2/W][N1] = natural order block
        torch.cuda.synchronize(x.device)
        if coset and inverse:
            out = self._coset(out, True)
        return out
