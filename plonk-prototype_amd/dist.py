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

import ctypes as C

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


def _fold_messages(stacked: np.ndarray, mine_failed: bool, empty):
    """[world, 1 + 288] gathered messages -> [k, 18] folded sums, or None when any rank sent the abort marker
    or the ranks are out of step (the same rule as the library's ``fold_gathered``, csrc/comm.hip)."""
    counts = stacked[:, 0]
    if mine_failed or (counts == 0).any() or (counts != counts[0]).any():
        return None
    world, k = stacked.shape[0], int(counts[0])
    pts = stacked[:, 1:1 + 18 * k].reshape(world, k, 18)
    return np.stack([g1_fold(np.ascontiguousarray(pts[:, j])) for j in range(k)]) if k else empty


def _message(partials_xyz) -> np.ndarray:
    """The fixed-size exchange message [count | 16 x 18 limbs]; ``None`` -> the abort marker (count 0)."""
    msg = np.zeros(1 + 18 * _MAX_PARTIALS, np.uint64)
    if partials_xyz is not None:
        p = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 18)
        if p.shape[0] > _MAX_PARTIALS:
            raise ValueError("at most 16 partial points per exchange")
        msg[0] = p.shape[0]
        msg[1:1 + p.size] = p.reshape(-1)
    return msg


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
    world = dist.get_world_size()
    mine = torch.from_numpy(_message(partials_xyz).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    stacked = torch.stack(parts).cpu().numpy().view(np.uint64)          # [world, 1 + 288]
    return _fold_messages(stacked, partials_xyz is None, p)


class LocalGroup:
    """W ranks as THREADS of one process, one :class:`Context` each: the exchange steps of the sharded prover
    (all-gather + fold of partial points) and of the four-step NTT (all-to-all of blocks) carried by a barrier
    and shared memory instead of a process group.  Two uses: a host that drives the GPUs of a node from one
    process (context r on device r; the all-to-all is then a peer copy per block), and the world-size-8
    rehearsal on a ONE-GPU box, whose process guard allows six GPU processes -- eight gloo ranks cannot run
    there, eight threads can (tests/test_gpu_world8.py).  Same fixed-size message, same abort marker and the
    same fold rule as the torch.distributed and the in-library RCCL exchanges."""

    def __init__(self, world: int, timeout: float = 600.0):
        import threading
        self.world = world
        self._barrier = threading.Barrier(world, timeout=timeout)
        self._msgs = np.zeros((world, 1 + 18 * _MAX_PARTIALS), np.uint64)
        self._send = [None] * world
        self._words = np.zeros((world, _lib.COMM_MSG_WORDS), np.uint64)

    def allgather_fold_many(self, rank: int, partials_xyz):
        """``allgather_fold_many`` among the threads of the group (``partials_xyz=None`` = abort marker)."""
        p = np.zeros((0, 18), np.uint64) if partials_xyz is None else \
            np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 18)
        self._msgs[rank] = _message(partials_xyz)
        self._barrier.wait()                       # every message is in place
        stacked = self._msgs.copy()
        self._barrier.wait()                       # every rank has its copy: the slots may be overwritten
        return _fold_messages(stacked, partials_xyz is None, p)

    def allgather_words(self, rank: int, words: np.ndarray) -> np.ndarray:
        """One fixed-size message per rank -> [world, words] on every rank (the distributed prover's all-gather)."""
        self._words[rank] = words
        self._barrier.wait()
        out = self._words.copy()
        self._barrier.wait()
        return out

    def alltoall_host(self, rank: int, send: np.ndarray) -> np.ndarray:
        """The same exchange on host arrays (equal length on every rank) -> this rank's receive buffer."""
        per = send.shape[0] // self.world
        self._send[rank] = send
        self._barrier.wait()
        out = np.concatenate([s[rank * per:(rank + 1) * per] for s in self._send])
        self._barrier.wait()
        return out

    def alltoall_tensors(self, rank: int, send, recv) -> int:
        """Block p of `recv` <- block `rank` of rank p's `send` (device tensors of equal size on every rank)."""
        import torch
        per = send.shape[0] // self.world
        self._send[rank] = send
        self._barrier.wait()
        for p_, s in enumerate(list(self._send)):
            recv[p_ * per:(p_ + 1) * per].copy_(s[rank * per:(rank + 1) * per])
        torch.cuda.synchronize()
        self._barrier.wait()                       # the peers have read this rank's send buffer
        return 0

    def alltoall_fn(self, rank: int, stage):
        """ALLTOALL_FN for ``pm_fr_ntt_fourstep_dev``: ``stage`` is this rank's [2 blk, 4] tensor (send | recv);
        recv block p = block `rank` of rank p's send half (a peer-to-peer copy per block on a node)."""
        import torch
        blk = stage.shape[0] // 2
        send, recv = stage[:blk], stage[blk:]
        per = blk // self.world

        def cb(_user, d_send, d_recv, _bytes_per_peer):
            try:
                ok = d_send == send.data_ptr() and d_recv == recv.data_ptr()
                self._send[rank] = send if ok else None
                self._barrier.wait()
                peers = list(self._send)
                if any(s is None for s in peers):
                    ok = False
                else:
                    for p_, s in enumerate(peers):
                        recv[p_ * per:(p_ + 1) * per].copy_(s[rank * per:(rank + 1) * per])
                    torch.cuda.synchronize(stage.device)
                self._barrier.wait()               # the peers have read this rank's send half
                return 0 if ok else 1
            except Exception:                      # never raise through the C frame (a broken barrier included)
                self._barrier.abort()              # the peers fail at once instead of waiting out the barrier's timeout
                return 1
        return _lib.ALLTOALL_FN(cb)


class _DevPtr:
    """A raw device pointer as a ``__cuda_array_interface__`` object: ``torch.as_tensor`` wraps it without a copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def _dev_tensor(ptr: int, nbytes: int, device):
    import torch
    return torch.as_tensor(_DevPtr(ptr, nbytes), device=device)


class DistGroup:
    """The two exchange callbacks of the distributed prover (``pm_plonk_*_dist``: coefficient-range ownership end to
    end) over one of two transports: the default ``torch.distributed`` process group (one
    process per rank; gloo rehearsals move the bytes through the host) or a :class:`LocalGroup` (ranks = threads of this
    process).  ``desc`` is the ``pm_dist`` to pass; with ``native=True`` both callbacks are NULL and the library uses
    its own RCCL communicator (``Context.comm_init`` first)."""

    def __init__(self, rank: int | None = None, world: int | None = None, local: "LocalGroup | None" = None,
                 device="cuda", native: bool = False, host_staging_ctx: Context | None = None):
        """host_staging_ctx (LocalGroup only): carry the all-to-all through host memory with this context's
        ``pm_dev_download`` / ``pm_dev_upload`` instead of torch device copies -- for runs where torch cannot touch the GPU
        (the host-AddressSanitizer build of the library: torch's CUDA initialisation fails under the preloaded runtime)."""
        self.local, self.device, self._stage_ctx = local, device, host_staging_ctx
        if local is not None:
            self.world, self.rank = local.world, int(rank)
            tdist = None
        else:
            import torch.distributed as tdist
            self.world = tdist.get_world_size() if tdist.is_initialized() else 1
            self.rank = tdist.get_rank() if tdist.is_initialized() else 0
        self._ag = None if native else _lib.ALLGATHER_FN(self._allgather)
        self._aa = None if native else _lib.ALLTOALL_FN(self._alltoall)
        self.desc = _lib.Dist(self.world, self.rank, C.cast(self._ag, C.c_void_p) if self._ag else None,
                              C.cast(self._aa, C.c_void_p) if self._aa else None, None)

    def _allgather(self, _user, msg, gathered):
        try:
            words = _lib.COMM_MSG_WORDS
            mine = np.ctypeslib.as_array(msg, shape=(words,)).copy()
            out = np.ctypeslib.as_array(gathered, shape=(self.world, words))
            if self.local is not None:
                out[:] = self.local.allgather_words(self.rank, mine)
            elif self.world == 1:
                out[0] = mine
            else:
                import torch
                import torch.distributed as tdist
                t = torch.from_numpy(mine.view(np.int64))
                dev = self.device if tdist.get_backend() == "nccl" else "cpu"
                parts = [torch.empty_like(t, device=dev) for _ in range(self.world)]
                tdist.all_gather(parts, t.to(dev))
                out[:] = torch.stack(parts).cpu().numpy().view(np.uint64)
            return 0
        except Exception:            # never unwind through the C frames
            self._abort_local()
            return 1

    def _abort_local(self):
        """A rank of a LocalGroup that fails outside the barrier (a bad pointer, a torch error) breaks the barrier for
        its peers: they get BrokenBarrierError -> 1 at once instead of after the barrier's 600 s timeout."""
        if self.local is not None:
            try:
                self.local._barrier.abort()
            except Exception:        # noqa: BLE001
                pass

    def _alltoall(self, _user, d_send, d_recv, bytes_per_peer):
        try:
            nbytes = bytes_per_peer * self.world
            if self.local is not None and self._stage_ctx is not None:
                c = self._stage_ctx
                host = np.empty(nbytes // 8, np.uint64)
                c._check(c._lib.pm_dev_download(c._h, host.ctypes.data_as(C.c_void_p), C.c_void_p(d_send), nbytes))
                got = self.local.alltoall_host(self.rank, host)
                c._check(c._lib.pm_dev_upload(c._h, C.c_void_p(d_recv), got.ctypes.data_as(C.c_void_p), nbytes))
                return 0
            import torch
            send, recv = _dev_tensor(d_send, nbytes, self.device), _dev_tensor(d_recv, nbytes, self.device)
            if self.local is not None:
                return self.local.alltoall_tensors(self.rank, send, recv)
            import torch.distributed as tdist
            if tdist.get_backend() != "nccl":                     # gloo rehearsal: exchange on the host
                s, r = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
                tdist.all_to_all_single(r, s)
                recv.copy_(r)
            else:
                tdist.all_to_all_single(recv, send)
            torch.cuda.synchronize()
            return 0
        except Exception:
            self._abort_local()
            return 1


class ShardedCommitKey:
    """``CommitKey`` whose ``powers_of_g`` are split by index range over the ranks of the default
    process group (BASELINE.json configs[4]): this rank holds powers [lo, lo + len(powers_slice)) of
    `total`.  ``commit_batch_dev`` runs the local slice of every MSM, then one all_gather + fold."""

    def __init__(self, powers_slice, lo: int, total: int, ctx: Context, device=None, precompute: bool = False,
                 native: bool = False, group: "LocalGroup | None" = None, rank: int = 0):
        """native: exchange through the library's own RCCL communicator (``Context.comm_init`` first) instead
        of torch.distributed -- what a host without PyTorch does.  group / rank: the ranks are threads of this
        process (:class:`LocalGroup`).  powers_slice: affine points [len, 12], or a ready ``Bases``."""
        self.ctx, self.lo, self.total, self.device, self.native = ctx, lo, total, device, native
        self.group, self.rank = group, rank
        self._bases = powers_slice if isinstance(powers_slice, Bases) else Bases(ctx, powers_slice)
        if precompute and self._bases.n:
            self._bases.precompute()
        one = np.array([0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745,
                        0x5C071A97A256EC6D, 0x15F65EC3FA80E493], dtype=np.uint64)   # Fp Montgomery 1
        self._identity = np.zeros(18, np.uint64)
        self._identity[6:12] = one

    @classmethod
    def setup(cls, total: int, tau_int: int, lo: int, hi: int, ctx: Context, **kw) -> "ShardedCommitKey":
        """This rank's slice tau^lo G .. tau^(hi-1) G of ``PublicParameters::setup``'s commit key, generated on
        its GPU (``CommitKey.setup`` with the powers started at tau^lo).  tau_int: the trapdoor as an integer --
        tests and benchmarks only, as upstream's ``setup`` is."""
        from .field import R_MOD, fr_to_limbs
        from .host import G1_GENERATOR, DeviceVector, _p
        cnt = hi - lo
        powers = DeviceVector(ctx, cnt)
        pts = DeviceVector(ctx, 3 * cnt)
        try:
            ctx.fr_powers(fr_to_limbs(tau_int % R_MOD), fr_to_limbs(pow(tau_int, lo, R_MOD)), cnt, powers.ptr)
            ctx._check(ctx._lib.pm_g1_fixed_base_mul_dev(ctx._h, _p(G1_GENERATOR), powers._p, cnt,
                                                         _lib.SCALAR_MONTGOMERY, pts._p, None))
            bases = Bases.from_device(ctx, pts.ptr, cnt)
        finally:
            pts.free()
            powers.free()
        return cls(bases, lo, total, ctx, **kw)

    def gather_fold(self, partials_xyz):
        """One exchange: [k, 18] partial points (None = abort marker) -> the sums over all ranks, or None when a
        rank gave up.  The transport is whatever this key was built for."""
        if self.group is not None:
            return self.group.allgather_fold_many(self.rank, partials_xyz)
        return allgather_fold_many(partials_xyz, self.device)

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
        folded = self.ctx.g1_allgather_fold(part) if self.native else self.gather_fold(part)
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
