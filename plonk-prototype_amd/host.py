"""Host-side mirror of the reference's dependency interface for the hot path.

Names, argument meaning and error behaviour follow dusk-plonk 0.8.2 / dusk-bls12_381 0.8
(the crates pinned by ref:Cargo.toml:19-20):

* ``EvaluationDomain::new(num_coeffs)`` -> :class:`EvaluationDomain`; ``fft``, ``ifft``,
  ``coset_fft``, ``coset_ifft`` zero-pad to ``size`` and return a new vector.
* ``msm_variable_base(points, scalars)`` -> :func:`msm_variable_base`.
* ``CommitKey { powers_of_g }.commit(poly)`` -> :class:`CommitKey`.

Vectors are ``numpy.uint64`` arrays in the Rust types' memory layout: Fr ``[n, 4]``
Montgomery limbs, affine G1 ``[n, 12]`` (x | y, (0, 0) = identity), projective ``[18]``.
Everything here is plumbing: the arithmetic is in ``csrc/`` behind ``include/plonk_mi355x.h``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import u64p


class Error(Exception):
    """Non-zero ``pm_status`` from the C ABI (``code``) with the library's message."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"plonk_mi355x error {code}: {msg}")
        self.code = code


def _p(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def _fr(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.size % 4:
        raise ValueError("Fr array must have 4 limbs per element")
    return a.reshape(-1, 4)


class Context:
    """One GPU, one stream, cached twiddle tables (``pm_ctx``).  Fails loudly without a
    gfx950 device or without the built HIP library -- there is no CPU path."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.pm_init(device, C.byref(h))
        if rc != _lib.PM_OK:
            raise Error(rc, "pm_init failed (no usable gfx950 device?)")
        self._h = h
        self.device = device

    def _check(self, rc: int):
        if rc != _lib.PM_OK:
            raise Error(rc, self._lib.pm_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pm_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self._lib.pm_sync(self._h))

    def stream_handle(self) -> int:
        """``pm_ctx_stream``: the context's own HIP stream (what ``stream = 0`` means in the ``*_dev`` calls), e.g. for
        ``torch.cuda.ExternalStream`` -- to order other work after the library's or to record timing events on it."""
        return int(self._lib.pm_ctx_stream(self._h) or 0)

    def trim(self) -> int:
        """``pm_trim``: release the cached workspaces and twiddle tables (regrown on demand) -> bytes freed."""
        freed = C.c_size_t(0)
        self._check(self._lib.pm_trim(self._h, C.byref(freed)))
        return freed.value

    def set_option(self, key: str, value: int):
        self._check(self._lib.pm_set_option(self._h, key.encode(), int(value)))

    def profile(self, on: bool, only: str | None = None):
        """Per-kernel event timers; `only` restricts them to one kernel name (less launch overhead)."""
        self._check(self._lib.pm_profile_select(self._h, only.encode() if only else None))
        self._check(self._lib.pm_profile_enable(self._h, 1 if on else 0))

    def profile_read(self) -> dict:
        """{kernel: (launches, total_ms)} accumulated since profile(True)."""
        buf = C.create_string_buffer(1 << 16)
        self._check(self._lib.pm_profile_read(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    # ---- raw ABI calls -------------------------------------------------------------
    # ---- the exchange step inside the library (RCCL; one process per GPU) ----------------------------
    def comm_init(self, rank: int | None = None, world: int | None = None, device=None):
        """Create this rank's RCCL communicator (``pm_comm_init``).  Rank 0 makes the unique id; its 128
        bytes travel to the other ranks through the default ``torch.distributed`` process group (any
        backend: the id is host data), which must already be initialised for world > 1."""
        if world is None:
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
        ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        status = 0
        if rank == 0:
            status = int(self._lib.pm_comm_unique_id(ident))
        if world > 1:
            # Rank 0 broadcasts (status, id) UNCONDITIONALLY and every rank checks the status afterwards: if the id could not
            # be made, all ranks leave this function the same way, in step -- r04 raised on rank 0 before the broadcast and left
            # the other ranks waiting in it.
            import torch
            import torch.distributed as dist
            t = torch.tensor([status & 0xff] + list(bytes(ident)), dtype=torch.uint8)
            if device is not None:
                t = t.to(device)
            dist.broadcast(t, src=0)
            got = t.cpu().tolist()
            if rank != 0:
                status = -(256 - got[0]) if got[0] else 0          # the error codes are small negative numbers
            ident = (C.c_uint8 * _lib.COMM_ID_BYTES)(*got[1:])
        if status:
            if rank == 0:
                self._check(status)
            raise Error(status, f"rank 0 could not create the communicator id (code {status})")
        self._check(self._lib.pm_comm_init(self._h, ident, rank, world))
        self.comm_world = world

    def comm_destroy(self):
        self._check(self._lib.pm_comm_destroy(self._h))
        self.comm_world = 1

    def g1_allgather_fold(self, partials_xyz) -> np.ndarray:
        """[k, 18] partial points of this rank -> [k, 18] sums over all ranks (``pm_g1_allgather_fold``)."""
        p = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 18).copy()
        self._check(self._lib.pm_g1_allgather_fold(self._h, _p(p), p.shape[0]))
        return p

    def fr_ntt(self, a, log_n: int, flags: int = 0, out=None) -> np.ndarray:
        a = _fr(a)
        n = 1 << log_n if log_n < 64 else 0
        if out is None:
            out = np.empty((n if log_n < 32 else 0, 4), dtype=np.uint64)
        src = a if a.shape[0] else np.zeros((1, 4), np.uint64)
        self._check(self._lib.pm_fr_ntt(self._h, _p(src), a.shape[0], _p(out), log_n, flags))
        return out

    def fr_ntt_batch(self, a, log_n: int, flags: int = 0, out=None) -> np.ndarray:
        """a: [batch, in_len, 4] -> [batch, 2^log_n, 4] (into `out` when given)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        batch, in_len = a.shape[0], a.shape[1]
        n = 1 << log_n
        if out is None:
            out = np.empty((batch, n, 4), dtype=np.uint64)
        self._check(self._lib.pm_fr_ntt_batch(self._h, _p(a), in_len, in_len, _p(out), n, log_n,
                                              batch, flags))
        return out

    def fr_ntt_dev(self, d_in: int, in_len: int, d_out: int, log_n: int, flags: int = 0,
                   batch: int = 1, in_stride: int | None = None, out_stride: int | None = None,
                   stream: int = 0):
        """Device pointers (ints, e.g. ``tensor.data_ptr()``); asynchronous on ``stream``."""
        n = 1 << log_n
        self._check(self._lib.pm_fr_ntt_dev(
            self._h, C.c_void_p(d_in), in_len, in_stride if in_stride is not None else in_len,
            C.c_void_p(d_out), out_stride if out_stride is not None else n, log_n, batch, flags,
            C.c_void_p(stream)))

    def fr_ntt_fourstep_dev(self, d_inout: int, d_stage: int, log_n: int, world: int, rank: int, flags: int = 0,
                            exchange=None):
        """``pm_fr_ntt_fourstep_dev``: this rank's N / world block of a 2^log_n-point transform, in place;
        ``exchange`` = an ``_lib.ALLTOALL_FN`` or None (the context's RCCL communicator)."""
        cb = exchange if exchange is not None else C.cast(None, _lib.ALLTOALL_FN)
        self._check(self._lib.pm_fr_ntt_fourstep_dev(self._h, C.c_void_p(d_inout), C.c_void_p(d_stage), log_n, world,
                                                     rank, flags, C.cast(cb, C.c_void_p), None))

    def fr_ntt_fourstep_batch_dev(self, d_inout: int, batch: int, d_stage: int, log_n: int, world: int, rank: int,
                                  flags: int = 0, exchange=None, d_halo: int = 0):
        """``pm_fr_ntt_fourstep_batch_dev``: `batch` contiguous N / world blocks through ONE sequence of exchanges;
        ``d_halo`` (forward + NTT_TRANSPOSED only): batch x N2 elements that receive the next rank's first row."""
        cb = exchange if exchange is not None else C.cast(None, _lib.ALLTOALL_FN)
        self._check(self._lib.pm_fr_ntt_fourstep_batch_dev(self._h, C.c_void_p(d_inout), batch, C.c_void_p(d_halo or None),
                                                           C.c_void_p(d_stage), log_n, world, rank, flags,
                                                           C.cast(cb, C.c_void_p), None))

    def comm_stats(self, reset: bool = False) -> dict:
        """``pm_comm_stats``: exchange counters of this context since the last reset."""
        out = np.zeros(4, np.uint64)
        self._check(self._lib.pm_comm_stats(self._h, _p(out), 1 if reset else 0))
        return {"alltoall_calls": int(out[0]), "alltoall_bytes": int(out[1]), "allgather_calls": int(out[2]),
                "transpose_steps": int(out[3])}

    # ---- device-pointer forms of the polynomial helpers (ints; used by prover.py) -------
    def fr_powers(self, base, scale, n: int, d_out: int):
        """out[i] = scale * base^i."""
        b, sc = (np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (base, scale))
        self._check(self._lib.pm_fr_powers_dev(self._h, _p(b), _p(sc), n, C.c_void_p(d_out), None))

    def fr_lincomb(self, d_vecs, coeffs, n: int, d_out: int):
        """out = sum_j coeffs[j] * vecs[j]."""
        k = len(d_vecs)
        ptrs = (C.c_void_p * k)(*[C.c_void_p(int(v)) for v in d_vecs])
        c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(k, 4)
        self._check(self._lib.pm_fr_lincomb_dev(self._h, k, ptrs, _p(c), n, C.c_void_p(d_out), None))

    def fr_vec_op(self, op: int, d_a: int, d_b: int, b_len: int, d_out: int, n: int):
        self._check(self._lib.pm_fr_vec_op_dev(self._h, op, C.c_void_p(d_a), C.c_void_p(d_b), b_len,
                                               C.c_void_p(d_out), n, None))

    def fr_evaluate(self, d_coeffs: int, n: int, point) -> np.ndarray:
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
        out = np.zeros(4, np.uint64)
        self._check(self._lib.pm_fr_poly_evaluate_dev(self._h, C.c_void_p(d_coeffs), n, _p(pt), _p(out), None))
        return out

    def fr_evaluate_many(self, d_polys, n: int, point) -> np.ndarray:
        """k polynomials of n coefficients at one point: one kernel pass, one synchronisation.  -> [k, 4]."""
        k = len(d_polys)
        ptrs = (C.c_void_p * k)(*[C.c_void_p(int(p)) for p in d_polys])
        out = np.zeros((k, 4), np.uint64)
        self._check(self._lib.pm_fr_poly_evaluate_many_dev(self._h, k, ptrs, n, _p(_fr(point)), _p(out), None))
        return out

    def fr_ruffini(self, d_coeffs: int, n: int, z, d_out: int):
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(4)
        self._check(self._lib.pm_fr_poly_ruffini_dev(self._h, C.c_void_p(d_coeffs), n, _p(zz), C.c_void_p(d_out), None))

    def fr_prefix_product(self, d_in: int, n: int, d_out: int):
        self._check(self._lib.pm_fr_prefix_product_dev(self._h, C.c_void_p(d_in), n, C.c_void_p(d_out), None))

    def fr_batch_inverse(self, d_inout: int, n: int):
        self._check(self._lib.pm_fr_batch_inverse_dev(self._h, C.c_void_p(d_inout), n, None))

    def plonk_perm_terms(self, args: "_lib.PermArgs", n: int, d_num: int, d_den: int):
        self._check(self._lib.pm_plonk_perm_terms_dev(self._h, C.byref(args), n, C.c_void_p(d_num),
                                                      C.c_void_p(d_den), None))

    def plonk_quotient(self, args: "_lib.QuotientArgs", n: int, d_out: int):
        self._check(self._lib.pm_plonk_quotient_dev(self._h, C.byref(args), n, C.c_void_p(d_out), None))

    def field_op(self, op: int, a, b) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.empty_like(a)
        limbs = 4 if (op < 3 or op == 6) else 6
        self._check(self._lib.pm_test_field_op(self._h, op, _p(a), _p(b), _p(out), a.size // limbs))
        return out


class DeviceVector:
    """Canonical Fr vector in device memory (``pm_dev_alloc``): the currency of the polynomial
    helpers.  ``DeviceVector.from_host(ctx, a)``, ``.to_host()``, ``.ptr`` (int)."""

    def __init__(self, ctx: Context, n: int):
        self.ctx, self.n = ctx, n
        h = C.c_void_p()
        ctx._check(ctx._lib.pm_dev_alloc(ctx._h, max(n, 1) * 32, C.byref(h)))
        self._p = h

    @property
    def ptr(self) -> int:
        return self._p.value or 0

    @classmethod
    def from_host(cls, ctx: Context, a) -> "DeviceVector":
        a = _fr(a)
        v = cls(ctx, a.shape[0])
        if a.shape[0]:
            ctx._check(ctx._lib.pm_dev_upload(ctx._h, v._p, a.ctypes.data_as(C.c_void_p), a.shape[0] * 32))
        return v

    def to_host(self) -> np.ndarray:
        out = np.empty((self.n, 4), np.uint64)
        if self.n:
            self.ctx._check(self.ctx._lib.pm_dev_download(self.ctx._h, out.ctypes.data_as(C.c_void_p), self._p,
                                                          self.n * 32))
        return out

    def view(self, offset: int, n: int) -> "DeviceVector":
        """Non-owning window [offset, offset + n) of this vector (keeps the parent alive)."""
        if offset < 0 or offset + n > self.n:
            raise ValueError("view out of range")
        v = object.__new__(_DeviceView)
        v.ctx, v.n, v._parent = self.ctx, n, self
        v._p = C.c_void_p(self.ptr + 32 * offset)
        return v

    def free(self):
        if getattr(self, "_p", None) and self.ctx._h:
            self.ctx._lib.pm_dev_free(self.ctx._h, self._p)
        self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _DeviceView(DeviceVector):
    def free(self):
        self._p = None
        self._parent = None


class Polynomial:
    """``dusk_plonk::fft::Polynomial`` / ``Evaluations`` on the device: coefficient-wise
    ``+ - *`` (a one-element operand is a scalar), ``evaluate`` and ``ruffini``; ``batch_inverse``
    is ``util::batch_inversion``.  Thin wrappers over the ``pm_fr_*_dev`` entry points."""

    OPS = {"add": 0, "sub": 1, "mul": 2}

    def __init__(self, vec: DeviceVector):
        self.vec = vec

    @classmethod
    def from_host(cls, ctx: Context, coeffs) -> "Polynomial":
        return cls(DeviceVector.from_host(ctx, coeffs))

    def to_host(self) -> np.ndarray:
        return self.vec.to_host()

    def _op(self, op: str, other: "Polynomial") -> "Polynomial":
        ctx = self.vec.ctx
        if other.vec.n not in (1, self.vec.n):
            raise Error(_lib.PM_ERR_LENGTH, "operands differ in length")
        out = DeviceVector(ctx, self.vec.n)
        ctx._check(ctx._lib.pm_fr_vec_op_dev(ctx._h, self.OPS[op], self.vec._p, other.vec._p, other.vec.n, out._p,
                                             self.vec.n, None))
        return Polynomial(out)

    def __add__(self, o):
        return self._op("add", o)

    def __sub__(self, o):
        return self._op("sub", o)

    def __mul__(self, o):
        return self._op("mul", o)

    def evaluate(self, point) -> np.ndarray:
        ctx = self.vec.ctx
        p = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
        out = np.zeros(4, np.uint64)
        ctx._check(ctx._lib.pm_fr_poly_evaluate_dev(ctx._h, self.vec._p, self.vec.n, _p(p), _p(out), None))
        return out

    def ruffini(self, z) -> "Polynomial":
        ctx = self.vec.ctx
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(4)
        out = DeviceVector(ctx, max(self.vec.n - 1, 0))
        ctx._check(ctx._lib.pm_fr_poly_ruffini_dev(ctx._h, self.vec._p, self.vec.n, _p(zz), out._p, None))
        return Polynomial(out)

    def prefix_product(self) -> "Polynomial":
        """[1, a0, a0 a1, ...]: the permutation argument's grand-product accumulator."""
        ctx = self.vec.ctx
        out = DeviceVector(ctx, self.vec.n)
        ctx._check(ctx._lib.pm_fr_prefix_product_dev(ctx._h, self.vec._p, self.vec.n, out._p, None))
        return Polynomial(out)

    def batch_inverse(self) -> "Polynomial":
        """In place; returns self."""
        ctx = self.vec.ctx
        ctx._check(ctx._lib.pm_fr_batch_inverse_dev(ctx._h, self.vec._p, self.vec.n, None))
        ctx.sync()
        return self


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def domain_info(log_n: int):
    """(group_gen, group_gen_inv, size_inv) of the 2^log_n domain.  Host arithmetic only."""
    lib = _lib.load()
    g, gi, si = (np.zeros(4, np.uint64) for _ in range(3))
    rc = lib.pm_domain_info(log_n, _p(g), _p(gi), _p(si))
    if rc != _lib.PM_OK:
        raise Error(rc, "log_size_of_group >= TWO_ADICITY (32)")
    return g, gi, si


def ntt_plan(log_n: int):
    lib = _lib.load()
    radix = (C.c_uint32 * 4)()
    npass = C.c_uint32()
    rc = lib.pm_ntt_plan(log_n, radix, C.byref(npass))
    if rc != _lib.PM_OK:
        raise Error(rc, "log_size_of_group >= TWO_ADICITY (32)")
    return [int(radix[i]) for i in range(npass.value)]


class EvaluationDomain:
    """``dusk_plonk::fft::EvaluationDomain`` (ark-poly ``Radix2EvaluationDomain``)."""

    def __init__(self, num_coeffs: int, ctx: Context | None = None):
        size = 1
        while size < num_coeffs:
            size <<= 1
        self.size = size
        self.log_size_of_group = size.bit_length() - 1
        # InvalidEvalDomainSize in the reference: log2(size) must be < TWO_ADICITY
        self.group_gen, self.group_gen_inv, self.size_inv = domain_info(self.log_size_of_group)
        self._ctx = ctx

    @property
    def ctx(self) -> Context:
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def _run(self, a, flags):
        a = _fr(a)
        if a.shape[0] > self.size:
            # the reference's resize() would silently truncate; no prover call site does that
            raise Error(_lib.PM_ERR_LENGTH, "input longer than the domain")
        return self.ctx.fr_ntt(a, self.log_size_of_group, flags)

    def fft(self, coeffs):
        return self._run(coeffs, 0)

    def ifft(self, evals):
        return self._run(evals, _lib.NTT_INVERSE)

    def coset_fft(self, coeffs):
        return self._run(coeffs, _lib.NTT_COSET)

    def coset_ifft(self, evals):
        return self._run(evals, _lib.NTT_INVERSE | _lib.NTT_COSET)

    def evaluate_vanishing_polynomial(self, tau) -> np.ndarray:
        """``EvaluationDomain::evaluate_vanishing_polynomial``: tau^size - 1 (host arithmetic)."""
        t, out = np.ascontiguousarray(tau, dtype=np.uint64).reshape(4), np.zeros(4, np.uint64)
        rc = _lib.load().pm_domain_evaluate_vanishing_polynomial(self.log_size_of_group, _p(t), _p(out))
        if rc != _lib.PM_OK:
            raise Error(rc, "pm_domain_evaluate_vanishing_polynomial")
        return out

    def evaluate_all_lagrange_coefficients(self, tau, device: bool = False):
        """``EvaluationDomain::evaluate_all_lagrange_coefficients``: [L_0(tau), .., L_(size-1)(tau)] -- on the host
        ([size, 4] limbs), or with ``device=True`` as a :class:`DeviceVector` filled by the library's kernels."""
        t = np.ascontiguousarray(tau, dtype=np.uint64).reshape(4)
        if device:
            v = DeviceVector(self.ctx, self.size)
            self.ctx._check(self.ctx._lib.pm_domain_evaluate_all_lagrange_coefficients_dev(
                self.ctx._h, self.log_size_of_group, _p(t), v._p, None))
            return v
        out = np.zeros((self.size, 4), np.uint64)
        rc = _lib.load().pm_domain_evaluate_all_lagrange_coefficients(self.log_size_of_group, _p(t), _p(out))
        if rc != _lib.PM_OK:
            raise Error(rc, "pm_domain_evaluate_all_lagrange_coefficients")
        return out

    def compute_vanishing_poly_over_coset(self, poly_degree: int, device: bool = False):
        """``fft::domain::compute_vanishing_poly_over_coset(domain, poly_degree)``: the evaluations over this
        domain of X^poly_degree - 1 on the coset GENERATOR * H (in the prover: this = the 4n domain, poly_degree = n).
        Upstream asserts ``size > poly_degree`` (here: PM_ERR_BAD_ARG)."""
        if device:
            v = DeviceVector(self.ctx, self.size)
            self.ctx._check(self.ctx._lib.pm_domain_vanishing_poly_over_coset_dev(
                self.ctx._h, self.log_size_of_group, poly_degree, v._p, None))
            return v
        out = np.zeros((self.size, 4), np.uint64)
        rc = _lib.load().pm_domain_vanishing_poly_over_coset(self.log_size_of_group, poly_degree, _p(out))
        if rc != _lib.PM_OK:
            raise Error(rc, "size > poly_degree violated" if rc == _lib.PM_ERR_BAD_ARG else "log_size_of_group >= 32")
        return out

    def elements(self):
        """All domain elements 1, g, g^2, ... = fft of X (the polynomial with coefficients [0, 1])."""
        if self.size == 1:
            return self.fft(_one_mont()[None, :])
        x = np.zeros((2, 4), np.uint64)
        x[1] = _one_mont()
        return self.fft(x)


def _one_mont() -> np.ndarray:
    return np.array([0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F],
                    dtype=np.uint64)


# G1 generator, affine Montgomery limbs (dusk_bls12_381::G1Affine::generator())
G1_GENERATOR = np.array([0x5CB38790FD530C16, 0x7817FC679976FFF5, 0x154F95C7143BA1C1, 0xF0AE6ACDF3D0E747,
                         0xEDCE6ECC21DBF440, 0x120177419E0BFB75, 0xBAAC93D50CE72271, 0x8C22631A7918FD8E,
                         0xDD595F13570725CE, 0x51AC582950405194, 0x0E1C8C3FAD0059C0, 0x0BBC3EFC5008A26A],
                        dtype=np.uint64)


# --------------------------------------------------------------------------- MSM
class Bases:
    """Device-resident affine bases (``pm_bases``)."""

    def __init__(self, ctx: Context, points_xy):
        p = np.ascontiguousarray(points_xy, dtype=np.uint64).reshape(-1, 12)
        self.ctx = ctx
        self.n = p.shape[0]
        h = C.c_void_p()
        src = p if self.n else np.zeros((1, 12), np.uint64)
        ctx._check(ctx._lib.pm_g1_bases_upload(ctx._h, _p(src), self.n, C.byref(h)))
        self._h = h

    @classmethod
    def from_device(cls, ctx: Context, d_xy: int, n: int) -> "Bases":
        """Affine points already in device memory (n x 96 bytes, ABI layout)."""
        self = object.__new__(cls)
        self.ctx, self.n = ctx, n
        h = C.c_void_p()
        ctx._check(ctx._lib.pm_g1_bases_from_dev(ctx._h, C.c_void_p(d_xy), n, C.byref(h)))
        self._h = h
        return self

    def precompute(self, window_bits: int = 0):
        """Build the resident table of window multiples (``pm_g1_bases_precompute``)."""
        self.ctx._check(self.ctx._lib.pm_g1_bases_precompute(self.ctx._h, self._h, window_bits))
        return self

    def free(self):
        if getattr(self, "_h", None) and self.ctx._h:
            self.ctx._lib.pm_g1_bases_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def msm(self, scalars, scalar_form: int = _lib.SCALAR_MONTGOMERY, n: int | None = None) -> np.ndarray:
        s = _fr(scalars)
        n = s.shape[0] if n is None else n
        out = np.zeros(18, np.uint64)
        src = s if s.shape[0] else np.zeros((1, 4), np.uint64)
        self.ctx._check(self.ctx._lib.pm_g1_msm(self.ctx._h, self._h, n, _p(src), scalar_form, _p(out)))
        return out

    def msm_dev(self, d_scalars: int, n: int, offset: int = 0,
                scalar_form: int = _lib.SCALAR_MONTGOMERY, stream: int = 0) -> np.ndarray:
        out = np.zeros(18, np.uint64)
        self.ctx._check(self.ctx._lib.pm_g1_msm_dev(self.ctx._h, self._h, offset, n, C.c_void_p(d_scalars),
                                                    scalar_form, _p(out), C.c_void_p(stream)))
        return out


def _msm_batch_dev(self, d_scalars: int, n: int, batch: int, stride: int | None = None, offset: int = 0,
                   scalar_form: int = _lib.SCALAR_MONTGOMERY, stream: int = 0) -> np.ndarray:
    """`batch` MSMs over the same bases in one pass -> [batch, 18]."""
    out = np.zeros((batch, 18), np.uint64)
    self.ctx._check(self.ctx._lib.pm_g1_msm_batch_dev(
        self.ctx._h, self._h, offset, n, C.c_void_p(d_scalars), stride if stride is not None else n, batch,
        scalar_form, _p(out), C.c_void_p(stream)))
    return out


Bases.msm_batch_dev = _msm_batch_dev


def msm_variable_base(points, scalars, ctx: Context | None = None,
                      scalar_form: int = _lib.SCALAR_MONTGOMERY) -> np.ndarray:
    """``dusk_bls12_381::multiscalar_mul::msm_variable_base(points, scalars) -> G1Projective``.
    Uploads the bases for this one call; use :class:`CommitKey` to keep an SRS resident."""
    ctx = ctx or default_context()
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    s = _fr(scalars)
    if p.shape[0] != s.shape[0]:
        raise Error(_lib.PM_ERR_LENGTH, "points and scalars differ in length")
    b = Bases(ctx, p)
    try:
        return b.msm(s, scalar_form)
    finally:
        b.free()


def g1_fold(parts) -> np.ndarray:
    lib = _lib.load()
    parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 18)
    out = np.zeros(18, np.uint64)
    rc = lib.pm_g1_fold(_p(parts), parts.shape[0], _p(out))
    if rc != _lib.PM_OK:
        raise Error(rc, "pm_g1_fold")
    return out


def g1_to_affine(xyz):
    """-> (xy[12], is_identity)"""
    lib = _lib.load()
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(18)
    out = np.zeros(12, np.uint64)
    ident = C.c_int()
    rc = lib.pm_g1_to_affine(_p(xyz), _p(out), C.byref(ident))
    if rc != _lib.PM_OK:
        raise Error(rc, "pm_g1_to_affine")
    return out, bool(ident.value)


class CommitKey:
    """``dusk_plonk::commitment_scheme::kzg10::CommitKey``: holds ``powers_of_g`` on the device;
    ``commit`` checks the degree (PolynomialDegreeTooLarge) and runs one MSM."""

    def __init__(self, powers_of_g, ctx: Context | None = None, precompute: bool = False):
        self.ctx = ctx or default_context()
        self._bases = Bases(self.ctx, powers_of_g)
        if precompute:      # a long-lived SRS: trade HBM for ~20 % faster commits
            self._bases.precompute()

    @classmethod
    def setup(cls, max_degree: int, tau, ctx: Context | None = None, precompute: bool = False,
              generator=None, host_copy: bool = False) -> "CommitKey":
        """``PublicParameters::setup(max_degree, rng).commit_key``: powers_of_g[i] = tau^i G for
        i <= max_degree, generated on the GPU (powers of tau, then a fixed-base multiplication) and kept
        there.  tau: Montgomery limbs [4] -- the toxic waste; for tests and benchmarks only, as upstream's
        ``setup`` is.  generator: affine [12], default the G1 generator.  host_copy=True also downloads
        the points into ``self.powers_of_g`` ([n, 12])."""
        ctx = ctx or default_context()
        n = max_degree + 1
        g = G1_GENERATOR if generator is None else np.ascontiguousarray(generator, dtype=np.uint64).reshape(12)
        powers = DeviceVector(ctx, n)
        pts = DeviceVector(ctx, 3 * n)                            # n affine points of 96 bytes
        try:
            ctx.fr_powers(tau, _one_mont(), n, powers.ptr)
            ctx._check(ctx._lib.pm_g1_fixed_base_mul_dev(ctx._h, _p(g), powers._p, n, _lib.SCALAR_MONTGOMERY, pts._p,
                                                         None))
            self = object.__new__(cls)
            self.ctx = ctx
            self._bases = Bases.from_device(ctx, pts.ptr, n)
            self.powers_of_g = pts.to_host().reshape(n, 12) if host_copy else None
        finally:
            pts.free()
            powers.free()
        if precompute:
            self._bases.precompute()
        return self

    def max_degree(self) -> int:
        return self._bases.n - 1

    def commit_many(self, polys) -> np.ndarray:
        """Commit to several polynomials of one length in one pass -> [k, 12] affine."""
        c = np.ascontiguousarray(polys, dtype=np.uint64)
        k, n = c.shape[0], c.shape[1]
        if n > self._bases.n:
            raise Error(_lib.PM_ERR_LENGTH, "PolynomialDegreeTooLarge")
        d = DeviceVector.from_host(self.ctx, c.reshape(-1, 4))
        try:
            xyz = self._bases.msm_batch_dev(d.ptr, n, k)
        finally:
            d.free()
        return np.stack([g1_to_affine(x)[0] for x in xyz])

    def commit_batch_dev(self, d_ptr: int, n: int, batch: int, stride: int | None = None) -> list:
        """Commit to `batch` device-resident coefficient vectors of n elements (vector j at
        d_ptr + 32 j stride) -> list of affine [12].  The prover's form of ``commit``."""
        if n > self._bases.n:
            raise Error(_lib.PM_ERR_LENGTH, "PolynomialDegreeTooLarge")
        xyz = self._bases.msm_batch_dev(d_ptr, n, batch, stride=stride)
        return [g1_to_affine(p)[0] for p in xyz]

    def commit(self, coeffs) -> np.ndarray:
        """-> Commitment as affine G1 [12] ((0, 0) for the zero polynomial)."""
        c = _fr(coeffs)
        if c.shape[0] > self._bases.n:
            raise Error(_lib.PM_ERR_LENGTH, "PolynomialDegreeTooLarge")
        xyz = self._bases.msm(c, _lib.SCALAR_MONTGOMERY)
        return g1_to_affine(xyz)[0]
