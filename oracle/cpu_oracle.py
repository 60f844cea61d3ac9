"""ctypes front-end of the C CPU restatement (``oracle/c/plonk_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see the header of ``bigint_oracle.py``.  PARITY UNPINNED
(no reference vectors exist; pinned by first-principles constants + identities).

All arrays are ``numpy.uint64``:  Fr = ``[..., 4]`` little-endian Montgomery limbs
(the memory layout of ``&[BlsScalar]``), affine G1 = ``[..., 12]`` = x[6] | y[6]
Montgomery with (0,0) for the identity, Jacobian G1 = ``[18]``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
INVERSE = 1
COSET = 2
SCALAR_MONTGOMERY = 0
SCALAR_CANONICAL = 1

_u64p = C.POINTER(C.c_uint64)


def build(native: bool = False) -> str:
    """Compile the restatement with gcc (seconds).  Returns the .so path."""
    out = "_build_native" if native else "_build"
    args = ["make", "-s", "-C", _HERE] + (["native"] if native else [])
    subprocess.check_call(args)
    return os.path.join(_HERE, out, "libplonk_oracle.so")


def _ptr(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


class CpuOracle:
    def __init__(self, native: bool = False):
        path = os.path.join(_HERE, "_build_native" if native else "_build", "libplonk_oracle.so")
        if not os.path.exists(path):
            path = build(native)
        self.path = path
        L = self.lib = C.CDLL(path)
        L.orc_init.restype = None
        L.orc_fr_ntt.restype = C.c_int
        L.orc_fr_ntt.argtypes = [_u64p, C.c_uint, C.c_uint, C.c_int]
        L.orc_g1_msm.restype = C.c_int
        L.orc_g1_msm.argtypes = [_u64p, _u64p, C.c_size_t, C.c_uint, _u64p, C.c_int]
        L.orc_msm_window_bits.restype = C.c_uint
        L.orc_msm_window_bits.argtypes = [C.c_size_t]
        L.orc_g1_is_on_curve.restype = C.c_int
        L.orc_g1_is_on_curve.argtypes = [_u64p]
        for name, at in {
            "orc_fr_to_mont": [_u64p, C.c_size_t],
            "orc_fr_from_mont": [_u64p, C.c_size_t],
            "orc_fp_to_mont": [_u64p, C.c_size_t],
            "orc_fp_from_mont": [_u64p, C.c_size_t],
            "orc_fr_mul": [_u64p, _u64p, _u64p, C.c_size_t],
            "orc_fp_mul": [_u64p, _u64p, _u64p, C.c_size_t],
            "orc_fr_sample": [C.c_uint64, C.c_size_t, _u64p],
            "orc_g1_jacobian_to_affine": [_u64p, _u64p],
            "orc_g1_projective_to_affine": [_u64p, _u64p],
            "orc_g1_generator": [_u64p],
            "orc_g1_mul": [_u64p, _u64p, _u64p],
            "orc_g1_add_affine": [_u64p, _u64p, _u64p],
            "orc_g1_bases_arith": [_u64p, _u64p, C.c_size_t, _u64p, C.c_int],
            "orc_expected_dlog": [_u64p, C.c_size_t, C.c_uint, _u64p, _u64p, _u64p],
            "orc_constants": [_u64p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p],
            "orc_fr_vec_op": [C.c_int, _u64p, _u64p, C.c_size_t, _u64p, C.c_size_t],
            "orc_fr_batch_inverse": [_u64p, C.c_size_t],
            "orc_fr_poly_evaluate": [_u64p, C.c_size_t, _u64p, _u64p],
            "orc_fr_poly_ruffini": [_u64p, C.c_size_t, _u64p, _u64p],
            "orc_fr_prefix_product": [_u64p, C.c_size_t, _u64p],
            "orc_fr_powers": [_u64p, _u64p, C.c_size_t, _u64p],
            "orc_fr_lincomb": [C.c_uint, C.POINTER(_u64p), _u64p, C.c_size_t, _u64p, C.c_int],
            "orc_fr_batch_inverse_trick": [_u64p, C.c_size_t, C.c_int],
            "orc_plonk_perm_terms": [C.POINTER(_u64p), C.POINTER(_u64p), _u64p, _u64p, _u64p, C.c_size_t, _u64p, _u64p,
                                     C.c_int],
            "orc_plonk_quotient": [C.POINTER(_u64p), C.c_size_t, _u64p, _u64p, _u64p, _u64p, _u64p, C.c_int],
            "orc_plonk_widget_values": [_u64p, _u64p, _u64p],
        }.items():
            getattr(L, name).restype = None
            getattr(L, name).argtypes = at
        L.orc_init()

    # ------------------------------------------------------------------ fields
    def fr_to_mont(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.orc_fr_to_mont(_ptr(a), a.size // 4)
        return a

    def fr_from_mont(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.orc_fr_from_mont(_ptr(a), a.size // 4)
        return a

    def fp_to_mont(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.orc_fp_to_mont(_ptr(a), a.size // 6)
        return a

    def fp_from_mont(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.orc_fp_from_mont(_ptr(a), a.size // 6)
        return a

    def fr_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.empty_like(a)
        self.lib.orc_fr_mul(_ptr(r), _ptr(a), _ptr(b), a.size // 4)
        return r

    def fp_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.empty_like(a)
        self.lib.orc_fp_mul(_ptr(r), _ptr(a), _ptr(b), a.size // 6)
        return r

    def fr_sample(self, seed: int, n: int) -> np.ndarray:
        out = np.empty((n, 4), dtype=np.uint64)
        self.lib.orc_fr_sample(seed, n, _ptr(out))
        return out

    def constants(self) -> dict:
        fr_one = np.zeros(4, np.uint64); fr_r2 = np.zeros(4, np.uint64)
        fr_inv = np.zeros(1, np.uint64); fr_root = np.zeros(4, np.uint64)
        fr_gen = np.zeros(4, np.uint64); fp_one = np.zeros(6, np.uint64)
        fp_inv = np.zeros(1, np.uint64)
        self.lib.orc_constants(_ptr(fr_one), _ptr(fr_r2), _ptr(fr_inv), _ptr(fr_root),
                               _ptr(fr_gen), _ptr(fp_one), _ptr(fp_inv))
        return dict(fr_one=fr_one, fr_r2=fr_r2, fr_inv=int(fr_inv[0]), fr_root=fr_root,
                    fr_gen=fr_gen, fp_one=fp_one, fp_inv=int(fp_inv[0]))

    # ------------------------------------------------------- EvaluationDomain
    def fr_ntt(self, a, log_n: int, flags: int = 0, threads: int = 1) -> np.ndarray:
        """a: [len<=2^log_n, 4] Montgomery.  Zero-pads like fft(); returns [2^log_n, 4]."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        if log_n >= 32:
            raise ValueError("log_n >= TWO_ADICITY")
        n = 1 << log_n
        if a.shape[0] > n:
            raise ValueError("input longer than domain")
        buf = np.zeros((n, 4), dtype=np.uint64)
        buf[: a.shape[0]] = a
        rc = self.lib.orc_fr_ntt(_ptr(buf), log_n, flags, threads)
        if rc:
            raise ValueError(f"orc_fr_ntt rc={rc}")
        return buf

    # ------------------------------------------------- Polynomial / Evaluations
    def fr_vec_op(self, op: int, a, b) -> np.ndarray:
        """op 0 add, 1 sub, 2 mul; b of one element broadcasts."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.empty_like(a)
        if a.shape[0]:
            self.lib.orc_fr_vec_op(op, _ptr(a), _ptr(b), b.shape[0], _ptr(out), a.shape[0])
        return out

    def fr_batch_inverse(self, a) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
        if a.shape[0]:
            self.lib.orc_fr_batch_inverse(_ptr(a), a.shape[0])
        return a

    def fr_poly_evaluate(self, coeffs, point) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        p = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
        out = np.zeros(4, np.uint64)
        if c.shape[0]:
            self.lib.orc_fr_poly_evaluate(_ptr(c), c.shape[0], _ptr(p), _ptr(out))
        return out

    def fr_poly_ruffini(self, coeffs, z) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(4)
        out = np.zeros((max(c.shape[0] - 1, 0), 4), np.uint64)
        if c.shape[0] > 1:
            self.lib.orc_fr_poly_ruffini(_ptr(c), c.shape[0], _ptr(zz), _ptr(out))
        return out

    def fr_prefix_product(self, a) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        out = np.empty_like(a)
        if a.shape[0]:
            self.lib.orc_fr_prefix_product(_ptr(a), a.shape[0], _ptr(out))
        return out

    # ------------------------------------------------- prover rounds
    @staticmethod
    def _ptr_array(arrs):
        keep = [np.ascontiguousarray(a, dtype=np.uint64) for a in arrs]
        return (_u64p * len(keep))(*[_ptr(a) for a in keep]), keep

    def fr_powers(self, base, scale, n: int) -> np.ndarray:
        b, sc = (np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (base, scale))
        out = np.zeros((n, 4), np.uint64)
        if n:
            self.lib.orc_fr_powers(_ptr(b), _ptr(sc), n, _ptr(out))
        return out

    def fr_lincomb(self, coeffs, vecs, threads: int = 1) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        ptrs, keep = self._ptr_array(vecs)
        n = keep[0].size // 4
        out = np.zeros((n, 4), np.uint64)
        self.lib.orc_fr_lincomb(len(keep), ptrs, _ptr(c), n, _ptr(out), threads)
        return out

    def fr_batch_inverse_trick(self, a, threads: int = 1) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
        if a.shape[0]:
            self.lib.orc_fr_batch_inverse_trick(_ptr(a), a.shape[0], threads)
        return a

    def plonk_perm_terms(self, wires, sigmas, roots, beta, gamma, threads: int = 1):
        wp, k1 = self._ptr_array(wires)
        sp, k2 = self._ptr_array(sigmas)
        r = np.ascontiguousarray(roots, dtype=np.uint64).reshape(-1, 4)
        n = r.shape[0]
        b, g = (np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (beta, gamma))
        num, den = np.zeros((n, 4), np.uint64), np.zeros((n, 4), np.uint64)
        self.lib.orc_plonk_perm_terms(wp, sp, _ptr(r), _ptr(b), _ptr(g), n, _ptr(num), _ptr(den), threads)
        return num, den

    def plonk_quotient(self, arrays23, n: int, alpha, beta, gamma, seps, threads: int = 1) -> np.ndarray:
        """arrays23: w0..w3, z, q_m, q_l, q_r, q_o, q_4, q_c, pi, s0..s3, l1, x, q_arith, q_range, q_logic,
        q_fixed_group_add, q_variable_group_add -- each [4n, 4]; seps: [4, 4] separation challenges."""
        ptrs, keep = self._ptr_array(arrays23)
        assert len(keep) == 23 and all(a.size == 16 * n for a in keep)
        a, b, g = (np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (alpha, beta, gamma))
        sp = np.ascontiguousarray(seps, dtype=np.uint64).reshape(16)
        out = np.zeros((4 * n, 4), np.uint64)
        self.lib.orc_plonk_quotient(ptrs, n, _ptr(a), _ptr(b), _ptr(g), _ptr(sp), _ptr(out), threads)
        return out

    def plonk_widget_values(self, seps, row) -> np.ndarray:
        """seps [4, 4]; row [10, 4] = a b c d a_next b_next d_next q_l q_r q_c -> [4, 4] range, logic, fixed, var."""
        sp = np.ascontiguousarray(seps, dtype=np.uint64).reshape(16)
        rw = np.ascontiguousarray(row, dtype=np.uint64).reshape(40)
        out = np.zeros((4, 4), np.uint64)
        self.lib.orc_plonk_widget_values(_ptr(sp), _ptr(rw), _ptr(out))
        return out

    # ----------------------------------------------------------------------- G1
    def g1_generator(self) -> np.ndarray:
        g = np.zeros(12, np.uint64)
        self.lib.orc_g1_generator(_ptr(g))
        return g

    def g1_mul(self, xy, k_canonical) -> np.ndarray:
        xy = np.ascontiguousarray(xy, dtype=np.uint64)
        k = np.ascontiguousarray(k_canonical, dtype=np.uint64)
        out = np.zeros(12, np.uint64)
        self.lib.orc_g1_mul(_ptr(xy), _ptr(k), _ptr(out))
        return out

    def g1_add(self, a, b) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros(12, np.uint64)
        self.lib.orc_g1_add_affine(_ptr(a), _ptr(b), _ptr(out))
        return out

    def g1_is_on_curve(self, xy) -> bool:
        xy = np.ascontiguousarray(xy, dtype=np.uint64)
        return bool(self.lib.orc_g1_is_on_curve(_ptr(xy)))

    def g1_jacobian_to_affine(self, xyz) -> np.ndarray:
        xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
        out = np.zeros(12, np.uint64)
        self.lib.orc_g1_jacobian_to_affine(_ptr(xyz), _ptr(out))
        return out

    def g1_projective_to_affine(self, xyz) -> np.ndarray:
        xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
        out = np.zeros(12, np.uint64)
        self.lib.orc_g1_projective_to_affine(_ptr(xyz), _ptr(out))
        return out

    def g1_bases_arith(self, k0, d, n: int, threads: int = 1) -> np.ndarray:
        """P_i = (k0 + i d) G, i < n.  k0, d: canonical 4-limb arrays."""
        k0 = np.ascontiguousarray(k0, dtype=np.uint64)
        d = np.ascontiguousarray(d, dtype=np.uint64)
        out = np.zeros((n, 12), np.uint64)
        self.lib.orc_g1_bases_arith(_ptr(k0), _ptr(d), n, _ptr(out), threads)
        return out

    def expected_dlog(self, scalars, scalar_form, k0, d) -> np.ndarray:
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        k0 = np.ascontiguousarray(k0, dtype=np.uint64)
        d = np.ascontiguousarray(d, dtype=np.uint64)
        out = np.zeros(4, np.uint64)
        self.lib.orc_expected_dlog(_ptr(scalars), scalars.shape[0], scalar_form, _ptr(k0),
                                   _ptr(d), _ptr(out))
        return out

    def msm_window_bits(self, n: int) -> int:
        return int(self.lib.orc_msm_window_bits(n))

    def g1_msm(self, points_xy, scalars, scalar_form: int = SCALAR_MONTGOMERY,
               threads: int = 1) -> np.ndarray:
        """msm_variable_base.  Returns the AFFINE result [12] ((0,0) = identity)."""
        p = np.ascontiguousarray(points_xy, dtype=np.uint64).reshape(-1, 12)
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        if p.shape[0] != s.shape[0]:
            raise ValueError("points/scalars length mismatch")
        out = np.zeros(18, np.uint64)
        if p.shape[0] == 0:  # numpy gives a dangling pointer for empty arrays
            p = np.zeros((1, 12), np.uint64)
            s = np.zeros((1, 4), np.uint64)
            n = 0
        else:
            n = p.shape[0]
        self.lib.orc_g1_msm(_ptr(p), _ptr(s), n, scalar_form, _ptr(out), threads)
        return self.g1_jacobian_to_affine(out)


# ---------------------------------------------------------------- int <-> limbs
def ints_to_limbs(vals, nlimbs: int) -> np.ndarray:
    out = np.zeros((len(vals), nlimbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(nlimbs):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a) -> list[int]:
    a = np.asarray(a, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    return [sum(int(a[i, j]) << (64 * j) for j in range(a.shape[1])) for i in range(a.shape[0])]
