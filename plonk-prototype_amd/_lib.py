"""ctypes declarations of the C ABI in ``include/plonk_mi355x.h`` (one entry per export)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIB_PATH: load another build of the same library (A/B runs of compiler flags)
LIB_PATH = os.environ.get("PM_LIB_PATH") or os.path.join(_HERE, "lib", "libplonk_mi355x.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)



class PermArgs(C.Structure):
    """``pm_plonk_perm_args``"""
    _fields_ = [("wires", C.c_void_p * 4), ("sigmas", C.c_void_p * 4), ("roots", C.c_void_p),
                ("beta", C.c_uint64 * 4), ("gamma", C.c_uint64 * 4), ("k", (C.c_uint64 * 4) * 3)]


class QuotientArgs(C.Structure):
    """``pm_plonk_quotient_args``"""
    _fields_ = [("wires", C.c_void_p * 4), ("z", C.c_void_p), ("q_m", C.c_void_p), ("q_l", C.c_void_p),
                ("q_r", C.c_void_p), ("q_o", C.c_void_p), ("q_4", C.c_void_p), ("q_c", C.c_void_p),
                ("pi", C.c_void_p), ("sigmas", C.c_void_p * 4), ("l1", C.c_void_p), ("x", C.c_void_p),
                ("alpha", C.c_uint64 * 4), ("beta", C.c_uint64 * 4), ("gamma", C.c_uint64 * 4),
                ("k", (C.c_uint64 * 4) * 3), ("zh_inv", (C.c_uint64 * 4) * 4),
                ("q_arith", C.c_void_p), ("q_range", C.c_void_p), ("q_logic", C.c_void_p),
                ("q_fixed_group_add", C.c_void_p), ("q_variable_group_add", C.c_void_p),
                ("range_sep", C.c_uint64 * 4), ("logic_sep", C.c_uint64 * 4), ("fixed_sep", C.c_uint64 * 4),
                ("var_sep", C.c_uint64 * 4)]


class PlonkProof(C.Structure):
    """``pm_plonk_proof``"""
    _fields_ = [("commitments", (C.c_uint64 * 12) * 11), ("evaluations", (C.c_uint64 * 4) * 17),
                ("challenges", (C.c_uint64 * 4) * 10)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, u64p, C.c_uint32)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, u64p, u64p)
COMM_MSG_WORDS = 289


class Dist(C.Structure):
    """``pm_dist``"""
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("allgather", C.c_void_p), ("alltoall", C.c_void_p),
                ("user", C.c_void_p)]


ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
VK_POINTS = (C.c_uint64 * 12) * 15
PLONK_SELECTORS, PLONK_PROOF_BYTES, PLONK_BIND_PUBLIC_INPUTS, PLONK_UPSTREAM_TRANSCRIPT = 11, 1040, 1, 2
COMM_ID_BYTES, COMM_MAX_POINTS = 128, 16
LINCOMB_MAX = 16

# name -> (restype, argtypes); must list every function the header declares
SIGNATURES = {
    "pm_version": (C.c_char_p, []),
    "pm_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pm_shutdown": (None, [C.c_void_p]),
    "pm_last_error": (C.c_char_p, [C.c_void_p]),
    "pm_sync": (C.c_int, [C.c_void_p]),
    "pm_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "pm_trim": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pm_domain_info": (C.c_int, [C.c_uint32, u64p, u64p, u64p]),
    "pm_domain_prepare": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pm_domain_evaluate_vanishing_polynomial": (C.c_int, [C.c_uint32, u64p, u64p]),
    "pm_domain_vanishing_poly_over_coset": (C.c_int, [C.c_uint32, C.c_uint64, u64p]),
    "pm_domain_vanishing_poly_over_coset_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pm_domain_evaluate_all_lagrange_coefficients": (C.c_int, [C.c_uint32, u64p, u64p]),
    "pm_domain_evaluate_all_lagrange_coefficients_dev": (C.c_int, [C.c_void_p, C.c_uint32, u64p, C.c_void_p, C.c_void_p]),
    "pm_fr_ntt": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, C.c_uint32, C.c_uint32]),
    "pm_fr_ntt_batch": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p, C.c_size_t,
                                  C.c_uint32, C.c_uint32, C.c_uint32]),
    "pm_fr_ntt_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "pm_fr_ntt_fourstep_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_void_p, C.c_void_p]),
    "pm_fr_ntt_fourstep_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pm_comm_stats": (C.c_int, [C.c_void_p, u64p, C.c_int]),
    "pm_test_host_field_op": (C.c_int, [C.c_int, u64p, u64p, u64p, C.c_size_t]),
    "pm_g1_bases_upload": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pm_g1_bases_from_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pm_g1_fixed_base_mul_dev": (C.c_int, [C.c_void_p, u64p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                           C.c_void_p]),
    "pm_g1_bases_precompute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "pm_g1_bases_free": (None, [C.c_void_p, C.c_void_p]),
    "pm_g1_bases_len": (C.c_size_t, [C.c_void_p]),
    "pm_g1_msm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, u64p, C.c_uint32, u64p]),
    "pm_g1_msm_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                C.c_uint32, u64p, C.c_void_p]),
    "pm_g1_msm_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.c_uint32, C.c_uint32, u64p, C.c_void_p]),
    "pm_g1_fold": (C.c_int, [u64p, C.c_size_t, u64p]),
    "pm_g1_to_affine": (C.c_int, [u64p, u64p, C.POINTER(C.c_int)]),
    "pm_g1_to_affine_batch": (C.c_int, [u64p, C.c_size_t, u64p, C.POINTER(C.c_int)]),
    "pm_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pm_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pm_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pm_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pm_fr_vec_op_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_size_t, C.c_void_p]),
    "pm_fr_poly_evaluate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.c_void_p]),
    "pm_fr_poly_ruffini_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, u64p, C.c_void_p, C.c_void_p]),
    "pm_fr_prefix_product_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pm_fr_batch_inverse_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_fr_powers_dev": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pm_fr_lincomb_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), u64p, C.c_size_t, C.c_void_p,
                                    C.c_void_p]),
    "pm_plonk_perm_terms_dev": (C.c_int, [C.c_void_p, C.POINTER(PermArgs), C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "pm_plonk_quotient_dev": (C.c_int, [C.c_void_p, C.POINTER(QuotientArgs), C.c_size_t, C.c_void_p, C.c_void_p]),
    "pm_plonk_preprocess": (C.c_int, [C.c_void_p, C.POINTER(u64p), C.POINTER(C.c_int64), C.c_size_t,
                                      C.POINTER(C.c_void_p)]),
    "pm_plonk_key_free": (None, [C.c_void_p, C.c_void_p]),
    "pm_plonk_key_commit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p]),
    "pm_plonk_key_commit_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                              C.c_char_p, C.c_void_p]),
    "pm_plonk_verifier_key": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pm_plonk_prove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64p, u64p, C.c_size_t, C.c_uint32,
                                 C.POINTER(PlonkProof)]),
    "pm_plonk_prove_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, u64p, u64p,
                                         C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(PlonkProof)]),
    "pm_plonk_transcript_labels": (C.c_char_p, []),
    "pm_plonk_preprocess_dist": (C.c_int, [C.c_void_p, C.POINTER(Dist), C.POINTER(u64p), C.POINTER(C.c_int64), C.c_size_t,
                                           C.POINTER(C.c_void_p)]),
    "pm_plonk_dist_key_free": (None, [C.c_void_p, C.c_void_p]),
    "pm_plonk_dist_key_bytes": (C.c_size_t, [C.c_void_p]),
    "pm_plonk_key_commit_dist": (C.c_int, [C.c_void_p, C.POINTER(Dist), C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p]),
    "pm_plonk_prove_dist": (C.c_int, [C.c_void_p, C.POINTER(Dist), C.c_void_p, C.c_void_p, C.c_void_p, u64p, u64p, C.c_size_t,
                                      C.c_uint32, C.POINTER(PlonkProof)]),
    "pm_plonk_proof_to_bytes": (C.c_int, [C.POINTER(PlonkProof), C.POINTER(C.c_uint8)]),
    "pm_fr_poly_evaluate_many_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.c_size_t, u64p, u64p,
                                               C.c_void_p]),
    "pm_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "pm_comm_init": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "pm_comm_destroy": (C.c_int, [C.c_void_p]),
    "pm_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pm_g1_allgather_fold": (C.c_int, [C.c_void_p, u64p, C.c_uint32]),
    "pm_test_fold_gathered": (C.c_int, [u64p, C.c_int, C.c_uint32, u64p]),
    "pm_test_comm_deadline": (C.c_int, [C.c_long, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "pm_keccak_f1600": (None, [C.c_char_p]),
    "pm_ntt_plan": (C.c_int, [C.c_uint32, u32p, u32p]),
    "pm_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    "pm_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pm_profile_select": (C.c_int, [C.c_void_p, C.c_char_p]),
    "pm_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "pm_test_field_op": (C.c_int, [C.c_void_p, C.c_int, u64p, u64p, u64p, C.c_size_t]),
    "pm_test_ntt_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_long, C.c_long, C.c_long, C.c_uint32, C.POINTER(C.c_uint32)]),
    "pm_test_msm_sizing": (C.c_int, [C.c_size_t, C.c_uint32, C.c_long, C.c_uint32, C.c_uint32, u64p]),
    "pm_test_msm_geometry": (C.c_int, [C.c_size_t, C.c_long, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}

PM_OK = 0
PM_ERR_BAD_ARG = -1
PM_ERR_DOMAIN_TOO_LARGE = -2
PM_ERR_OOM = -3
PM_ERR_HIP = -4
PM_ERR_NO_DEVICE = -5
PM_ERR_LENGTH = -6
PM_ERR_EXCHANGE = -7
PM_ERR_BUSY = -8

NTT_INVERSE = 1
NTT_COSET = 2
NTT_TRANSPOSED = 4   # pm_fr_ntt_fourstep_dev: block-transposed order between a forward and an inverse transform
SCALAR_MONTGOMERY = 0
SCALAR_CANONICAL = 1

_lib = None


class BackendMissing(RuntimeError):
    """The HIP extension is not built / not loadable.  There is no CPU fallback."""


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so with
    the same SONAME (libamdhip64.so.7) as /opt/rocm's.  Whichever loads first serves both, but
    if the system copy wins, torch later loads its bundled copy *as well* (it asks for it by
    file name) and that second runtime finds no GPU.  So when torch is installed, load its
    copy first; our library's DT_NEEDED then resolves to it and torch tensors, streams and
    our kernels share one runtime.  Without torch the library uses /opt/rocm via its RUNPATH."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """dlopen the in-tree HIP library and bind every symbol.  Fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C plonk-prototype_amd/csrc`).  This package has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared export
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
