"""Host-side BLS12-381 scalar/base field constants and the int <-> limb conversions the prover's
host logic needs (challenges, a handful of per-proof scalars).  Bulk data never goes through
here: vectors live on the device as ``dusk_bls12_381::BlsScalar`` memory (4 x u64 Montgomery)."""
from __future__ import annotations

import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_R256 = (1 << 256) % R_MOD
_R256_INV = pow(_R256, -1, R_MOD)
_R384_INV = pow((1 << 384) % P_MOD, -1, P_MOD)
GENERATOR = 7              # multiplicative generator = coset shift (dusk_bls12_381::GENERATOR)
K1, K2, K3 = 7, 13, 17     # coset representatives of the permutation argument (dusk_plonk::permutation::constants)


def fr_to_limbs(v: int) -> np.ndarray:
    """int -> Montgomery limbs [4] (the memory of a ``BlsScalar``)."""
    m = (v % R_MOD) * _R256 % R_MOD
    return np.frombuffer(m.to_bytes(32, "little"), dtype=np.uint64).copy()


def fr_from_limbs(a) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(4)
    return int.from_bytes(a.tobytes(), "little") * _R256_INV % R_MOD


def fr_vec_to_limbs(vals) -> np.ndarray:
    """list of ints -> [n, 4] Montgomery limbs (host loop: for circuit construction, not hot)."""
    buf = bytearray(32 * len(vals))
    for i, v in enumerate(vals):
        buf[32 * i:32 * i + 32] = ((v % R_MOD) * _R256 % R_MOD).to_bytes(32, "little")
    return np.frombuffer(bytes(buf), dtype=np.uint64).reshape(-1, 4).copy()


def fr_vec_from_limbs(a) -> list[int]:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") * _R256_INV % R_MOD for i in range(a.shape[0])]


def fp_from_limbs(a) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(6)
    return int.from_bytes(a.tobytes(), "little") * _R384_INV % P_MOD


def fp_to_limbs(v: int) -> np.ndarray:
    m = (v % P_MOD) * ((1 << 384) % P_MOD) % P_MOD
    return np.frombuffer(m.to_bytes(48, "little"), dtype=np.uint64).copy()
