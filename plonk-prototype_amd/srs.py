"""Raw (uncompressed, unchecked) commit-key bytes, the on-disk form either side of the MSM
(SURVEY.md section 8f row N4).

Layout restated from dusk-plonk 0.8.2 ``CommitKey::to_raw_var_bytes`` / ``from_slice_unchecked`` and
dusk-bls12_381 0.8 ``G1Affine::to_raw_bytes`` (ref:Cargo.toml:19-20; neither crate is in the
reference tree -- UNPINNED, verified by round trip only):

    u64 little-endian   number of points n
    n x 97 bytes        x: 6 x u64 LE Montgomery limbs | y: 6 x u64 LE Montgomery limbs | infinity: 1 byte

i.e. the in-memory ``G1Affine`` without padding -- which is also this backend's affine layout (the
identity is (0, 0) here and carries infinity = 1 in the file).
"""
from __future__ import annotations

import numpy as np

G1_RAW = 97


def commit_key_to_raw_bytes(powers_of_g) -> bytes:
    p = np.ascontiguousarray(powers_of_g, dtype=np.uint64).reshape(-1, 12)
    n = p.shape[0]
    out = np.zeros((n, G1_RAW), np.uint8)
    out[:, :96] = p.view(np.uint8).reshape(n, 96)
    out[:, 96] = (~p.any(axis=1)).astype(np.uint8)
    return n.to_bytes(8, "little") + out.tobytes()


def commit_key_from_raw_bytes(data: bytes) -> np.ndarray:
    """-> powers_of_g [n, 12] Montgomery limbs.  Unchecked like upstream's ``from_slice_unchecked``
    (no on-curve / subgroup test); raises ValueError only for a malformed length."""
    if len(data) < 8:
        raise ValueError("truncated commit key")
    n = int.from_bytes(data[:8], "little")
    if len(data) != 8 + n * G1_RAW:
        raise ValueError("commit key length does not match its point count")
    raw = np.frombuffer(data, dtype=np.uint8, offset=8).reshape(n, G1_RAW)
    pts = raw[:, :96].copy().view(np.uint64).reshape(n, 12)
    pts[raw[:, 96] != 0] = 0
    return pts
