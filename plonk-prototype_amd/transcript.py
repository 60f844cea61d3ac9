"""Fiat-Shamir transcript of the prover: Merlin over STROBE-128 / Keccak-f[1600], plus the
``TranscriptProtocol`` extension dusk-plonk layers on top.

merlin 2.x is a dependency of dusk-plonk 0.8.2 (ref:Cargo.toml:19); neither is in the reference
tree, so this restates the published constructions:

* STROBE-128 as Merlin uses it (``strobe.rs``: R = 166, operations meta-AD, AD, PRF, KEY);
* ``Transcript::new / append_message / challenge_bytes`` framing (label, little-endian u32 length);
* dusk's ``append_commitment`` (48-byte zcash-compressed G1), ``append_scalar`` (32-byte
  little-endian canonical), ``challenge_scalar`` (64 bytes reduced mod r, ``from_bytes_wide``).

Pinned by Merlin's own known-answer vector (tests/test_transcript.py) and by SHA3-256 of the same
Keccak permutation against hashlib.  The labels the prover uses live in prover.py and are NOT
pinned (upstream's label strings are not available here).  Host-side, a few hundred bytes per proof.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .field import P_MOD, R_MOD, fp_from_limbs, fr_from_limbs

_MASK = (1 << 64) - 1
_native = None
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
    0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
    0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
    0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]


def _rol(x: int, n: int) -> int:
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _MASK if n else x


def keccak_f1600(state: bytearray) -> None:
    """In-place Keccak-f[1600] on a 200-byte state: the library's ``pm_keccak_f1600``."""
    global _native
    if _native is None:
        from . import _lib
        try:
            _native = _lib.load().pm_keccak_f1600
        except _lib.BackendMissing:          # host-only hashing: usable before the HIP library is built
            _native = False
    if _native is False:
        return keccak_f1600_py(state)
    buf = (ctypes.c_char * 200).from_buffer(state)
    _native(buf)


def keccak_f1600_py(state: bytearray) -> None:
    """The same permutation in plain Python (lane (x, y) at byte 8 (x + 5 y), little-endian); kept as
    the cross-check of the native one in tests/test_transcript.py."""
    a = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


class Strobe128:
    """The subset of STROBE-128 (v1.0.2) Merlin needs."""
    R = 166
    FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32

    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[self.R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos, self.pos_begin = 0, 0

    def _absorb(self, data: bytes):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _overwrite(self, data: bytes):
        for byte in data:
            self.state[self.pos] = byte
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            if self.cur_flags != flags:
                raise ValueError("continued operation with different flags")
            return
        if flags & self.FLAG_T:
            raise ValueError("transport operations are not used by Merlin")
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (self.FLAG_C | self.FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_M | self.FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool) -> bytes:
        self._begin_op(self.FLAG_I | self.FLAG_A | self.FLAG_C, more)
        return self._squeeze(n)

    def key(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_A | self.FLAG_C, more)
        self._overwrite(data)


def g1_compress(xy) -> bytes:
    """Affine G1 (Montgomery limbs [12], (0, 0) = identity) -> 48-byte zcash compressed encoding
    (``G1Affine::to_compressed``): big-endian x, bit 7 = compressed, bit 6 = infinity, bit 5 = y is
    the lexicographically larger root."""
    xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(12)
    if not xy.any():
        return bytes([0xC0]) + bytes(47)
    x, y = fp_from_limbs(xy[:6]), fp_from_limbs(xy[6:])
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80
    if y > (P_MOD - 1) // 2:
        out[0] |= 0x20
    return bytes(out)


def g1_decompress(data: bytes) -> np.ndarray:
    """48-byte zcash compressed encoding -> affine Montgomery limbs [12] (``G1Affine::from_compressed``
    without the subgroup check).  Raises ValueError for malformed input / x not on the curve."""
    from .field import fp_to_limbs
    if len(data) != 48 or not data[0] & 0x80:
        raise ValueError("not a 48-byte compressed G1 point")
    inf, big = bool(data[0] & 0x40), bool(data[0] & 0x20)
    x = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:], "big")
    if inf:
        if x or big:
            raise ValueError("malformed point at infinity")
        return np.zeros(12, np.uint64)
    if x >= P_MOD:
        raise ValueError("x is not reduced")
    rhs = (pow(x, 3, P_MOD) + 4) % P_MOD
    y = pow(rhs, (P_MOD + 1) // 4, P_MOD)                 # p = 3 mod 4
    if y * y % P_MOD != rhs:
        raise ValueError("x is not on the curve")
    if (y > (P_MOD - 1) // 2) != big:
        y = P_MOD - y
    return np.concatenate([fp_to_limbs(x), fp_to_limbs(y)])


class Transcript:
    """``merlin::Transcript`` with dusk-plonk's ``TranscriptProtocol`` methods."""

    def __init__(self, label: bytes):
        self._strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    # ---- merlin ------------------------------------------------------------------------
    def append_message(self, label: bytes, message: bytes):
        self._strobe.meta_ad(label, False)
        self._strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self._strobe.ad(message, False)

    def append_u64(self, label: bytes, x: int):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self._strobe.meta_ad(label, False)
        self._strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self._strobe.prf(n, False)

    # ---- dusk_plonk::transcript::TranscriptProtocol ---------------------------------------
    def append_commitment(self, label: bytes, xy):
        self.append_message(label, g1_compress(xy))

    def append_scalar(self, label: bytes, limbs):
        self.append_message(label, fr_from_limbs(limbs).to_bytes(32, "little"))

    def challenge_scalar(self, label: bytes) -> int:
        """64 challenge bytes as a little-endian integer, reduced mod r (``from_bytes_wide``)."""
        return int.from_bytes(self.challenge_bytes(label, 64), "little") % R_MOD

    def circuit_domain_sep(self, n: int):
        self.append_message(b"dom-sep", b"circuit_size")
        self.append_u64(b"n", n)
