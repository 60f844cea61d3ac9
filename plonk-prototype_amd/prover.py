"""Host-side mirror of ``dusk_plonk::proof_system::{Prover::preprocess, Prover::prove_with_preprocessed,
Proof}`` (dusk-plonk 0.8.2, ref:Cargo.toml:19 -- the crate is not in the reference tree, so the formulas
and transcript labels are restated from the published 0.8 design: "parity unpinned") over the native
prover of the library (``pm_plonk_preprocess`` / ``pm_plonk_key_commit`` / ``pm_plonk_prove``:
``csrc/prover.hip``).  SURVEY.md section 8f rows N1 + N2 + N3, BASELINE.json configs[3].

  gates     q_arith (q_m a b + q_l a + q_r b + q_o c + q_4 d + q_c) + PI
            + q_range R + q_logic L + q_fixed_group_add F + q_variable_group_add V = 0
            (the gate kinds the reference's gadgets emit: ref:src/zk/gadgets.rs:34,37,40,88-91,211)
  copy      sigma over the cosets {1, K1, K2, K3} H            (K = 7, 13, 17)
  round 1   wire polynomials (iNTT), commitments
  round 2   beta, gamma; grand product z; commitment
  round 3   alpha + four separation challenges; quotient t on the 4n coset, split in four, commitments
  round 4   evaluation challenge; 16 openings + t(z); linearisation polynomial r
  round 5   two aggregate opening witnesses W_z, W_zw; commitments

Every polynomial stays in HBM from the witness upload to the last commitment: the five rounds run
inside ONE C-ABI call (what a Rust prover binds).  This module only marshals arguments, and holds the
verifier's side of the transcript (``derive_challenges``) for the end-to-end checks in tests/.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .field import R_MOD, fr_from_limbs, fr_to_limbs
from .host import CommitKey, Context, DeviceVector
from .transcript import Transcript

# order of pm_plonk_preprocess's selector array (= dusk's ProverKey widgets)
SELECTORS = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
             "q_variable_group_add")
# VerifierKey::seed_transcript order
_SEED_ORDER = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "q_range", "q_logic", "q_variable_group_add",
               "q_fixed_group_add")
VK_NAMES = SELECTORS + ("sigma_1", "sigma_2", "sigma_3", "sigma_4")
CHALLENGES = ("beta", "gamma", "alpha", "range_sep", "logic_sep", "fixed_sep", "var_sep", "z", "aw", "aw_shifted")

_labels_cache: dict | None = None


def transcript_labels() -> dict:
    """Every label string of the transcript, from the ONE table both sides use (csrc/prover.hip, namespace tl, through
    ``pm_plonk_transcript_labels``): {key: bytes}.  The labels are restated from the published dusk-plonk 0.8 design --
    parity-unpinned -- and that table is the single place to edit when upstream vectors become available."""
    global _labels_cache
    if _labels_cache is None:
        text = _lib.load().pm_plonk_transcript_labels().decode()
        _labels_cache = {k: v.encode() for k, v in (line.split("=", 1) for line in text.splitlines() if line)}
    return _labels_cache


@dataclass
class Circuit:
    """Selector evaluations on H ([n, 4] Montgomery limbs each; None = identically zero) and the copy
    permutation: ``sigma_index[j, i] = j' * n + i'`` means wire j of gate i is followed by wire j' of gate i'."""
    sigma_index: np.ndarray
    q_m: np.ndarray
    q_l: np.ndarray
    q_r: np.ndarray
    q_o: np.ndarray
    q_c: np.ndarray
    q_4: np.ndarray
    q_arith: np.ndarray
    q_range: np.ndarray | None = None
    q_logic: np.ndarray | None = None
    q_fixed_group_add: np.ndarray | None = None
    q_variable_group_add: np.ndarray | None = None

    @property
    def n(self) -> int:
        return self.q_m.shape[0]


@dataclass
class Proof:
    """``dusk_plonk::proof_system::Proof``: 11 commitments (affine [12]) and the opening evaluations ([4]
    Montgomery limbs), plus t(z) (which the verifier recomputes) and the challenges for the tests."""
    commitments: dict = field(default_factory=dict)
    evaluations: dict = field(default_factory=dict)
    challenges: dict = field(default_factory=dict)   # ints; recomputable from the transcript
    native_bytes: bytes = b""                        # pm_plonk_proof_to_bytes of the same proof

    COMMITMENTS = ("a", "b", "c", "d", "z", "t_1", "t_2", "t_3", "t_4", "w_z", "w_zw")
    # transcript order (pm_plonk_proof.evaluations) / ProofEvaluations::to_bytes order
    TRANSCRIPT_EVALS = ("a", "b", "c", "d", "a_next", "b_next", "d_next", "sigma_1", "sigma_2", "sigma_3", "q_arith",
                        "q_c", "q_l", "q_r", "z_next", "t", "r")
    EVALUATIONS = ("a", "b", "c", "d", "a_next", "b_next", "d_next", "q_arith", "q_c", "q_l", "q_r", "sigma_1",
                   "sigma_2", "sigma_3", "r", "z_next")

    def to_bytes(self) -> bytes:
        """``Proof::to_bytes``: 11 x 48-byte compressed G1 then the 16 x 32-byte little-endian canonical
        scalars of ``ProofEvaluations::to_bytes`` (1040 bytes)."""
        from .transcript import g1_compress
        return b"".join(g1_compress(self.commitments[k]) for k in self.COMMITMENTS) + \
            b"".join(fr_from_limbs(self.evaluations[k]).to_bytes(32, "little") for k in self.EVALUATIONS)

    @classmethod
    def from_bytes(cls, data: bytes) -> "Proof":
        from .transcript import g1_decompress
        if len(data) != 48 * len(cls.COMMITMENTS) + 32 * len(cls.EVALUATIONS):
            raise ValueError("wrong proof length")
        p, off = cls(), 0
        for k in cls.COMMITMENTS:
            p.commitments[k] = g1_decompress(data[off:off + 48])
            off += 48
        for k in cls.EVALUATIONS:
            v = int.from_bytes(data[off:off + 32], "little")
            if v >= R_MOD:
                raise ValueError("scalar is not reduced")
            p.evaluations[k] = fr_to_limbs(v)
            off += 32
        return p


class ProverKey:
    """``pm_prover_key``: selector and sigma polynomials as coefficients and on the 4n coset, the coset
    points, L_1, 1/Z_H and the per-proof workspace -- built and owned by the library, all in HBM."""

    def __init__(self, circuit: Circuit, ctx: Context):
        self.ctx, self.n = ctx, circuit.n
        keep = []
        ptrs = (_lib.u64p * len(SELECTORS))()
        for i, s in enumerate(SELECTORS):
            a = getattr(circuit, s)
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
                if a.shape[0] != self.n:
                    raise ValueError(f"selector {s} has the wrong length")
                keep.append(a)
                ptrs[i] = a.ctypes.data_as(_lib.u64p)
        idx = np.ascontiguousarray(circuit.sigma_index, dtype=np.int64).reshape(-1)
        h = C.c_void_p()
        ctx._check(ctx._lib.pm_plonk_preprocess(ctx._h, ptrs, idx.ctypes.data_as(C.POINTER(C.c_int64)), self.n, C.byref(h)))
        self._h = h
        self.verifier_key: dict | None = None
        self.label = b"plonk"

    def commit(self, ck, label: bytes = b"plonk") -> dict:
        """``Prover::preprocess``'s second half: commit to the 15 polynomials of the key (the verifier
        key's G1 part) and seed the transcript every proof starts from.  -> {name: affine [12]}."""
        ctx = self.ctx
        vk = _lib.VK_POINTS()
        if hasattr(ck, "lo"):                   # dist.ShardedCommitKey
            cb = None if ck.native else _exchange_callback(ck)      # None: the library's RCCL communicator
            ctx._check(ctx._lib.pm_plonk_key_commit_sharded(ctx._h, self._h, ck._bases._h, ck.lo,
                                                            C.cast(cb, C.c_void_p) if cb else None, None, label,
                                                            C.byref(vk)))
        else:
            ctx._check(ctx._lib.pm_plonk_key_commit(ctx._h, self._h, ck._bases._h, label, C.byref(vk)))
        self.verifier_key = {nm: np.array(vk[i], dtype=np.uint64) for i, nm in enumerate(VK_NAMES)}
        self.label = label
        return self.verifier_key

    def free(self):
        if getattr(self, "_h", None) and self.ctx._h:
            self.ctx._lib.pm_plonk_key_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DistProverKey:
    """``pm_dist_key``: this rank's share of a prover key with coefficient-range ownership end to end (rows /
    coefficients [rank n / world, (rank + 1) n / world) of every vector; SURVEY.md section 8f N5, configs[4]).
    ``circuit`` is the WHOLE circuit here (the slices are cut from it; a host that never holds the whole circuit
    passes slices to ``pm_plonk_preprocess_dist`` directly); ``group`` a ``dist.DistGroup``."""

    def __init__(self, circuit: Circuit, ctx: Context, group):
        self.ctx, self.n, self.group = ctx, circuit.n, group
        W, r = group.world, group.rank
        if self.n % W:
            raise ValueError("the ranks must divide the circuit size")
        m = self.n // W
        self.m, self.lo = m, r * m
        keep = []
        ptrs = (_lib.u64p * len(SELECTORS))()
        for i, s in enumerate(SELECTORS):
            a = getattr(circuit, s)
            if a is not None:
                a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64).reshape(-1, 4)[self.lo:self.lo + m])
                keep.append(a)
                ptrs[i] = a.ctypes.data_as(_lib.u64p)
        idx = np.ascontiguousarray(np.asarray(circuit.sigma_index, dtype=np.int64).reshape(4, self.n)[:, self.lo:self.lo + m])
        h = C.c_void_p()
        ctx._check(ctx._lib.pm_plonk_preprocess_dist(ctx._h, C.byref(group.desc), ptrs, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     self.n, C.byref(h)))
        self._h = h
        self.verifier_key: dict | None = None

    @property
    def device_bytes(self) -> int:
        return int(self.ctx._lib.pm_plonk_dist_key_bytes(self._h))

    def commit(self, bases_slice, label: bytes = b"plonk") -> dict:
        """bases_slice: ``host.Bases`` holding powers [rank m, (rank + 1) m) of the commit key."""
        vk = _lib.VK_POINTS()
        self.ctx._check(self.ctx._lib.pm_plonk_key_commit_dist(self.ctx._h, C.byref(self.group.desc), self._h, bases_slice._h,
                                                               label, C.byref(vk)))
        self.verifier_key = {nm: np.array(vk[i], dtype=np.uint64) for i, nm in enumerate(VK_NAMES)}
        return self.verifier_key

    def prove(self, bases_slice, witness, public_inputs=None, bind_public_inputs: bool = True) -> "Proof":
        """witness: the WHOLE [4, n, 4] wire values (this rank uploads its [4, m] slices) or a DeviceVector of this
        rank's 4m elements; public inputs as for ``prove`` (global positions)."""
        ctx, m = self.ctx, self.m
        if isinstance(witness, DeviceVector):
            d_wit, own = witness, False
        else:
            w = np.asarray(witness, dtype=np.uint64).reshape(4, self.n, 4)[:, self.lo:self.lo + m]
            d_wit, own = DeviceVector.from_host(ctx, np.ascontiguousarray(w).reshape(4 * m, 4)), True
        if isinstance(public_inputs, tuple):
            pos = np.ascontiguousarray(public_inputs[0], dtype=np.uint64).reshape(-1)
            val = np.ascontiguousarray(public_inputs[1], dtype=np.uint64).reshape(-1, 4)
        else:
            pos, val = sparse_public_inputs(public_inputs)
        raw = _lib.PlonkProof()
        flags = 0 if bind_public_inputs else _lib.PLONK_UPSTREAM_TRANSCRIPT
        try:
            ctx._check(ctx._lib.pm_plonk_prove_dist(ctx._h, C.byref(self.group.desc), self._h, bases_slice._h, d_wit._p,
                                                    pos.ctypes.data_as(_lib.u64p) if pos.size else None,
                                                    val.ctypes.data_as(_lib.u64p) if pos.size else None, pos.size, flags,
                                                    C.byref(raw)))
        finally:
            if own:
                d_wit.free()
        return _proof_from_raw(ctx, raw)

    def free(self):
        if getattr(self, "_h", None) and self.ctx._h:
            self.ctx._lib.pm_plonk_dist_key_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _proof_from_raw(ctx: Context, raw) -> "Proof":
    proof = Proof()
    for i, name in enumerate(Proof.COMMITMENTS):
        proof.commitments[name] = np.array(raw.commitments[i], dtype=np.uint64)
    for i, name in enumerate(Proof.TRANSCRIPT_EVALS):
        proof.evaluations[name] = np.array(raw.evaluations[i], dtype=np.uint64)
    for i, name in enumerate(CHALLENGES):
        proof.challenges[name] = fr_from_limbs(np.array(raw.challenges[i], dtype=np.uint64))
    buf = (C.c_uint8 * _lib.PLONK_PROOF_BYTES)()
    ctx._check(ctx._lib.pm_plonk_proof_to_bytes(C.byref(raw), buf))
    proof.native_bytes = bytes(buf)
    return proof


def preprocess(circuit: Circuit, ctx: Context, ck=None, label: bytes = b"plonk") -> ProverKey:
    """``Prover::preprocess``.  With a commit key the verifier key is committed and the transcript seeded
    right away; otherwise ``prove`` does it with its commit key before the first proof."""
    pk = ProverKey(circuit, ctx)
    if ck is not None:
        pk.commit(ck, label)
    return pk


def _exchange_callback(ck):
    """pm_exchange_fn over the key's transport (torch.distributed, or the threads of a LocalGroup): all-gather the k partial points, fold with the group law.
    k = 0 marks a rank that failed locally: the marker is exchanged so that no rank blocks."""
    def exchange(_user, xyz, k):
        try:
            if k == 0:
                ck.gather_fold(None)
                return 1
            buf = np.ctypeslib.as_array(xyz, shape=(k, 18))
            res = ck.gather_fold(buf.copy())
            if res is None:          # a peer gave up
                return 1
            buf[:] = res
            return 0
        except Exception:            # never unwind through the C frames
            return 1
    return _lib.EXCHANGE_FN(exchange)


def sparse_public_inputs(public_inputs) -> tuple[np.ndarray, np.ndarray]:
    """Dense PI evaluations on H ([n, 4] limbs or None) -> (positions [k], values [k, 4]) of the non-zero ones."""
    if public_inputs is None:
        return np.zeros(0, np.uint64), np.zeros((0, 4), np.uint64)
    pi = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    pos = np.flatnonzero(pi.any(axis=1)).astype(np.uint64)
    return pos, np.ascontiguousarray(pi[pos.astype(np.int64)])


def prove(pk: ProverKey, ck: CommitKey, witness, public_inputs=None, bind_public_inputs: bool = True) -> Proof:
    """``Prover::prove_with_preprocessed``: one ``pm_plonk_prove`` call.

    witness: [4, n, 4] wire values (a, b, c, d rows) in Montgomery limbs, or a DeviceVector of 4n elements
    already in HBM.  public_inputs: dense [n, 4] evaluations of PI on H, or a (positions, values) pair, or None.
    bind_public_inputs: absorb the public inputs into the transcript before round 1 (dusk-plonk 0.8.2 does
    not; False reproduces the restated upstream transcript).

    Multi-GPU: every rank calls prove() with the same inputs and a ``dist.ShardedCommitKey``; the polynomial
    work is replicated, each MSM is split by coefficient range, and the ranks exchange 144-byte partial points.
    All ranks return the same proof."""
    ctx, n = pk.ctx, pk.n
    if ck.max_degree() + 1 < n:
        raise ValueError("commit key shorter than the circuit")
    if pk.verifier_key is None:
        pk.commit(ck)
    if isinstance(witness, DeviceVector):
        if witness.n != 4 * n:
            raise ValueError("device witness must hold 4n elements")
        d_wit, own = witness, False
    else:
        d_wit, own = DeviceVector.from_host(ctx, np.ascontiguousarray(witness, dtype=np.uint64).reshape(4 * n, 4)), True
    if isinstance(public_inputs, tuple):
        pos, val = public_inputs
        pos = np.ascontiguousarray(pos, dtype=np.uint64).reshape(-1)
        val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4)
    else:
        if isinstance(public_inputs, DeviceVector):
            public_inputs = public_inputs.to_host()
        pos, val = sparse_public_inputs(public_inputs)
    raw = _lib.PlonkProof()
    flags = 0 if bind_public_inputs else _lib.PLONK_UPSTREAM_TRANSCRIPT   # binding is the library's default
    p_pos = pos.ctypes.data_as(_lib.u64p) if pos.size else None
    p_val = val.ctypes.data_as(_lib.u64p) if pos.size else None
    try:
        if hasattr(ck, "lo"):       # dist.ShardedCommitKey: this rank's slice of the SRS, partial sums exchanged
            cb = None if ck.native else _exchange_callback(ck)      # None: the library's RCCL communicator
            ctx._check(ctx._lib.pm_plonk_prove_sharded(ctx._h, pk._h, ck._bases._h, ck.lo, d_wit._p, p_pos, p_val, pos.size,
                                                       flags, C.cast(cb, C.c_void_p) if cb else None, None, C.byref(raw)))
        else:
            ctx._check(ctx._lib.pm_plonk_prove(ctx._h, pk._h, ck._bases._h, d_wit._p, p_pos, p_val, pos.size, flags,
                                               C.byref(raw)))
    finally:
        if own:
            d_wit.free()
    return _proof_from_raw(ctx, raw)


def seeded_transcript(verifier_key: dict, n: int, label: bytes | None = None) -> Transcript:
    """``Prover::preprocess`` / ``Verifier::preprocess``: the transcript after ``VerifierKey::seed_transcript``."""
    L = transcript_labels()
    ts = Transcript(label if label is not None else L["protocol"])
    for i, nm in enumerate(_SEED_ORDER):
        ts.append_commitment(L[f"selector_{i}"], verifier_key[nm])
    for j in range(4):
        ts.append_commitment(L[f"sigma_{j}"], verifier_key[f"sigma_{j + 1}"])
    ts.append_message(L["dom_sep"], L["dom_sep_value"])
    ts.append_u64(L["circuit_size"], n)
    return ts


def derive_challenges(proof: Proof, verifier_key: dict, n: int, public_inputs=None, bind_public_inputs: bool = True,
                      label: bytes | None = None, t_eval=None) -> dict:
    """The verifier's side of Fiat-Shamir: replay the transcript over the verifier key, the public inputs
    and the proof's commitments and evaluations, and return the challenges (plus "batch", the one that
    folds the two opening checks).  t_eval: the verifier's own t(z); default = the prover's.  Labels and message
    order come from the library's table (``transcript_labels``)."""
    L = transcript_labels()
    ts = seeded_transcript(verifier_key, n, label)
    if bind_public_inputs:
        pos, val = public_inputs if isinstance(public_inputs, tuple) else sparse_public_inputs(public_inputs)
        ts.append_u64(L["pi_len"], len(pos))
        for p_, v_ in zip(pos, val):
            ts.append_u64(L["pi_pos"], int(p_))
            ts.append_scalar(L["pi_value"], v_)
    for j, name in enumerate("abcd"):
        ts.append_commitment(L[f"wire_{j}"], proof.commitments[name])
    ch = {"beta": ts.challenge_scalar(L["beta"])}
    ts.append_scalar(L["beta"], fr_to_limbs(ch["beta"]))
    ch["gamma"] = ts.challenge_scalar(L["gamma"])
    ts.append_commitment(L["perm"], proof.commitments["z"])
    ch["alpha"] = ts.challenge_scalar(L["alpha"])
    ch["range_sep"] = ts.challenge_scalar(L["range_sep"])
    ch["logic_sep"] = ts.challenge_scalar(L["logic_sep"])
    ch["fixed_sep"] = ts.challenge_scalar(L["fixed_sep"])
    ch["var_sep"] = ts.challenge_scalar(L["var_sep"])
    for i in range(4):
        ts.append_commitment(L[f"quotient_{i}"], proof.commitments[f"t_{i + 1}"])
    ch["z"] = ts.challenge_scalar(L["z_challenge"])
    for i, name in enumerate(Proof.TRANSCRIPT_EVALS):
        v = proof.evaluations[name] if not (name == "t" and t_eval is not None) else fr_to_limbs(t_eval)
        ts.append_scalar(L[f"eval_{i}"], v)
    ch["aw"] = ts.challenge_scalar(L["aggregate"])
    ch["aw_shifted"] = ts.challenge_scalar(L["aggregate"])
    ts.append_commitment(L["w_z"], proof.commitments["w_z"])
    ts.append_commitment(L["w_zw"], proof.commitments["w_zw"])
    ch["batch"] = ts.challenge_scalar(L["batch"])
    return ch


def check_identity(proof: Proof, n: int, pi_eval: int = 0) -> bool:
    """The verifier's scalar equation  t(z) Z_H(z) = r(z) + PI(z) - alpha (a + beta s1 + gamma)(b + beta s2 +
    gamma)(c + beta s3 + gamma)(d + gamma) z_w - alpha^2 L_1(z)  on the proof's evaluations
    (``Proof::compute_quotient_evaluation``).  The pairing checks live in oracle/plonk_verifier_oracle.py."""
    ch, ev = proof.challenges, {k: fr_from_limbs(v) for k, v in proof.evaluations.items()}
    beta, gamma, alpha, zc = ch["beta"], ch["gamma"], ch["alpha"], ch["z"]
    zn = pow(zc, n, R_MOD)
    l1_z = (zn - 1) * pow(n * (zc - 1) % R_MOD, -1, R_MOD) % R_MOD
    rhs = (ev["r"] + pi_eval
           - alpha * (ev["a"] + beta * ev["sigma_1"] + gamma) % R_MOD * (ev["b"] + beta * ev["sigma_2"] + gamma) % R_MOD
           * (ev["c"] + beta * ev["sigma_3"] + gamma) % R_MOD * (ev["d"] + gamma) % R_MOD * ev["z_next"]
           - alpha * alpha % R_MOD * l1_z) % R_MOD
    return ev["t"] * (zn - 1) % R_MOD == rhs
