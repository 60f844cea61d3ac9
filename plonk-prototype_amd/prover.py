"""PLONK prover rounds with every polynomial resident in HBM (SURVEY.md section 8f rows N1 + N2;
BASELINE.json configs[3] "Full PLONK prove (wire polys + permutation + quotient + KZG opening)").

Restates the round structure of ``dusk_plonk::proof_system::Prover::prove_with_preprocessed``
(dusk-plonk 0.8.2, ref:Cargo.toml:19 -- the crate is not in the reference tree, so this is the
published PLONK protocol in dusk's 4-wire arrangement, "parity unpinned"):

  gate      q_m a b + q_l a + q_r b + q_o c + q_4 d + q_c + PI = 0
  copy      sigma over the cosets {1, K1, K2, K3} H            (K = 7, 13, 17)
  round 1   wire polynomials (iNTT), commitments
  round 2   beta, gamma; grand product z; commitment
  round 3   alpha; quotient t on the 4n coset, split in four, commitments
  round 4   evaluation challenge; openings; linearisation polynomial r
  round 5   aggregation challenge; W_z, W_zw; commitments

Only the arithmetic and permutation identities are built (no range / logic / curve widgets, no
blinding -- 0.8.2 has none).  The host code below only sequences C-ABI calls and does scalar
arithmetic on a dozen challenges; vectors never leave the device between rounds.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .field import GENERATOR, K1, K2, K3, R_MOD, fr_from_limbs, fr_to_limbs
from .host import CommitKey, Context, DeviceVector, domain_info
from .transcript import Transcript

SELECTORS = ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")
OP_ADD, OP_SUB, OP_MUL = 0, 1, 2


@dataclass
class Circuit:
    """Selector evaluations on H ([n, 4] Montgomery limbs each) and the copy permutation:
    ``sigma_index[j, i] = j' * n + i'`` means wire j of gate i is followed by wire j' of gate i'."""
    q_m: np.ndarray
    q_l: np.ndarray
    q_r: np.ndarray
    q_o: np.ndarray
    q_4: np.ndarray
    q_c: np.ndarray
    sigma_index: np.ndarray

    @property
    def n(self) -> int:
        return self.q_m.shape[0]


@dataclass
class Proof:
    """11 commitments (affine [12]) and the opening evaluations ([4] Montgomery limbs)."""
    commitments: dict = field(default_factory=dict)
    evaluations: dict = field(default_factory=dict)
    challenges: dict = field(default_factory=dict)   # ints; recomputable from the transcript

    COMMITMENTS = ("a", "b", "c", "d", "z", "t_1", "t_2", "t_3", "t_4", "w_z", "w_zw")
    EVALUATIONS = ("a", "b", "c", "d", "sigma_1", "sigma_2", "sigma_3", "z_next", "t", "r")

    def to_bytes(self) -> bytes:
        """11 x 48-byte compressed G1 then 10 x 32-byte little-endian canonical scalars (the layout of
        dusk's ``Proof::to_bytes``, which carries 6 more evaluations for the gates not built here)."""
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


def _arr4(v) -> "C.Array":
    return (C.c_uint64 * 4)(*[int(x) for x in np.asarray(v, dtype=np.uint64).reshape(4)])


class ProverKey:
    """``dusk_plonk::proof_system::ProverKey``: selector and sigma polynomials as coefficients (n) and
    as evaluations on the 4n coset, the coset points, L_1 on the coset and 1/Z_H -- all in HBM."""

    def __init__(self, circuit: Circuit, ctx: Context):
        n = circuit.n
        if n < 4 or n & (n - 1):
            raise ValueError("the circuit size must be a power of two >= 4")
        self.ctx, self.n, self.log_n = ctx, n, n.bit_length() - 1
        omega, _, n_inv = domain_info(self.log_n)
        omega4, _, _ = domain_info(self.log_n + 2)
        self.omega, self.omega4, self.n_inv = omega, omega4, n_inv
        one = fr_to_limbs(1)
        self.k = [fr_to_limbs(K1), fr_to_limbs(K2), fr_to_limbs(K3)]
        # H and the 4n coset
        self.roots = DeviceVector(ctx, n)
        ctx.fr_powers(omega, one, n, self.roots.ptr)
        self.x4 = DeviceVector(ctx, 4 * n)
        ctx.fr_powers(omega4, fr_to_limbs(GENERATOR), 4 * n, self.x4.ptr)
        # selectors: evaluations -> coefficients -> 4n coset, one batched transform each way
        ns = len(SELECTORS)
        ev = DeviceVector.from_host(ctx, np.concatenate([getattr(circuit, s) for s in SELECTORS]))
        self.sel_coeffs = DeviceVector(ctx, ns * n)
        ctx.fr_ntt_dev(ev.ptr, n, self.sel_coeffs.ptr, self.log_n, _lib.NTT_INVERSE, batch=ns)
        self.sel_coset = DeviceVector(ctx, ns * 4 * n)
        ctx.fr_ntt_dev(self.sel_coeffs.ptr, n, self.sel_coset.ptr, self.log_n + 2, _lib.NTT_COSET, batch=ns,
                       in_stride=n, out_stride=4 * n)
        ev.free()
        # sigma_j(w^i) = k_j' w^i' : gather from the table of all 4n coset-of-H points
        table = DeviceVector(ctx, 4 * n)
        for j, kj in enumerate([one] + self.k):
            ctx.fr_powers(omega, kj, n, table.ptr + 32 * j * n)
        tab = table.to_host()
        table.free()
        idx = np.ascontiguousarray(circuit.sigma_index, dtype=np.int64).reshape(4 * n)
        if idx.min() < 0 or idx.max() >= 4 * n or np.unique(idx).size != 4 * n:
            raise ValueError("sigma_index is not a permutation of the 4n wire positions")
        self.sigma_evals = DeviceVector.from_host(ctx, tab[idx])
        self.sigma_coeffs = DeviceVector(ctx, 4 * n)
        ctx.fr_ntt_dev(self.sigma_evals.ptr, n, self.sigma_coeffs.ptr, self.log_n, _lib.NTT_INVERSE, batch=4)
        self.sigma_coset = DeviceVector(ctx, 16 * n)
        ctx.fr_ntt_dev(self.sigma_coeffs.ptr, n, self.sigma_coset.ptr, self.log_n + 2, _lib.NTT_COSET, batch=4,
                       in_stride=n, out_stride=4 * n)
        # L_1(X) = (X^n - 1) / (n (X - 1)) = (1/n) sum_i X^i
        l1c = DeviceVector(ctx, n)
        ctx.fr_powers(one, n_inv, n, l1c.ptr)
        self.l1_coset = DeviceVector(ctx, 4 * n)
        ctx.fr_ntt_dev(l1c.ptr, n, self.l1_coset.ptr, self.log_n + 2, _lib.NTT_COSET)
        ctx.sync()
        l1c.free()
        # Z_H(g w4^i) = g^n (w4^n)^i - 1 has period 4 in i
        gn = pow(GENERATOR, n, R_MOD)
        i4 = pow(fr_from_limbs(omega4), n, R_MOD)
        self.zh_inv = [fr_to_limbs(pow((gn * pow(i4, k, R_MOD) - 1) % R_MOD, -1, R_MOD)) for k in range(4)]

    def workspace(self, name: str, n: int) -> DeviceVector:
        """Per-proof scratch, allocated on the first proof and reused by later ones (a prover
        service proves many witnesses against one key; hipMalloc/hipFree of GB-sized buffers
        would otherwise sit inside every proof)."""
        ws = self.__dict__.setdefault("_ws", {})
        v = ws.get(name)
        if v is None or v.n != n:
            if v is not None:
                v.free()
            v = ws[name] = DeviceVector(self.ctx, n)
        return v

    def selector_coeffs(self, name: str) -> int:
        return self.sel_coeffs.ptr + 32 * self.n * SELECTORS.index(name)

    def selector_coset(self, name: str) -> int:
        return self.sel_coset.ptr + 32 * 4 * self.n * SELECTORS.index(name)


def preprocess(circuit: Circuit, ctx: Context) -> ProverKey:
    return ProverKey(circuit, ctx)


def _commit_batch(ck, d_ptr: int, n: int, batch: int, stride: int) -> list:
    # CommitKey on one GPU; dist.ShardedCommitKey when the SRS is split over the ranks of a node
    return ck.commit_batch_dev(d_ptr, n, batch, stride)


def prove(pk: ProverKey, ck: CommitKey, witness, public_inputs=None, transcript: Transcript | None = None) -> Proof:
    """witness: [4, n, 4] wire values (a, b, c, d rows) in Montgomery limbs; public_inputs: [n, 4]
    evaluations of PI on H (None = no public inputs).  Either may be a DeviceVector already in HBM.

    Multi-GPU: every rank calls prove() with the same inputs and a ``dist.ShardedCommitKey``; the
    O(n log n) polynomial work is replicated (13 % of a proof), each of the 11 MSMs is split by
    coefficient range, and the ranks exchange 144-byte partial points.  All ranks return the same proof."""
    ctx, n, log_n = pk.ctx, pk.n, pk.log_n
    if ck.max_degree() + 1 < n:
        raise ValueError("commit key shorter than the circuit")
    ts = transcript or Transcript(b"plonk")
    ts.circuit_domain_sep(n)
    proof = Proof()
    fr = fr_to_limbs

    # ---- round 1: wire polynomials ----------------------------------------------------------------
    if isinstance(witness, DeviceVector):                          # already resident: [a | b | c | d], 4n elements
        if witness.n != 4 * n:
            raise ValueError("device witness must hold 4n elements")
        wire_evals, own_witness = witness, False
    else:
        wire_evals = DeviceVector.from_host(ctx, np.ascontiguousarray(witness, dtype=np.uint64).reshape(4 * n, 4))
        own_witness = True
    # coefficient buffer [a, b, c, d, z, pi], each n long
    coeffs = pk.workspace("coeffs", 6 * n)
    ctx.fr_ntt_dev(wire_evals.ptr, n, coeffs.ptr, log_n, _lib.NTT_INVERSE, batch=4)
    for name, c in zip(("a", "b", "c", "d"), _commit_batch(ck, coeffs.ptr, n, 4, n)):
        proof.commitments[name] = c
        ts.append_commitment(b"w_" + name.encode(), c)

    # ---- round 2: permutation grand product --------------------------------------------------------
    beta, gamma = ts.challenge_scalar(b"beta"), ts.challenge_scalar(b"gamma")
    num, den = pk.workspace("num", n), pk.workspace("den", n)
    pa = _lib.PermArgs()
    for j in range(4):
        pa.wires[j] = wire_evals.ptr + 32 * j * n
        pa.sigmas[j] = pk.sigma_evals.ptr + 32 * j * n
    pa.roots = pk.roots.ptr
    pa.beta, pa.gamma = _arr4(fr(beta)), _arr4(fr(gamma))
    for j in range(3):
        pa.k[j] = _arr4(pk.k[j])
    ctx.plonk_perm_terms(pa, n, num.ptr, den.ptr)
    ctx.fr_batch_inverse(den.ptr, n)
    ctx.fr_vec_op(OP_MUL, num.ptr, den.ptr, n, num.ptr, n)
    ctx.fr_prefix_product(num.ptr, n, den.ptr)                   # den now holds z on H
    z_coeffs = coeffs.ptr + 32 * 4 * n
    ctx.fr_ntt_dev(den.ptr, n, z_coeffs, log_n, _lib.NTT_INVERSE)
    proof.commitments["z"] = _commit_batch(ck, z_coeffs, n, 1, n)[0]
    ts.append_commitment(b"z", proof.commitments["z"])

    # ---- round 3: quotient ---------------------------------------------------------------------------
    alpha = ts.challenge_scalar(b"alpha")
    pi_coeffs = coeffs.ptr + 32 * 5 * n
    if isinstance(public_inputs, DeviceVector):
        pi_ev, own_pi = public_inputs, False
    elif public_inputs is None:                                    # PI = 0: a device-side fill, no upload
        pi_ev, own_pi = pk.workspace("pi_zero", n), False
        ctx.fr_powers(fr(0), fr(0), n, pi_ev.ptr)
    else:
        pi_host = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(n, 4)
        pi_ev, own_pi = DeviceVector.from_host(ctx, pi_host), True
    ctx.fr_ntt_dev(pi_ev.ptr, n, pi_coeffs, log_n, _lib.NTT_INVERSE)
    coset = pk.workspace("coset", 6 * 4 * n)                         # a, b, c, d, z, pi on the 4n coset
    ctx.fr_ntt_dev(coeffs.ptr, n, coset.ptr, log_n + 2, _lib.NTT_COSET, batch=6, in_stride=n, out_stride=4 * n)
    qa = _lib.QuotientArgs()
    for j in range(4):
        qa.wires[j] = coset.ptr + 32 * 4 * n * j
        qa.sigmas[j] = pk.sigma_coset.ptr + 32 * 4 * n * j
    qa.z = coset.ptr + 32 * 4 * n * 4
    qa.pi = coset.ptr + 32 * 4 * n * 5
    for s in SELECTORS:
        setattr(qa, s, pk.selector_coset(s))
    qa.l1, qa.x = pk.l1_coset.ptr, pk.x4.ptr
    qa.alpha, qa.beta, qa.gamma = _arr4(fr(alpha)), _arr4(fr(beta)), _arr4(fr(gamma))
    for j in range(3):
        qa.k[j] = _arr4(pk.k[j])
    for j in range(4):
        qa.zh_inv[j] = _arr4(pk.zh_inv[j])
    t = pk.workspace("t", 4 * n)
    ctx.plonk_quotient(qa, n, t.ptr)
    ctx.fr_ntt_dev(t.ptr, 4 * n, t.ptr, log_n + 2, _lib.NTT_INVERSE | _lib.NTT_COSET)
    for i, c in enumerate(_commit_batch(ck, t.ptr, n, 4, n)):
        proof.commitments[f"t_{i + 1}"] = c
        ts.append_commitment(f"t_{i + 1}".encode(), c)

    # ---- round 4: openings and the linearisation polynomial -----------------------------------------
    zc = ts.challenge_scalar(b"z")
    zc_l = fr(zc)
    zw_l = fr(zc * fr_from_limbs(pk.omega) % R_MOD)
    ev = {}
    for j, name in enumerate(("a", "b", "c", "d")):
        ev[name] = ctx.fr_evaluate(coeffs.ptr + 32 * j * n, n, zc_l)
    for j in range(3):
        ev[f"sigma_{j + 1}"] = ctx.fr_evaluate(pk.sigma_coeffs.ptr + 32 * j * n, n, zc_l)
    ev["z_next"] = ctx.fr_evaluate(z_coeffs, n, zw_l)
    zn = pow(zc, n, R_MOD)
    t_parts = [fr_from_limbs(ctx.fr_evaluate(t.ptr + 32 * i * n, n, zc_l)) for i in range(4)]
    ev["t"] = fr((t_parts[0] + zn * (t_parts[1] + zn * (t_parts[2] + zn * t_parts[3]))) % R_MOD)
    a_, b_, c_, d_ = (fr_from_limbs(ev[k]) for k in ("a", "b", "c", "d"))
    s1, s2, s3 = (fr_from_limbs(ev[f"sigma_{j}"]) for j in (1, 2, 3))
    z_next = fr_from_limbs(ev["z_next"])
    l1_z = (zn - 1) * pow(n * (zc - 1) % R_MOD, -1, R_MOD) % R_MOD
    ident = ((a_ + beta * zc + gamma) * (b_ + beta * K1 * zc + gamma) % R_MOD
             * (c_ + beta * K2 * zc + gamma) % R_MOD * (d_ + beta * K3 * zc + gamma)) % R_MOD
    copy3 = (a_ + beta * s1 + gamma) * (b_ + beta * s2 + gamma) % R_MOD * (c_ + beta * s3 + gamma) % R_MOD
    lin_terms = [
        ("q_m", a_ * b_), ("q_l", a_), ("q_r", b_), ("q_o", c_), ("q_4", d_), ("q_c", 1),
    ]
    vec_ptrs = [pk.selector_coeffs(s) for s, _ in lin_terms]
    lin_coeffs = [c % R_MOD for _, c in lin_terms]
    vec_ptrs += [z_coeffs, pk.sigma_coeffs.ptr + 32 * 3 * n]
    lin_coeffs += [(alpha * ident + alpha * alpha % R_MOD * l1_z) % R_MOD,
                   (-alpha * copy3 % R_MOD * beta % R_MOD * z_next) % R_MOD]
    r_poly = pk.workspace("r", n)
    ctx.fr_lincomb(vec_ptrs, np.stack([fr(c) for c in lin_coeffs]), n, r_poly.ptr)
    ev["r"] = ctx.fr_evaluate(r_poly.ptr, n, zc_l)
    for name in ("a", "b", "c", "d", "sigma_1", "sigma_2", "sigma_3", "z_next", "t", "r"):
        ts.append_scalar(name.encode() + b"_eval", ev[name])
    proof.evaluations = ev

    # ---- round 5: aggregated opening witnesses -------------------------------------------------------
    v = ts.challenge_scalar(b"v")
    agg_ptrs = [t.ptr + 32 * i * n for i in range(4)] + [r_poly.ptr] + [coeffs.ptr + 32 * j * n for j in range(4)] \
        + [pk.sigma_coeffs.ptr + 32 * j * n for j in range(3)]
    agg_coeffs = [1, zn, zn * zn % R_MOD, pow(zn, 3, R_MOD)] + [pow(v, e, R_MOD) for e in range(1, 9)]
    agg = pk.workspace("agg", n)
    ctx.fr_lincomb(agg_ptrs, np.stack([fr(c) for c in agg_coeffs]), n, agg.ptr)
    wit = pk.workspace("wit", 2 * n)                                # [W_z | W_zw], n - 1 coefficients each
    ctx.fr_ruffini(agg.ptr, n, zc_l, wit.ptr)
    ctx.fr_ruffini(z_coeffs, n, zw_l, wit.ptr + 32 * n)
    for name, c in zip(("w_z", "w_zw"), _commit_batch(ck, wit.ptr, n - 1, 2, n)):
        proof.commitments[name] = c
        ts.append_commitment(name.encode(), c)
    u = ts.challenge_scalar(b"u")                                  # separates the two opening checks of the verifier
    proof.challenges = {"beta": beta, "gamma": gamma, "alpha": alpha, "z": zc, "v": v, "u": u}
    if own_witness:
        wire_evals.free()
    if own_pi:
        pi_ev.free()
    return proof


def derive_challenges(proof: Proof, n: int, transcript: Transcript | None = None) -> dict:
    """The verifier's side of Fiat-Shamir: replay the transcript over the proof's commitments and
    evaluations (same labels and order as prove()) and return the challenges."""
    ts = transcript or Transcript(b"plonk")
    ts.circuit_domain_sep(n)
    for name in ("a", "b", "c", "d"):
        ts.append_commitment(b"w_" + name.encode(), proof.commitments[name])
    ch = {"beta": ts.challenge_scalar(b"beta"), "gamma": ts.challenge_scalar(b"gamma")}
    ts.append_commitment(b"z", proof.commitments["z"])
    ch["alpha"] = ts.challenge_scalar(b"alpha")
    for i in range(4):
        ts.append_commitment(f"t_{i + 1}".encode(), proof.commitments[f"t_{i + 1}"])
    ch["z"] = ts.challenge_scalar(b"z")
    for name in Proof.EVALUATIONS:
        ts.append_scalar(name.encode() + b"_eval", proof.evaluations[name])
    ch["v"] = ts.challenge_scalar(b"v")
    for name in ("w_z", "w_zw"):
        ts.append_commitment(name.encode(), proof.commitments[name])
    ch["u"] = ts.challenge_scalar(b"u")
    return ch


def verifier_key(pk: ProverKey, ck) -> dict:
    """Commitments to the selector and sigma polynomials (``dusk_plonk::proof_system::VerifierKey``'s
    G1 part), computed from the prover key's device-resident coefficients."""
    out = dict(zip(SELECTORS, ck.commit_batch_dev(pk.sel_coeffs.ptr, pk.n, len(SELECTORS), pk.n)))
    for j, c in enumerate(ck.commit_batch_dev(pk.sigma_coeffs.ptr, pk.n, 4, pk.n)):
        out[f"sigma_{j + 1}"] = c
    return out


def check_identity(proof: Proof, n: int, pi_eval: int = 0) -> bool:
    """The verifier's scalar equation  t(z) Z_H(z) = r(z) + PI(z) - alpha (a + beta s1 + gamma)(b + beta s2 +
    gamma)(c + beta s3 + gamma)(d + gamma) z_w - alpha^2 L_1(z)  on the proof's evaluations.  (The pairing
    checks of the two opening witnesses need G2 arithmetic, which is outside this backend.)"""
    ch, ev = proof.challenges, {k: fr_from_limbs(v) for k, v in proof.evaluations.items()}
    beta, gamma, alpha, zc = ch["beta"], ch["gamma"], ch["alpha"], ch["z"]
    zn = pow(zc, n, R_MOD)
    l1_z = (zn - 1) * pow(n * (zc - 1) % R_MOD, -1, R_MOD) % R_MOD
    rhs = (ev["r"] + pi_eval
           - alpha * (ev["a"] + beta * ev["sigma_1"] + gamma) % R_MOD * (ev["b"] + beta * ev["sigma_2"] + gamma) % R_MOD
           * (ev["c"] + beta * ev["sigma_3"] + gamma) % R_MOD * (ev["d"] + gamma) % R_MOD * ev["z_next"]
           - alpha * alpha % R_MOD * l1_z) % R_MOD
    return ev["t"] * (zn - 1) % R_MOD == rhs


# ---------------------------------------------------------------------------------------------
class NativeProverKey:
    """``pm_prover_key``: the same key and per-proof workspace built and owned by the library
    (``pm_plonk_preprocess``), for ``prove_native``."""

    def __init__(self, circuit: Circuit, ctx: Context):
        self.ctx, self.n = ctx, circuit.n
        sels = [np.ascontiguousarray(getattr(circuit, s), dtype=np.uint64).reshape(-1, 4) for s in SELECTORS]
        ptrs = (_lib.u64p * 6)(*[a.ctypes.data_as(_lib.u64p) for a in sels])
        idx = np.ascontiguousarray(circuit.sigma_index, dtype=np.int64).reshape(-1)
        h = C.c_void_p()
        ctx._check(ctx._lib.pm_plonk_preprocess(ctx._h, ptrs, idx.ctypes.data_as(C.POINTER(C.c_int64)), self.n, C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None) and self.ctx._h:
            self.ctx._lib.pm_plonk_key_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def prove_native(pk: NativeProverKey, ck: CommitKey, witness: DeviceVector, public_inputs: DeviceVector | None = None,
                 label: bytes = b"plonk") -> Proof:
    """The five rounds sequenced inside the library (``pm_plonk_prove``): one C-ABI call per proof, what a
    Rust prover binds.  Same transcript, same proof as :func:`prove`."""
    ctx = pk.ctx
    raw = _lib.PlonkProof()
    d_pi = public_inputs._p if public_inputs is not None else None
    if hasattr(ck, "lo"):       # dist.ShardedCommitKey: this rank's slice of the SRS, partial sums exchanged
        from .dist import allgather_fold_many

        def exchange(_user, xyz, k):
            try:
                buf = np.ctypeslib.as_array(xyz, shape=(k, 18))
                buf[:] = allgather_fold_many(buf.copy(), ck.device)
                return 0
            except Exception:            # never unwind through the C frames
                return 1
        cb = _lib.EXCHANGE_FN(exchange)
        ctx._check(ctx._lib.pm_plonk_prove_sharded(ctx._h, pk._h, ck._bases._h, ck.lo, witness._p, d_pi, label,
                                                   C.cast(cb, C.c_void_p), None, C.byref(raw)))
    else:
        ctx._check(ctx._lib.pm_plonk_prove(ctx._h, pk._h, ck._bases._h, witness._p, d_pi, label, C.byref(raw)))
    proof = Proof()
    for i, name in enumerate(Proof.COMMITMENTS):
        proof.commitments[name] = np.array(raw.commitments[i], dtype=np.uint64)
    for i, name in enumerate(Proof.EVALUATIONS):
        proof.evaluations[name] = np.array(raw.evaluations[i], dtype=np.uint64)
    for i, name in enumerate(("beta", "gamma", "alpha", "z", "v", "u")):
        proof.challenges[name] = fr_from_limbs(np.array(raw.challenges[i], dtype=np.uint64))
    return proof
