"""TEST INFRASTRUCTURE ONLY -- the PLONK verifier for the proofs of the native prover
(plonk-prototype_amd/csrc/prover.hip), in plain Python integers with the slow pairing of
oracle/pairing_oracle.py.  Restates ``dusk_plonk::proof_system::Proof::verify`` of dusk-plonk 0.8
(quotient evaluation, linearisation commitment with all five widgets, the two aggregate opening
proofs, the batched pairing check); PARITY UNPINNED like the rest of the prover rows (the crate is
not in the reference tree).  The verifier is out of scope for the product:
this file exists so that a GPU-made proof can be checked end to end without trusting the
prover-side oracle.

All points are affine (x, y) int tuples or None; scalars are canonical ints.
"""
from __future__ import annotations

from . import bigint_oracle as B
from . import pairing_oracle as PG
from . import plonk_rounds_oracle as PO

R = B.R_MOD
K = (1, 7, 13, 17)


def _lin(terms):
    """sum_i k_i P_i in G1."""
    acc = None
    for k, pt in terms:
        acc = B.g1_add(acc, B.g1_mul(k % R, pt))
    return acc


def quotient_evaluation(n, ev, ch, pi_z) -> int:
    """``Proof::compute_quotient_evaluation``: t(z) from the other evaluations."""
    beta, gamma, alpha, z = (ch[k] for k in ("beta", "gamma", "alpha", "z"))
    zn = pow(z, n, R)
    zh = (zn - 1) % R
    l1 = zh * pow(n * (z - 1) % R, -1, R) % R
    copy3 = (ev["a"] + beta * ev["sigma_1"] + gamma) * (ev["b"] + beta * ev["sigma_2"] + gamma) % R \
        * (ev["c"] + beta * ev["sigma_3"] + gamma) % R
    num = (ev["r"] + pi_z - alpha * copy3 % R * (ev["d"] + gamma) % R * ev["z_next"] - alpha * alpha % R * l1) % R
    return num * pow(zh, -1, R) % R


def verify(n, vk, proof_comms, ev, ch, pi_z, tau_g2):
    """vk: the 11 selector commitments + sigma_1..4; proof_comms: the 11 commitments; ev: the 16 evaluations
    (a "t" entry, if present, is compared with the verifier's own); ch: beta gamma alpha range_sep logic_sep
    fixed_sep var_sep z aw aw_shifted batch; pi_z: PI(z); tau_g2 = [tau] G2.  Returns (identity_ok, pairing_ok)."""
    z, aw, aws, u = (ch[k] for k in ("z", "aw", "aw_shifted", "batch"))
    omega = B.Domain(n).group_gen
    zn = pow(z, n, R)
    # 1. the quotient evaluation the verifier derives; a prover-supplied t must agree
    t_eval = quotient_evaluation(n, ev, ch, pi_z)
    identity_ok = ev.get("t", t_eval) == t_eval
    # 2. the commitment to the linearisation polynomial, from the verifier key (compute_linearisation_commitment)
    lc = PO.linearisation_coeffs(ev, ch, n)
    r_comm = _lin([(lc[k], vk[k]) for k in PO.SELECTORS[:6]] + [(lc[k], vk[k]) for k in PO.WIDGET_SELECTORS]
                  + [(lc["z"], proof_comms["z"]), (lc["sigma_4"], vk["sigma_4"])])
    # 3. AggregateProof::flatten twice, then OpeningKey::batch_check
    t_comm = _lin([(1, proof_comms["t_1"]), (zn, proof_comms["t_2"]), (zn * zn, proof_comms["t_3"]),
                   (pow(zn, 3, R), proof_comms["t_4"])])
    parts_a = [(t_eval, t_comm), (ev["r"], r_comm), (ev["a"], proof_comms["a"]), (ev["b"], proof_comms["b"]),
               (ev["c"], proof_comms["c"]), (ev["d"], proof_comms["d"]), (ev["sigma_1"], vk["sigma_1"]),
               (ev["sigma_2"], vk["sigma_2"]), (ev["sigma_3"], vk["sigma_3"])]
    parts_b = [(ev["z_next"], proof_comms["z"]), (ev["a_next"], proof_comms["a"]), (ev["b_next"], proof_comms["b"]),
               (ev["d_next"], proof_comms["d"])]
    f_a = _lin([(pow(aw, i, R), c) for i, (_, c) in enumerate(parts_a)])
    e_a = sum(pow(aw, i, R) * v for i, (v, _) in enumerate(parts_a)) % R
    f_b = _lin([(pow(aws, i, R), c) for i, (_, c) in enumerate(parts_b)])
    e_b = sum(pow(aws, i, R) * v for i, (v, _) in enumerate(parts_b)) % R
    zw = z * omega % R
    lhs = _lin([(1, proof_comms["w_z"]), (u, proof_comms["w_zw"])])
    rhs_pt = _lin([(z, proof_comms["w_z"]), (u * zw, proof_comms["w_zw"]), (1, f_a), (-e_a, B.G1_GEN),
                   (u, f_b), (-u * e_b, B.G1_GEN)])
    # e(tau G2, lhs) = e(G2, rhs)
    pairing_ok = PG.pairing_product_is_one([(tau_g2, lhs), (PG.G2_GEN, B.g1_neg(rhs_pt))])
    return identity_ok, pairing_ok
