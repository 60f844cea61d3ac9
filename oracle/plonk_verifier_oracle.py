"""TEST INFRASTRUCTURE ONLY -- the PLONK verifier for the proofs of plonk-prototype_amd/prover.py, in
plain Python integers with the slow pairing of oracle/pairing_oracle.py.  Restates
``dusk_plonk::proof_system::Proof::verify`` for the identities the prover builds (arithmetic gate +
4-wire permutation); PARITY UNPINNED like the rest of the prover rows (dusk-plonk 0.8.2 is not in
the reference tree).  The verifier is out of scope for the product: this file
exists so that a GPU-made proof can be checked end to end without trusting the prover-side oracle.

All points are affine (x, y) int tuples or None; scalars are canonical ints.
"""
from __future__ import annotations

from . import bigint_oracle as B
from . import pairing_oracle as PG

R = B.R_MOD
K = (1, 7, 13, 17)


def _lin(terms):
    """sum_i k_i P_i in G1."""
    acc = None
    for k, pt in terms:
        acc = B.g1_add(acc, B.g1_mul(k % R, pt))
    return acc


def verify(n, vk, proof_comms, ev, ch, pi_z, tau_g2):
    """vk: {q_m, q_l, q_r, q_o, q_4, q_c, sigma_1..4} commitments; proof_comms: the 11 commitments;
    ev: the 10 evaluations; ch: {beta, gamma, alpha, z, v, u}; pi_z: PI(z); tau_g2 = [tau] G2.
    Returns (identity_ok, pairing_ok)."""
    beta, gamma, alpha, z, v, u = (ch[k] for k in ("beta", "gamma", "alpha", "z", "v", "u"))
    omega = B.Domain(n).group_gen
    zn = pow(z, n, R)
    zh = (zn - 1) % R
    l1 = zh * pow(n * (z - 1) % R, -1, R) % R
    a, b, c, d = ev["a"], ev["b"], ev["c"], ev["d"]
    s1, s2, s3, zw_eval = ev["sigma_1"], ev["sigma_2"], ev["sigma_3"], ev["z_next"]
    # 1. the quotient identity at z
    copy3 = (a + beta * s1 + gamma) * (b + beta * s2 + gamma) % R * (c + beta * s3 + gamma) % R
    rhs = (ev["r"] + pi_z - alpha * copy3 % R * (d + gamma) % R * zw_eval - alpha * alpha % R * l1) % R
    identity_ok = ev["t"] * zh % R == rhs
    # 2. the commitment to the linearisation polynomial, from the verifier key
    ident = 1
    for kj, w in zip(K, (a, b, c, d)):
        ident = ident * (w + beta * kj * z + gamma) % R
    r_comm = _lin([(a * b, vk["q_m"]), (a, vk["q_l"]), (b, vk["q_r"]), (c, vk["q_o"]), (d, vk["q_4"]), (1, vk["q_c"]),
                   (alpha * ident + alpha * alpha * l1, proof_comms["z"]),
                   (-alpha * copy3 * beta * zw_eval, vk["sigma_4"])])
    # 3. the batched opening at z and the opening of z(X) at z w, folded with u into one pairing equation
    f_comm = _lin([(1, proof_comms["t_1"]), (zn, proof_comms["t_2"]), (zn * zn, proof_comms["t_3"]),
                   (pow(zn, 3, R), proof_comms["t_4"]), (v, r_comm), (v ** 2, proof_comms["a"]),
                   (v ** 3, proof_comms["b"]), (v ** 4, proof_comms["c"]), (v ** 5, proof_comms["d"]),
                   (v ** 6, vk["sigma_1"]), (v ** 7, vk["sigma_2"]), (v ** 8, vk["sigma_3"])])
    e_val = (ev["t"] + v * ev["r"] + v ** 2 * a + v ** 3 * b + v ** 4 * c + v ** 5 * d + v ** 6 * s1 + v ** 7 * s2
             + v ** 8 * s3) % R
    zw = z * omega % R
    lhs = _lin([(1, proof_comms["w_z"]), (u, proof_comms["w_zw"])])
    rhs_pt = _lin([(z, proof_comms["w_z"]), (u * zw, proof_comms["w_zw"]), (1, f_comm), (-e_val, B.G1_GEN),
                   (u, proof_comms["z"]), (-u * zw_eval, B.G1_GEN)])
    # e(tau G2, lhs) = e(G2, rhs)
    pairing_ok = PG.pairing_product_is_one([(tau_g2, lhs), (PG.G2_GEN, B.g1_neg(rhs_pt))])
    return identity_ok, pairing_ok
