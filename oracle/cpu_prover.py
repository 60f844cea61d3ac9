"""TEST INFRASTRUCTURE ONLY -- the five PLONK prover rounds on the CPU, composed from the C
restatement (oracle/c/plonk_oracle.c: radix-2 NTT, Pippenger MSM, Horner, Ruffini, batch inversion,
the pointwise permutation / quotient steps with the four widgets).  Two uses: the ``cpu_baseline`` of
bench.py's full-prove leg (what the reference's dusk-plonk prover does on host cores, restated -- kind
"port"), and a second oracle for the GPU prover at sizes the big-int restatement
(plonk_rounds_oracle.py) is too slow for.  PARITY UNPINNED: dusk-plonk 0.8.2 (ref:Cargo.toml:19) is not
in the reference tree; formulas restated from the published 0.8 design.

All vectors are [n, 4] uint64 Montgomery limbs; challenges are canonical ints.
"""
from __future__ import annotations

import numpy as np

from . import bigint_oracle as B
from .cpu_oracle import COSET, INVERSE, CpuOracle, ints_to_limbs, limbs_to_ints
from .plonk_rounds_oracle import SELECTORS, WIDGET_SELECTORS

R = B.R_MOD
K = (1, 7, 13, 17)


def _m(o: CpuOracle, v: int) -> np.ndarray:
    return o.fr_to_mont(ints_to_limbs([v % R], 4))[0]


def _i(o: CpuOracle, limbs) -> int:
    return limbs_to_ints(o.fr_from_mont(np.ascontiguousarray(limbs).reshape(1, 4)))[0]


def preprocess(o: CpuOracle, sel: dict, sigma_index, threads: int = 1) -> dict:
    """sel: selector evaluations on H (missing / None = zero); sigma_index [4, n].
    -> coefficient forms, coset forms, tables."""
    n = sel["q_m"].shape[0]
    log_n = n.bit_length() - 1
    dom, dom4 = B.Domain(n), B.Domain(4 * n)
    zero = np.zeros((n, 4), np.uint64)
    sel = {k: (np.ascontiguousarray(sel[k]) if sel.get(k) is not None else zero) for k in SELECTORS}
    roots = o.fr_powers(_m(o, dom.group_gen), _m(o, 1), n)
    table = np.concatenate([o.fr_powers(_m(o, dom.group_gen), _m(o, k), n) for k in K])
    sig_ev = [np.ascontiguousarray(table[np.asarray(sigma_index[j], dtype=np.int64)]) for j in range(4)]
    pk = {"n": n, "log_n": log_n, "omega": dom.group_gen, "roots": roots, "sig_ev": sig_ev}
    pk["sel_c"] = {k: o.fr_ntt(v, log_n, INVERSE, threads) for k, v in sel.items()}
    pk["sig_c"] = [o.fr_ntt(s, log_n, INVERSE, threads) for s in sig_ev]
    pk["sel_4"] = {k: o.fr_ntt(v, log_n + 2, COSET, threads) for k, v in pk["sel_c"].items()}
    pk["sig_4"] = [o.fr_ntt(s, log_n + 2, COSET, threads) for s in pk["sig_c"]]
    pk["l1_4"] = o.fr_ntt(np.tile(_m(o, dom.size_inv), (n, 1)), log_n + 2, COSET, threads)
    pk["x4"] = o.fr_powers(_m(o, dom4.group_gen), _m(o, 7), 4 * n)
    return pk


def verifier_key(o: CpuOracle, pk: dict, srs, threads: int = 1) -> dict:
    """Commitments to the 11 selector and 4 sigma polynomials."""
    n = pk["n"]
    vk = {k: o.g1_msm(srs[:n], pk["sel_c"][k], 0, threads) for k in SELECTORS}
    for j in range(4):
        vk[f"sigma_{j + 1}"] = o.g1_msm(srs[:n], pk["sig_c"][j], 0, threads)
    return vk


def prove(o: CpuOracle, pk: dict, srs, witness, pi, ch: dict, threads: int = 1) -> dict:
    """-> {"commitments": {name: affine [12]}, "evaluations": {name: limbs [4]}} for given challenges
    (keys of plonk_rounds_oracle.CHALLENGES)."""
    n, log_n = pk["n"], pk["log_n"]
    beta, gamma, alpha, zc = (ch[k] for k in ("beta", "gamma", "alpha", "z"))
    seps = np.stack([_m(o, ch[k]) for k in ("range_sep", "logic_sep", "fixed_sep", "var_sep")])
    commit = lambda c: o.g1_msm(srs[:c.shape[0]], c, 0, threads)   # noqa: E731
    out = {"commitments": {}, "evaluations": {}}
    # round 1
    wc = [o.fr_ntt(witness[j], log_n, INVERSE, threads) for j in range(4)]
    for j, nm in enumerate("abcd"):
        out["commitments"][nm] = commit(wc[j])
    # round 2
    num, den = o.plonk_perm_terms(list(witness), pk["sig_ev"], pk["roots"], _m(o, beta), _m(o, gamma), threads)
    ratio = o.fr_vec_op(2, num, o.fr_batch_inverse_trick(den, threads))
    z_c = o.fr_ntt(o.fr_prefix_product(ratio), log_n, INVERSE, threads)
    out["commitments"]["z"] = commit(z_c)
    # round 3
    pi_c = o.fr_ntt(pi, log_n, INVERSE, threads)
    cos = lambda c: o.fr_ntt(c, log_n + 2, COSET, threads)          # noqa: E731
    s4 = pk["sel_4"]
    arrays = [cos(c) for c in wc] + [cos(z_c)] + [s4[k] for k in ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")] \
        + [cos(pi_c)] + pk["sig_4"] + [pk["l1_4"], pk["x4"]] \
        + [s4[k] for k in ("q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add")]
    t_ev = o.plonk_quotient(arrays, n, _m(o, alpha), _m(o, beta), _m(o, gamma), seps, threads)
    t = o.fr_ntt(t_ev, log_n + 2, INVERSE | COSET, threads)
    for i in range(4):
        out["commitments"][f"t_{i + 1}"] = commit(t[i * n:(i + 1) * n])
    # round 4
    zm, zwm = _m(o, zc), _m(o, zc * pk["omega"] % R)
    ev = {nm: o.fr_poly_evaluate(wc[j], zm) for j, nm in enumerate("abcd")}
    for j, nm in ((0, "a_next"), (1, "b_next"), (3, "d_next")):
        ev[nm] = o.fr_poly_evaluate(wc[j], zwm)
    for j in range(3):
        ev[f"sigma_{j + 1}"] = o.fr_poly_evaluate(pk["sig_c"][j], zm)
    for nm in ("q_arith", "q_c", "q_l", "q_r"):
        ev[nm] = o.fr_poly_evaluate(pk["sel_c"][nm], zm)
    ev["z_next"] = o.fr_poly_evaluate(z_c, zwm)
    ev["t"] = o.fr_poly_evaluate(t, zm)
    a_, b_, c_, d_ = (_i(o, ev[k]) for k in "abcd")
    s1, s2, s3, zw_e, qar = (_i(o, ev[k]) for k in ("sigma_1", "sigma_2", "sigma_3", "z_next", "q_arith"))
    zn = pow(zc, n, R)
    l1_z = (zn - 1) * pow(n * (zc - 1) % R, -1, R) % R
    ident = 1
    for kj, wv in zip(K, (a_, b_, c_, d_)):
        ident = ident * (wv + beta * kj * zc + gamma) % R
    copy3 = (a_ + beta * s1 + gamma) * (b_ + beta * s2 + gamma) % R * (c_ + beta * s3 + gamma) % R
    row = np.stack([ev[k] for k in ("a", "b", "c", "d", "a_next", "b_next", "d_next", "q_l", "q_r", "q_c")])
    wv4 = o.plonk_widget_values(seps, row)
    lin_c = [_m(o, c) for c in (qar * a_ * b_, qar * a_, qar * b_, qar * c_, qar * d_, qar)] + [wv4[i] for i in range(4)] \
        + [_m(o, alpha * ident + alpha * alpha * l1_z), _m(o, -alpha * copy3 * beta * zw_e)]
    lin_v = [pk["sel_c"][k] for k in ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")] \
        + [pk["sel_c"][k] for k in WIDGET_SELECTORS] + [z_c, pk["sig_c"][3]]
    r = o.fr_lincomb(np.stack(lin_c), lin_v, threads)
    ev["r"] = o.fr_poly_evaluate(r, zm)
    out["evaluations"] = ev
    # round 5
    aw, aws = ch["aw"], ch["aw_shifted"]
    agg_c = [1, zn, zn * zn, zn ** 3] + [pow(aw, e, R) for e in range(1, 9)]
    agg_v = [t[i * n:(i + 1) * n] for i in range(4)] + [r] + wc + pk["sig_c"][:3]
    agg = o.fr_lincomb(np.stack([_m(o, c) for c in agg_c]), agg_v, threads)
    out["commitments"]["w_z"] = commit(o.fr_poly_ruffini(agg, zm))
    agg_s = o.fr_lincomb(np.stack([_m(o, pow(aws, e, R)) for e in range(4)]), [z_c, wc[0], wc[1], wc[3]], threads)
    out["commitments"]["w_zw"] = commit(o.fr_poly_ruffini(agg_s, zwm))
    return out
