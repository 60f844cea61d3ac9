"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the PLONK prover rounds (SURVEY.md section 8f rows
N1 / N2) in plain Python integers, written from the protocol equations rather than from the device
code so that the two can disagree.  Only tests/, smoke() and bench.py's checker may import it.

PARITY UNPINNED: the rounds live in dusk-plonk 0.8.2 (ref:Cargo.toml:19), which is not in the
reference tree and cannot be built here; there are no upstream vectors.  The formulas below are
restated from the published dusk-plonk 0.8 design (the gates the reference's gadgets emit:
ref:src/zk/gadgets.rs:34,37 fixed_base_scalar_mul, :40 point_addition_gate, :211 boolean_gate /
range via :88-91, ref:src/zk/circuits.rs:64-70):

  permutation   dusk_plonk::permutation::Permutation::compute_permutation_poly
                z(w^0) = 1,  z(w^{i+1}) = z(w^i) prod_j (w_j + beta k_j w^i + gamma) / (w_j + beta sigma_j + gamma)
  quotient      dusk_plonk::proof_system::quotient_poly::compute with the widgets
                arithmetic (x q_arith), range, logic, fixed-base scalar mul, variable-base curve addition, permutation
  linearisation dusk_plonk::proof_system::linearisation_poly::compute (16 evaluations)
  opening       dusk_plonk::commitment_scheme::kzg10::CommitKey::compute_aggregate_witness

Structural pins (tests/test_oracle_plonk.py): the logic identity vanishes on exactly the valid AND
(q_c = 1) / XOR (q_c = -1) quad tuples, the curve identities on exactly the JubJub sums, EDWARDS_D
is -(10240/10241); the quotient of a satisfied circuit is a polynomial (exact division by Z_H).

Sizes: pure-Python loops, meant for n <= 2^10.
"""
from __future__ import annotations

from . import bigint_oracle as B

R = B.R_MOD
K = (1, 7, 13, 17)
GEN = 7
EDWARDS_D = (-(10240 * pow(10241, -1, R))) % R      # JubJub: -x^2 + y^2 = 1 + d x^2 y^2
SELECTORS = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
             "q_variable_group_add")
WIDGET_SELECTORS = SELECTORS[7:]
# evaluations in the order the transcript absorbs them; PROOF_EVALS is the order of ProofEvaluations::to_bytes
TRANSCRIPT_EVALS = ("a", "b", "c", "d", "a_next", "b_next", "d_next", "sigma_1", "sigma_2", "sigma_3", "q_arith",
                    "q_c", "q_l", "q_r", "z_next", "t", "r")
PROOF_EVALS = ("a", "b", "c", "d", "a_next", "b_next", "d_next", "q_arith", "q_c", "q_l", "q_r", "sigma_1", "sigma_2",
               "sigma_3", "r", "z_next")
CHALLENGES = ("beta", "gamma", "alpha", "range_sep", "logic_sep", "fixed_sep", "var_sep", "z", "aw", "aw_shifted")


def inv(x: int) -> int:
    return pow(x % R, -1, R)


def powers(base: int, scale: int, n: int) -> list[int]:
    out, cur = [], scale % R
    for _ in range(n):
        out.append(cur)
        cur = cur * base % R
    return out


def lincomb(coeffs, vecs) -> list[int]:
    n = len(vecs[0])
    return [sum(c * v[i] for c, v in zip(coeffs, vecs)) % R for i in range(n)]


# ---------------------------------------------------------------------------------- widgets
def delta(f: int) -> int:
    """f (f - 1)(f - 2)(f - 3): zero exactly on the quads 0..3."""
    return f * (f - 1) % R * (f - 2) % R * (f - 3) % R


def delta_xor_and(a: int, b: int, w: int, c: int, q_c: int) -> int:
    """widget::logic::delta_xor_and: zero iff w = a b and c = a & b (q_c = 1) resp. a ^ b (q_c = -1), on quads."""
    f = w * (w * (4 * w - 18 * (a + b) + 81) + 18 * (a * a + b * b) - 81 * (a + b) + 83)
    e = 3 * (a + b + c) - 2 * f
    return (q_c * (9 * c - 3 * (a + b)) + e) % R


def widget_range(sep, a, b, c, d, d_next):
    k = sep * sep % R
    return (delta(c - 4 * d) + delta(b - 4 * c) * k + delta(a - 4 * b) * k % R * k
            + delta(d_next - 4 * a) * k % R * k % R * k) % R * sep % R


def widget_logic(sep, a, a_next, b, b_next, c, d, d_next, q_c):
    k = sep * sep % R
    qa, qb, qd = (a_next - 4 * a) % R, (b_next - 4 * b) % R, (d_next - 4 * d) % R
    k2, k3, k4 = k * k % R, pow(k, 3, R), pow(k, 4, R)
    return (delta(qa) + delta(qb) * k + delta(qd) * k2 + (c - qa * qb) * k3
            + delta_xor_and(qa, qb, c, qd, q_c) * k4) % R * sep % R


def widget_fixed_base(sep, a, a_next, b, b_next, c, d, d_next, q_l, q_r, q_c):
    """One round of the fixed-base scalar multiplication: (acc_x, acc_y) = (a, b), xy_alpha = c,
    accumulated bits d; table point (x_beta, y_beta) = (q_l, q_r), x_beta y_beta = q_c."""
    k = sep * sep % R
    bit = (d_next - 2 * d) % R
    bit_consistency = bit * (bit - 1) % R * (bit + 1) % R
    y_alpha = (bit * bit % R * (q_r - 1) + 1) % R
    x_alpha = q_l * bit % R
    xy_consistency = (bit * q_c - c) % R * k % R
    dxy = c * a % R * b % R * EDWARDS_D % R
    x_acc = (a_next + a_next * dxy - (a * y_alpha + b * x_alpha)) % R * k % R * k % R
    y_acc = (b_next - b_next * dxy - (b * y_alpha + a * x_alpha)) % R * pow(k, 3, R) % R
    return (bit_consistency + x_acc + y_acc + xy_consistency) % R * sep % R


def widget_variable_base(sep, a, a_next, b, b_next, c, d, d_next):
    """JubJub addition (x_1, y_1) + (x_2, y_2) = (x_3, y_3): (a, b) + (c, d) = (a_next, b_next), x_1 y_2 = d_next."""
    k = sep * sep % R
    x1y2, y1x2, y1y2, x1x2 = d_next, b * c % R, b * d % R, a * c % R
    xy_consistency = (a * d - x1y2) % R
    dd = EDWARDS_D * x1y2 % R * y1x2 % R
    x3 = (x1y2 + y1x2 - (a_next + a_next * dd)) % R * k % R
    y3 = (y1y2 + x1x2 - (b_next - b_next * dd)) % R * k % R * k % R
    return (xy_consistency + x3 + y3) % R * sep % R


def widget_values(ch, a, b, c, d, a_next, b_next, d_next, q_l, q_r, q_c) -> dict:
    """The factor each widget selector is multiplied with (pointwise on the coset, or at the evaluations)."""
    return {"q_range": widget_range(ch["range_sep"], a, b, c, d, d_next),
            "q_logic": widget_logic(ch["logic_sep"], a, a_next, b, b_next, c, d, d_next, q_c),
            "q_fixed_group_add": widget_fixed_base(ch["fixed_sep"], a, a_next, b, b_next, c, d, d_next, q_l, q_r, q_c),
            "q_variable_group_add": widget_variable_base(ch["var_sep"], a, a_next, b, b_next, c, d, d_next)}


def gate_value(selr: dict, a, b, c, d, a_next, b_next, d_next, ch) -> int:
    """Sum of all gate identities of one row (without PI): what must vanish on H."""
    arith = (selr["q_m"] * a % R * b + selr["q_l"] * a + selr["q_r"] * b + selr["q_o"] * c + selr["q_4"] * d
             + selr["q_c"]) % R
    g = selr["q_arith"] * arith
    wv = widget_values(ch, a, b, c, d, a_next, b_next, d_next, selr["q_l"], selr["q_r"], selr["q_c"])
    for k_ in WIDGET_SELECTORS:
        g += selr[k_] * wv[k_]
    return g % R


# ---------------------------------------------------------------------------------- rounds
def perm_terms(wires, sigmas, roots, beta, gamma):
    n = len(roots)
    num, den = [], []
    for i in range(n):
        a, b = 1, 1
        for j in range(4):
            a = a * (wires[j][i] + beta * K[j] * roots[i] + gamma) % R
            b = b * (wires[j][i] + beta * sigmas[j][i] + gamma) % R
        num.append(a)
        den.append(b)
    return num, den


def grand_product(num, den) -> list[int]:
    z, out = 1, []
    for a, b in zip(num, den):
        out.append(z)
        z = z * a % R * inv(b) % R
    return out


def quotient_evals(n, w, z, sel, pi, sig, l1, x, ch):
    """All arguments are evaluations on the 4n coset x_i = 7 w_4n^i; sel = dict of the 11 selector evals.
    "next" = the value one step of H further = index + 4 on the 4n coset."""
    n4 = 4 * n
    alpha, beta, gamma = ch["alpha"], ch["beta"], ch["gamma"]
    out = []
    for i in range(n4):
        nx = (i + 4) % n4
        a, b, c, d = (w[j][i] for j in range(4))
        gate = (gate_value({k_: sel[k_][i] for k_ in SELECTORS}, a, b, c, d, w[0][nx], w[1][nx], w[3][nx], ch)
                + pi[i]) % R
        ident, copy = z[i], z[nx]
        for j in range(4):
            ident = ident * (w[j][i] + beta * K[j] * x[i] + gamma) % R
            copy = copy * (w[j][i] + beta * sig[j][i] + gamma) % R
        first = (z[i] - 1) * l1[i] % R
        zh = (pow(x[i], n, R) - 1) % R
        out.append((gate + alpha * (ident - copy) + alpha * alpha * first) % R * inv(zh) % R)
    return out


def ruffini(coeffs, z) -> list[int]:
    """Quotient of coeffs(X) / (X - z), remainder dropped."""
    q, acc = [0] * (len(coeffs) - 1), 0
    for i in range(len(coeffs) - 1, 0, -1):
        acc = (coeffs[i] + acc * z) % R
        q[i - 1] = acc
    return q


def linearisation_coeffs(ev: dict, ch: dict, n: int) -> dict:
    """Scalar in front of every polynomial of the linearisation polynomial r(X) (prover) resp. of every
    commitment of [r] (verifier): selector name / "z" / "sigma_4" -> int."""
    alpha, beta, gamma, zz = ch["alpha"], ch["beta"], ch["gamma"], ch["z"]
    a, b, c, d = ev["a"], ev["b"], ev["c"], ev["d"]
    qa = ev["q_arith"]
    out = {"q_m": qa * a % R * b % R, "q_l": qa * a % R, "q_r": qa * b % R, "q_o": qa * c % R, "q_4": qa * d % R,
           "q_c": qa}
    out.update(widget_values(ch, a, b, c, d, ev["a_next"], ev["b_next"], ev["d_next"], ev["q_l"], ev["q_r"], ev["q_c"]))
    zn = pow(zz, n, R)
    l1_z = (zn - 1) * inv(n * (zz - 1)) % R
    ident = 1
    for j, nm in enumerate("abcd"):
        ident = ident * (ev[nm] + beta * K[j] * zz + gamma) % R
    copy3 = 1
    for j, nm in enumerate("abc"):
        copy3 = copy3 * (ev[nm] + beta * ev[f"sigma_{j + 1}"] + gamma) % R
    out["z"] = (alpha * ident + alpha * alpha * l1_z) % R
    out["sigma_4"] = (-alpha * copy3 * beta * ev["z_next"]) % R
    return out


def prove(n, sel, sigma_index, witness, pi, ch):
    """Every intermediate of the five rounds for given challenges (keys: CHALLENGES).
    sel: the 11 selector evaluations on H (ints, missing = zero); sigma_index[j][i] = j' n + i';
    witness[j][i]; pi[i]."""
    log_n = n.bit_length() - 1
    dom, dom4 = B.Domain(n), B.Domain(4 * n)
    roots = powers(dom.group_gen, 1, n)
    x4 = powers(dom4.group_gen, GEN, 4 * n)
    sel = {k_: list(sel.get(k_, [0] * n)) for k_ in SELECTORS}
    table = [K[j] * roots[i] % R for j in range(4) for i in range(n)]
    sigmas = [[table[sigma_index[j][i]] for i in range(n)] for j in range(4)]
    out = {}
    # round 1
    wc = [B.ifft(witness[j], log_n) for j in range(4)]
    out["wire_coeffs"] = wc
    # round 2
    num, den = perm_terms(witness, sigmas, roots, ch["beta"], ch["gamma"])
    z_ev = grand_product(num, den)
    zc = B.ifft(z_ev, log_n)
    out["z_evals"], out["z_coeffs"] = z_ev, zc
    # round 3
    cos = lambda c: B.coset_fft(c, log_n + 2)   # noqa: E731
    sel_c = {k_: B.ifft(v, log_n) for k_, v in sel.items()}
    sig_c = [B.ifft(s, log_n) for s in sigmas]
    out["sel_coeffs"], out["sigma_coeffs"] = sel_c, sig_c
    pic = B.ifft(pi, log_n)
    l1c = [dom.size_inv] * n
    t_ev = quotient_evals(n, [cos(c) for c in wc], cos(zc), {k_: cos(v) for k_, v in sel_c.items()}, cos(pic),
                          [cos(s) for s in sig_c], cos(l1c), x4, ch)
    t = B.coset_ifft(t_ev, log_n + 2)
    out["t_coeffs"] = t
    # round 4
    zz = ch["z"]
    zw = zz * dom.group_gen % R
    ev = {nm: B.horner(wc[j], zz) for j, nm in enumerate("abcd")}
    for j, nm in ((0, "a_next"), (1, "b_next"), (3, "d_next")):
        ev[nm] = B.horner(wc[j], zw)
    for j in range(3):
        ev[f"sigma_{j + 1}"] = B.horner(sig_c[j], zz)
    for nm in ("q_arith", "q_c", "q_l", "q_r"):
        ev[nm] = B.horner(sel_c[nm], zz)
    ev["z_next"] = B.horner(zc, zw)
    ev["t"] = B.horner(t, zz)
    lc = linearisation_coeffs(ev, ch, n)
    names = list(SELECTORS[:6]) + list(WIDGET_SELECTORS)
    r = lincomb([lc[k_] for k_ in names] + [lc["z"], lc["sigma_4"]], [sel_c[k_] for k_ in names] + [zc, sig_c[3]])
    ev["r"] = B.horner(r, zz)
    out["r_coeffs"], out["evals"] = r, ev
    # round 5: compute_aggregate_witness twice, each with its own "aggregate_witness" challenge
    zn = pow(zz, n, R)
    parts = [t[i * n:(i + 1) * n] for i in range(4)]
    aw, aws = ch["aw"], ch["aw_shifted"]
    agg = lincomb([1, zn, zn * zn, zn ** 3] + [pow(aw, e, R) for e in range(1, 9)], parts + [r] + wc + sig_c[:3])
    out["w_z"] = ruffini(agg, zz)
    agg_s = lincomb([pow(aws, e, R) for e in range(4)], [zc, wc[0], wc[1], wc[3]])
    out["w_zw"] = ruffini(agg_s, zw)
    out["agg"], out["agg_shifted"] = agg, agg_s
    return out


def check_identity(ev: dict, ch: dict, n: int, pi_z: int) -> bool:
    """The verifier's scalar equation: t(z) Z_H(z) = r(z) + PI(z) - alpha (a + beta s1 + gamma)(b + beta s2 + gamma)
    (c + beta s3 + gamma)(d + gamma) z_w - alpha^2 L_1(z)  (Proof::compute_quotient_evaluation)."""
    alpha, beta, gamma, zz = ch["alpha"], ch["beta"], ch["gamma"], ch["z"]
    zn = pow(zz, n, R)
    l1_z = (zn - 1) * inv(n * (zz - 1)) % R
    rhs = (ev["r"] + pi_z
           - alpha * (ev["a"] + beta * ev["sigma_1"] + gamma) % R * (ev["b"] + beta * ev["sigma_2"] + gamma) % R
           * (ev["c"] + beta * ev["sigma_3"] + gamma) % R * (ev["d"] + gamma) % R * ev["z_next"]
           - alpha * alpha % R * l1_z) % R
    return ev["t"] * (zn - 1) % R == rhs
