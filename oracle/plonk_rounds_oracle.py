"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the PLONK prover rounds (SURVEY.md section 8f rows
N1 / N2) in plain Python integers, written from the protocol equations rather than from the device
code so that the two can disagree.  Only tests/, smoke() and bench.py's checker may import it.

PARITY UNPINNED: the rounds live in dusk-plonk 0.8.2 (ref:Cargo.toml:19), which is not in the
reference tree and cannot be built here; there are no upstream vectors.  What this file follows:

  permutation   dusk_plonk::permutation::Permutation::compute_permutation_poly
                z(w^0) = 1,  z(w^{i+1}) = z(w^i) prod_j (w_j + beta k_j w^i + gamma) / (w_j + beta sigma_j + gamma)
  quotient      dusk_plonk::proof_system::quotient_poly::compute  (arithmetic + permutation widgets)
  linearisation dusk_plonk::proof_system::linearisation_poly::compute
  opening       dusk_plonk::commitment_scheme::kzg10::CommitKey::compute_aggregate_witness

Sizes: pure-Python loops, meant for n <= 2^10.
"""
from __future__ import annotations

from . import bigint_oracle as B

R = B.R_MOD
K = (1, 7, 13, 17)
GEN = 7


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


def quotient_evals(n, w, z, sel, pi, sig, l1, x, alpha, beta, gamma):
    """All arguments are evaluations on the 4n coset x_i = 7 w_4n^i; sel = dict of selector evals."""
    n4 = 4 * n
    out = []
    for i in range(n4):
        a, b, c, d = (w[j][i] for j in range(4))
        gate = (sel["q_m"][i] * a * b + sel["q_l"][i] * a + sel["q_r"][i] * b + sel["q_o"][i] * c
                + sel["q_4"][i] * d + sel["q_c"][i] + pi[i]) % R
        ident, copy = z[i], z[(i + 4) % n4]
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


def prove(n, sel, sigma_index, witness, pi, ch):
    """Every intermediate of the five rounds for given challenges ch = {beta, gamma, alpha, z, v}.
    sel: selector evaluations on H (ints); sigma_index[j][i] = j' n + i'; witness[j][i]; pi[i]."""
    log_n = n.bit_length() - 1
    dom, dom4 = B.Domain(n), B.Domain(4 * n)
    roots = powers(dom.group_gen, 1, n)
    x4 = powers(dom4.group_gen, GEN, 4 * n)
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
    sel_c = {k: B.ifft(v, log_n) for k, v in sel.items()}
    sig_c = [B.ifft(s, log_n) for s in sigmas]
    pic = B.ifft(pi, log_n)
    l1c = [dom.size_inv] * n
    t_ev = quotient_evals(n, [cos(c) for c in wc], cos(zc), {k: cos(v) for k, v in sel_c.items()}, cos(pic),
                          [cos(s) for s in sig_c], cos(l1c), x4, ch["alpha"], ch["beta"], ch["gamma"])
    t = B.coset_ifft(t_ev, log_n + 2)
    out["t_coeffs"] = t
    # round 4
    zz, alpha, beta, gamma = ch["z"], ch["alpha"], ch["beta"], ch["gamma"]
    ev = {nm: B.horner(wc[j], zz) for j, nm in enumerate("abcd")}
    for j in range(3):
        ev[f"sigma_{j + 1}"] = B.horner(sig_c[j], zz)
    ev["z_next"] = B.horner(zc, zz * dom.group_gen % R)
    ev["t"] = B.horner(t, zz)
    zn = pow(zz, n, R)
    l1_z = (zn - 1) * inv(n * (zz - 1)) % R
    ident = 1
    for j, nm in enumerate("abcd"):
        ident = ident * (ev[nm] + beta * K[j] * zz + gamma) % R
    copy3 = 1
    for j, nm in enumerate("abc"):
        copy3 = copy3 * (ev[nm] + beta * ev[f"sigma_{j + 1}"] + gamma) % R
    r = lincomb([ev["a"] * ev["b"], ev["a"], ev["b"], ev["c"], ev["d"], 1,
                 alpha * ident + alpha * alpha * l1_z, -alpha * copy3 * beta * ev["z_next"]],
                [sel_c["q_m"], sel_c["q_l"], sel_c["q_r"], sel_c["q_o"], sel_c["q_4"], sel_c["q_c"], zc, sig_c[3]])
    ev["r"] = B.horner(r, zz)
    out["r_coeffs"], out["evals"] = r, ev
    # round 5
    v = ch["v"]
    parts = [t[i * n:(i + 1) * n] for i in range(4)]
    agg = lincomb([1, zn, zn * zn, zn ** 3] + [pow(v, e, R) for e in range(1, 9)],
                  parts + [r] + wc + sig_c[:3])
    out["w_z"] = ruffini(agg, zz)
    out["w_zw"] = ruffini(zc, zz * dom.group_gen % R)
    out["agg"] = agg
    return out
