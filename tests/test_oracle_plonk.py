"""CPU: the restated prover rounds (oracle/plonk_rounds_oracle.py) are internally sound -- the
quotient is exact, the verifier's scalar identity holds, the opening witnesses divide exactly -- and
break when the witness is tampered with.  Also covers the host-only pieces of the prover package
(synthetic circuit, field conversions) that need no GPU."""
import random

import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle import plonk_rounds_oracle as PO

R = B.R_MOD
CH = {k: pow(0x1234567 + 977 * i, 5 + i, R) for i, k in enumerate(PO.CHALLENGES)}


def _circuit_ints(n, seed=3, mixed=False):
    from plonk_prototype_amd.field import fr_vec_from_limbs
    from plonk_prototype_amd.synthetic import chain_circuit, mixed_circuit
    c, w, pi = (mixed_circuit if mixed else chain_circuit)(n, seed)
    sel = {k: fr_vec_from_limbs(getattr(c, k)) if getattr(c, k) is not None else [0] * n for k in PO.SELECTORS}
    wit = [fr_vec_from_limbs(w[j]) for j in range(4)]
    return sel, c.sigma_index.tolist(), wit, fr_vec_from_limbs(pi)


def test_widget_identities_vanish_exactly_on_valid_tuples():
    """Structural pins of the restated dusk-plonk 0.8 widget formulas (no upstream vectors exist)."""
    # logic: zero on exactly the AND (q_c = 1) / XOR (q_c = -1) quad tuples
    for q_c, op in ((1, lambda a, b: a & b), (R - 1, lambda a, b: a ^ b)):
        for a in range(4):
            for b in range(4):
                for c in range(4):
                    assert (PO.delta_xor_and(a, b, a * b, c, q_c) == 0) == (c == op(a, b))
    assert all((PO.delta(f) == 0) == (f < 4) for f in range(16))
    assert PO.EDWARDS_D == 0x2a9318e74bfa2b48f5fd9207e6bd7fd4292d7f6d37579d2601065fd6d6343eb1
    # curve identities: zero on a JubJub sum, non-zero when any coordinate is off by one
    from plonk_prototype_amd.synthetic import jubjub_add, jubjub_point
    p1, p2 = jubjub_point(11), jubjub_point(29)
    p3 = jubjub_add(p1, p2)
    sep = CH["var_sep"]
    good = (p1[0], p3[0], p1[1], p3[1], p2[0], p2[1], p1[0] * p2[1] % R)     # a a_next b b_next c d d_next
    assert PO.widget_variable_base(sep, *good) == 0
    for k in range(7):
        bad = list(good)
        bad[k] = (bad[k] + 1) % R
        assert PO.widget_variable_base(sep, *bad) != 0
    # fixed base: acc + bit * table point
    for bit in (0, 1, R - 1):
        sb = {0: 0, 1: 1}.get(bit, -1)
        xa, ya = p2[0] * sb % R, (sb * sb * (p2[1] - 1) + 1) % R
        q3 = jubjub_add(p1, (xa, ya))
        args = dict(a=p1[0], a_next=q3[0], b=p1[1], b_next=q3[1], c=sb * p2[0] * p2[1] % R, d=5, d_next=(10 + sb) % R,
                    q_l=p2[0], q_r=p2[1], q_c=p2[0] * p2[1] % R)
        assert PO.widget_fixed_base(CH["fixed_sep"], **args) == 0
        assert PO.widget_fixed_base(CH["fixed_sep"], **dict(args, a_next=(q3[0] + 1) % R)) != 0
        assert PO.widget_fixed_base(CH["fixed_sep"], **dict(args, d_next=(12 + sb) % R)) != 0   # bit = 2: not in {-1, 0, 1}


@pytest.mark.parametrize("n,mixed", [(4, False), (16, False), (64, False), (32, True), (64, True)])
def test_rounds_are_sound(n, mixed):
    sel, sigma, wit, pi = _circuit_ints(n, mixed=mixed)
    # the synthetic witness satisfies every gate (all five kinds) and every copy constraint
    for i in range(n):
        nx = (i + 1) % n
        g = PO.gate_value({k: sel[k][i] for k in PO.SELECTORS}, wit[0][i], wit[1][i], wit[2][i], wit[3][i],
                          wit[0][nx], wit[1][nx], wit[3][nx], CH)
        assert (g + pi[i]) % R == 0
    if mixed:
        assert all(any(sel[k]) for k in PO.WIDGET_SELECTORS)
    flat = [wit[j][i] for j in range(4) for i in range(n)]
    assert sorted(p for row in sigma for p in row) == list(range(4 * n))
    assert all(flat[j * n + i] == flat[sigma[j][i]] for j in range(4) for i in range(n))
    assert any(sigma[j][i] != j * n + i for j in range(4) for i in range(n))
    out = PO.prove(n, sel, sigma, wit, pi, CH)
    # the grand product closes: z(w^n) = z(1) = 1
    assert out["z_evals"][0] == 1
    # exact division by Z_H: deg t <= 5 (n - 1) - n, so the top four coefficients vanish
    assert not any(out["t_coeffs"][4 * n - 4:])
    pi_z = B.horner(B.ifft(pi, n.bit_length() - 1), CH["z"])
    assert PO.check_identity(out["evals"], CH, n, pi_z)
    # openings: agg(X) - agg(z) = W_z(X) (X - z) at a random point, and the shifted aggregate at z w
    rng = random.Random(5)
    x = rng.randrange(R)
    assert (B.horner(out["agg"], x) - B.horner(out["agg"], CH["z"])) % R == B.horner(out["w_z"], x) * (x - CH["z"]) % R
    zw = CH["z"] * B.Domain(n).group_gen % R
    assert (B.horner(out["agg_shifted"], x) - B.horner(out["agg_shifted"], zw)) % R == \
        B.horner(out["w_zw"], x) * (x - zw) % R
    ev = out["evals"]
    assert B.horner(out["agg_shifted"], zw) == (ev["z_next"] + CH["aw_shifted"] * ev["a_next"]
                                                + CH["aw_shifted"] ** 2 * ev["b_next"]
                                                + CH["aw_shifted"] ** 3 * ev["d_next"]) % R


@pytest.mark.parametrize("mixed,pos", [(False, (2, 5)), (True, (0, 1)), (True, (1, 4)), (True, (0, 8)), (True, (3, 13)),
                                       (True, (0, 15))])
def test_tampered_witness_breaks_the_identity(mixed, pos):
    """One wrong wire value in an arithmetic, range, logic, fixed-base or variable-base row is caught."""
    n = 32 if mixed else 16
    sel, sigma, wit, pi = _circuit_ints(n, mixed=mixed)
    pi_z = B.horner(B.ifft(pi, n.bit_length() - 1), CH["z"])
    j, i = pos
    wit[j][i] = (wit[j][i] + 1) % R
    out = PO.prove(n, sel, sigma, wit, pi, CH)
    assert any(out["t_coeffs"][4 * n - 4:]) or not PO.check_identity(out["evals"], CH, n, pi_z)
    assert not PO.check_identity(out["evals"], CH, n, pi_z)


def test_c_restatement_matches_the_big_int_restatement():
    """oracle/cpu_prover.py (C: widgets in oracle/c/plonk_oracle.c) == oracle/plonk_rounds_oracle.py on a circuit
    with every gate kind: all 17 evaluations and the polynomials behind the 11 commitments."""
    from oracle import cpu_prover as CP
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs, limbs_to_ints
    from plonk_prototype_amd.synthetic import mixed_circuit
    o = CpuOracle()
    n = 32
    c, w, pi = mixed_circuit(n, 3)
    sel, sigma, wit, pii = _circuit_ints(n, mixed=True)
    big = PO.prove(n, sel, sigma, wit, pii, CH)
    pk = CP.preprocess(o, {k: getattr(c, k) for k in PO.SELECTORS}, c.sigma_index, 2)
    srs = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0xABCDEF], 4)[0], n, 2)
    got = CP.prove(o, pk, srs, w, pi, CH, 2)
    to_int = lambda v: limbs_to_ints(o.fr_from_mont(np.ascontiguousarray(v).reshape(1, 4)))[0]   # noqa: E731
    assert set(got["evaluations"]) == set(PO.TRANSCRIPT_EVALS)
    assert all(to_int(v) == big["evals"][k] for k, v in got["evaluations"].items())
    commit = lambda coeffs: o.g1_msm(srs[:len(coeffs)], o.fr_to_mont(ints_to_limbs(coeffs, 4)), 0, 2)   # noqa: E731
    assert np.array_equal(got["commitments"]["t_3"], commit(big["t_coeffs"][2 * n:3 * n]))
    assert np.array_equal(got["commitments"]["w_z"], commit(big["w_z"]))
    assert np.array_equal(got["commitments"]["w_zw"], commit(big["w_zw"]))


def test_field_conversions_roundtrip():
    from plonk_prototype_amd import field as F
    vals = [0, 1, R - 1, 7, 2 ** 255 % R]
    limbs = F.fr_vec_to_limbs(vals)
    assert F.fr_vec_from_limbs(limbs) == vals
    assert np.array_equal(limbs[1], np.array([0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5,
                                              0x1824B159ACC5056F], dtype=np.uint64))   # R mod r
    assert all(F.fr_from_limbs(F.fr_to_limbs(v)) == v for v in vals)


@pytest.mark.parametrize("n", [4, 32])
def test_c_prover_equals_the_bigint_prover(oracle, n):
    """The two CPU restatements of the rounds (Python integers; C + OpenMP) agree on every output."""
    from oracle import cpu_prover as CP
    from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints
    from plonk_prototype_amd.synthetic import chain_circuit
    c, w, pi = chain_circuit(n, 12)
    sel_l = {k: getattr(c, k) for k in CP.SELECTORS}
    sel, sigma, wit, pii = _circuit_ints(n, 12)
    big = PO.prove(n, sel, sigma, wit, pii, CH)
    srs = oracle.g1_bases_arith(ints_to_limbs([3], 4)[0], ints_to_limbs([11], 4)[0], n, 1)
    pk = CP.preprocess(oracle, sel_l, c.sigma_index, threads=2)
    got = CP.prove(oracle, pk, srs, w, pi, CH, threads=2)
    ev = {k: limbs_to_ints(oracle.fr_from_mont(v.reshape(1, 4)))[0] for k, v in got["evaluations"].items()}
    assert ev == big["evals"]
    commit = lambda cf: oracle.g1_msm(srs[:len(cf)], oracle.fr_to_mont(ints_to_limbs(cf, 4)))   # noqa: E731
    want = {nm: commit(big["wire_coeffs"][j]) for j, nm in enumerate("abcd")}
    want["z"] = commit(big["z_coeffs"])
    for i in range(4):
        want[f"t_{i + 1}"] = commit(big["t_coeffs"][i * n:(i + 1) * n])
    want["w_z"], want["w_zw"] = commit(big["w_z"]), commit(big["w_zw"])
    assert set(want) == set(got["commitments"])
    assert all(np.array_equal(got["commitments"][k], want[k]) for k in want)
