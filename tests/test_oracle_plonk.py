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
CH = {"beta": 0x1234567 ** 5 % R, "gamma": 0x89ABCDEF ** 7 % R, "alpha": 0xDEADBEEF ** 6 % R,
      "z": 0xC0FFEE ** 9 % R, "v": 0xFACADE ** 8 % R}


def _circuit_ints(n, seed=3):
    from plonk_prototype_amd.field import fr_vec_from_limbs
    from plonk_prototype_amd.synthetic import chain_circuit
    c, w, pi = chain_circuit(n, seed)
    sel = {k: fr_vec_from_limbs(getattr(c, k)) for k in ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")}
    wit = [fr_vec_from_limbs(w[j]) for j in range(4)]
    return sel, c.sigma_index.tolist(), wit, fr_vec_from_limbs(pi)


def _check_identity(ev, n, pi_z):
    zz, a, b, g = CH["z"], CH["alpha"], CH["beta"], CH["gamma"]
    zn = pow(zz, n, R)
    l1 = (zn - 1) * pow(n * (zz - 1) % R, -1, R) % R
    rhs = (ev["r"] + pi_z - a * (ev["a"] + b * ev["sigma_1"] + g) * (ev["b"] + b * ev["sigma_2"] + g)
           * (ev["c"] + b * ev["sigma_3"] + g) * (ev["d"] + g) * ev["z_next"] - a * a * l1) % R
    return ev["t"] * (zn - 1) % R == rhs


@pytest.mark.parametrize("n", [4, 16, 64])
def test_rounds_are_sound(n):
    sel, sigma, wit, pi = _circuit_ints(n)
    # the synthetic witness satisfies every gate and every copy constraint
    for i in range(n):
        a, b, c, d = (wit[j][i] for j in range(4))
        assert (sel["q_m"][i] * a * b + sel["q_l"][i] * a + sel["q_r"][i] * b + sel["q_o"][i] * c
                + sel["q_4"][i] * d + sel["q_c"][i] + pi[i]) % R == 0
    flat = [wit[j][i] for j in range(4) for i in range(n)]
    assert sorted(p for row in sigma for p in row) == list(range(4 * n))
    assert all(flat[j * n + i] == flat[sigma[j][i]] for j in range(4) for i in range(n))
    assert any(sigma[j][i] != j * n + i for j in range(4) for i in range(n))
    out = PO.prove(n, sel, sigma, wit, pi, CH)
    # the grand product closes: z(w^n) = z(1) = 1
    assert out["z_evals"][0] == 1
    # exact division by Z_H: deg t = 5 (n - 1) - n, so the top four coefficients vanish
    assert not any(out["t_coeffs"][4 * n - 4:])
    pi_z = B.horner(B.ifft(pi, n.bit_length() - 1), CH["z"])
    assert _check_identity(out["evals"], n, pi_z)
    # openings: agg(X) - agg(z) = W_z(X) (X - z) at a random point
    rng = random.Random(5)
    x = rng.randrange(R)
    assert (B.horner(out["agg"], x) - B.horner(out["agg"], CH["z"])) % R == B.horner(out["w_z"], x) * (x - CH["z"]) % R
    zw = CH["z"] * B.Domain(n).group_gen % R
    assert (B.horner(out["z_coeffs"], x) - out["evals"]["z_next"]) % R == B.horner(out["w_zw"], x) * (x - zw) % R


def test_tampered_witness_breaks_the_identity():
    n = 16
    sel, sigma, wit, pi = _circuit_ints(n)
    pi_z = B.horner(B.ifft(pi, 4), CH["z"])
    wit[2][5] = (wit[2][5] + 1) % R                       # breaks gate 5 and a copy constraint
    out = PO.prove(n, sel, sigma, wit, pi, CH)
    assert any(out["t_coeffs"][4 * n - 4:]) or not _check_identity(out["evals"], n, pi_z)
    assert not _check_identity(out["evals"], n, pi_z)


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
