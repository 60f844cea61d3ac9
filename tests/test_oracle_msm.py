"""C restatement of msm_variable_base (Pippenger, oracle/c) against the big-int oracle's
double-and-add MSM, the golden vectors and the known-answer identities of SURVEY.md 8c."""
import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import SCALAR_CANONICAL, SCALAR_MONTGOMERY, ints_to_limbs, limbs_to_ints
from conftest import hex_to_fr_mont, points_to_mont


def aff_ints(oracle, xy):
    x, y = limbs_to_ints(oracle.fp_from_mont(np.asarray(xy).reshape(2, 6)))
    return None if (x, y) == (0, 0) else (x, y)


@pytest.mark.parametrize("threads", [1, 4])
def test_golden_vectors(oracle, golden, threads):
    for v in golden["msm"]:
        pts = points_to_mont(oracle, v["points"])
        sc = hex_to_fr_mont(oracle, v["scalars"])
        exp = None if v["result"] is None else (int(v["result"][0], 16), int(v["result"][1], 16))
        assert aff_ints(oracle, oracle.g1_msm(pts, sc, SCALAR_MONTGOMERY, threads)) == exp, v["n"]
        canon = oracle.fr_from_mont(sc) if len(sc) else sc
        assert aff_ints(oracle, oracle.g1_msm(pts, canon, SCALAR_CANONICAL, threads)) == exp, v["n"]


def test_generator_and_group_law(oracle):
    G = oracle.g1_generator()
    assert aff_ints(oracle, G) == B.G1_GEN and oracle.g1_is_on_curve(G)
    k = ints_to_limbs([0xdeadbeefcafebabe1234], 4)[0]
    assert aff_ints(oracle, oracle.g1_mul(G, k)) == B.g1_mul(0xdeadbeefcafebabe1234, B.G1_GEN)
    P = oracle.g1_mul(G, k)
    assert aff_ints(oracle, oracle.g1_add(P, P)) == B.g1_mul(2 * 0xdeadbeefcafebabe1234, B.G1_GEN)
    negP = P.copy()
    negP[6:] = oracle.fp_to_mont(ints_to_limbs([B.P_MOD - aff_ints(oracle, P)[1]], 6))[0]
    assert aff_ints(oracle, oracle.g1_add(P, negP)) is None
    assert aff_ints(oracle, oracle.g1_add(P, np.zeros(12, np.uint64))) == aff_ints(oracle, P)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 300])
def test_discrete_log_identity(oracle, n):
    """bases (k0 + i d) G  =>  MSM = (sum s_i (k0 + i d)) G, checked with ONE scalar mul."""
    k0 = ints_to_limbs([0x1234567], 4)[0]
    d = ints_to_limbs([0xabcdef0123456789abcdef], 4)[0]
    pts = oracle.g1_bases_arith(k0, d, n, 2)
    assert all(oracle.g1_is_on_curve(p) for p in pts[:: max(1, n // 7)])
    sc = oracle.fr_sample(42 + n, n)
    got = oracle.g1_msm(pts, sc, SCALAR_MONTGOMERY, 4)
    dl = oracle.expected_dlog(sc, SCALAR_MONTGOMERY, k0, d)
    assert np.array_equal(got, oracle.g1_mul(oracle.g1_generator(), dl))
    assert aff_ints(oracle, got) == B.g1_mul(limbs_to_ints(dl)[0], B.G1_GEN)


def test_edge_cases(oracle):
    k0 = ints_to_limbs([5], 4)[0]
    d = ints_to_limbs([11], 4)[0]
    n = 40
    pts = oracle.g1_bases_arith(k0, d, n, 1)
    zero = np.zeros((n, 4), np.uint64)
    assert aff_ints(oracle, oracle.g1_msm(pts, zero)) is None                     # all s = 0
    assert aff_ints(oracle, oracle.g1_msm(pts[:0], zero[:0])) is None             # n = 0
    ones = oracle.fr_to_mont(ints_to_limbs([1] * n, 4))
    exp = None
    for i in range(n):
        exp = B.g1_add(exp, B.g1_mul(5 + 11 * i, B.G1_GEN))
    assert aff_ints(oracle, oracle.g1_msm(pts, ones)) == exp                      # all s = 1 -> sum P_i
    rm1 = oracle.fr_to_mont(ints_to_limbs([B.R_MOD - 1] * n, 4))
    assert aff_ints(oracle, oracle.g1_msm(pts, rm1)) == B.g1_neg(exp)             # s = r-1 -> -P_i
    with pytest.raises(ValueError):
        oracle.g1_msm(pts, ones[:-1])
