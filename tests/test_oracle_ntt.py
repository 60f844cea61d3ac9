"""C restatement of EvaluationDomain::{fft,ifft,coset_fft,coset_ifft} (oracle/c) against the
big-int oracle, the committed golden vectors and the algebraic identities of SURVEY.md 8c."""
import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import COSET, INVERSE, ints_to_limbs, limbs_to_ints
from conftest import hex_to_fr_mont

FLAGS = {"fft": 0, "ifft": INVERSE, "coset_fft": COSET, "coset_ifft": INVERSE | COSET}


def canon(oracle, a):
    return limbs_to_ints(oracle.fr_from_mont(a))


@pytest.mark.parametrize("threads", [1, 4])
def test_golden_vectors(oracle, golden, threads):
    for v in golden["ntt"]:
        a = hex_to_fr_mont(oracle, v["input"])
        for name, flags in FLAGS.items():
            got = canon(oracle, oracle.fr_ntt(a, v["log_n"], flags, threads))
            assert got == [int(h, 16) for h in v[name]], (v["log_n"], name)


@pytest.mark.parametrize("k", [0, 1, 2, 5, 9, 12])
def test_against_bigint_recursive(oracle, k):
    n = 1 << k
    vals = B.sample_fr(77 + k, n)
    a = oracle.fr_sample(77 + k, n)
    assert canon(oracle, a) == vals                      # both samplers agree
    for flags, fn in [(0, B.fft), (INVERSE, B.ifft), (COSET, B.coset_fft), (INVERSE | COSET, B.coset_ifft)]:
        assert canon(oracle, oracle.fr_ntt(a, k, flags, 1)) == fn(vals, k)
        assert canon(oracle, oracle.fr_ntt(a, k, flags, 8)) == fn(vals, k)   # parallel_fft shape


def test_identities(oracle):
    k, n = 10, 1 << 10
    d = B.Domain(n)
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))
    # NTT(delta_0) = all ones ; NTT(delta_1)[j] = w^j
    ones = canon(oracle, oracle.fr_ntt(one, k, 0))
    assert ones == [1] * n
    d1 = np.zeros((2, 4), np.uint64)
    d1[1] = one[0]
    assert canon(oracle, oracle.fr_ntt(d1, k, 0)) == list(d.elements())
    # round trips, zero padding, Horner evaluation, linearity
    a = oracle.fr_sample(5, n - 37)
    av = canon(oracle, a)
    f = oracle.fr_ntt(a, k, 0)
    assert canon(oracle, oracle.fr_ntt(f, k, INVERSE)) == av + [0] * 37
    cf = oracle.fr_ntt(a, k, COSET)
    assert canon(oracle, oracle.fr_ntt(cf, k, INVERSE | COSET)) == av + [0] * 37
    fv, cfv = canon(oracle, f), canon(oracle, cf)
    for j in (0, 1, 17, n - 1):
        wj = pow(d.group_gen, j, B.R_MOD)
        assert fv[j] == B.horner(av, wj)
        assert cfv[j] == B.horner(av, 7 * wj % B.R_MOD)
    b = oracle.fr_sample(6, n - 37)
    s = ints_to_limbs([(x + y) % B.R_MOD for x, y in zip(av, canon(oracle, b))], 4)
    fs = canon(oracle, oracle.fr_ntt(oracle.fr_to_mont(s), k, 0))
    assert fs == [(x + y) % B.R_MOD for x, y in zip(fv, canon(oracle, oracle.fr_ntt(b, k, 0)))]


def test_domain_too_large(oracle):
    with pytest.raises(ValueError):
        oracle.fr_ntt(np.zeros((1, 4), np.uint64), 32, 0)
    with pytest.raises(ValueError):
        B.Domain((1 << 32))
    with pytest.raises(ValueError):
        oracle.fr_ntt(np.zeros((5, 4), np.uint64), 2, 0)   # longer than the domain
