"""The oracle's restatements of the polynomial helpers against big-int arithmetic (CPU)."""
import numpy as np

from oracle import bigint_oracle as B
from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints


def test_poly_helpers_against_bigint(oracle):
    n = 300
    av, bv = B.sample_fr(1, n), B.sample_fr(2, n)
    av[3] = 0
    a = oracle.fr_to_mont(ints_to_limbs(av, 4))
    b = oracle.fr_to_mont(ints_to_limbs(bv, 4))
    f = lambda x: limbs_to_ints(oracle.fr_from_mont(x))
    assert f(oracle.fr_vec_op(0, a, b)) == [(x + y) % B.R_MOD for x, y in zip(av, bv)]
    assert f(oracle.fr_vec_op(1, a, b)) == [(x - y) % B.R_MOD for x, y in zip(av, bv)]
    assert f(oracle.fr_vec_op(2, a, b)) == [x * y % B.R_MOD for x, y in zip(av, bv)]
    assert f(oracle.fr_vec_op(2, a, b[:1])) == [x * bv[0] % B.R_MOD for x in av]
    assert f(oracle.fr_batch_inverse(a)) == [pow(x, -1, B.R_MOD) if x else 0 for x in av]
    z = bv[7]
    assert f(oracle.fr_poly_evaluate(a, b[7]).reshape(1, 4))[0] == B.horner(av, z)
    q = f(oracle.fr_poly_ruffini(a, b[7]))
    prod = [0] * n                                   # q(X) (X - z) + a(z) == a(X)
    for i, c in enumerate(q):
        prod[i + 1] = (prod[i + 1] + c) % B.R_MOD
        prod[i] = (prod[i] - c * z) % B.R_MOD
    prod[0] = (prod[0] + B.horner(av, z)) % B.R_MOD
    assert prod == av
    pp, acc = f(oracle.fr_prefix_product(a)), 1
    for i in range(n):
        assert pp[i] == acc
        acc = acc * av[i] % B.R_MOD
