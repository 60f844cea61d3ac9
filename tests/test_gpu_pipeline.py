"""The primitives composed the way a PLONK prover round composes them (upstream's KZG
commit/open/check round trip): a trapdoor SRS [tau^i] G lets the test verify
commitments and openings with plain scalar arithmetic instead of pairings."""
import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints

pytestmark = pytest.mark.gpu
TAU = 0x1F2E3D4C5B6A79788796A5B4C3D2E1F00112233445566778899AABBCCDDEEFF % B.R_MOD


def _srs(oracle, n):
    """powers_of_g = [tau^i] G (the shape of dusk's PublicParameters::setup)."""
    G = oracle.g1_generator()
    out = np.zeros((n, 12), np.uint64)
    t = 1
    for i in range(n):
        out[i] = oracle.g1_mul(G, ints_to_limbs([t], 4)[0])
        t = t * TAU % B.R_MOD
    return out


def _scalar_times_g(oracle, k):
    return oracle.g1_mul(oracle.g1_generator(), ints_to_limbs([k % B.R_MOD], 4)[0])


def test_kzg_commit_open_check(ctx, oracle):
    import plonk_prototype_amd as pa
    n = 512
    ck = pa.CommitKey(_srs(oracle, n), ctx, precompute=True)
    dom = pa.EvaluationDomain(n, ctx)
    # a "wire polynomial": given by its evaluations on the domain, interpolated on the GPU
    evals = oracle.fr_sample(2024, n)
    coeffs = dom.ifft(evals)
    assert np.array_equal(dom.fft(coeffs), evals)
    cv = limbs_to_ints(oracle.fr_from_mont(coeffs))
    # commit(p) == p(tau) G
    comm = ck.commit(coeffs)
    assert np.array_equal(comm, _scalar_times_g(oracle, B.horner(cv, TAU)))
    # opening at a challenge z: witness q = (p - p(z)) / (X - z); check  commit(q) (tau - z) + p(z) G == commit(p)
    z = oracle.fr_sample(7, 1)[0]
    zv = limbs_to_ints(oracle.fr_from_mont(z.reshape(1, 4)))[0]
    p = pa.Polynomial.from_host(ctx, coeffs)
    pz = p.evaluate(z)
    pzv = limbs_to_ints(oracle.fr_from_mont(pz.reshape(1, 4)))[0]
    assert pzv == B.horner(cv, zv)
    q = p.ruffini(z).to_host()
    wit = ck.commit(q)
    qv = limbs_to_ints(oracle.fr_from_mont(q))
    assert np.array_equal(wit, _scalar_times_g(oracle, B.horner(qv, TAU)))
    lhs = _scalar_times_g(oracle, B.horner(qv, TAU) * (TAU - zv) + pzv)
    assert np.array_equal(lhs, comm)
    # commitments are linear: commit(a + s b) == commit(a) + s commit(b)
    b = oracle.fr_sample(9, n)
    s = oracle.fr_sample(10, 1)
    combo = (pa.Polynomial.from_host(ctx, coeffs) + pa.Polynomial.from_host(ctx, b) * pa.Polynomial.from_host(ctx, s)).to_host()
    bv = limbs_to_ints(oracle.fr_from_mont(b))
    sv = limbs_to_ints(oracle.fr_from_mont(s))[0]
    assert np.array_equal(ck.commit(combo), _scalar_times_g(oracle, B.horner(cv, TAU) + sv * B.horner(bv, TAU)))


def test_quotient_on_the_4n_coset(ctx, oracle):
    """The quotient-polynomial pattern: evaluate on the 4n coset, combine pointwise, divide by the
    vanishing polynomial there, come back with coset_ifft.  With t = a*b - c and c := (a*b) mod Z_H the
    division is exact, so the recovered quotient must satisfy a b = c + q Z_H at a random point."""
    import plonk_prototype_amd as pa
    n, k = 256, 8
    d4 = pa.EvaluationDomain(4 * n, ctx)
    dn = pa.EvaluationDomain(n, ctx)
    a, b = oracle.fr_sample(31, n), oracle.fr_sample(32, n)
    av, bv = (limbs_to_ints(oracle.fr_from_mont(v)) for v in (a, b))
    # c = a*b reduced mod (X^n - 1): fold the product's upper half onto the lower
    prod = [0] * (2 * n)
    for i, x in enumerate(av):
        for j, y in enumerate(bv):
            prod[i + j] = (prod[i + j] + x * y) % B.R_MOD
    cvals = [(prod[i] + prod[i + n]) % B.R_MOD for i in range(n)]
    qvals = prod[n:]                                          # a b = c + q (X^n - 1) with q = upper half
    c = oracle.fr_to_mont(ints_to_limbs(cvals, 4))
    ea, eb, ec = (pa.Polynomial.from_host(ctx, d4.coset_fft(v)) for v in (a, b, c))
    num = ea * eb - ec
    # Z_H on the coset: (7 w^i)^n - 1 takes only 4 values; build its evaluations with the same kernels
    zh = np.zeros((n + 1, 4), np.uint64)
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
    zh[0] = oracle.fr_to_mont(ints_to_limbs([B.R_MOD - 1], 4))[0]
    zh[n] = one
    ezh = pa.Polynomial.from_host(ctx, d4.coset_fft(zh))
    quot = d4.coset_ifft((num * ezh.batch_inverse()).to_host())
    assert not quot[n:].any()                                 # degree < n, as predicted
    assert limbs_to_ints(oracle.fr_from_mont(quot[:n])) == qvals
    assert dn.size == n
