"""The small helpers of dusk_plonk::fft::EvaluationDomain (VERDICT r03 missing #4):
evaluate_vanishing_polynomial, compute_vanishing_poly_over_coset, evaluate_all_lagrange_coefficients -- the host
forms of the C ABI against the big-int oracle (CPU), the device forms against the host forms and against the
Lagrange-interpolation property (GPU)."""
import random

import numpy as np
import pytest

from oracle import bigint_oracle as B

R = B.R_MOD


def _mont(v):
    from plonk_prototype_amd.field import fr_to_limbs
    return fr_to_limbs(v)


def _ints(a):
    from plonk_prototype_amd.field import fr_vec_from_limbs
    return fr_vec_from_limbs(a)


@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 10])
def test_host_helpers_match_the_oracle(log_n):
    import plonk_prototype_amd as pa
    rng = random.Random(100 + log_n)
    n = 1 << log_n
    dom, ref = pa.EvaluationDomain(n), B.Domain(n)
    for tau in (rng.randrange(R), 0, 1, ref.group_gen, pow(ref.group_gen, n - 1, R), R - 1):
        assert _ints(dom.evaluate_vanishing_polynomial(_mont(tau)))[0] == ref.evaluate_vanishing_polynomial(tau)
        assert _ints(dom.evaluate_all_lagrange_coefficients(_mont(tau))) == ref.evaluate_all_lagrange_coefficients(tau), tau
    for deg in sorted({0, 1, n // 4, n - 1}):
        if deg < n:
            assert _ints(dom.compute_vanishing_poly_over_coset(deg)) == ref.compute_vanishing_poly_over_coset(deg), deg


def test_lagrange_coefficients_interpolate():
    """sum_i f(w^i) L_i(tau) = f(tau) for a polynomial of degree < n: ties the helper to the transforms' domain."""
    import plonk_prototype_amd as pa
    rng = random.Random(5)
    n = 64
    dom, ref = pa.EvaluationDomain(n), B.Domain(n)
    coeffs = [rng.randrange(R) for _ in range(n)]
    evals = [B.horner(coeffs, e) for e in ref.elements()]
    tau = rng.randrange(R)
    lag = _ints(dom.evaluate_all_lagrange_coefficients(_mont(tau)))
    assert sum(a * b for a, b in zip(evals, lag)) % R == B.horner(coeffs, tau)


def test_helper_errors():
    import plonk_prototype_amd as pa
    from plonk_prototype_amd import _lib
    dom = pa.EvaluationDomain(16)
    with pytest.raises(pa.Error) as e:
        dom.compute_vanishing_poly_over_coset(16)            # upstream: assert!(domain.size() > poly_degree)
    assert e.value.code == _lib.PM_ERR_BAD_ARG
    lib = pa.load()
    out = np.zeros(4, np.uint64)
    u64p = _lib.u64p
    assert lib.pm_domain_evaluate_vanishing_polynomial(32, _mont(3).ctypes.data_as(u64p), out.ctypes.data_as(u64p)) == -2
    assert lib.pm_domain_vanishing_poly_over_coset(32, 1, out.ctypes.data_as(u64p)) == -2
    assert lib.pm_domain_evaluate_all_lagrange_coefficients(40, _mont(3).ctypes.data_as(u64p), out.ctypes.data_as(u64p)) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [0, 1, 5, 12, 16])
def test_device_helpers_match_the_host_forms(ctx, log_n):
    import plonk_prototype_amd as pa
    rng = random.Random(7 + log_n)
    n = 1 << log_n
    dom, ref = pa.EvaluationDomain(n, ctx), B.Domain(n)
    for tau in (rng.randrange(R), ref.group_gen, pow(ref.group_gen, n - 1, R), 1):
        host = dom.evaluate_all_lagrange_coefficients(_mont(tau))
        dev = dom.evaluate_all_lagrange_coefficients(_mont(tau), device=True)
        assert np.array_equal(dev.to_host(), host), tau
        dev.free()
    for deg in sorted({0, 1, n // 4, n - 1}):
        if deg < n:
            dev = dom.compute_vanishing_poly_over_coset(deg, device=True)
            assert np.array_equal(dev.to_host(), dom.compute_vanishing_poly_over_coset(deg)), deg
            dev.free()
    with pytest.raises(pa.Error):
        dom.compute_vanishing_poly_over_coset(n, device=True)


@pytest.mark.gpu
def test_vanishing_poly_over_coset_prover_shape(ctx, oracle):
    """The prover's call: the 4n domain, poly_degree = n (2^18 gates): the device vector against the closed form --
    (g w_4n^i)^n - 1 takes only four values, g^n i_4^(i mod 4) - 1 with i_4 = w_4n^n a primitive fourth root of unity."""
    import plonk_prototype_amd as pa
    n = 1 << 18
    dom, ref = pa.EvaluationDomain(4 * n, ctx), B.Domain(4 * n)
    v = dom.compute_vanishing_poly_over_coset(n, device=True)
    got = v.to_host()
    v.free()
    gn, i4 = pow(B.FR_GENERATOR, n, R), pow(ref.group_gen, n, R)
    four = np.stack([_mont((gn * pow(i4, k, R) - 1) % R) for k in range(4)])
    assert np.array_equal(got, np.tile(four, (n, 1)))
