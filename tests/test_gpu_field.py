"""Elementwise parity of the carry-free Fr / Fp device arithmetic with the oracle."""
import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints

pytestmark = pytest.mark.gpu


def _edge_values(mod):
    return [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << 200) % mod, (1 << 255) % mod, (1 << 380) % mod]


@pytest.mark.parametrize("field", ["fr", "fp"])
def test_mul_add_sub(ctx, oracle, field):
    mod, nl, base = (B.R_MOD, 4, 0) if field == "fr" else (B.P_MOD, 6, 3)
    rng = B.sample_fr(1 if field == "fr" else 2, 4000)
    vals = [v * 0x9E3779B97F4A7C15 % mod for v in rng] if field == "fp" else rng
    edge = _edge_values(mod)
    a = edge * len(edge) + vals[:2000]
    b = [e for e in edge for _ in edge] + vals[2000:]
    to_mont = oracle.fr_to_mont if field == "fr" else oracle.fp_to_mont
    from_mont = oracle.fr_from_mont if field == "fr" else oracle.fp_from_mont
    am, bm = to_mont(ints_to_limbs(a, nl)), to_mont(ints_to_limbs(b, nl))
    for op, fn in [(0, lambda x, y: x * y % mod), (1, lambda x, y: (x + y) % mod), (2, lambda x, y: (x - y) % mod)]:
        got = limbs_to_ints(from_mont(ctx.field_op(base + op, am, bm)))
        assert got == [fn(x, y) for x, y in zip(a, b)], (field, op)


@pytest.mark.parametrize("field", ["fr", "fp"])
def test_inverse(ctx, oracle, field):
    """The binary-GCD inversion of csrc/field_inv.hip.h against pow(x, -1, m): edge values (0 stays 0, 1, m - 1, powers of
    two, small numbers whose approximations are exact from the first round on) and random ones."""
    mod, nl, op = (B.R_MOD, 4, 6) if field == "fr" else (B.P_MOD, 6, 7)
    rng = B.sample_fr(5 if field == "fr" else 6, 3000)
    vals = [v * 0x9E3779B97F4A7C15 % mod for v in rng] if field == "fp" else rng
    a = _edge_values(mod) + list(range(2, 40)) + [mod - k for k in range(3, 20)] + [(1 << k) % mod for k in range(1, 384, 7)] + vals
    to_mont = oracle.fr_to_mont if field == "fr" else oracle.fp_to_mont
    from_mont = oracle.fr_from_mont if field == "fr" else oracle.fp_from_mont
    am = to_mont(ints_to_limbs(a, nl))
    got = limbs_to_ints(from_mont(ctx.field_op(op, am, am)))
    assert got == [pow(x, -1, mod) if x else 0 for x in a], field


def test_inverse_ignores_b(ctx, oracle):
    """ops 6 / 7 document `b ignored`: NULL must be accepted; the binary ops still need both operands."""
    import ctypes as C
    from plonk_prototype_amd import _lib
    a = oracle.fr_to_mont(ints_to_limbs([5, 7, 11], 4))
    out = np.empty_like(a)
    u64p = _lib.u64p
    assert ctx._lib.pm_test_field_op(ctx._h, 6, a.ctypes.data_as(u64p), None, out.ctypes.data_as(u64p), 3) == 0
    assert limbs_to_ints(oracle.fr_from_mont(out)) == [pow(x, -1, B.R_MOD) for x in (5, 7, 11)]
    assert ctx._lib.pm_test_field_op(ctx._h, 0, a.ctypes.data_as(u64p), None, out.ctypes.data_as(u64p), 3) == _lib.PM_ERR_BAD_ARG
