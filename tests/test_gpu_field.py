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
