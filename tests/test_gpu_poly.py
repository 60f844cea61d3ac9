"""Parity of the device-resident polynomial helpers (Polynomial /
Evaluations ops, evaluate, ruffini, batch_inversion) with the oracle's restatements."""
import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints

pytestmark = pytest.mark.gpu


def _poly(ctx, a):
    import plonk_prototype_amd as pa
    return pa.Polynomial.from_host(ctx, a)


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 5000, 1 << 16, (1 << 18) + 3])
def test_vec_ops(ctx, oracle, n):
    a, b = oracle.fr_sample(1, n), oracle.fr_sample(2, n)
    edge = oracle.fr_to_mont(ints_to_limbs([0, 1, B.R_MOD - 1], 4))
    a[: min(n, 3)] = edge[: min(n, 3)]
    b[-min(n, 3):] = edge[: min(n, 3)]
    pa_, pb_ = _poly(ctx, a), _poly(ctx, b)
    for op, py in ((0, pa_ + pb_), (1, pa_ - pb_), (2, pa_ * pb_)):
        assert np.array_equal(py.to_host(), oracle.fr_vec_op(op, a, b)), (n, op)
    s = _poly(ctx, b[:1])                                           # scalar broadcast
    for op, py in ((0, pa_ + s), (1, pa_ - s), (2, pa_ * s)):
        assert np.array_equal(py.to_host(), oracle.fr_vec_op(op, a, b[:1])), (n, op)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 256, 257, 4097, 1 << 16, (1 << 20) + 17])
def test_evaluate(ctx, oracle, n):
    c = oracle.fr_sample(3 + n, n)
    for seed in (5, 6):
        x = oracle.fr_sample(seed, 1)[0]
        assert np.array_equal(_poly(ctx, c).evaluate(x), oracle.fr_poly_evaluate(c, x)), n
    zero = np.zeros(4, np.uint64)
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
    assert np.array_equal(_poly(ctx, c).evaluate(zero), c[0] if n else zero)                 # p(0) = c_0
    assert np.array_equal(_poly(ctx, c).evaluate(one), oracle.fr_poly_evaluate(c, one))      # p(1) = sum c_i


@pytest.mark.parametrize("n", [1, 2, 3, 256, 257, 258, 2048, 2049, 2050, 4097, 5000, (1 << 16) + 1, (1 << 20) + 5, (1 << 21) + 3,
                               (1 << 22) + 1, (1 << 22) + 77, (1 << 23) + 5])
def test_ruffini(ctx, oracle, n):
    """(r06: one scheme for every size -- the scaled prefix sum; the tile carries are summed inside the replay kernel up to 512
    tiles = 2^20 elements and by a kernel of their own beyond, in sweeps of 2048 tiles: 2^22 + 77 and 2^23 + 5 take two and three.
    z = 1 and z = -1 make every power table entry +-1; z = 0 is the shift.)"""
    c = oracle.fr_sample(7 + n, n)
    for z in (oracle.fr_sample(9, 1)[0], np.zeros(4, np.uint64), oracle.fr_to_mont(ints_to_limbs([1], 4))[0],
              oracle.fr_to_mont(ints_to_limbs([B.R_MOD - 1], 4))[0]):
        q = _poly(ctx, c).ruffini(z).to_host()
        assert q.shape[0] == n - 1
        assert np.array_equal(q, oracle.fr_poly_ruffini(c, z)), n
    if n > 1:   # q(X) (X - z) + c(z) == c(X) at a random point (the defining identity)
        z = oracle.fr_sample(9, 1)[0]
        q = _poly(ctx, c).ruffini(z)
        x = oracle.fr_sample(10, 1)[0]
        xv, zv = limbs_to_ints(oracle.fr_from_mont(np.stack([x, z])))
        qx, cx, cz = (limbs_to_ints(oracle.fr_from_mont(v.reshape(1, 4)))[0]
                      for v in (q.evaluate(x), _poly(ctx, c).evaluate(x), _poly(ctx, c).evaluate(z)))
        assert (qx * (xv - zv) + cz) % B.R_MOD == cx


@pytest.mark.parametrize("n,quads", [(1, 0), (2, 0), (3, 1), (63, 0), (64, 0), (65, 0), (1000, 0), (1000, 3), (5000, 7), (1 << 16, 0),
                                     (1 << 16, 2), ((1 << 18) + 11, 0), ((1 << 18) + 11, 64), ((1 << 20) + 3, 0), ((1 << 22) + 5, 0)])
def test_batch_inverse(ctx, oracle, n, quads):
    """quads: option binv_quads (0 = the library's rule: one quad per thread up to 2^18 elements, 4 at 2^20, 16 at 2^22) -- small
    sizes with several quads per thread exercise the scratch chain between the quads and ragged last quads."""
    ctx.set_option("binv_quads", quads)
    try:
        _batch_inverse_case(ctx, oracle, n)
    finally:
        ctx.set_option("binv_quads", 0)


def _batch_inverse_case(ctx, oracle, n):
    a = oracle.fr_sample(11 + n, n)
    a[:: max(1, n // 7)] = 0                                       # zeros stay zero
    if n > 5:
        a[5] = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
        a[4] = oracle.fr_to_mont(ints_to_limbs([B.R_MOD - 1], 4))[0]
    got = _poly(ctx, a).batch_inverse().to_host()
    assert np.array_equal(got, oracle.fr_batch_inverse(a)), n
    # a * a^-1 == 1 wherever a != 0
    prod = (_poly(ctx, a) * _poly(ctx, got)).to_host()
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
    nz = a.any(axis=1)
    assert np.array_equal(prod[nz], np.broadcast_to(one, (int(nz.sum()), 4))) and not prod[~nz].any()


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 2047, 2048, 2049, 4096, 4097, 5000, 1 << 16, (1 << 20) + 3, (1 << 21) + (1 << 19),
                               (1 << 22) + 9, (1 << 23) + 3])
def test_prefix_product(ctx, oracle, n):
    """(above 2^21 the library's choice is the chunked scan with ONE look-back launch over the first inner level that is
    resident at once -- level 1 up to 2^23, level 2 beyond: r05)"""
    a = oracle.fr_sample(21 + n, n)
    if n > 6000:
        a[5000 if n < (1 << 20) else n - 777] = 0         # everything after a zero factor is zero
    want = oracle.fr_prefix_product(a)
    try:
        for mode in (1, 0, 2):                            # the library's choice, the three-stage scan, the one-pass look-back
            ctx.set_option("poly_lookback", mode)
            assert np.array_equal(_poly(ctx, a).prefix_product().to_host(), want), (n, mode)
    finally:
        ctx.set_option("poly_lookback", 1)


@pytest.mark.parametrize("log_n", [24, 26])
def test_full_size_properties(ctx, oracle, log_n):
    """BASELINE.json's sizes (2^24 = configs[4]'s vectors; 2^26 = its 4n domain) through size-independent properties, all on
    the device: Ruffini through q(x)(x - z) + p(z) = p(x) at a random x (8193 / 32769 tiles: the carry kernel in several
    sweeps); batch inversion through a * a^-1 = 1 with zeros staying zero (quads per thread at the rule's cap); the prefix
    product through out[0] = 1 and out[k + 1] = out[k] a[k], checked on the whole vector with two vector products."""
    import plonk_prototype_amd as pa
    n = 1 << log_n
    host = oracle.fr_sample(4242 + log_n, n)
    host[7] = 0
    host[n - 2] = 0
    a = pa.DeviceVector.from_host(ctx, host)
    out = pa.DeviceVector(ctx, n)
    tmp = pa.DeviceVector(ctx, n)
    fi = pa.field.fr_from_limbs
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
    # Ruffini
    z, x = oracle.fr_sample(1, 1)[0], oracle.fr_sample(2, 1)[0]
    ctx.fr_ruffini(a.ptr, n, z, out.ptr)
    px, pz, qx = fi(ctx.fr_evaluate(a.ptr, n, x)), fi(ctx.fr_evaluate(a.ptr, n, z)), fi(ctx.fr_evaluate(out.ptr, n - 1, x))
    assert (qx * (fi(x) - fi(z)) + pz - px) % B.R_MOD == 0
    # batch inversion (in place on a copy)
    ctx._check(ctx._lib.pm_dev_upload(ctx._h, out._p, host.ctypes.data, n * 32))
    ctx.fr_batch_inverse(out.ptr, n)
    ctx.fr_vec_op(2, a.ptr, out.ptr, n, tmp.ptr, n)
    prod = tmp.to_host()
    assert not prod[7].any() and not prod[n - 2].any()
    prod[7] = one
    prod[n - 2] = one
    assert (prod == one).all()
    del prod
    # prefix product: out[k + 1] == out[k] * a[k] for every k, out[0] == 1 (zeros in `a` make everything after them zero)
    ctx.fr_prefix_product(a.ptr, n, out.ptr)
    ctx.fr_vec_op(2, out.ptr, a.ptr, n, tmp.ptr, n)                  # tmp[k] = out[k] a[k]
    got, want = out.to_host(), tmp.to_host()
    assert np.array_equal(got[0], one) and np.array_equal(got[1:], want[:-1])
    assert got[7].any() and not got[8:].any()                        # a[7] = 0: the product is zero from index 8 on
    for v in (a, out, tmp):
        v.free()


def test_errors(ctx, oracle):
    import plonk_prototype_amd as pa
    a, b = _poly(ctx, oracle.fr_sample(1, 10)), _poly(ctx, oracle.fr_sample(2, 7))
    with pytest.raises(pa.Error) as e:
        a + b
    assert e.value.code == -6
