"""Parity of the HIP NTT (through the C ABI) with the oracle: bit-exact, every transform kind,
every pass plan, the reference's edge cases, and size-independent properties at full size."""
import ctypes as C

import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import COSET, INVERSE, ints_to_limbs, limbs_to_ints
from conftest import hex_to_fr_mont

pytestmark = pytest.mark.gpu
ALL_FLAGS = [0, INVERSE, COSET, INVERSE | COSET]
NAMES = {0: "fft", INVERSE: "ifft", COSET: "coset_fft", INVERSE | COSET: "coset_ifft"}


def test_golden_vectors(ctx, oracle, golden):
    for v in golden["ntt"]:
        a = hex_to_fr_mont(oracle, v["input"])
        for flags, name in NAMES.items():
            got = limbs_to_ints(oracle.fr_from_mont(ctx.fr_ntt(a, v["log_n"], flags)))
            assert got == [int(h, 16) for h in v[name]], (v["log_n"], name)


@pytest.mark.parametrize("k", list(range(0, 15)) + [16, 17, 18, 19, 20])
def test_every_plan_against_oracle(ctx, oracle, k):
    n = 1 << k
    a = oracle.fr_sample(0x504C4F4E4B + k, n)
    for flags in ALL_FLAGS:
        assert np.array_equal(ctx.fr_ntt(a, k, flags), oracle.fr_ntt(a, k, flags, 8)), (k, flags)
    # ragged input: zero padding is part of the transform (the reference's resize)
    for in_len in sorted({0, 1, n // 4 + 1, n - 1} & set(range(n + 1))):
        for flags in (0, COSET):
            assert np.array_equal(ctx.fr_ntt(a[:in_len], k, flags), oracle.fr_ntt(a[:in_len], k, flags, 8))


@pytest.mark.parametrize("radix", [4, 8])
@pytest.mark.parametrize("k,tile,maxr", [(12, 12, 10), (16, 11, 10), (16, 12, 8), (20, 11, 10), (20, 12, 10),
                                           (20, 11, 7), (18, 11, 6), (21, 12, 9), (20, 10, 8), (19, 10, 7),
                                           (20, 10, 10), (19, 10, 10), (18, 10, 10)])
def test_plan_options(ctx, oracle, k, tile, maxr, radix):
    """Every kernel family / tile shape / pass split the tunables can select gives the same bits."""
    a = oracle.fr_sample(12 + k, 1 << k)
    ctx.set_option("ntt_tile_log", tile)
    ctx.set_option("ntt_max_radix", maxr)
    ctx.set_option("ntt_radix", radix)
    try:
        for flags in ALL_FLAGS:
            assert np.array_equal(ctx.fr_ntt(a, k, flags), oracle.fr_ntt(a, k, flags, 8))
    finally:
        ctx.set_option("ntt_tile_log", 0)
        ctx.set_option("ntt_max_radix", 10)
        ctx.set_option("ntt_radix", 4)


@pytest.mark.parametrize("k,maxr", [(25, 6), (26, 6), (29, 7)])
def test_four_pass_plans(ctx, k, maxr):
    """ntt_max_radix 6 / 7 on large domains would ask for five or six passes; a plan holds four (the CPU test of the plan
    hook found the overrun).  The four-pass plan and the default three-pass plan give the same bits, both directions."""
    import torch
    n = 1 << k
    g = torch.Generator(device="cuda").manual_seed(77 + k)
    a = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    a[:, 3] &= 0x3FFFFFFFFFFFFFFF
    ref, out = torch.empty_like(a), torch.empty_like(a)
    torch.cuda.synchronize()
    try:
        for flags in (0, INVERSE | COSET):
            ctx.fr_ntt_dev(a.data_ptr(), n, ref.data_ptr(), k, flags)
            ctx.set_option("ntt_max_radix", maxr)
            ctx.fr_ntt_dev(a.data_ptr(), n, out.data_ptr(), k, flags)
            ctx.set_option("ntt_max_radix", 10)
            ctx.sync()
            assert torch.equal(ref, out), flags
    finally:
        ctx.set_option("ntt_max_radix", 10)
        del a, ref, out
        torch.cuda.empty_cache()
        ctx.trim()


def test_xcd_mapping_is_result_neutral(ctx, oracle):
    """blockIdx -> tile placement is a speed knob only."""
    a = oracle.fr_sample(5, 1 << 16)
    ctx.set_option("ntt_xcd", 0)
    try:
        off = [ctx.fr_ntt(a, 16, f) for f in ALL_FLAGS]
    finally:
        ctx.set_option("ntt_xcd", 1)
    for f, x in zip(ALL_FLAGS, off):
        assert np.array_equal(x, ctx.fr_ntt(a, 16, f)) and np.array_equal(x, oracle.fr_ntt(a, 16, f, 8))


@pytest.mark.parametrize("k", [3, 4, 7, 9, 10, 11, 13, 14, 17])
def test_radix8_family(ctx, oracle, k):
    """The radix-8 kernels (non-default) over single- and multi-pass sizes."""
    a = oracle.fr_sample(99 + k, (1 << k) - (1 << k) // 3)
    ctx.set_option("ntt_radix", 8)
    try:
        for flags in ALL_FLAGS:
            assert np.array_equal(ctx.fr_ntt(a, k, flags), oracle.fr_ntt(a, k, flags, 8))
    finally:
        ctx.set_option("ntt_radix", 4)


@pytest.mark.parametrize("k", [2, 7, 10, 13, 20])
def test_in_place(ctx, oracle, k):
    a = oracle.fr_sample(3 + k, 1 << k)
    for flags in ALL_FLAGS:
        buf = a.copy()
        out = ctx.fr_ntt(buf, k, flags, out=buf)     # in == out: the *_in_place forms
        assert out is buf and np.array_equal(buf, oracle.fr_ntt(a, k, flags, 8))


@pytest.mark.parametrize("k,batch,in_len", [(1, 3, 2), (5, 4, 20), (10, 6, 1024), (12, 5, 1000), (16, 4, 1 << 14)])
def test_batch(ctx, oracle, k, batch, in_len):
    a = oracle.fr_sample(1000 + k, batch * in_len).reshape(batch, in_len, 4)
    for flags in ALL_FLAGS:
        got = ctx.fr_ntt_batch(a, k, flags)
        for b in range(batch):
            assert np.array_equal(got[b], oracle.fr_ntt(a[b], k, flags, 8)), (k, b, flags)


def test_errors(ctx):
    import plonk_prototype_amd as pa
    x = np.zeros((8, 4), np.uint64)
    with pytest.raises(pa.Error) as e:
        ctx.fr_ntt(x, 32, 0)
    assert e.value.code == -2                    # log_n >= TWO_ADICITY
    with pytest.raises(pa.Error) as e:
        ctx.fr_ntt(x, 2, 0)
    assert e.value.code == -6                    # input longer than the domain
    with pytest.raises(pa.Error):
        ctx._check(ctx._lib.pm_fr_ntt(ctx._h, x.ctypes.data_as(C.POINTER(C.c_uint64)), 8,
                                       x.ctypes.data_as(C.POINTER(C.c_uint64)), 3, 64))   # unknown flag


def test_evaluation_domain_mirror(ctx, oracle):
    import plonk_prototype_amd as pa
    d = pa.EvaluationDomain(3000, ctx)
    assert d.size == 4096 and d.log_size_of_group == 12
    a = oracle.fr_sample(9, 3000)
    assert np.array_equal(d.fft(a), oracle.fr_ntt(a, 12, 0))
    assert np.array_equal(d.ifft(d.fft(a))[:3000], a) and not d.ifft(d.fft(a))[3000:].any()
    assert np.array_equal(d.coset_ifft(d.coset_fft(a))[:3000], a)
    # elements() == successive powers of group_gen (upstream's `elements_contents` test)
    els = limbs_to_ints(oracle.fr_from_mont(d.elements()))
    assert els == list(B.Domain(4096).elements())
    assert limbs_to_ints(d.group_gen)[0] == B.fr_to_mont(B.Domain(4096).group_gen)


def test_device_resident_api(ctx, oracle):
    import torch
    k, n = 14, 1 << 14
    a = oracle.fr_sample(4, n)
    t_in = torch.from_numpy(a.view(np.int64)).cuda()
    t_out = torch.empty_like(t_in)
    st = torch.cuda.current_stream().cuda_stream
    ctx.fr_ntt_dev(t_in.data_ptr(), n, t_out.data_ptr(), k, 0, stream=st)
    ctx.fr_ntt_dev(t_out.data_ptr(), n, t_out.data_ptr(), k, INVERSE, stream=st)   # in place
    torch.cuda.synchronize()
    assert np.array_equal(t_out.cpu().numpy().view(np.uint64), a)


@pytest.mark.parametrize("k", [22, 24])
def test_full_size_properties(ctx, oracle, k):
    """BASELINE sizes (2^20 is compared to the oracle above; 4n = 2^22 and 2^24 here):
    round trips, delta -> ones, linearity and Horner spot checks."""
    n = 1 << k
    a = oracle.fr_sample(0x504C4F4E4B, n)
    f = ctx.fr_ntt(a, k, 0)
    assert np.array_equal(ctx.fr_ntt(f, k, INVERSE), a)
    cf = ctx.fr_ntt(a[: n // 4], k, COSET)                         # the prover's 4n coset shape
    back = ctx.fr_ntt(cf, k, INVERSE | COSET)
    assert np.array_equal(back[: n // 4], a[: n // 4]) and not back[n // 4:].any()
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))
    ones = ctx.fr_ntt(one, k, 0)
    assert np.array_equal(ones, np.broadcast_to(one, (n, 4)))
    # Horner: NTT(a)[j] == a(w^j) on a short polynomial (cheap on the CPU), coset: a(7 w^j)
    short = a[:257]
    sv = limbs_to_ints(oracle.fr_from_mont(short))
    fs = ctx.fr_ntt(short, k, 0)
    cs = ctx.fr_ntt(short, k, COSET)
    w = B.Domain(n).group_gen
    for j in (0, 1, 12345, n // 2 + 3, n - 1):
        wj = pow(w, j, B.R_MOD)
        assert limbs_to_ints(oracle.fr_from_mont(fs[j:j + 1]))[0] == B.horner(sv, wj)
        assert limbs_to_ints(oracle.fr_from_mont(cs[j:j + 1]))[0] == B.horner(sv, 7 * wj % B.R_MOD)
    # linearity on a sample of outputs: NTT(a + b) == NTT(a) + NTT(b)
    b = oracle.fr_sample(77, n)
    idx = np.array([0, 1, 2, 4097, n // 3, n - 2, n - 1])
    ab = ctx.field_op(1, a, b)
    lhs = ctx.fr_ntt(ab, k, 0)[idx]
    rhs = ctx.field_op(1, np.ascontiguousarray(f[idx]), np.ascontiguousarray(ctx.fr_ntt(b, k, 0)[idx]))
    assert np.array_equal(lhs, rhs)


@pytest.mark.parametrize("log_n,batch,in_len", [(10, 2, 1024), (12, 5, 3000), (16, 3, 65536), (3, 4, 8)])
def test_host_batch_pipeline_matches_the_serial_path(ctx, oracle, log_n, batch, in_len):
    """pm_fr_ntt_batch overlaps upload / transform / download per vector on three streams; the result
    must equal the serial path (option ntt_pipeline = 0) and the oracle, for every flag combination."""
    a = np.stack([oracle.fr_sample(700 + 10 * log_n + b, in_len) for b in range(batch)])
    for flags in (0, 1, 2, 3):
        ctx.set_option("ntt_pipeline", 1)
        piped = ctx.fr_ntt_batch(a, log_n, flags)
        ctx.set_option("ntt_pipeline", 0)
        try:
            serial = ctx.fr_ntt_batch(a, log_n, flags)
        finally:
            ctx.set_option("ntt_pipeline", 1)
        assert np.array_equal(piped, serial)
        for b in range(batch):
            assert np.array_equal(piped[b], oracle.fr_ntt(a[b], log_n, flags, 4)), (flags, b)


@pytest.mark.parametrize("k", [26, 28, 30])
def test_large_domains_device_resident(ctx, oracle, k):
    """Domains well above BASELINE's (2^28 x 32 B = 8.6 GB per vector; EvaluationDomain::new allows up to 2^31): three
    passes, 64-bit element offsets, the two-level twiddle path above 2^26.  Device-resident, checked through
    size-independent properties: inverse(forward(a)) == a, coset round trip, NTT(a)[j] == a(w^j) with the evaluation done
    by the (independent) Horner kernel pm_fr_poly_evaluate_dev, ragged input, delta -> ones."""
    import torch
    n = 1 << k
    g = torch.Generator(device="cuda").manual_seed(1234 + k)
    a = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    a[:, 3] &= 0x3FFFFFFFFFFFFFFF                                  # any value below 2^254 < r is an element
    out = torch.empty_like(a)
    torch.cuda.synchronize()                                       # torch filled `a` on ITS stream; the library uses its own
    ctx.fr_ntt_dev(a.data_ptr(), n, out.data_ptr(), k, 0)
    w = B.Domain(n).group_gen
    for j in (0, 1, n // 2 + 12345, n - 1):
        wj = oracle.fr_to_mont(ints_to_limbs([pow(w, j, B.R_MOD)], 4))[0]
        got = ctx.fr_evaluate(a.data_ptr(), n, wj)
        assert np.array_equal(out[j].cpu().numpy().view(np.uint64), got), j
    ctx.fr_ntt_dev(out.data_ptr(), n, out.data_ptr(), k, INVERSE)   # in place
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    # the prover's shape: n / 4 coefficients onto the coset of size n, and back
    ctx.fr_ntt_dev(a.data_ptr(), n // 4, out.data_ptr(), k, COSET)
    pt = oracle.fr_to_mont(ints_to_limbs([7 * pow(w, 3, B.R_MOD) % B.R_MOD], 4))[0]
    got = ctx.fr_evaluate(a.data_ptr(), n // 4, pt)               # synchronises the context's stream
    assert np.array_equal(out[3].cpu().numpy().view(np.uint64), got)
    ctx.fr_ntt_dev(out.data_ptr(), n, out.data_ptr(), k, INVERSE | COSET)
    torch.cuda.synchronize()
    assert torch.equal(out[: n // 4], a[: n // 4]) and not bool(out[n // 4:].any())
    # delta -> all ones (Montgomery one everywhere)
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))
    d_one = torch.from_numpy(one.view(np.int64)).cuda()
    ctx.fr_ntt_dev(d_one.data_ptr(), 1, out.data_ptr(), k, 0)
    torch.cuda.synchronize()
    assert bool((out == d_one).all())
    del a, out
    torch.cuda.empty_cache()
    assert ctx.trim() >= 2 * n * 36                                # the two pass buffers at least; the shared context stays small
