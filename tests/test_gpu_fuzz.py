"""Seeded random sweeps of the ABI against the oracle: shapes, flags, tunables and scalar/base
structure nobody hand-picked.  Deterministic (fixed seeds), a few seconds on the GPU."""
import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle.cpu_oracle import COSET, INVERSE, SCALAR_CANONICAL, SCALAR_MONTGOMERY, ints_to_limbs, limbs_to_ints

pytestmark = pytest.mark.gpu
# one-off stress runs: PM_FUZZ_SCALE multiplies the number of cases, PM_FUZZ_SEED shifts every seed
import os  # noqa: E402
SCALE = int(os.environ.get("PM_FUZZ_SCALE", "1"))
SEED = int(os.environ.get("PM_FUZZ_SEED", "0"))


def test_ntt_random_shapes(ctx, oracle):
    rng = np.random.default_rng(20261003 + SEED)
    for case in range(80 * SCALE):
        k = int(rng.integers(0, 15))
        n = 1 << k
        in_len = int(rng.integers(0, n + 1))
        flags = int(rng.integers(0, 4))
        a = oracle.fr_sample(1000 + case + 100000 * SEED, n)[:in_len]
        if in_len and rng.random() < 0.3:                      # sparse / structured inputs
            a[rng.integers(0, in_len, size=max(1, in_len // 2))] = 0
        ctx.set_option("ntt_radix", int(rng.choice([4, 8])))
        ctx.set_option("ntt_max_radix", int(rng.integers(6, 11)))
        try:
            got = ctx.fr_ntt(a, k, flags)
        finally:
            ctx.set_option("ntt_radix", 4)
            ctx.set_option("ntt_max_radix", 10)
        assert np.array_equal(got, oracle.fr_ntt(a, k, flags, 4)), (case, k, in_len, flags)


def test_ntt_extreme_values(ctx, oracle):
    """Inputs at the edges of the canonical range (0, 1, r-1, r-2, 2^255 mod r, ...) in every slot."""
    k, n = 11, 1 << 11
    edge = [0, 1, 2, B.R_MOD - 1, B.R_MOD - 2, (1 << 255) % B.R_MOD, (1 << 254), B.R_MOD // 2, B.R_MOD // 2 + 1]
    vals = [edge[(i * 7 + i // 5) % len(edge)] for i in range(n)]
    a = oracle.fr_to_mont(ints_to_limbs(vals, 4))
    raw = ints_to_limbs(vals, 4)                               # the same limbs read as Montgomery data
    for x in (a, raw):
        for flags in (0, INVERSE, COSET, INVERSE | COSET):
            assert np.array_equal(ctx.fr_ntt(x, k, flags), oracle.fr_ntt(x, k, flags, 4))
    top = oracle.fr_to_mont(ints_to_limbs([B.R_MOD - 1] * n, 4))   # every element r-1
    assert np.array_equal(ctx.fr_ntt(top, k, 0), oracle.fr_ntt(top, k, 0, 4))


def test_msm_random_structure(ctx, oracle):
    import plonk_prototype_amd as pa
    rng = np.random.default_rng(777 + SEED)
    k0 = ints_to_limbs([0x77777], 4)[0]
    dd = ints_to_limbs([0x123456789abcdef0123], 4)[0]
    pool = oracle.g1_bases_arith(k0, dd, 4096, 8)
    one = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
    for case in range(40 * SCALE):
        n = int(rng.integers(1, 3500))
        pts = pool[rng.integers(0, 4096 if rng.random() < 0.6 else 8, size=n)].copy()   # many duplicates sometimes
        sc = oracle.fr_sample(5000 + case + 100000 * SEED, n)
        mode = case % 5
        if mode == 1:
            sc[rng.random(n) < 0.5] = 0
            sc[rng.random(n) < 0.3] = one
        elif mode == 2:
            small = [int(v) for v in rng.integers(0, 1 << 16, size=n)]
            sc = oracle.fr_to_mont(ints_to_limbs(small, 4))
        elif mode == 3:
            sc[:] = sc[0]
        elif mode == 4:                                        # negated copies and points at infinity
            neg = pts.copy()
            ys = limbs_to_ints(oracle.fp_from_mont(np.ascontiguousarray(pts[:, 6:])))
            neg[:, 6:] = oracle.fp_to_mont(ints_to_limbs([(B.P_MOD - y) % B.P_MOD for y in ys], 6))
            flip = rng.random(n) < 0.4
            pts[flip] = neg[flip]
            pts[rng.random(n) < 0.1] = 0
        exp = oracle.g1_msm(pts, sc, SCALAR_MONTGOMERY, 8)
        c = int(rng.choice([0, 6, 9, 12, 15]))
        ctx.set_option("msm_window_bits", c)
        ctx.set_option("msm_chunk", int(rng.choice([0, 16, 32, 64])))
        try:
            bases = pa.host.Bases(ctx, pts)
            got, _ = pa.g1_to_affine(bases.msm(sc))
            assert np.array_equal(got, exp), (case, n, mode, c)
            if case % 3 == 0:
                bases.precompute(int(rng.choice([0, 9, 14])))
                got, _ = pa.g1_to_affine(bases.msm(oracle.fr_from_mont(sc), SCALAR_CANONICAL))
                assert np.array_equal(got, exp), (case, n, mode, "table")
            bases.free()
        finally:
            ctx.set_option("msm_window_bits", 0)
            ctx.set_option("msm_chunk", 0)


def test_field_limb_patterns(ctx, oracle):
    """Saturated limb patterns (all-ones words, alternating words) that stress the 29/28-bit re-limbing."""
    pats = []
    for mod, nl in ((B.R_MOD, 4), (B.P_MOD, 6)):
        words = [0xFFFFFFFFFFFFFFFF, 0, 0xAAAAAAAAAAAAAAAA, 0x5555555555555555, 0x00000000FFFFFFFF, 0xFFFFFFFF00000000,
                 0x1FFFFFFF1FFFFFFF, 0xE0000000E0000000]
        vals = []
        for w in words:
            for sh in range(nl):
                v = sum(w << (64 * j) for j in range(nl) if j != sh) % mod
                vals.append(v)
        pats.append((mod, nl, vals))
    for (mod, nl, vals), base in zip(pats, (0, 3)):
        a = ints_to_limbs(vals, nl)
        b = ints_to_limbs(list(reversed(vals)), nl)
        fm = oracle.fr_from_mont if nl == 4 else oracle.fp_from_mont
        tm = oracle.fr_to_mont if nl == 4 else oracle.fp_to_mont
        am, bm = tm(a), tm(b)
        for op, fn in ((0, lambda x, y: x * y % mod), (1, lambda x, y: (x + y) % mod), (2, lambda x, y: (x - y) % mod)):
            got = limbs_to_ints(fm(ctx.field_op(base + op, am, bm)))
            assert got == [fn(x, y) for x, y in zip(vals, reversed(vals))], (nl, op)
            got2 = ctx.field_op(base + op, a, b)               # raw limbs as Montgomery residues
            exp2 = tm(ints_to_limbs([fn(x, y) for x, y in zip(limbs_to_ints(fm(a)), limbs_to_ints(fm(b)))], nl))
            assert np.array_equal(got2, exp2), (nl, op, "raw")


def test_prover_round_kernels_random_shapes(ctx, oracle):
    """powers / lincomb / permutation terms on random lengths, with edge values salted in."""
    import ctypes as C
    import random
    import plonk_prototype_amd as pa
    from oracle import plonk_rounds_oracle as PO
    from plonk_prototype_amd import _lib
    R = B.R_MOD
    rng = random.Random(424242 + SEED)
    edge = [0, 1, R - 1, R - 2, (1 << 255) % R, R // 2]

    def rand_vec(n):
        return [rng.choice(edge) if rng.random() < 0.1 else rng.randrange(R) for _ in range(n)]

    def dev(vals):
        return pa.DeviceVector.from_host(ctx, oracle.fr_to_mont(ints_to_limbs(vals, 4)))

    def ints(dv):
        return limbs_to_ints(oracle.fr_from_mont(dv.to_host()))

    def mont(v):
        return oracle.fr_to_mont(ints_to_limbs([v % R], 4))[0]

    for case in range(25 * SCALE):
        n = rng.randrange(1, 2500)
        base, scale = rng.choice(edge + [rng.randrange(R)]), rng.choice(edge + [rng.randrange(R)])
        out = pa.DeviceVector(ctx, n)
        ctx.fr_powers(mont(base), mont(scale), n, out.ptr)
        assert ints(out) == PO.powers(base, scale, n), ("powers", case, n)
        k = rng.randrange(1, 17)
        vecs, coeffs = [rand_vec(n) for _ in range(k)], rand_vec(k)
        dv = [dev(v) for v in vecs]
        ctx.fr_lincomb([d.ptr for d in dv], oracle.fr_to_mont(ints_to_limbs(coeffs, 4)), n, out.ptr)
        assert ints(out) == PO.lincomb(coeffs, vecs), ("lincomb", case, n, k)
        wires, sigmas, roots = [rand_vec(n) for _ in range(4)], [rand_vec(n) for _ in range(4)], rand_vec(n)
        beta, gamma = rng.choice(edge + [rng.randrange(R)]), rng.randrange(R)
        dw, ds, dr = [dev(v) for v in wires], [dev(v) for v in sigmas], dev(roots)
        a = _lib.PermArgs()
        for j in range(4):
            a.wires[j], a.sigmas[j] = dw[j].ptr, ds[j].ptr
        a.roots = dr.ptr
        u = lambda v: (C.c_uint64 * 4)(*[int(x) for x in mont(v)])   # noqa: E731
        a.beta, a.gamma = u(beta), u(gamma)
        for j, kk in enumerate((7, 13, 17)):
            a.k[j] = u(kk)
        num, den = pa.DeviceVector(ctx, n), pa.DeviceVector(ctx, n)
        ctx.plonk_perm_terms(a, n, num.ptr, den.ptr)
        en, ed = PO.perm_terms(wires, sigmas, roots, beta, gamma)
        assert ints(num) == en and ints(den) == ed, ("perm", case, n)


def test_poly_helpers_random_shapes(ctx, oracle):
    """Ruffini (scaled prefix sum: tiles of 2048, tile carries inside the replay kernel up to 512 tiles), batch inversion
    (quad-blocked, several quads per thread forced through option binv_quads, random zero patterns) and prefix product (zero
    factors, all three scan modes) on random lengths around the tile / chunk / quad boundaries, against the C oracle."""
    import plonk_prototype_amd as pa
    rng = np.random.default_rng(6060842 + SEED)
    edge = oracle.fr_to_mont(ints_to_limbs([0, 1, 2, B.R_MOD - 1, B.R_MOD - 2, (1 << 255) % B.R_MOD], 4))
    near = [1, 2, 3, 4, 5, 7, 8, 9, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6145]
    try:
        for case in range(40 * SCALE):
            n = int(rng.choice(near)) if rng.random() < 0.4 else int(rng.integers(1, 70000))
            a = oracle.fr_sample(7000 + case + 100000 * SEED, n)
            salt = rng.random(n) < rng.choice([0.0, 0.01, 0.3])
            a[salt] = edge[rng.integers(0, len(edge), size=int(salt.sum()))]
            # Ruffini at a random, a tiny and a huge point
            z = [oracle.fr_sample(9000 + case, 1)[0], edge[int(rng.integers(0, len(edge)))]][int(rng.integers(0, 2))]
            q = pa.Polynomial.from_host(ctx, a).ruffini(z).to_host()
            assert np.array_equal(q, oracle.fr_poly_ruffini(a, z)), ("ruffini", case, n)
            # batch inversion: zeros stay, everything else inverts; the quads per thread forced
            quads = int(rng.choice([0, 1, 2, 3, 5, 16]))
            ctx.set_option("binv_quads", quads)
            got = pa.Polynomial.from_host(ctx, a).batch_inverse().to_host()
            assert np.array_equal(got, oracle.fr_batch_inverse(a)), ("batch_inverse", case, n, quads)
            ctx.set_option("binv_quads", 0)
            # prefix product
            ctx.set_option("poly_lookback", int(rng.integers(0, 3)))
            assert np.array_equal(pa.Polynomial.from_host(ctx, a).prefix_product().to_host(), oracle.fr_prefix_product(a)), ("prefix", case, n)
            ctx.set_option("poly_lookback", 1)
    finally:
        ctx.set_option("binv_quads", 0)
        ctx.set_option("poly_lookback", 1)


def test_prove_many_seeds(ctx, oracle):
    """Several circuits (arithmetic-only and with every widget) and sizes: every proof must satisfy the
    verifier identity, the serialised proof must round-trip, and a second proof from the same key
    (workspace reuse) must be identical."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    srs = oracle.g1_bases_arith(ints_to_limbs([11], 4)[0], ints_to_limbs([0x10001], 4)[0], 2048, 8)
    for seed, n in [(1 + SEED, 8), (2 + SEED, 32), (3 + SEED, 128), (4 + SEED, 512), (5 + SEED, 2048)] * min(SCALE, 4):
        gen = pa.synthetic.mixed_circuit if (n >= 32 and seed % 2) else pa.synthetic.chain_circuit
        circuit, wit, pub = gen(n, seed)
        ck = pa.CommitKey(srs[:n], ctx, precompute=(seed % 2 == 0))
        pk = PR.preprocess(circuit, ctx, ck)
        proof = PR.prove(pk, ck, wit, pub)
        pub_c = oracle.fr_ntt(pub, n.bit_length() - 1, INVERSE)
        pub_z = limbs_to_ints(oracle.fr_from_mont(
            oracle.fr_poly_evaluate(pub_c, oracle.fr_to_mont(ints_to_limbs([proof.challenges["z"]], 4))[0]).reshape(1, 4)))[0]
        assert PR.check_identity(proof, n, pub_z), (seed, n)
        blob = proof.to_bytes()
        assert PR.Proof.from_bytes(blob).to_bytes() == blob and proof.native_bytes == blob
        d_w = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
        assert PR.prove(pk, ck, d_w, PR.sparse_public_inputs(pub)).to_bytes() == blob
        pk.free()
