"""GPU parity of the prover-round kernels and of the five-round pipeline (SURVEY.md section 8f rows
N1 / N2, BASELINE.json configs[3]) against oracle/plonk_rounds_oracle.py -- bit-exact -- plus
oracle-independent checks: the verifier's scalar identity, and the KZG opening equations in the
exponent with a trapdoor SRS."""
import random

import numpy as np
import pytest

from oracle import bigint_oracle as B
from oracle import plonk_rounds_oracle as PO
from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints

pytestmark = pytest.mark.gpu
R = B.R_MOD
TAU = 0x2B7E151628AED2A6ABF7158809CF4F3C762E7160F38B4DA56A784D9045190CFE % R


def _to_dev(pa, ctx, oracle, vals):
    return pa.DeviceVector.from_host(ctx, oracle.fr_to_mont(ints_to_limbs(vals, 4)))


def _ints(oracle, limbs):
    return limbs_to_ints(oracle.fr_from_mont(np.ascontiguousarray(limbs).reshape(-1, 4)))


def _mont(oracle, v):
    return oracle.fr_to_mont(ints_to_limbs([v % R], 4))[0]


def _rand(rng, n):
    return [rng.randrange(R) for _ in range(n)]


@pytest.mark.parametrize("n", [1, 2, 255, 256, 4097, 70001])
def test_powers(ctx, oracle, n):
    import plonk_prototype_amd as pa
    rng = random.Random(n)
    base, scale = rng.randrange(R), rng.randrange(R)
    out = pa.DeviceVector(ctx, n)
    ctx.fr_powers(_mont(oracle, base), _mont(oracle, scale), n, out.ptr)
    assert _ints(oracle, out.to_host()) == PO.powers(base, scale, n)


def test_powers_gives_the_domain(ctx, oracle):
    import plonk_prototype_amd as pa
    dom = pa.EvaluationDomain(1 << 12, ctx)
    out = pa.DeviceVector(ctx, 1 << 12)
    ctx.fr_powers(dom.group_gen, _mont(oracle, 1), 1 << 12, out.ptr)
    assert np.array_equal(out.to_host(), dom.elements())


@pytest.mark.parametrize("k,n", [(1, 100), (2, 1), (8, 1000), (16, 5000)])
def test_lincomb(ctx, oracle, k, n):
    import plonk_prototype_amd as pa
    rng = random.Random(100 * k + n)
    vecs = [_rand(rng, n) for _ in range(k)]
    coeffs = _rand(rng, k)
    if k > 2:
        coeffs[1], coeffs[2] = 0, R - 1
    dv = [_to_dev(pa, ctx, oracle, v) for v in vecs]
    out = pa.DeviceVector(ctx, n)
    ctx.fr_lincomb([d.ptr for d in dv], oracle.fr_to_mont(ints_to_limbs(coeffs, 4)), n, out.ptr)
    assert _ints(oracle, out.to_host()) == PO.lincomb(coeffs, vecs)


def test_lincomb_rejects_bad_k(ctx):
    import plonk_prototype_amd as pa
    v = pa.DeviceVector(ctx, 4)
    with pytest.raises(pa.Error) as e:
        ctx.fr_lincomb([v.ptr] * 17, np.zeros((17, 4), np.uint64), 4, v.ptr)
    assert e.value.code == -1


def _perm_args(pa, oracle, wires, sigmas, roots, beta, gamma):
    from plonk_prototype_amd import _lib
    a = _lib.PermArgs()
    for j in range(4):
        a.wires[j], a.sigmas[j] = wires[j].ptr, sigmas[j].ptr
    a.roots = roots.ptr
    u = lambda v: (pa.prover.C.c_uint64 * 4)(*[int(x) for x in _mont(oracle, v)])   # noqa: E731
    a.beta, a.gamma = u(beta), u(gamma)
    for j, k in enumerate((7, 13, 17)):
        a.k[j] = u(k)
    return a


@pytest.mark.parametrize("n", [1, 300, 8192])
def test_perm_terms(ctx, oracle, n):
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover  # noqa: F401
    rng = random.Random(n)
    wires, sigmas, roots = [_rand(rng, n) for _ in range(4)], [_rand(rng, n) for _ in range(4)], _rand(rng, n)
    wires[0][0], sigmas[1][0] = 0, R - 1
    beta, gamma = rng.randrange(R), rng.randrange(R)
    dw, ds = [_to_dev(pa, ctx, oracle, v) for v in wires], [_to_dev(pa, ctx, oracle, v) for v in sigmas]
    dr = _to_dev(pa, ctx, oracle, roots)
    num, den = pa.DeviceVector(ctx, n), pa.DeviceVector(ctx, n)
    ctx.plonk_perm_terms(_perm_args(pa, oracle, dw, ds, dr, beta, gamma), n, num.ptr, den.ptr)
    en, ed = PO.perm_terms(wires, sigmas, roots, beta, gamma)
    assert _ints(oracle, num.to_host()) == en
    assert _ints(oracle, den.to_host()) == ed


@pytest.mark.parametrize("n", [4, 64, 1024])
def test_quotient_kernel(ctx, oracle, n):
    """Random (unsatisfied) inputs on the true 4n coset: the kernel is a pointwise map, so parity needs
    no valid circuit."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd import _lib
    rng = random.Random(n)
    n4 = 4 * n
    names = ["w0", "w1", "w2", "w3", "z", "q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "pi", "s0", "s1", "s2", "s3", "l1"]
    v = {k: _rand(rng, n4) for k in names}
    v["q_m"][0], v["z"][1], v["w0"][2] = 0, 1, R - 1
    x = PO.powers(B.Domain(n4).group_gen, 7, n4)
    alpha, beta, gamma = (rng.randrange(R) for _ in range(3))
    d = {k: _to_dev(pa, ctx, oracle, val) for k, val in v.items()}
    dx = _to_dev(pa, ctx, oracle, x)
    qa = _lib.QuotientArgs()
    u = lambda s: (PR.C.c_uint64 * 4)(*[int(t) for t in _mont(oracle, s)])   # noqa: E731
    for j in range(4):
        qa.wires[j], qa.sigmas[j] = d[f"w{j}"].ptr, d[f"s{j}"].ptr
    for k in ("z", "q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "pi", "l1"):
        setattr(qa, k, d[k].ptr)
    qa.x = dx.ptr
    qa.alpha, qa.beta, qa.gamma = u(alpha), u(beta), u(gamma)
    for j, k in enumerate((7, 13, 17)):
        qa.k[j] = u(k)
    for j in range(4):
        qa.zh_inv[j] = u(PO.inv(pow(x[j], n, R) - 1))
    out = pa.DeviceVector(ctx, n4)
    ctx.plonk_quotient(qa, n, out.ptr)
    exp = PO.quotient_evals(n, [v[f"w{j}"] for j in range(4)], v["z"],
                            {k: v[k] for k in ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")}, v["pi"],
                            [v[f"s{j}"] for j in range(4)], v["l1"], x, alpha, beta, gamma)
    assert _ints(oracle, out.to_host()) == exp


def _srs(oracle, n):
    G = oracle.g1_generator()
    out = np.zeros((n, 12), np.uint64)
    t = 1
    for i in range(n):
        out[i] = oracle.g1_mul(G, ints_to_limbs([t], 4)[0])
        t = t * TAU % R
    return out


def _g(oracle, k):
    return oracle.g1_mul(oracle.g1_generator(), ints_to_limbs([k % R], 4)[0])


def _setup(ctx, oracle, n, seed):
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.field import fr_vec_from_limbs
    from plonk_prototype_amd.synthetic import chain_circuit
    circuit, wit, pi = chain_circuit(n, seed)
    srs = _srs(oracle, n)
    ck = pa.CommitKey(srs, ctx, precompute=(n >= 64))
    pk = pa.prover.preprocess(circuit, ctx)
    sel = {k: fr_vec_from_limbs(getattr(circuit, k)) for k in ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")}
    ints = (sel, circuit.sigma_index.tolist(), [fr_vec_from_limbs(wit[j]) for j in range(4)], fr_vec_from_limbs(pi))
    return circuit, wit, pi, srs, ck, pk, ints


@pytest.mark.parametrize("n", [4, 16, 256])
def test_prove_matches_the_oracle(ctx, oracle, n):
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    circuit, wit, pi, srs, ck, pk, (sel, sigma, wi, pii) = _setup(ctx, oracle, n, seed=n)
    proof = PR.prove(pk, ck, wit, pi)
    exp = PO.prove(n, sel, sigma, wi, pii, proof.challenges)
    # openings
    got = {k: _ints(oracle, v)[0] for k, v in proof.evaluations.items()}
    assert got == exp["evals"]
    # commitments: the oracle's Pippenger over the oracle's coefficient vectors
    def commit(c):
        return oracle.g1_msm(srs[:len(c)], oracle.fr_to_mont(ints_to_limbs(c, 4)))
    want = {nm: commit(exp["wire_coeffs"][j]) for j, nm in enumerate("abcd")}
    want["z"] = commit(exp["z_coeffs"])
    for i in range(4):
        want[f"t_{i + 1}"] = commit(exp["t_coeffs"][i * n:(i + 1) * n])
    want["w_z"], want["w_zw"] = commit(exp["w_z"]), commit(exp["w_zw"])
    assert set(want) == set(proof.commitments)
    for k in want:
        assert np.array_equal(proof.commitments[k], want[k]), k
    # oracle-independent: the verifier's scalar identity ...
    pi_z = B.horner(B.ifft(pii, n.bit_length() - 1), proof.challenges["z"])
    assert PR.check_identity(proof, n, pi_z)
    # ... and both KZG opening equations in the exponent (tau is known to the test):
    #   W(tau) (tau - z) = F(tau) - F(z)   with commitments checked as  [p(tau)] G
    ch, ev = proof.challenges, got
    zz, v = ch["z"], ch["v"]
    zn = pow(zz, n, R)
    agg_eval = (ev["t"] + v * ev["r"] + v ** 2 * ev["a"] + v ** 3 * ev["b"] + v ** 4 * ev["c"] + v ** 5 * ev["d"]
                + v ** 6 * ev["sigma_1"] + v ** 7 * ev["sigma_2"] + v ** 8 * ev["sigma_3"]) % R
    tau_of = lambda c: B.horner(c, TAU)   # noqa: E731
    parts = [exp["t_coeffs"][i * n:(i + 1) * n] for i in range(4)]
    agg_tau = (sum(pow(zn, i, R) * tau_of(parts[i]) for i in range(4)) + v * tau_of(exp["r_coeffs"])
               + sum(pow(v, 2 + j, R) * tau_of(exp["wire_coeffs"][j]) for j in range(4))) % R
    sig_c = [B.ifft([PO.K[p // n] * pow(B.Domain(n).group_gen, p % n, R) % R for p in sigma[j]], n.bit_length() - 1)
             for j in range(3)]
    agg_tau = (agg_tau + sum(pow(v, 6 + j, R) * tau_of(sig_c[j]) for j in range(3))) % R
    assert np.array_equal(proof.commitments["w_z"], _g(oracle, (agg_tau - agg_eval) * PO.inv(TAU - zz)))
    zw = zz * B.Domain(n).group_gen % R
    assert np.array_equal(proof.commitments["w_zw"],
                          _g(oracle, (tau_of(exp["z_coeffs"]) - ev["z_next"]) * PO.inv(TAU - zw)))


def test_prove_is_deterministic_and_transcript_bound(ctx, oracle):
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.transcript import Transcript
    n = 64
    circuit, wit, pi, srs, ck, pk, _ = _setup(ctx, oracle, n, seed=9)
    p1, p2 = PR.prove(pk, ck, wit, pi), PR.prove(pk, ck, wit, pi)
    assert p1.challenges == p2.challenges
    assert all(np.array_equal(p1.commitments[k], p2.commitments[k]) for k in p1.commitments)
    p3 = PR.prove(pk, ck, wit, pi, transcript=Transcript(b"another protocol"))
    assert p3.challenges["beta"] != p1.challenges["beta"]
    assert np.array_equal(p3.commitments["a"], p1.commitments["a"])          # round 1 has no challenge
    assert not np.array_equal(p3.commitments["z"], p1.commitments["z"])


def test_tampered_witness_fails_the_identity(ctx, oracle):
    import plonk_prototype_amd.prover as PR
    n = 64
    circuit, wit, pi, srs, ck, pk, (sel, sigma, wi, pii) = _setup(ctx, oracle, n, seed=4)
    bad = wit.copy()
    bad[2, 7] = bad[2, 8]
    proof = PR.prove(pk, ck, bad, pi)
    pi_z = B.horner(B.ifft(pii, 6), proof.challenges["z"])
    assert not PR.check_identity(proof, n, pi_z)
    good = PR.prove(pk, ck, wit, pi)
    assert PR.check_identity(good, n, B.horner(B.ifft(pii, 6), good.challenges["z"]))


def test_prover_key_rejects_bad_circuits(ctx, oracle):
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.synthetic import chain_circuit
    circuit, wit, pi = chain_circuit(16, 1)
    circuit.sigma_index = circuit.sigma_index.copy()
    circuit.sigma_index[0, 0] = circuit.sigma_index[0, 1]                      # not a permutation
    with pytest.raises(ValueError):
        PR.preprocess(circuit, ctx)


def test_prove_2_16_gates(ctx, oracle):
    """A mid-size run (4n = 2^18 coset): scalar identity, and commitments against the known discrete
    logs of an arithmetic-progression SRS (no 2^16 scalar multiplications on the CPU)."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.synthetic import chain_circuit
    n = 1 << 16
    circuit, wit, pi = chain_circuit(n, 11)
    k0, d = ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0]
    srs = oracle.g1_bases_arith(k0, d, n, threads=8)
    ck = pa.CommitKey(srs, ctx, precompute=True)
    pk = PR.preprocess(circuit, ctx)
    proof = PR.prove(pk, ck, wit, pi)
    pi_coeffs = oracle.fr_ntt(pi, 16, 1)
    pi_z = _ints(oracle, oracle.fr_poly_evaluate(pi_coeffs, _mont(oracle, proof.challenges["z"])))[0]
    assert PR.check_identity(proof, n, pi_z)
    # commitment to wire a: sum_i a_i (k0 + i d) G
    a_coeffs = oracle.fr_ntt(wit[0], 16, 1)
    dlog = oracle.expected_dlog(a_coeffs, 0, k0, d)
    assert np.array_equal(proof.commitments["a"], oracle.g1_mul(oracle.g1_generator(), dlog))


def test_wide_circuit_is_satisfied_and_proves(ctx, oracle):
    """The GPU-assisted generator used for the large runs: its witness satisfies every gate and copy
    constraint (checked with big-int arithmetic), and the proof passes the verifier identity."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    n = 1 << 10
    circuit, d_wit, _ = pa.synthetic.wide_circuit(n, ctx, seed=3)
    w = _ints(oracle, d_wit.to_host())
    wit = [w[j * n:(j + 1) * n] for j in range(4)]
    sel = {k: _ints(oracle, getattr(circuit, k)) for k in ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c")}
    for i in range(n):
        a, b, c, d = (wit[j][i] for j in range(4))
        assert (sel["q_m"][i] * a * b + sel["q_l"][i] * a + sel["q_r"][i] * b + sel["q_o"][i] * c
                + sel["q_4"][i] * d + sel["q_c"][i]) % R == 0
    sig = circuit.sigma_index.reshape(-1)
    assert sorted(sig.tolist()) == list(range(4 * n)) and (sig != np.arange(4 * n)).sum() > 2 * n
    assert all(w[p] == w[sig[p]] for p in range(4 * n))
    srs = oracle.g1_bases_arith(ints_to_limbs([3], 4)[0], ints_to_limbs([5], 4)[0], n, 4)
    proof = PR.prove(PR.preprocess(circuit, ctx), pa.CommitKey(srs, ctx), d_wit, None)
    assert PR.check_identity(proof, n, 0)
    assert PR.Proof.from_bytes(proof.to_bytes()).to_bytes() == proof.to_bytes()


def _pt(oracle, xy):
    """affine Montgomery limbs [12] -> (x, y) ints or None."""
    if not np.asarray(xy).any():
        return None
    v = limbs_to_ints(oracle.fp_from_mont(np.ascontiguousarray(xy).reshape(2, 6)))
    return (v[0], v[1])


@pytest.mark.parametrize("n", [16, 128])
def test_gpu_proof_passes_the_pairing_verifier(ctx, oracle, n):
    """End to end without the prover-side oracle: SRS generated on the GPU, proof made on the GPU,
    challenges re-derived from the proof bytes by replaying the transcript, then the verifier's
    scalar identity and the KZG pairing equation (oracle/plonk_verifier_oracle.py, plain-Python
    pairing).  Tampered proofs must fail."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from oracle import pairing_oracle as PG
    from oracle import plonk_verifier_oracle as PV
    tau = TAU
    circuit, wit, pub = pa.synthetic.chain_circuit(n, 77 + n)
    ck = pa.CommitKey.setup(n - 1, _mont(oracle, tau), ctx, precompute=(n > 16))
    pk = PR.preprocess(circuit, ctx)
    proof = PR.Proof.from_bytes(PR.prove(pk, ck, wit, pub).to_bytes())        # what a verifier receives
    ch = PR.derive_challenges(proof, n)
    vk = {k: _pt(oracle, v) for k, v in PR.verifier_key(pk, ck).items()}
    comms = {k: _pt(oracle, v) for k, v in proof.commitments.items()}
    ev = {k: _ints(oracle, v)[0] for k, v in proof.evaluations.items()}
    pub_z = B.horner(B.ifft(_ints(oracle, pub), n.bit_length() - 1), ch["z"])
    tau_g2 = PG.g2_mul(tau, PG.G2_GEN)
    assert PV.verify(n, vk, comms, ev, ch, pub_z, tau_g2) == (True, True)
    # the prover derived the same challenges
    assert ch == PR.prove(pk, ck, wit, pub).challenges
    if n == 16:
        bad = dict(ev, c=(ev["c"] + 1) % R)
        assert PV.verify(n, vk, comms, bad, ch, pub_z, tau_g2) == (False, False)
        swapped = dict(comms, t_1=comms["t_2"], t_2=comms["t_1"])
        assert PV.verify(n, vk, swapped, ev, ch, pub_z, tau_g2)[1] is False
        # a proof for a different witness does not verify against tampered public inputs
        assert PV.verify(n, vk, comms, ev, ch, (pub_z + 1) % R, tau_g2)[0] is False


@pytest.mark.parametrize("log_n", [10, 14])
def test_prove_matches_the_c_prover(ctx, oracle, log_n):
    """Sizes beyond the big-int oracle: every commitment and evaluation of the GPU proof equals the
    CPU prover composed from the C restatement (oracle/cpu_prover.py), same challenges."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from oracle import cpu_prover as CP
    n = 1 << log_n
    circuit, wit, pub = pa.synthetic.chain_circuit(n, 100 + log_n)
    srs = oracle.g1_bases_arith(ints_to_limbs([0xA5A5], 4)[0], ints_to_limbs([0x7FFFFFFF], 4)[0], n, 8)
    proof = PR.prove(PR.preprocess(circuit, ctx), pa.CommitKey(srs, ctx, precompute=True), wit, pub)
    cpk = CP.preprocess(oracle, {k: getattr(circuit, k) for k in CP.SELECTORS}, circuit.sigma_index, threads=8)
    exp = CP.prove(oracle, cpk, srs, wit, pub, proof.challenges, threads=8)
    assert set(exp["commitments"]) == set(proof.commitments) and set(exp["evaluations"]) == set(proof.evaluations)
    for k, v in exp["evaluations"].items():
        assert np.array_equal(proof.evaluations[k], v), k
    for k, v in exp["commitments"].items():
        assert np.array_equal(proof.commitments[k], v), k


def test_round_kernels_empty_and_bad_arguments(ctx, oracle):
    """n == 0 is a no-op for every entry point; null pointers and bad sizes are reported, not executed."""
    import ctypes as C
    import plonk_prototype_amd as pa
    from plonk_prototype_amd import _lib
    lib, h = ctx._lib, ctx._h
    one = _mont(oracle, 1)
    u64p = C.POINTER(C.c_uint64)
    p1 = one.ctypes.data_as(u64p)
    v = pa.DeviceVector(ctx, 8)
    assert lib.pm_fr_powers_dev(h, p1, p1, 0, None, None) == 0
    assert lib.pm_fr_powers_dev(h, p1, p1, 4, None, None) == -1
    ptrs = (C.c_void_p * 1)(v._p)
    assert lib.pm_fr_lincomb_dev(h, 1, ptrs, p1, 0, None, None) == 0
    assert lib.pm_fr_lincomb_dev(h, 0, ptrs, p1, 4, v._p, None) == -1
    assert lib.pm_fr_lincomb_dev(h, 1, (C.c_void_p * 1)(None), p1, 4, v._p, None) == -1
    pargs = _lib.PermArgs()
    assert lib.pm_plonk_perm_terms_dev(h, C.byref(pargs), 0, None, None, None) == 0
    assert lib.pm_plonk_perm_terms_dev(h, C.byref(pargs), 4, v._p, v._p, None) == -1     # null wire pointers
    assert lib.pm_plonk_perm_terms_dev(h, None, 4, v._p, v._p, None) == -1
    qargs = _lib.QuotientArgs()
    assert lib.pm_plonk_quotient_dev(h, C.byref(qargs), 0, None, None) == 0
    assert lib.pm_plonk_quotient_dev(h, C.byref(qargs), 2, v._p, None) == -1             # null operands
    assert lib.pm_plonk_quotient_dev(h, C.byref(qargs), 3, v._p, None) == -6             # not a power of two
    G = oracle.g1_generator()
    assert lib.pm_g1_fixed_base_mul_dev(h, G.ctypes.data_as(u64p), None, 0, 0, None, None) == 0
    assert lib.pm_g1_fixed_base_mul_dev(h, G.ctypes.data_as(u64p), None, 4, 0, v._p, None) == -1
    assert lib.pm_g1_fixed_base_mul_dev(h, G.ctypes.data_as(u64p), v._p, 2, 7, v._p, None) == -1   # scalar_form
    assert b"" != lib.pm_last_error(h)
    out = C.c_void_p()
    assert lib.pm_g1_bases_from_dev(h, None, 0, C.byref(out)) == 0 and lib.pm_g1_bases_len(out) == 0
    lib.pm_g1_bases_free(h, out)


@pytest.mark.parametrize("n,with_pi", [(4, True), (64, True), (1024, False), (1 << 14, True)])
def test_native_prover_equals_the_python_sequence(ctx, oracle, n, with_pi):
    """pm_plonk_prove (rounds and Merlin transcript in C++ inside the library) must return the very
    proof prover.prove() returns -- commitments, evaluations and challenges."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    circuit, wit, pub = pa.synthetic.chain_circuit(n, 500 + n)
    srs = oracle.g1_bases_arith(ints_to_limbs([0xBEEF], 4)[0], ints_to_limbs([0x10000001], 4)[0], n, 8)
    ck = pa.CommitKey(srs, ctx, precompute=(n >= 64))
    d_wit = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
    d_pub = pa.DeviceVector.from_host(ctx, pub) if with_pi else None
    ref = PR.prove(PR.preprocess(circuit, ctx), ck, d_wit, d_pub)
    npk = PR.NativeProverKey(circuit, ctx)
    got = PR.prove_native(npk, ck, d_wit, d_pub)
    assert got.challenges == ref.challenges
    assert got.to_bytes() == ref.to_bytes()
    again = PR.prove_native(npk, ck, d_wit, d_pub)                  # workspace reuse
    assert again.to_bytes() == ref.to_bytes()
    other = PR.prove_native(npk, ck, d_wit, d_pub, label=b"another protocol")
    assert other.challenges["beta"] != ref.challenges["beta"]
    npk.free()


def test_native_prover_rejects_bad_input(ctx, oracle):
    import ctypes as C
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    circuit, wit, pub = pa.synthetic.chain_circuit(16, 3)
    bad = PR.Circuit(**{**circuit.__dict__, "sigma_index": circuit.sigma_index.copy()})
    bad.sigma_index[0, 0] = bad.sigma_index[0, 1]
    with pytest.raises(pa.Error):
        PR.NativeProverKey(bad, ctx)
    npk = PR.NativeProverKey(circuit, ctx)
    short = pa.CommitKey(oracle.g1_bases_arith(ints_to_limbs([1], 4)[0], ints_to_limbs([1], 4)[0], 8, 1), ctx)
    with pytest.raises(pa.Error) as e:
        PR.prove_native(npk, short, pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4)))
    assert e.value.code == -6
    npk.free()
