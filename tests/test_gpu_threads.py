"""Host threads against the C ABI (include/plonk_mi355x.h: "calls on one ctx are serialised by an internal mutex;
use one ctx per thread").  ctypes drops the GIL for the duration of a call, so the threads below really are inside the
library at the same time: one shared context with callers on different HIP streams (the library's workspaces, twiddle
caches and pinned buffers are per context and must be ordered across streams), and one context per thread on the same
device (nothing shared but the GPU)."""
import threading

import numpy as np
import pytest

from oracle.cpu_oracle import ints_to_limbs

pytestmark = pytest.mark.gpu

SCALAR_MONTGOMERY = 0
INVERSE, COSET = 1, 2


def _run_threads(workers):
    errs = []

    def guard(fn):
        def run():
            try:
                fn()
            except BaseException as e:   # noqa: BLE001 -- reported in the main thread
                errs.append(e)
        return run

    ts = [threading.Thread(target=guard(w)) for w in workers]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a worker is still inside the library (deadlock?)"
    if errs:
        raise errs[0]


def _msm_inputs(oracle, n, seed):
    k0, d = ints_to_limbs([0x1234567 + seed], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0]
    return oracle.g1_bases_arith(k0, d, n, 4), oracle.fr_sample(900 + seed, n)


def test_one_context_shared_by_four_threads(ctx, oracle):
    """Four host threads on ONE context, each on its own torch stream, mixing transforms of different sizes (they share the
    context's pass buffers), MSMs with and without the window table (one MSM workspace, one control block, one pinned
    result buffer), polynomial evaluations and blocking host-pointer calls.  Every result equals the one computed
    alone beforehand."""
    import plonk_prototype_amd as pa
    import torch
    sizes = [10, 12, 13, 16]
    vecs = {k: oracle.fr_sample(4000 + k, 1 << k) for k in sizes}
    exp_ntt = {(k, f): ctx.fr_ntt(vecs[k], k, f) for k in sizes for f in (0, INVERSE, COSET)}
    n = 6000
    pts, sc = _msm_inputs(oracle, n, 1)
    plain, table = pa.host.Bases(ctx, pts), pa.host.Bases(ctx, pts).precompute(13)
    exp_msm = {m: plain.msm(sc[:m]) for m in (n, 777, 1)}
    pt = oracle.fr_sample(5, 1)[0]
    exp_eval = {k: oracle.fr_poly_evaluate(vecs[k], pt) for k in sizes}

    def worker(tid):
        def run():
            st = torch.cuda.Stream()
            d_in = {k: torch.from_numpy(vecs[k].view(np.int64)).cuda() for k in sizes}
            d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
            torch.cuda.synchronize()
            for it in range(12):
                k = sizes[(tid + it) % len(sizes)]
                f = (0, INVERSE, COSET)[(tid + 2 * it) % 3]
                out = torch.empty_like(d_in[k])
                ctx.fr_ntt_dev(d_in[k].data_ptr(), 1 << k, out.data_ptr(), k, f, stream=st.cuda_stream)
                m = (n, 777, 1)[(tid + it) % 3]
                b = table if (tid + it) % 2 else plain
                r = b.msm_dev(d_sc.data_ptr(), m, stream=st.cuda_stream)          # returns when its result is on the host
                assert np.array_equal(r, exp_msm[m]), (tid, it, m)
                st.synchronize()
                assert np.array_equal(out.cpu().numpy().view(np.uint64), exp_ntt[(k, f)]), (tid, it, k, f)
                assert np.array_equal(ctx.fr_evaluate(d_in[k].data_ptr(), 1 << k, pt), exp_eval[k]), (tid, it, k)
                if it % 4 == tid:                                                   # host-pointer entry points too
                    assert np.array_equal(ctx.fr_ntt(vecs[k], k, f), exp_ntt[(k, f)])
                    assert np.array_equal(plain.msm(sc[:777]), exp_msm[777])
        return run

    _run_threads([worker(t) for t in range(4)])
    plain.free()
    table.free()


def test_one_context_per_thread_proves_the_same_bytes(ctx, oracle):
    """Three threads with a context each on the same device, proving at the same time: the proofs are the bytes a single
    context produces (and differ between the circuits)."""
    import plonk_prototype_amd as pa
    from test_gpu_prover import _srs
    n = 256
    srs = _srs(oracle, n)
    cases = [pa.synthetic.chain_circuit(n, 3), pa.synthetic.mixed_circuit(n, 4), pa.synthetic.chain_circuit(n, 5)]

    def prove_on(c, case, reps):
        circuit, wit, pi = case
        ck = pa.CommitKey(srs, c, precompute=True)
        pk = pa.prover.preprocess(circuit, c, ck)
        return [pa.prover.prove(pk, ck, wit, pi).to_bytes() for _ in range(reps)]

    want = [prove_on(ctx, case, 1)[0] for case in cases]
    assert len(set(want)) == len(cases)
    got = [None] * len(cases)

    def worker(i):
        def run():
            c = pa.Context(0)
            try:
                got[i] = prove_on(c, cases[i], 5)
            finally:
                c.close()
        return run

    _run_threads([worker(i) for i in range(len(cases))])
    for i in range(len(cases)):
        assert got[i] == [want[i]] * 5, i


def test_contexts_share_one_resident_srs(ctx, oracle):
    """include/plonk_mi355x.h: a pm_bases is read-only after precompute and may be used by every context of its device at the
    same time.  Four contexts, one window table: commitments and proofs equal the lone context's."""
    import plonk_prototype_amd as pa
    n = 1024
    srs = oracle.g1_bases_arith(ints_to_limbs([0x51], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, 4)
    ck0 = pa.CommitKey(srs, ctx, precompute=True)
    circuit, wit, pi = pa.synthetic.mixed_circuit(n, 8)
    want_proof = pa.prover.prove(pa.prover.preprocess(circuit, ctx, ck0), ck0, wit, pi).to_bytes()
    polys = [oracle.fr_sample(60 + i, n) for i in range(4)]
    want_commit = [ck0.commit(p) for p in polys]
    results = [None] * 4

    def worker(i):
        def run():
            c = pa.Context(0)
            view = object.__new__(pa.host.Bases)          # the same pm_bases handle, driven through THIS thread's context
            view.ctx, view.n, view._h = c, ck0._bases.n, ck0._bases._h
            try:
                ck = pa.CommitKey.__new__(pa.CommitKey)
                ck.ctx, ck._bases = c, view
                pk = pa.prover.preprocess(circuit, c, ck)
                out = []
                for _ in range(6):
                    out.append((pa.prover.prove(pk, ck, wit, pi).to_bytes(), [ck.commit(p) for p in polys]))
                results[i] = out
                pk.free()
            finally:
                view._h = None                            # not this view's to free
                c.close()
        return run

    _run_threads([worker(i) for i in range(4)])
    for i in range(4):
        for proof, commits in results[i]:
            assert proof == want_proof, i
            assert all(np.array_equal(a, b) for a, b in zip(commits, want_commit)), i
