"""BASELINE.json configs[4] in its own shape, rehearsed on the ONE GPU of the test box: world size 8 -- the whole
target machine -- for the sharded prover, the four-step NTT and the fold, and a 2-rank sharded proof of a
2^24-gate circuit.

The ranks here are THREADS of this process with one context each (dist.LocalGroup): the box's process guard
allows six GPU processes, so eight spawned gloo ranks (how tests/test_gpu_dist_prover.py runs world 2) cannot
exist on it.  Everything the library does per rank -- its slice of every MSM, the slice-cover check, the
fixed-size exchange message, the abort marker, the pack / unpack indexing of the all-to-all -- is the same code
whichever transport carries the messages; the transport itself (RCCL over xGMI) is NOT exercised here."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_ranks(world, body, timeout=900):
    """body(rank, group) on `world` threads -> list of results; an exception on one rank breaks the group's
    barrier (its peers' exchanges then fail instead of waiting) and is re-raised here."""
    from plonk_prototype_amd.dist import LocalGroup
    group = LocalGroup(world, timeout=timeout)
    res, errs = [None] * world, [None] * world

    def run(r):
        try:
            res[r] = body(r, group)
        except BaseException as e:          # noqa: BLE001 -- reported below
            errs[r] = e
            group._barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout)
        assert not t.is_alive(), "a rank is stuck"
    for e in errs:
        if e is not None:
            raise e
    return res


def _blob(proof, pk):
    return proof.to_bytes() + b"".join(pk.verifier_key[k].tobytes() for k in sorted(pk.verifier_key))


def test_local_group_gather_fold_world8(oracle):
    """The transport of this file by itself: eight threads, 16 partial points each, abort marker."""
    import plonk_prototype_amd as pa
    from oracle.cpu_oracle import ints_to_limbs
    G = oracle.g1_generator()
    one = oracle.fp_to_mont(ints_to_limbs([1], 6))[0]
    mult = [oracle.g1_mul(G, ints_to_limbs([m], 4)[0]) for m in range(1, 9)]

    def body(r, g):
        part = np.zeros((16, 18), np.uint64)
        part[:, :12], part[:, 12:] = mult[r], one
        ok = g.allgather_fold_many(r, part)
        gave_up = g.allgather_fold_many(r, None if r == 6 else part)
        return ok, gave_up
    for ok, gave_up in run_ranks(8, body):
        assert gave_up is None
        for j in range(16):
            assert np.array_equal(pa.g1_to_affine(ok[j])[0], oracle.g1_mul(G, ints_to_limbs([36], 4)[0]))


def test_sharded_prover_world8_at_2_20_gates(ctx, oracle):
    """(a) 8 ranks, the configs[3] circuit (2^20 gates): pm_plonk_key_commit_sharded + pm_plonk_prove_sharded on
    each rank's 2^17-point slice of a powers-of-tau key must give, on EVERY rank, the single-context proof and
    verifier key byte for byte.  The ranks hold the slices in a scrambled order (any tiling of [0, n) is a valid
    sharding), half of them with the window table."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import ShardedCommitKey, shard_range
    from test_gpu_prover import TAU, _mont, full_size_checks
    gk, world = 20, 8
    n = 1 << gk
    circuit, d_wit, _ = pa.synthetic.wide_circuit(n, ctx, seed=5)
    wit = d_wit.to_host()
    ck = pa.CommitKey.setup(n - 1, _mont(oracle, TAU), ctx, precompute=True)
    pk = PR.preprocess(circuit, ctx, ck)
    proof = PR.prove(pk, ck, d_wit, None)
    full_size_checks(ctx, oracle, proof, d_wit, gk)
    single = _blob(proof, pk)
    pk.free()
    ck._bases.free()
    d_wit.free()
    owner = [3, 0, 7, 1, 6, 2, 5, 4]                      # rank r holds slice owner[r]

    def body(r, g):
        c = pa.Context(0)
        try:
            lo, hi = shard_range(n, owner[r], world)
            ckr = ShardedCommitKey.setup(n, TAU, lo, hi, c, group=g, rank=r, precompute=bool(r & 1))
            pkr = PR.preprocess(circuit, c, ckr)
            out = _blob(PR.prove(pkr, ckr, wit, None), pkr)
            again = _blob(PR.prove(pkr, ckr, wit, None), pkr)
            pkr.free()
            return out, again
        finally:
            c.close()
    for r, (out, again) in enumerate(run_ranks(world, body)):
        assert out == single, r
        assert again == single, r


@pytest.mark.parametrize("mode", ["pi", "gap", "overlap"])
def test_world8_failure_paths(mode):
    """One abort path and the slice-cover check at world 8 (2^12 gates): "pi" -- rank 5 alone passes a public
    input outside the circuit: it gets PM_ERR_LENGTH, its seven peers PM_ERR_EXCHANGE, nobody blocks, and the
    keys prove normally afterwards; "gap" -- rank 3's slice starts one coefficient late; "overlap" -- ranks 2
    and 5 both hold slice 2 shifted by half a slice and slice 5 is held shifted back (counts and index sums
    still add up to n and n (n - 1) / 2: the two-moment check of r03 accepted this)."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import ShardedCommitKey, shard_range
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs
    n, world = 1 << 12, 8
    circuit, wit, pi = pa.synthetic.mixed_circuit(n, 21)
    srs = CpuOracle().g1_bases_arith(ints_to_limbs([77], 4)[0], ints_to_limbs([0x10001], 4)[0], n, 4)

    def body(r, g):
        c = pa.Context(0)
        codes = []
        try:
            lo, hi = shard_range(n, r, world)
            if mode == "gap" and r == 3:
                lo += 1
            if mode == "overlap":
                half = (hi - lo) // 2
                if r == 2:
                    lo, hi = lo + half, hi + half
                if r == 5:
                    lo, hi = lo - half, hi - half
            ckr = ShardedCommitKey(srs[lo:hi], lo, n, c, group=g, rank=r)
            try:
                pkr = PR.preprocess(circuit, c, ckr)
                codes.append(0)
            except pa.Error as e:
                return [e.code]
            if mode == "pi":
                pos, val = PR.sparse_public_inputs(pi)
                bad = pos.copy()
                if r == 5:
                    bad[0] = n + 5
                try:
                    PR.prove(pkr, ckr, wit, (bad, val))
                    codes.append(0)
                except pa.Error as e:
                    codes.append(e.code)
                codes.append(len(PR.prove(pkr, ckr, wit, pi).to_bytes()))
            return codes
        finally:
            c.close()
    res = run_ranks(world, body)
    if mode == "pi":
        for r, codes in enumerate(res):
            assert codes == [0, -6 if r == 5 else -7, 1040], (r, codes)
    else:
        assert all(codes == [-6] for codes in res), res


def test_sharded_prover_2_ranks_at_2_24_gates(ctx, oracle):
    """(b) The circuit size of configs[4]: a 2^24-gate proof with every MSM split over two ranks (2^23-point
    slices, each with its own window table) -- byte-equal to the unsharded proof of the same circuit, which
    itself passes the size-independent checks (identity, [a(tau)] G, the KZG equation of W_zw)."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import ShardedCommitKey, shard_range
    from test_gpu_prover import TAU, _mont, full_size_checks
    gk, world = 24, 2
    n = 1 << gk
    ctx.trim()
    circuit, d_wit, _ = pa.synthetic.wide_circuit(n, ctx, seed=9)
    wit = d_wit.to_host()
    ck = pa.CommitKey.setup(n - 1, _mont(oracle, TAU), ctx, precompute=True)
    pk = PR.preprocess(circuit, ctx, ck)
    proof = PR.prove(pk, ck, d_wit, None)
    full_size_checks(ctx, oracle, proof, d_wit, gk)
    single = _blob(proof, pk)
    pk.free()
    ck._bases.free()
    d_wit.free()
    ctx.trim()

    def body(r, g):
        c = pa.Context(0)
        try:
            lo, hi = shard_range(n, r, world)
            ckr = ShardedCommitKey.setup(n, TAU, lo, hi, c, group=g, rank=r, precompute=True)
            pkr = PR.preprocess(circuit, c, ckr)
            out = _blob(PR.prove(pkr, ckr, wit, None), pkr)
            pkr.free()
            ckr._bases.free()
            return out
        finally:
            c.close()
    for r, out in enumerate(run_ranks(world, body, timeout=1200)):
        assert out == single, r


@pytest.mark.parametrize("log_ns", [(6, 7, 9, 12), (13, 16)])
def test_four_step_ntt_world8(ctx, oracle, log_ns):
    """(c) pm_fr_ntt_fourstep_dev at world 8, 2^6 (one row and one column block per rank) .. 2^16, every flag
    combination incl. PM_NTT_TRANSPOSED: every rank's block equals the same block of the single-GPU transform."""
    import torch
    import plonk_prototype_amd as pa
    world = 8
    fulls = {k: oracle.fr_sample(888 + k, 1 << k) for k in log_ns}
    exp = {(k, f): ctx.fr_ntt(fulls[k], k, f) for k in log_ns for f in (0, 1, 2, 3)}

    def body(r, g):
        c = pa.Context(0)
        bad = []
        try:
            for k in log_ns:
                n = 1 << k
                blk = n // world
                l1 = k // 2
                n1, n2 = 1 << l1, 1 << (k - l1)
                mine = torch.from_numpy(fulls[k][r * blk:(r + 1) * blk].view(np.int64).copy()).cuda()
                stage = torch.empty((2 * blk, 4), dtype=torch.int64, device="cuda")
                cb = g.alltoall_fn(r, stage)

                def run(x, flags):
                    y = x.clone()
                    torch.cuda.synchronize()
                    c.fr_ntt_fourstep_dev(y.data_ptr(), stage.data_ptr(), k, world, r, flags, cb)
                    c.sync()
                    return y
                for f in (0, 1, 2, 3):
                    if not np.array_equal(run(mine, f).cpu().numpy().view(np.uint64), exp[(k, f)][r * blk:(r + 1) * blk]):
                        bad.append((k, f))
                for cos in (0, 2):
                    tr = run(mine, cos | 4)
                    nat = exp[(k, cos)].reshape(n2, n1, 4)
                    exp_t = np.ascontiguousarray(nat.transpose(1, 0, 2)).reshape(n, 4)[r * blk:(r + 1) * blk]
                    if not np.array_equal(tr.cpu().numpy().view(np.uint64), exp_t):
                        bad.append((k, cos | 4))
                    if not torch.equal(run(tr, cos | 1 | 4), mine):
                        bad.append((k, cos | 5))
            return bad
        finally:
            c.close()
    assert run_ranks(world, body) == [[]] * world


@pytest.mark.parametrize("gk", [20, 24])
def test_dist_prover_world8(ctx, oracle, gk):
    """configs[4]'s world size (and, for 24, its circuit size: configs[4] in its stated shape but for the single GPU
    under the eight ranks) with NOTHING replicated (pm_plonk_*_dist): 8 ranks, 2^20 gates, every rank
    holding 2^17 rows / coefficients of every vector and a 2^17-point slice of the commit key -- proof and verifier key
    byte-equal to the single-context ones, and each rank's device memory an eighth of a single-GPU key's."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup, ShardedCommitKey
    from test_gpu_prover import TAU, _mont
    world = 8
    n = 1 << gk
    m = n // world
    ctx.trim()
    circuit, d_wit, _ = pa.synthetic.wide_circuit(n, ctx, seed=5)
    wit = d_wit.to_host()
    ck = pa.CommitKey.setup(n - 1, _mont(oracle, TAU), ctx, precompute=True)
    pk = PR.preprocess(circuit, ctx, ck)
    single = _blob(PR.prove(pk, ck, d_wit, None), pk)
    pk.free()
    ck._bases.free()
    d_wit.free()
    ctx.trim()

    def body(r, g):
        c = pa.Context(0)
        try:
            grp = DistGroup(rank=r, local=g)
            bases = ShardedCommitKey.setup(n, TAU, r * m, (r + 1) * m, c, group=g, rank=r, precompute=bool(r & 1))._bases
            key = PR.DistProverKey(circuit, c, grp)
            key.commit(bases)
            proof = key.prove(bases, wit.reshape(4, n, 4), None)
            out = proof.to_bytes() + b"".join(key.verifier_key[k].tobytes() for k in sorted(key.verifier_key))
            nbytes = key.device_bytes
            key.free()
            return out, nbytes
        finally:
            c.close()
    res = run_ranks(world, body, timeout=1500)
    for r, (out, nbytes) in enumerate(res):
        assert out == single, r
        # key + workspace + the stage buffer of the batched transforms (2 x 20 vectors): 0.68 GB per rank at 2^20 gates on 8
        # ranks, 10.9 GB at 2^24 (a single-GPU key at 2^24: ~70 GB)
        assert nbytes <= (163 * m + 64 * (1 << (gk - gk // 2)) + 8) * 32, (r, nbytes)
    print(f"[n5] 2^{gk} gates on 8 ranks: {res[0][1] / 2**20:.0f} MiB of key + workspace per rank")
