"""The prover with coefficient-range ownership end to end (pm_plonk_*_dist; SURVEY.md section 8e row 3 + 8f N5, VERDICT
r03 #5): rank r of W holds rows / coefficients [r n / W, (r + 1) n / W) of every vector, the size-n transforms run as
the four-step transform over the ranks, and NOTHING is replicated -- yet the proof and the verifier key must equal the
single-GPU prover's byte for byte on every rank.  Ranks are threads of this process (dist.LocalGroup, as in
tests/test_gpu_world8.py) and, once, spawned gloo processes."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_world8 import run_ranks

pytestmark = pytest.mark.gpu


def _blob(proof, vk):
    return proof.to_bytes() + b"".join(vk[k].tobytes() for k in sorted(vk))


def _inputs(n, mixed, seed):
    import plonk_prototype_amd as pa
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs
    circuit, wit, pi = (pa.synthetic.mixed_circuit if mixed else pa.synthetic.chain_circuit)(n, seed)
    srs = CpuOracle().g1_bases_arith(ints_to_limbs([77], 4)[0], ints_to_limbs([0x10001], 4)[0], n, 4)
    return circuit, wit, pi, srs


@pytest.mark.parametrize("log_n,world,mixed", [(12, 2, True), (12, 4, False), (14, 4, True), (16, 2, False), (16, 4, True),
                                               (8, 4, True), (6, 8, False), (10, 8, True), (10, 1, True), (11, 2, True)])
def test_dist_prover_equals_single_gpu(ctx, log_n, world, mixed):
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    n = 1 << log_n
    circuit, wit, pi, srs = _inputs(n, mixed, 31 + log_n)
    ck = pa.CommitKey(srs, ctx)
    pk = PR.preprocess(circuit, ctx, ck)
    single = _blob(PR.prove(pk, ck, wit, pi), pk.verifier_key)
    upstream = PR.prove(pk, ck, wit, pi, bind_public_inputs=False).to_bytes()
    full_bytes = None
    m = n // world

    def body(r, g):
        c = pa.Context(0)
        try:
            grp = DistGroup(rank=r, local=g)
            key = PR.DistProverKey(circuit, c, grp)
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            if r & 1:
                bases.precompute()                      # table and no table side by side
            key.commit(bases)
            c.comm_stats(reset=True)
            out = _blob(key.prove(bases, wit, pi), key.verifier_key)
            stats = c.comm_stats()
            again = key.prove(bases, wit, pi, bind_public_inputs=False).to_bytes()
            nbytes = key.device_bytes
            key.free()
            return out, again, nbytes, stats
        finally:
            c.close()
    n2 = 1 << (log_n - log_n // 2)
    for r, (out, again, nbytes, stats) in enumerate(run_ranks(world, body)):
        assert out == single, r
        assert again == upstream, r
        # per-rank memory: what a rank holds is 1 / world of the single-GPU key's arrays, plus the stage buffer of the batched
        # transforms (2 x 20 vectors), the halo rows and a scalar
        assert nbytes <= (186 * m + 64 * n2 + 8) * 32, (r, nbytes)
        # exchanges of ONE proof (was 3 per transform x 34 transforms): 3 + 2 for the wires and public inputs,
        # 3 + 2 for z, 2 for the quotient's way back; the all-to-all of a batch is one call whatever the batch; 8 all-gathers
        assert stats["alltoall_calls"] == (12 if world > 1 else 0) and stats["transpose_steps"] == 12, stats
        assert stats["allgather_calls"] == 8, stats


@pytest.mark.parametrize("log_n,world", [(10, 4), (12, 2), (12, 8)])
def test_dist_prover_against_the_c_prover(oracle, log_n, world):
    """Not only "equal to the library's other prover": every commitment and evaluation of the distributed proof, and the
    verifier key, against the CPU prover composed from the C restatement (oracle/cpu_prover.py) with the same challenges --
    and those challenges are the ones the verifier's side of the transcript derives from the proof bytes."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    from oracle import cpu_prover as CP
    from oracle.cpu_oracle import ints_to_limbs
    n = 1 << log_n
    circuit, wit, pub = pa.synthetic.mixed_circuit(n, 200 + log_n)
    srs = oracle.g1_bases_arith(ints_to_limbs([0xA5A5], 4)[0], ints_to_limbs([0x7FFFFFFF], 4)[0], n, 8)
    m = n // world

    def body(r, g):
        c = pa.Context(0)
        try:
            key = PR.DistProverKey(circuit, c, DistGroup(rank=r, local=g))
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            vk = key.commit(bases)
            proof = key.prove(bases, wit, pub)
            key.free()
            return proof, vk
        finally:
            c.close()
    proof, vk = run_ranks(world, body)[world - 1]
    replay = PR.derive_challenges(PR.Proof.from_bytes(proof.to_bytes()), vk, n, pub, t_eval=PR.fr_from_limbs(proof.evaluations["t"]))
    assert all(replay[k] == v for k, v in proof.challenges.items())
    cpk = CP.preprocess(oracle, {k: getattr(circuit, k) for k in CP.SELECTORS}, circuit.sigma_index, threads=8)
    exp = CP.prove(oracle, cpk, srs, wit, pub, proof.challenges, threads=8)
    for k, v in exp["evaluations"].items():
        assert np.array_equal(proof.evaluations[k], v), k
    for k, v in exp["commitments"].items():
        assert np.array_equal(proof.commitments[k], v), k
    for k, v in CP.verifier_key(oracle, cpk, srs, threads=8).items():
        assert np.array_equal(vk[k], v), k


def test_sharded_prover_world8_against_the_c_prover(oracle):
    """The sharded prover's twin of the test above at the world size of the target machine: eight ranks,
    2^12 gates (the configs[0] domain), pm_plonk_key_commit_sharded + pm_plonk_prove_sharded -- commitments, all 17
    evaluations and the verifier key against oracle/cpu_prover.py, not against the library's own single-GPU proof."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import ShardedCommitKey, shard_range
    from oracle import cpu_prover as CP
    from oracle.cpu_oracle import ints_to_limbs
    n, world = 1 << 12, 8
    circuit, wit, pub = pa.synthetic.mixed_circuit(n, 212)
    srs = oracle.g1_bases_arith(ints_to_limbs([0xA5A5], 4)[0], ints_to_limbs([0x7FFFFFFF], 4)[0], n, 8)

    def body(r, g):
        c = pa.Context(0)
        try:
            lo, hi = shard_range(n, r, world)
            ckr = ShardedCommitKey(srs[lo:hi], lo, n, c, group=g, rank=r)
            pkr = PR.preprocess(circuit, c, ckr)
            proof = PR.prove(pkr, ckr, wit, pub)
            vk = dict(pkr.verifier_key)
            pkr.free()
            return proof, vk
        finally:
            c.close()
    res = run_ranks(world, body)
    assert len({p.to_bytes() for p, _ in res}) == 1
    proof, vk = res[5]
    replay = PR.derive_challenges(PR.Proof.from_bytes(proof.to_bytes()), vk, n, pub, t_eval=PR.fr_from_limbs(proof.evaluations["t"]))
    assert all(replay[k] == v for k, v in proof.challenges.items())
    cpk = CP.preprocess(oracle, {k: getattr(circuit, k) for k in CP.SELECTORS}, circuit.sigma_index, threads=8)
    exp = CP.prove(oracle, cpk, srs, wit, pub, proof.challenges, threads=8)
    for k, v in exp["evaluations"].items():
        assert np.array_equal(proof.evaluations[k], v), k
    for k, v in exp["commitments"].items():
        assert np.array_equal(proof.commitments[k], v), k
    for k, v in CP.verifier_key(oracle, cpk, srs, threads=8).items():
        assert np.array_equal(vk[k], v), k


def test_dist_prover_public_inputs_on_every_rank(ctx):
    """32 public inputs spread over all four ranks' rows (each rank scatters the ones in its slice; all bind the whole
    list into the transcript), one of them with value zero and one position repeated."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs
    n, world = 1 << 10, 4
    rows = tuple(range(3, n, n // 32))
    circuit, wit, pi = pa.synthetic.chain_circuit(n, 9, public_rows=rows)
    srs = CpuOracle().g1_bases_arith(ints_to_limbs([77], 4)[0], ints_to_limbs([0x10001], 4)[0], n, 4)
    pos, val = PR.sparse_public_inputs(pi)
    assert len(pos) == 32
    ck = pa.CommitKey(srs, ctx)
    pk = PR.preprocess(circuit, ctx, ck)
    single = PR.prove(pk, ck, wit, (pos, val)).to_bytes()
    m = n // world

    def body(r, g):
        c = pa.Context(0)
        try:
            key = PR.DistProverKey(circuit, c, DistGroup(rank=r, local=g))
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            key.commit(bases)
            out = key.prove(bases, wit, (pos, val)).to_bytes()
            key.free()
            return out
        finally:
            c.close()
    assert all(out == single for out in run_ranks(world, body))


def test_dist_prover_with_host_staged_alltoall(ctx):
    """The all-to-all carried through host memory by the library's own copies (no torch on the device): the variant the
    host-AddressSanitizer run uses, where torch's CUDA initialisation is not available."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    n, world = 1 << 10, 4
    circuit, wit, pi, srs = _inputs(n, True, 41)
    ck = pa.CommitKey(srs, ctx)
    pk = PR.preprocess(circuit, ctx, ck)
    single = _blob(PR.prove(pk, ck, wit, pi), pk.verifier_key)
    m = n // world

    def body(r, g):
        c = pa.Context(0)
        try:
            key = PR.DistProverKey(circuit, c, DistGroup(rank=r, local=g, host_staging_ctx=c))
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            key.commit(bases)
            out = _blob(key.prove(bases, wit, pi), key.verifier_key)
            key.free()
            return out
        finally:
            c.close()
    assert all(out == single for out in run_ranks(world, body))


def test_dist_prover_failure_does_not_block(ctx):
    """A public input outside the circuit on ONE rank: that rank gets PM_ERR_LENGTH, its peers PM_ERR_EXCHANGE from their
    next all-gather, nobody blocks, and the keys prove normally afterwards; a short commit-key slice likewise."""
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    n, world = 1 << 10, 4
    circuit, wit, pi, srs = _inputs(n, True, 5)
    m = n // world

    def body(r, g):
        c = pa.Context(0)
        try:
            grp = DistGroup(rank=r, local=g)
            key = PR.DistProverKey(circuit, c, grp)
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            short = pa.host.Bases(c, srs[r * m:(r + 1) * m - (1 if r == 2 else 0)])
            codes = []
            try:
                key.commit(short)
                codes.append(0)
            except pa.Error as e:
                codes.append(e.code)
            key.commit(bases)
            pos, val = PR.sparse_public_inputs(pi)
            bad = pos.copy()
            if r == 1:
                bad[0] = n + 3
            try:
                key.prove(bases, wit, (bad, val))
                codes.append(0)
            except pa.Error as e:
                codes.append(e.code)
            codes.append(len(key.prove(bases, wit, pi).to_bytes()))
            key.free()
            return codes
        finally:
            c.close()
    res = run_ranks(world, body)
    for r, codes in enumerate(res):
        assert codes == [-6 if r == 2 else -7, -6 if r == 1 else -7, 1040], (r, codes)


@pytest.mark.parametrize("variant", ["q4_qc_zero", "q_arith_none"])
def test_dist_prover_with_identically_zero_selectors(ctx, variant):
    """a base selector that is zero on every rank -- or a q_arith that is absent -- has a coset array
    the quotient kernel reads; it must hold zeros, not whatever the allocation held before.  A key with random selectors
    is built and freed first on every rank's context so that the next key's allocations reuse dirty memory."""
    import dataclasses
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    n, world = 1 << 10, 4
    dirty, _, _, srs = _inputs(n, False, 3)
    if variant == "q4_qc_zero":
        circuit, wit, pi = pa.synthetic.chain_circuit(n, 19, zero_selectors=("q_4", "q_c"))
    else:
        circuit, wit, pi = pa.synthetic.chain_circuit(n, 23)
        circuit = dataclasses.replace(circuit, q_arith=None)
    ck = pa.CommitKey(srs, ctx)
    pk = PR.preprocess(circuit, ctx, ck)
    single = _blob(PR.prove(pk, ck, wit, pi), pk.verifier_key)
    m = n // world

    def body(r, g):
        c = pa.Context(0)
        try:
            grp = DistGroup(rank=r, local=g)
            PR.DistProverKey(dirty, c, grp).free()
            key = PR.DistProverKey(circuit, c, grp)
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            key.commit(bases)
            out = _blob(key.prove(bases, wit, pi), key.verifier_key)
            key.free()
            return out
        finally:
            c.close()
    for r, out in enumerate(run_ranks(world, body)):
        assert out == single, r


def test_dist_preprocess_failure_on_one_rank_does_not_block(ctx):
    """a sigma index outside the circuit, or one repeated inside a slice, on ONE rank of four: that rank
    gets PM_ERR_BAD_ARG before the first exchange, its peers PM_ERR_EXCHANGE from it, nobody reaches an all-to-all; slices
    whose union is not a permutation (two ranks given the same rows) fail on every rank at the agreement."""
    import dataclasses
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    n, world = 1 << 8, 4
    circuit, wit, pi, srs = _inputs(n, False, 2)
    m = n // world

    def run(mutate):
        def body(r, g):
            c = pa.Context(0)
            try:
                sig = np.array(circuit.sigma_index, dtype=np.int64).reshape(4, n).copy()
                mutate(r, sig)
                try:
                    PR.DistProverKey(dataclasses.replace(circuit, sigma_index=sig), c, DistGroup(rank=r, local=g)).free()
                    return 0
                except pa.Error as e:
                    return e.code
            finally:
                c.close()
        return run_ranks(world, body)

    def outside(r, sig):
        if r == 2:
            sig[1, 2 * m + 5] = 4 * n
    assert run(outside) == [-7, -7, -1, -7]

    def repeated(r, sig):
        if r == 1:
            sig[0, m + 1] = sig[3, m + 7]
    assert run(repeated) == [-7, -1, -7, -7]

    def overlapping(r, sig):
        if r == 3:
            sig[:, 3 * m:4 * m] = sig[:, 2 * m:3 * m]      # rank 3 was given rank 2's rows
    assert run(overlapping) == [-1, -1, -1, -1]
    assert run(lambda r, sig: None) == [0, 0, 0, 0]


def test_dist_rejects_bad_groups(ctx):
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup, LocalGroup
    circuit, wit, pi, srs = _inputs(16, False, 1)
    for world in (3, 8):                      # not a power of two; world^2 > n
        grp = DistGroup(rank=0, local=LocalGroup(world))
        with pytest.raises((pa.Error, ValueError)):
            PR.DistProverKey(circuit, ctx, grp)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    from plonk_prototype_amd.dist import DistGroup
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << 12
        circuit, wit, pi, srs = _inputs(n, True, 77)
        c = pa.Context(0)
        grp = DistGroup()                      # the default process group
        key = PR.DistProverKey(circuit, c, grp)
        m = n // world
        bases = pa.host.Bases(c, srs[rank * m:(rank + 1) * m])
        key.commit(bases)
        q.put((rank, list(_blob(key.prove(bases, wit, pi), key.verifier_key))))
        key.free()
        c.close()
    finally:
        dist.destroy_process_group()


def test_dist_prover_over_gloo_processes(ctx):
    """One process per rank (how a node runs it), gloo carrying both exchanges through the host."""
    import torch.multiprocessing as mp
    import plonk_prototype_amd as pa
    import plonk_prototype_amd.prover as PR
    n = 1 << 12
    circuit, wit, pi, srs = _inputs(n, True, 77)
    ck = pa.CommitKey(srs, ctx)
    pk = PR.preprocess(circuit, ctx, ck)
    single = list(_blob(PR.prove(pk, ck, wit, pi), pk.verifier_key))
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, blob in res:
        assert blob == single, rank
