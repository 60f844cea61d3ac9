"""Two ranks on the one GPU of the test box (gloo for the 144-byte exchange): the sharded prover
(dist.ShardedCommitKey -- BASELINE.json configs[4]: MSMs split over the GPUs of a node, partial sums
all-gathered and folded) must return, on every rank, exactly the proof a single GPU returns."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
N = 1 << 10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    import plonk_prototype_amd as pa
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs
    circuit, wit, pi = pa.synthetic.mixed_circuit(N, 21)
    srs = CpuOracle().g1_bases_arith(ints_to_limbs([77], 4)[0], ints_to_limbs([0x10001], 4)[0], N, 4)
    return circuit, wit, pi, srs


def _proof_blob(proof):
    keys = sorted(proof.commitments)
    return np.concatenate([proof.commitments[k] for k in keys] + [proof.evaluations[k] for k in sorted(proof.evaluations)])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import ShardedCommitKey, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        circuit, wit, pi, srs = _inputs()
        ctx = pa.Context(0)
        lo, hi = shard_range(N, rank, world)
        ck = ShardedCommitKey(srs[lo:hi], lo, N, ctx, precompute=(rank == 0))   # mixed table / no table
        pk = pa.preprocess(circuit, ctx, ck)        # the verifier key's 15 commitments go through the exchange too
        proof = pa.prove(pk, ck, wit, pi)
        again = pa.prove(pk, ck, pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4)), pi)
        assert again.to_bytes() == proof.to_bytes()
        vk = np.concatenate([pk.verifier_key[k] for k in sorted(pk.verifier_key)])
        q.put((rank, np.concatenate([_proof_blob(proof), vk]).tolist()))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_sharded_prover_world2_equals_single_gpu(ctx):
    import torch.multiprocessing as mp
    import plonk_prototype_amd as pa
    circuit, wit, pi, srs = _inputs()
    ck = pa.CommitKey(srs, ctx)
    pk = pa.preprocess(circuit, ctx, ck)
    single = np.concatenate([_proof_blob(pa.prove(pk, ck, wit, pi))]
                            + [pk.verifier_key[k] for k in sorted(pk.verifier_key)]).tolist()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    for rank, blob in res:
        assert blob == single, rank


def _failing_worker(rank, world, port, q, mode):
    """mode "pi": rank 1 passes a public-input position outside the circuit (an argument error on ONE rank, far from
    any MSM); mode "gap": rank 1's slice starts one coefficient late; mode "overlap": both ranks hold the middle half.  No rank may block: both must get an error."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import ShardedCommitKey, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        circuit, wit, pi, srs = _inputs()
        ctx = pa.Context(0)
        lo, hi = shard_range(N, rank, world)
        codes = []
        if mode in ("gap", "overlap"):
            shift = 1 if rank == 1 else 0
            if mode == "overlap":            # both ranks hold [n/4, 3n/4): counts and index sums still add up
                lo, hi, shift = N // 4, 3 * N // 4, 0
            ck = ShardedCommitKey(srs[lo + shift:hi], lo + shift, N, ctx)
            try:
                pa.preprocess(circuit, ctx, ck)
                codes.append(0)
            except pa.Error as e:
                codes.append(e.code)
        else:
            ck = ShardedCommitKey(srs[lo:hi], lo, N, ctx)
            pk = pa.preprocess(circuit, ctx, ck)
            pos, val = pa.prover.sparse_public_inputs(pi)
            bad = pos.copy()
            if rank == 1:
                bad[0] = N + 5
            try:
                pa.prove(pk, ck, wit, (bad, val))
                codes.append(0)
            except pa.Error as e:
                codes.append(e.code)
            # the key is usable again afterwards: both ranks prove normally
            codes.append(len(pa.prove(pk, ck, wit, pi).to_bytes()))
        q.put((rank, codes))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["pi", "gap", "overlap"])
def test_one_rank_failing_outside_an_msm_does_not_block_its_peer(mode):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_failing_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode in ("gap", "overlap"):
        assert res[0] == [-6] and res[1] == [-6]            # PM_ERR_LENGTH on every rank: the slices do not tile [0, n)
    else:
        assert res[1][0] == -6 and res[0][0] == -7          # the bad rank: PM_ERR_LENGTH; its peer: PM_ERR_EXCHANGE
        assert res[0][1] == 1040 and res[1][1] == 1040


@pytest.mark.parametrize("timeout_ms", [0, 20000])
def test_library_communicator_world1(ctx, oracle, timeout_ms):
    """The in-library RCCL exchange on the one GPU of the test box (a one-rank communicator: two RCCL ranks
    cannot share a device): unique id, ncclCommInitRank, ncclAllGather through pm_g1_allgather_fold, and the
    sharded prover entry points with exchange = NULL -- the proof must equal the unsharded one byte for byte.
    timeout_ms > 0: the same with every exchange under the deadline of option comm_timeout_ms (the watch thread armed and
    disarmed around the real ncclAllGather; exchanges that complete are left alone -- tests/test_comm_deadline.py has the
    other case, on a stub table)."""
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import ShardedCommitKey
    from oracle.cpu_oracle import ints_to_limbs
    circuit, wit, pi, srs = _inputs()
    ctx.set_option("comm_timeout_ms", timeout_ms)
    ctx.comm_init(0, 1)
    try:
        part = np.zeros((2, 18), np.uint64)
        one = oracle.fp_to_mont(ints_to_limbs([1], 6))[0]
        part[0, :12], part[0, 12:] = srs[3], one
        part[1, 6:12] = one
        got = ctx.g1_allgather_fold(part)
        assert np.array_equal(pa.g1_to_affine(got[0])[0], srs[3]) and pa.g1_to_affine(got[1])[1]
        with pytest.raises(pa.Error) as e:
            ctx.g1_allgather_fold(np.zeros((0, 18), np.uint64))      # the abort marker comes back as an error
        assert e.value.code == -7
        ck_plain = pa.CommitKey(srs, ctx)
        ref = pa.prove(pa.preprocess(circuit, ctx, ck_plain), ck_plain, wit, pi)
        ck = ShardedCommitKey(srs, 0, N, ctx, native=True)
        pk = pa.preprocess(circuit, ctx, ck)
        assert pa.prove(pk, ck, wit, pi).to_bytes() == ref.to_bytes()
    finally:
        ctx.comm_destroy()
        ctx.set_option("comm_timeout_ms", 0)
    with pytest.raises(pa.Error):
        ctx.g1_allgather_fold(np.zeros((1, 18), np.uint64))          # no communicator any more
