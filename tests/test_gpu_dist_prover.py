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
    circuit, wit, pi = pa.synthetic.chain_circuit(N, 21)
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
        pk = pa.preprocess(circuit, ctx)
        proof = pa.prove(pk, ck, wit, pi)
        # the native sequence with the exchange callback must agree with the Python one on every rank
        npk = pa.NativeProverKey(circuit, ctx)
        nat = pa.prove_native(npk, ck, pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4)), pa.DeviceVector.from_host(ctx, pi))
        assert nat.to_bytes() == proof.to_bytes()
        npk.free()
        q.put((rank, _proof_blob(proof).tolist()))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_sharded_prover_world2_equals_single_gpu(ctx):
    import torch.multiprocessing as mp
    import plonk_prototype_amd as pa
    circuit, wit, pi, srs = _inputs()
    single = _proof_blob(pa.prove(pa.preprocess(circuit, ctx), pa.CommitKey(srs, ctx), wit, pi)).tolist()
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
