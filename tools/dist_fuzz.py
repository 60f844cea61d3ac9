"""GPU stress (not a test), r05: random shapes through the batched rank-split transform and the distributed prover (ranks = threads of this
process), every result compared with the single-GPU path.  usage: python tools/dist_fuzz.py [seconds=200] [seed=1]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import plonk_prototype_amd as pa
import plonk_prototype_amd.prover as PR
from plonk_prototype_amd.dist import DistGroup
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
from test_gpu_world8 import run_ranks
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o = CpuOracle(); ctx = pa.Context(0)
t_end = time.time() + budget
n_ntt = n_prove = 0
while time.time() < t_end:
    # (a) batched transform
    world = rng.choice([1, 2, 4, 8]); log_n = rng.randint(max(4, 2 * world.bit_length()), 13); batch = rng.randint(1, 7)
    n = 1 << log_n; blk = n // world; n2 = 1 << (log_n - log_n // 2); n1 = n // n2
    if n1 % world or n2 % world:
        continue
    flags = rng.choice([0, 1, 2, 3, 4, 6]); halo = flags in (4, 6) and rng.random() < 0.7
    full = [o.fr_sample(rng.getrandbits(30), n) for _ in range(batch)]
    ref = [ctx.fr_ntt(x, log_n, flags & 3) for x in full]
    if flags & 4:
        ref = [np.ascontiguousarray(y.reshape(n2, n1, 4).transpose(1, 0, 2)).reshape(n, 4) for y in ref]

    def body(r, g):
        c = pa.Context(0)
        try:
            grp = DistGroup(rank=r, local=g)
            x = torch.from_numpy(np.concatenate([v[r * blk:(r + 1) * blk] for v in full]).view(np.int64).copy()).cuda()
            stage = torch.empty((2 * batch * (blk + n2), 4), dtype=torch.int64, device="cuda")
            h = torch.zeros((batch * n2, 4), dtype=torch.int64, device="cuda")
            c.fr_ntt_fourstep_batch_dev(x.data_ptr(), batch, stage.data_ptr(), log_n, world, r, flags, grp._aa if world > 1 else None,
                                        d_halo=h.data_ptr() if halo else 0)
            c.sync()
            return x.cpu().numpy().view(np.uint64), h.cpu().numpy().view(np.uint64)
        finally:
            c.close()
    for r, (got, gh) in enumerate(run_ranks(world, body)):
        assert np.array_equal(got, np.concatenate([y[r * blk:(r + 1) * blk] for y in ref])), ("ntt", world, log_n, batch, flags, r)
        if halo:
            nxt = ((r + 1) % world) * blk
            assert np.array_equal(gh, np.concatenate([y[nxt:nxt + n2] for y in ref])), ("halo", world, log_n, batch, flags, r)
    n_ntt += 1
    # (b) distributed proof
    world = rng.choice([1, 2, 4, 8]); log_n = rng.randint(max(6, 2 * world.bit_length()), 12); n = 1 << log_n
    if (1 << (log_n // 2)) % world:
        continue
    mixed = log_n >= 8 and rng.random() < 0.6
    circuit, wit, pi = (pa.synthetic.mixed_circuit if mixed else pa.synthetic.chain_circuit)(n, rng.getrandbits(20))
    srs = o.g1_bases_arith(ints_to_limbs([rng.getrandbits(60) | 1], 4)[0], ints_to_limbs([rng.getrandbits(60) | 1], 4)[0], n, 8)
    ck = pa.CommitKey(srs, ctx); pk = PR.preprocess(circuit, ctx, ck)
    bind = rng.random() < 0.5
    single = PR.prove(pk, ck, wit, pi, bind_public_inputs=bind).to_bytes()
    vk = b"".join(pk.verifier_key[k].tobytes() for k in sorted(pk.verifier_key))
    pk.free(); ck._bases.free()
    m = n // world

    def body2(r, g):
        c = pa.Context(0)
        try:
            key = PR.DistProverKey(circuit, c, DistGroup(rank=r, local=g))
            bases = pa.host.Bases(c, srs[r * m:(r + 1) * m])
            if rng.random() < 0.5:
                bases.precompute()
            key.commit(bases)
            out = key.prove(bases, wit, pi, bind_public_inputs=bind).to_bytes()
            v = b"".join(key.verifier_key[k].tobytes() for k in sorted(key.verifier_key))
            key.free()
            return out, v
        finally:
            c.close()
    for r, (out, v) in enumerate(run_ranks(world, body2)):
        assert out == single and v == vk, ("prove", world, log_n, mixed, bind, r)
    n_prove += 1
    if (n_ntt + n_prove) % 10 == 0:
        print(f"{n_ntt} batched transforms, {n_prove} distributed proofs: all equal", flush=True)
print(f"done: {n_ntt} batched transforms and {n_prove} distributed proofs over random worlds / sizes / flags, all equal to the single-GPU results", flush=True)
