"""GPU probe (not a test): the distributed prover (pm_plonk_*_dist) on ONE rank at 2^K gates -- what its decomposition costs in kernels
(transposes of the rank-split transforms, sub-transforms, expansion, planar quotient) beside the single-GPU prover.  Run it under
rocprofv3 --kernel-trace --stats.  usage: python tools/dist_prover_profile.py [K=20] [reps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
import plonk_prototype_amd.prover as PR
from plonk_prototype_amd.dist import DistGroup, LocalGroup
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << k
ctx = pa.Context(0)
circuit, d_wit, _ = pa.synthetic.wide_circuit(n, ctx, seed=5)
tau = pa.field.fr_to_limbs(0xABCDEF123)
ck = pa.CommitKey.setup(n - 1, tau, ctx, precompute=True)
grp = DistGroup(rank=0, local=LocalGroup(1))
t0 = time.perf_counter()
key = PR.DistProverKey(circuit, ctx, grp)
key.commit(ck._bases)
ctx.sync()
print(f"preprocess + key commit: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
key.prove(ck._bases, d_wit, None)
ctx.comm_stats(reset=True)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    key.prove(ck._bases, d_wit, None)
    ts.append(time.perf_counter() - t0)
st = ctx.comm_stats()
print(f"prove_dist 2^{k} on one rank: best {min(ts) * 1e3:.2f} ms; per proof: {st['transpose_steps'] // reps} transpose steps, "
      f"{st['allgather_calls'] // reps} all-gathers", flush=True)
