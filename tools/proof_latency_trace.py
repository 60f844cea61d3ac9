import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
o = CpuOracle(); ctx = pa.Context(0)
k = 16; n = 1 << k
circuit, wit, pi = pa.synthetic.chain_circuit(n, 5)
srs = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, threads=16)
ck = pa.CommitKey(srs, ctx, precompute=True)
pk = pa.preprocess(circuit, ctx, ck); dw = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4)); dpi = pa.prover.sparse_public_inputs(pi)
for rep in range(3):
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); pa.prove(pk, ck, dw, dpi); ts.append(time.perf_counter() - t0)
    print(" ".join(f"{t*1e3:.2f}" for t in ts), flush=True)
    time.sleep(0.5)
