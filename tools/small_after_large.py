"""GPU probe: does the context's history (a 2^24-point MSM: large control block, workspaces) slow SMALL proofs down?
Times a 2^12- and a 2^16-gate proof on a fresh context, after a 2^20 MSM and after a 2^24 MSM in the same context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
o = CpuOracle()
ctx = pa.Context(0)
keys = {}
for k in (12, 16):
    n = 1 << k
    circuit, wit, pi = pa.synthetic.chain_circuit(n, 5)
    srs = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, threads=16)
    ck = pa.CommitKey(srs, ctx, precompute=True)
    keys[k] = (pa.preprocess(circuit, ctx, ck), ck, pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4)), pa.prover.sparse_public_inputs(pi))


def timed(label):
    for k, (pk, ck, dw, dpi) in keys.items():
        pa.prove(pk, ck, dw, dpi)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            pa.prove(pk, ck, dw, dpi)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{label}: 2^{k} best {ts[0]*1e3:.3f} median {ts[7]*1e3:.3f} ms", flush=True)


timed("fresh")
tau = pa.field.fr_to_limbs(0xABCDEF123)
for big_k in (20, 24):
    big = pa.CommitKey.setup((1 << big_k) - 1, tau, ctx, precompute=True)
    sc = torch.from_numpy(np.random.default_rng(1).integers(0, 1 << 62, size=(1 << big_k, 4), dtype=np.uint64).view(np.int64)).cuda()
    big._bases.msm_dev(sc.data_ptr(), 1 << big_k)
    big._bases.msm_batch_dev(sc.data_ptr(), 1 << (big_k - 2), 4)
    ctx.sync()
    timed(f"after a 2^{big_k} MSM")
    big._bases.free()
    del sc
    torch.cuda.empty_cache()
    timed(f"after freeing its key")
