"""GPU probe (not a test): K contexts (own stream, key and workspace each) proving the same 2^LOG_N-gate circuit from K host
threads over one resident SRS -- throughput of a proving service on one GPU.  usage: python tools/multi_ctx_prover.py [LOG_N] [K ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ks = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]
n = 1 << log_n
orc = CpuOracle()
circuit, wit, pi = pa.synthetic.chain_circuit(n, 1)
srs = orc.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, threads=16)
ctx0 = pa.Context(0)
ck0 = pa.CommitKey(srs, ctx0, precompute=True)
dpi = pa.prover.sparse_public_inputs(pi)
ctxs, cks, keys, wits = [ctx0], [ck0], [], []
for i in range(max(ks)):
    if i:
        c = pa.Context(0)
        ck = pa.CommitKey.__new__(pa.CommitKey)
        ck.__dict__.update(ck0.__dict__)            # the same resident SRS table
        ctxs.append(c); cks.append(ck)
    keys.append(pa.preprocess(circuit, ctxs[i], cks[i]))
    wits.append(pa.DeviceVector.from_host(ctxs[i], wit.reshape(-1, 4)))
ref = pa.prove(keys[0], cks[0], wits[0], dpi).to_bytes()
for i in range(1, max(ks)):
    assert pa.prove(keys[i], cks[i], wits[i], dpi).to_bytes() == ref
per = 8
for k in ks:
    def worker(i):
        for _ in range(per):
            pa.prove(keys[i], cks[i], wits[i], dpi)
    best = 1e9
    for rep in range(3):
        th = [threading.Thread(target=worker, args=(i,)) for i in range(k)]
        for c in ctxs: c.sync()
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        for c in ctxs: c.sync()
        best = min(best, (time.perf_counter() - t0) / (k * per))
    print(f"2^{log_n} gates, {k} context(s) in flight: {best * 1e3:7.2f} ms per proof  ({1 / best:6.1f} proofs/s)", flush=True)
