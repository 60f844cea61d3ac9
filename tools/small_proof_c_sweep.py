"""GPU probe (not a test): latency of a whole proof of a SMALL circuit against the window width of the commit key's table.
usage: python tools/small_proof_c_sweep.py [LOGS] [CS]      (c = 0: the library's choice, -1: no table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
logs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10, 12, 14, 16, 17]
cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 11, 13, 14, 16]
o = CpuOracle()
ctx = pa.Context(0)
for k in logs:
    n = 1 << k
    circuit, wit, pi = pa.synthetic.chain_circuit(n, 1)
    srs = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, threads=16)
    dw = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
    dpi = pa.prover.sparse_public_inputs(pi)
    ref = None
    for c in cs:
        ck = pa.CommitKey(srs, ctx)
        if c >= 0:
            ck._bases.precompute(c)
        pk = pa.preprocess(circuit, ctx, ck)
        b = pa.prove(pk, ck, dw, dpi).to_bytes()
        if ref is None: ref = b
        assert b == ref
        ts = []
        for _ in range(15):
            ctx.sync(); t0 = time.perf_counter()
            pa.prove(pk, ck, dw, dpi)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"2^{k} gates table c={c:2d}: best {ts[0]*1e3:6.3f} ms  median {ts[len(ts)//2]*1e3:6.3f} ms", flush=True)
        pk.free(); ck._bases.free()
    dw.free()
