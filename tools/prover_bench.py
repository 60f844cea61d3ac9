"""Wall time and per-kernel time of prove() on a synthetic chain circuit.
usage: python tools/prover_bench.py LOG_N [REPS] [wide|mixed]
"wide" uses synthetic.wide_circuit (generated with the GPU's help in seconds; for 2^22 and up), "mixed"
synthetic.wide_mixed_circuit (the same on half of the rows plus blocks of rows under each widget selector: all 11
selector polynomials present, the quotient / linearisation run the full widget arithmetic).""" 
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402
from oracle.cpu_oracle import CpuOracle, ints_to_limbs  # noqa: E402  (input synthesis + checking only)

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << log_n
orc = CpuOracle()
t0 = time.time()
mode = sys.argv[3] if len(sys.argv) > 3 else ""
wide = mode in ("wide", "mixed")
ctx = pa.Context(0)
if os.environ.get("PM_MSM_PIPELINE") == "0":       # A/B: batched MSMs as one piece (r02 behaviour)
    ctx.set_option("msm_pipeline", 0)
if wide:
    circuit, dw, _ = (pa.synthetic.wide_mixed_circuit if mode == "mixed" else pa.synthetic.wide_circuit)(n, ctx, 1)
    pi = np.zeros((n, 4), np.uint64)
else:
    circuit, wit, pi = pa.synthetic.chain_circuit(n, 1)
print(f"circuit 2^{log_n}: {time.time() - t0:.1f}s host", flush=True)
k0, d = ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0]
t0 = time.time()
srs = orc.g1_bases_arith(k0, d, n, threads=16)
print(f"srs stand-in: {time.time() - t0:.1f}s host", flush=True)
ck = pa.CommitKey(srs, ctx, precompute=True)
t0 = time.time()
pk = pa.preprocess(circuit, ctx, ck)
ctx.sync()
print(f"preprocess: {time.time() - t0:.3f}s", flush=True)
if not wide:
    dw = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
del circuit
dpi = pa.prover.sparse_public_inputs(pi)
proof = pa.prove(pk, ck, dw, dpi)
pi_z = 0 if wide else pa.field.fr_from_limbs(orc.fr_poly_evaluate(orc.fr_ntt(pi, log_n, 1, threads=16),
                                                                  pa.field.fr_to_limbs(proof.challenges["z"])))
print("identity:", pa.prover.check_identity(proof, n, pi_z), flush=True)
best = 1e9
for _ in range(reps):
    ctx.sync()
    t0 = time.time()
    pa.prove(pk, ck, dw, dpi)
    ctx.sync()
    best = min(best, time.time() - t0)
print(f"prove 2^{log_n}: best {best * 1e3:.2f} ms", flush=True)
ctx.profile(True)
pa.prove(pk, ck, dw, dpi)
prof = ctx.profile_read()
ctx.profile(False)
tot = sum(ms for _, ms in prof.values())
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} x{cnt:3d}  {ms:9.3f} ms")
print(f"  kernels total {tot:.2f} ms")
