"""cProfile of prove() host side.  usage: python tools/prover_profile.py LOG_N"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402
from oracle.cpu_oracle import CpuOracle, ints_to_limbs  # noqa: E402  (input synthesis only)

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
orc = CpuOracle()
circuit, wit, pi = pa.synthetic.chain_circuit(n, 1)
srs = orc.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, threads=16)
ctx = pa.Context(0)
ck = pa.CommitKey(srs, ctx, precompute=True)
pk = pa.preprocess(circuit, ctx)
dw = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
dpi = pa.DeviceVector.from_host(ctx, pi)
pa.prove(pk, ck, dw, dpi)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    pa.prove(pk, ck, dw, dpi)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
