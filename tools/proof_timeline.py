"""GPU probe (not a test): one proof of a small circuit, for a kernel timeline.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/proof_timeline.py LOG_N
then            :  python3 tools/proof_timeline.py --report OUT
The driver proves three times, sleeps 0.2 s and proves once more; the report lists the kernels after the last long gap
with their start offset, duration and the idle time in front of each."""
import os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import csv, glob
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 100_000_000:
            cut = i
    rows = rows[cut:]
    t0 = int(rows[0]["Start_Timestamp"]); prev = t0; busy = 0; agg = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("pm::", "")[:48]
        print(f"{(s - t0) / 1e3:9.1f} us  +{max(0, s - prev) / 1e3:6.1f} idle  {(e - s) / 1e3:8.1f} us  {name}")
        busy += e - s; prev = max(prev, e)
        a = agg.setdefault(name, [0, 0]); a[0] += 1; a[1] += e - s
    print(f"span {(prev - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(rows)} launches")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {k:50s} x{c:3d} {t / 1e3:9.1f} us")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = 1 << k
o = CpuOracle()
ctx = pa.Context(0)
circuit, wit, pi = pa.synthetic.chain_circuit(n, 1)
srs = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, threads=16)
ck = pa.CommitKey(srs, ctx, precompute=True)
pk = pa.preprocess(circuit, ctx, ck)
dw = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
dpi = pa.prover.sparse_public_inputs(pi)
for _ in range(3):
    pa.prove(pk, ck, dw, dpi)
ctx.sync(); time.sleep(0.2)
t0 = time.perf_counter()
pa.prove(pk, ck, dw, dpi)
ctx.sync()
print(f"prove 2^{k}: {(time.perf_counter() - t0) * 1e3:.3f} ms (under the profiler)")
