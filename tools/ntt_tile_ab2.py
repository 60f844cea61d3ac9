"""GPU A/B (not a test), r05: tile width of the radix-2^10 passes with the chain product -- 4 columns per workgroup (one 1024-thread
workgroup per CU) against 2 (two 512-thread workgroups per CU whose barrier phases can overlap); timers off, alternating.
usage: python tools/ntt_tile_ab2.py [log_n,batch ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle
ctx = pa.Context(0)
o = CpuOracle()
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(20, 1), (20, 4), (22, 1), (22, 5), (24, 1)]
for k, batch in cases:
    n = 1 << k
    a = torch.from_numpy(o.fr_sample(k, n * batch).view(np.int64)).cuda()
    b, ref = torch.empty_like(a), None
    res = {0: [], 10: [], 11: [], 12: []}
    for rnd in range(5):
        for tile in (0, 10, 11, 12):
            ctx.set_option("ntt_tile_log", tile)
            for _ in range(3):
                ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0, batch=batch)
                ctx.fr_ntt_dev(b.data_ptr(), n, b.data_ptr(), k, 1, batch=batch)
            ctx.sync()
            if ref is None:
                ref = b.clone()
            assert torch.equal(b, ref), (k, tile)
            reps = 40 if k <= 20 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0, batch=batch)
                ctx.fr_ntt_dev(b.data_ptr(), n, b.data_ptr(), k, 1, batch=batch)
            ctx.sync()
            res[tile].append((time.perf_counter() - t0) / reps / batch * 1e6)
    print(f"2^{k} batch {batch} plan={pa.ntt_plan(k)}: us per forward + inverse, median of 5: " +
          "  ".join(f"tile_log {t}: {sorted(v)[2]:.1f}" for t, v in res.items()), flush=True)
ctx.set_option("ntt_tile_log", 0)
