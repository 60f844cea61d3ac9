import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle
ctx = pa.Context(0); o = CpuOracle()
for k, batch in ((20, 1), (20, 4), (22, 1), (24, 1), (16, 4)):
    n = 1 << k
    a = torch.from_numpy(o.fr_sample(k, n * batch).view(np.int64)).cuda()
    b, ref = torch.empty_like(a), None
    res = {}
    for rnd in range(3):
        for radix, tile in ((4, 0), (8, 0), (8, 11), (8, 12)):
            ctx.set_option("ntt_radix", radix); ctx.set_option("ntt_tile_log", tile)
            try:
                for _ in range(3):
                    ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0, batch=batch)
                    ctx.fr_ntt_dev(b.data_ptr(), n, b.data_ptr(), k, 1, batch=batch)
                ctx.sync()
                if ref is None: ref = b.clone()
                assert torch.equal(b, ref), (k, radix)
                reps = 40 if k <= 20 else 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0, batch=batch)
                    ctx.fr_ntt_dev(b.data_ptr(), n, b.data_ptr(), k, 1, batch=batch)
                ctx.sync()
                res.setdefault((radix, tile), []).append((time.perf_counter() - t0) / reps / batch * 1e6)
            except Exception as e:
                res.setdefault((radix, tile), []).append(float("nan"))
    print(f"2^{k} batch {batch}: " + "  ".join(f"radix {r} tile {t}: {sorted(v)[1]:.1f}" for (r, t), v in res.items()), flush=True)
