"""Time dist.FourStepNTT against the library's own plan on one GPU (world 1: the exchanges become
local transposes), and under torch.distributed.run for W ranks.  usage: [torchrun ...] four_step_bench.py LOG_N"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402
from plonk_prototype_amd.dist import FourStepNTT  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
dev = int(os.environ.get("PM_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("PM_BENCH_BACKEND", "nccl"))
ctx = pa.Context(dev)
n = 1 << log_n
rng = np.random.default_rng(rank)
x = torch.from_numpy(rng.integers(0, 1 << 62, size=(n // world, 4), dtype=np.uint64).view(np.int64)).cuda()
plan = FourStepNTT(ctx, log_n)
plan(x, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    y = plan(x, 0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
if rank == 0:
    print(f"four-step 2^{log_n} over {world} rank(s): {dt * 1e3:.3f} ms  ({(n // 2) * log_n / dt:.3e} butterflies/s, "
          f"{3 * 32 * n / world * (world - 1) / world / 1e6:.1f} MB exchanged per rank)")
    if world == 1:
        out = torch.empty_like(x)
        ctx.fr_ntt_dev(x.data_ptr(), n, out.data_ptr(), log_n, 0)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.fr_ntt_dev(x.data_ptr(), n, out.data_ptr(), log_n, 0)
        ctx.sync()
        d1 = (time.perf_counter() - t0) / reps
        print(f"library plan 2^{log_n}: {d1 * 1e3:.3f} ms; equal: {torch.equal(out, y)}")
        # the native entry point alone (no clone of the input, no torch synchronisation), per kernel
        stage = torch.empty_like(x)
        w = x.clone()
        ctx.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.fr_ntt_fourstep_dev(w.data_ptr(), stage.data_ptr(), log_n, 1, 0, 0)
        ctx.sync()
        d2 = (time.perf_counter() - t0) / reps
        prof = ctx.profile_read(); ctx.profile(False)
        print(f"pm_fr_ntt_fourstep_dev 2^{log_n}: {d2 * 1e3:.3f} ms  kernels_us=" +
              str({k: round(v[1] / v[0] * 1e3, 1) for k, v in prof.items()}))
