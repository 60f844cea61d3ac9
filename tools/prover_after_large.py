"""Does a large allocation history slow later proofs down?  Times prove() at 2^20 before and after a
2^24 MSM with its 18 GB window table has come and gone."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402

ctx = pa.Context(0)
n = 1 << 20
tau = pa.field.fr_to_limbs(0xABCDEF123)
ck = pa.CommitKey.setup(n - 1, tau, ctx, precompute=True)
circuit, dw, _ = pa.synthetic.wide_circuit(n, ctx, 1)
pk = pa.preprocess(circuit, ctx)


def timed(label):
    pa.prove(pk, ck, dw, None)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        pa.prove(pk, ck, dw, None)
    ctx.sync()
    print(f"{label}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per proof", flush=True)


timed("fresh")
big = pa.CommitKey.setup((1 << 24) - 1, tau, ctx, precompute=True)
sc = torch.from_numpy(np.random.default_rng(1).integers(0, 1 << 62, size=(1 << 24, 4), dtype=np.uint64).view(np.int64)).cuda()
big._bases.msm_dev(sc.data_ptr(), 1 << 24)
timed("with the 2^24 key alive")
big._bases.free()
del sc
torch.cuda.empty_cache()
timed("after freeing it")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    pa.prove(pk, ck, dw, None)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
