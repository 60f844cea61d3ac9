"""How much of a forward+inverse 2^k NTT step is launch gap?  Times the bench step with the per-kernel
event timers on, off, and replayed from a HIP graph (captured through torch's graph API on the stream
the library launches on)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
steps = 200
ctx = pa.Context(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
d_a = torch.from_numpy(a.view(np.int64)).to(dev)
d_b, d_c = torch.empty_like(d_a), torch.empty_like(d_a)


def step(stream=0):
    ctx.fr_ntt_dev(d_a.data_ptr(), n, d_b.data_ptr(), k, 0, stream=stream)
    ctx.fr_ntt_dev(d_b.data_ptr(), n, d_c.data_ptr(), k, pa.NTT_INVERSE, stream=stream)


def timed(fn, label):
    for _ in range(10):
        fn()
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label:28s} {dt * 1e6:8.1f} us/step   {n * k / dt:.3e} butterflies/s", flush=True)


timed(step, "events off, ctx stream")
ctx.profile(True)
timed(step, "events on")
ctx.profile_read()
ctx.profile(False)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sp = s.cuda_stream
    timed(lambda: step(sp), "events off, torch stream")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step(s.cuda_stream)
    timed(g.replay, "graph replay")
    ref = d_c.clone()
    step(sp)
    torch.cuda.synchronize()
    print("graph result equals eager:", torch.equal(ref, d_c), "roundtrip:", torch.equal(d_c, d_a))
