"""GPU probe for the PMC passes of the 2^24 shapes (tools/collect_pmc.sh, prefix big_): one forward + inverse 2^24 transform
and one 2^24-point MSM over the resident table -- the launches behind bench.py's ntt_extra.fwd_inv_2^24 and msm_large rooflines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << k
o = CpuOracle()
ctx = pa.Context(0)
x = torch.from_numpy(o.fr_sample(0x504C4F4E4B, n).view(np.int64)).cuda()
y = torch.empty_like(x)
for _ in range(2):
    ctx.fr_ntt_dev(x.data_ptr(), n, y.data_ptr(), k, 0)
    ctx.fr_ntt_dev(y.data_ptr(), n, x.data_ptr(), k, 1)
ctx.sync()
pts = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0xabcdef123456789abcdef], 4)[0], n, 16)
bases = pa.host.Bases(ctx, pts).precompute()
for _ in range(2):
    bases.msm_dev(x.data_ptr(), n)
ctx.sync()
print("done")
