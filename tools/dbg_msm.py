import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
from conftest import hex_to_fr_mont, points_to_mont
o = CpuOracle(); ctx = pa.Context(0)
g = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "msm.json")))
for v in g:
    pts = points_to_mont(o, v["points"]); sc = hex_to_fr_mont(o, v["scalars"])
    got, ident = pa.g1_to_affine(pa.msm_variable_base(pts, sc, ctx))
    exp = o.g1_msm(pts, sc, 0, 1)
    print(v["n"], np.array_equal(got, exp), ident, flush=True)
for c in (5, 8, 16):
    ctx.set_option("msm_window_bits", c)
    pts = o.g1_bases_arith(ints_to_limbs([5], 4)[0], ints_to_limbs([7], 4)[0], 1, 1); sc = o.fr_sample(3, 1)
    got, _ = pa.g1_to_affine(pa.msm_variable_base(pts, sc, ctx)); print("c", c, np.array_equal(got, o.g1_msm(pts, sc)))
