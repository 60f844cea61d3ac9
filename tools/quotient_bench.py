"""GPU timing (not a test): the fused quotient kernel at n = 2^K gates (4n coset points) on random operands, for the
arithmetic-only circuit (quotient_kernel<false>) and with widget selectors present (quotient_kernel<true>).
usage: quotient_bench.py [K=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import plonk_prototype_amd as pa
from plonk_prototype_amd import _lib
from oracle.cpu_oracle import CpuOracle
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, n4 = 1 << k, 4 << k
ctx, o = pa.Context(0), CpuOracle()
names = ["w0", "w1", "w2", "w3", "z", "q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "pi", "s0", "s1", "s2", "s3", "l1", "x",
         "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"]
d = {nm: pa.DeviceVector.from_host(ctx, o.fr_sample(100 + i, n4)) for i, nm in enumerate(names)}
out = pa.DeviceVector(ctx, n4)
one = o.fr_sample(1, 1)[0]
u = lambda: (C.c_uint64 * 4)(*[int(t) for t in one])   # noqa: E731
for label, widgets in (("arithmetic only", ()), ("+ q_arith", ("q_arith",)), ("+ q_arith, q_range", ("q_arith", "q_range")),
                       ("+ q_arith, q_logic", ("q_arith", "q_logic")), ("+ q_arith, q_fixed_group_add", ("q_arith", "q_fixed_group_add")),
                       ("+ q_arith, q_variable_group_add", ("q_arith", "q_variable_group_add")),
                       ("all five", ("q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"))):
    qa = _lib.QuotientArgs()
    for j in range(4):
        qa.wires[j], qa.sigmas[j] = d[f"w{j}"].ptr, d[f"s{j}"].ptr
    for nm in ("z", "q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "pi", "l1", "x") + tuple(widgets):
        setattr(qa, nm, d[nm].ptr)
    qa.alpha, qa.beta, qa.gamma = u(), u(), u()
    qa.range_sep, qa.logic_sep, qa.fixed_sep, qa.var_sep = u(), u(), u(), u()
    for j in range(3):
        qa.k[j] = u()
    for j in range(4):
        qa.zh_inv[j] = u()
    for _ in range(3):
        ctx.plonk_quotient(qa, n, out.ptr)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.plonk_quotient(qa, n, out.ptr)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    arrays = 19 + len(widgets)
    print(f"2^{k} gates, {label:34s} {dt * 1e3:7.3f} ms   {arrays} operand arrays + 1 result: {(arrays + 1) * 32 * n4 / dt / 1e12:.2f} TB/s", flush=True)
