"""GPU sweep (not a test): MSM with the resident-SRS table, window width x bucket chunk."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
o = CpuOracle()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
ctx = pa.Context(0)
for kv in os.environ.get("PM_OPTS", "").split(","):          # PM_OPTS=msm_reduce_wg=1,...
    if kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
pts = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0xabcdef123456789abcdef], 4)[0], n, 16)
sc = o.fr_sample(0x5343414C, n)
d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
ref = None
cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (17, 18, 19, 20, 21)
for c in cs:
    bases = pa.host.Bases(ctx, pts).precompute(c)
    for lb in ([int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (4, 8, 16, 32)):
        ctx.set_option("msm_lb", lb)
        r = bases.msm_dev(d_sc.data_ptr(), n)
        if ref is None: ref = r
        assert np.array_equal(r, ref)
        ctx.sync(); ctx.profile(True)
        t0 = time.perf_counter()
        for _ in range(3): bases.msm_dev(d_sc.data_ptr(), n)
        dt = (time.perf_counter() - t0) / 3
        prof = ctx.profile_read(); ctx.profile(False)
        ks = {s.replace("msm_", ""): round(v[1] / 3 * 1e3) for s, v in prof.items()}
        print(f"c={c} lb={lb:2d} {dt*1e3:7.3f} ms  {ks}", flush=True)
    ctx.set_option("msm_lb", 0)
    bases.free()

# skewed scalar distributions (long runs in a few buckets): all equal, and 30 % equal to one value
if len(sys.argv) > 4 and sys.argv[4] == "skew":
    bases = pa.host.Bases(ctx, pts).precompute(cs[-1])
    rs = np.random.default_rng(7)
    for tag, frac in (("all-equal", 1.0), ("30%-equal", 0.3), ("1%-one", 0.0)):
        s2 = sc.copy()
        if frac:
            s2[rs.random(n) < frac] = sc[0]
        else:
            s2[rs.random(n) < 0.01] = o.fr_to_mont(ints_to_limbs([1], 4))[0]
        d2 = torch.from_numpy(np.ascontiguousarray(s2).view(np.int64)).cuda()
        r = bases.msm_dev(d2.data_ptr(), n)
        if n <= 1 << 17:
            assert np.array_equal(pa.g1_to_affine(r)[0], o.g1_msm(pts, s2, 0, 16)), tag
        ctx.sync(); ctx.profile(True)
        t0 = time.perf_counter()
        for _ in range(3): bases.msm_dev(d2.data_ptr(), n)
        dt = (time.perf_counter() - t0) / 3
        prof = ctx.profile_read(); ctx.profile(False)
        ks = {s.replace("msm_", ""): round(v[1] / 3 * 1e3) for s, v in prof.items()}
        print(f"{tag:10s} c={cs[-1]} {dt*1e3:7.3f} ms  {ks}", flush=True)
