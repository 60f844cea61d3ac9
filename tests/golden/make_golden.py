"""Generates tests/golden/*.json with the pure-Python big-int oracle (oracle/bigint_oracle.py).

No reference code is involved: the reference (/root/reference) ships no NTT/MSM code, tests
or vectors, so these are first-principles vectors: O(n^2) DFT
definitions and double-and-add MSMs.  Values are hex strings of CANONICAL integers; the
tests convert to Montgomery limbs.   Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bigint_oracle as B  # noqa: E402


def hx(v):
    return format(v, "x")


def ntt_vectors():
    out = []
    for k, in_len in [(0, 1), (1, 2), (2, 4), (2, 3), (3, 8), (3, 5), (4, 16), (5, 32), (6, 64), (8, 256), (8, 100)]:
        a = B.sample_fr(0x474F4C44 + 131 * k + in_len, in_len)
        d = B.Domain(1 << k)
        n = 1 << k
        pad = a + [0] * (n - in_len)
        gi = pow(B.FR_GENERATOR, -1, B.R_MOD)
        fwd = B.naive_dft(pad, d.group_gen)
        inv = [x * d.size_inv % B.R_MOD for x in B.naive_dft(pad, d.group_gen_inv)]
        cfwd = B.naive_dft([x * pow(B.FR_GENERATOR, i, B.R_MOD) % B.R_MOD for i, x in enumerate(pad)], d.group_gen)
        cinv = [x * pow(gi, i, B.R_MOD) % B.R_MOD for i, x in enumerate(inv)]
        out.append(dict(log_n=k, input=[hx(v) for v in a], fft=[hx(v) for v in fwd], ifft=[hx(v) for v in inv],
                        coset_fft=[hx(v) for v in cfwd], coset_ifft=[hx(v) for v in cinv]))
    return out


def msm_vectors():
    out = []
    G = B.G1_GEN
    for n, seed in [(0, 1), (1, 2), (2, 3), (5, 4), (31, 5), (32, 6), (33, 7), (64, 8)]:
        ks = [v % (1 << 64) + 1 for v in B.sample_fr(1000 + seed, n)]
        pts = [B.g1_mul(k, G) for k in ks]
        sc = B.sample_fr(2000 + seed, n)
        if n >= 5:
            sc[0] = 0
            sc[1] = 1
            sc[2] = B.R_MOD - 1
            pts[3] = None                 # identity among the bases
            pts[4] = pts[1]               # duplicate base
        if n >= 31:
            pts[7] = B.g1_neg(pts[6])     # P and -P ...
            sc[7] = sc[6]                 # ... with equal scalars (cancels)
            sc[9] = sc[8] = 5             # small equal scalars
        res = B.naive_msm(pts, sc)
        out.append(dict(n=n,
                        points=[None if p is None else [hx(p[0]), hx(p[1])] for p in pts],
                        scalars=[hx(s) for s in sc],
                        result=None if res is None else [hx(res[0]), hx(res[1])]))
    return out


def constants():
    return dict(
        r=hx(B.R_MOD), p=hx(B.P_MOD), fr_R=hx(B.FR_MONT_R), fr_R2=hx(B.FR_MONT_R2),
        fr_R3=hx(B.FR_MONT_R2 * B.FR_MONT_R % B.R_MOD), fr_inv64=hx(B.FR_INV64), fp_R=hx(B.FP_MONT_R),
        fp_inv64=hx(B.FP_INV64), root_of_unity=hx(B.ROOT_OF_UNITY),
        omega={str(k): hx(B.Domain(1 << k).group_gen) for k in (12, 20, 22, 24, 26)},
        size_inv={str(k): hx(B.Domain(1 << k).size_inv) for k in (20, 24)},
        g1_compressed=B.g1_compress(B.G1_GEN).hex())


if __name__ == "__main__":
    for name, fn in [("ntt", ntt_vectors), ("msm", msm_vectors), ("constants", constants)]:
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=0)
        print("wrote", name)
