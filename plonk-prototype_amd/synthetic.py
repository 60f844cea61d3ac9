"""Synthetic arithmetic circuits for the prover path (the reference has no
runnable circuit of benchmark size, so configs[3]/[4] of BASELINE.json are quoted on a synthetic
2^k-gate circuit).  Random-init selectors, a satisfying witness, a non-trivial copy permutation.

    gate i:   v[i+1] = q_m v[i] v[i-1] + q_l v[i] + q_r v[i-1] + q_4 v[i-2] + q_c + PI[i]
    wires:    a = v[i],  b = v[max(i-1, 0)],  c = v[i+1],  d = v[max(i-2, 0)],  q_o = -1

Every variable sits in up to four wire positions, so sigma has cycles of length 1..6 across all four
wire columns.  PI[0] is the only public input.
"""
from __future__ import annotations

import random

import numpy as np

from .field import R_MOD, fr_to_limbs, fr_vec_to_limbs
from .host import Context, DeviceVector
from .prover import Circuit


def chain_circuit(n: int, seed: int = 1):
    """-> (Circuit, witness [4, n, 4], public_inputs [n, 4]) for a power-of-two n >= 4."""
    rng = random.Random(seed)
    rnd = lambda: rng.getrandbits(256) % R_MOD   # noqa: E731
    q = {k: [rnd() for _ in range(n)] for k in ("q_m", "q_l", "q_r", "q_4", "q_c")}
    pi = [0] * n
    pi[0] = rnd()
    v = [0] * (n + 1)
    v[0] = rnd()
    qm, ql, qr, q4, qc = q["q_m"], q["q_l"], q["q_r"], q["q_4"], q["q_c"]
    for i in range(n):
        a, b, d = v[i], v[i - 1 if i else 0], v[i - 2 if i >= 2 else 0]
        v[i + 1] = (qm[i] * a % R_MOD * b + ql[i] * a + qr[i] * b + q4[i] * d + qc[i] + pi[i]) % R_MOD
    vl = fr_vec_to_limbs(v)
    ar = np.arange(n, dtype=np.int64)
    var = np.concatenate([ar, np.maximum(ar - 1, 0), ar + 1, np.maximum(ar - 2, 0)])   # variable of position j n + i
    witness = vl[var].reshape(4, n, 4)
    # sigma: each variable's positions form one cycle
    order = np.argsort(var, kind="stable")
    sv = var[order]
    nxt = np.roll(order, -1)
    starts = np.flatnonzero(np.r_[True, sv[1:] != sv[:-1]])
    ends = np.r_[starts[1:] - 1, var.size - 1]
    nxt[ends] = order[starts]
    sigma = np.empty(4 * n, np.int64)
    sigma[order] = nxt
    minus_one = fr_vec_to_limbs([R_MOD - 1])[0]
    circuit = Circuit(q_m=fr_vec_to_limbs(qm), q_l=fr_vec_to_limbs(ql), q_r=fr_vec_to_limbs(qr),
                      q_o=np.tile(minus_one, (n, 1)), q_4=fr_vec_to_limbs(q4), q_c=fr_vec_to_limbs(qc),
                      sigma_index=sigma.reshape(4, n))
    return circuit, witness, fr_vec_to_limbs(pi)


def _random_fr(rng, m: int) -> np.ndarray:
    """m uniformly random 254-bit limb patterns: each is the Montgomery form of some field element."""
    a = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(m, 4),
                                                                                             dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


def wide_circuit(n: int, ctx: Context, seed: int = 1):
    """A large circuit in seconds: independent gates over a pool of n shared variables, the output
    column computed on the GPU with the library's own vector ops (no per-gate host arithmetic).

        a = v[i],  b = v[(5 i + 1) mod n],  d = v[i // 2],  c = q_m a b + q_l a + q_r b + q_4 d + q_c

    so every pool variable sits in 2-4 positions of the a / b / d columns (copy cycles across three
    columns) and each c is its own variable.  -> (Circuit, witness DeviceVector [a | b | c | d], None)."""
    rng = np.random.default_rng(seed)
    v = _random_fr(rng, n)
    ar = np.arange(n, dtype=np.int64)
    ia, ib, idd = ar, (5 * ar + 1) % n, ar // 2
    q = {k: _random_fr(rng, n) for k in ("q_m", "q_l", "q_r", "q_4", "q_c")}
    wit = DeviceVector(ctx, 4 * n)
    A, B, Cc, D = (wit.ptr + 32 * j * n for j in range(4))
    for ptr, idx in ((A, ia), (B, ib), (D, idd)):
        col = np.ascontiguousarray(v[idx])                           # keep the gathered column alive over the call
        ctx._check(ctx._lib.pm_dev_upload(ctx._h, ptr, col.ctypes.data, n * 32))
    tmp, sel = DeviceVector(ctx, n), DeviceVector(ctx, n)
    MUL, ADD = 2, 0

    def load(name):
        ctx._check(ctx._lib.pm_dev_upload(ctx._h, sel._p, q[name].ctypes.data, n * 32))

    load("q_m")
    ctx.fr_vec_op(MUL, A, B, n, Cc, n)
    ctx.fr_vec_op(MUL, Cc, sel.ptr, n, Cc, n)
    for name, w in (("q_l", A), ("q_r", B), ("q_4", D)):
        load(name)
        ctx.fr_vec_op(MUL, sel.ptr, w, n, tmp.ptr, n)
        ctx.fr_vec_op(ADD, Cc, tmp.ptr, n, Cc, n)
    load("q_c")
    ctx.fr_vec_op(ADD, Cc, sel.ptr, n, Cc, n)
    ctx.sync()
    tmp.free()
    sel.free()
    var = np.concatenate([ia, ib, n + ar, idd])                      # variable id of position j n + i
    order = np.argsort(var, kind="stable")
    sv = var[order]
    nxt = np.roll(order, -1)
    starts = np.flatnonzero(np.r_[True, sv[1:] != sv[:-1]])
    ends = np.r_[starts[1:] - 1, var.size - 1]
    nxt[ends] = order[starts]
    sigma = np.empty(4 * n, np.int64)
    sigma[order] = nxt
    circuit = Circuit(q_m=q["q_m"], q_l=q["q_l"], q_r=q["q_r"], q_o=np.tile(fr_to_limbs(R_MOD - 1), (n, 1)),
                      q_4=q["q_4"], q_c=q["q_c"], sigma_index=sigma.reshape(4, n))
    return circuit, wit, None
