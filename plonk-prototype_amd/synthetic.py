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

from .field import R_MOD, fr_vec_to_limbs
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
