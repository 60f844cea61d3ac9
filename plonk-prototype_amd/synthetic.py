"""Synthetic circuits for the prover path (the reference has no runnable
circuit of benchmark size, so configs[3]/[4] of BASELINE.json are quoted on a synthetic 2^k-gate
circuit).  Random-init selectors, a satisfying witness, a non-trivial copy permutation.
``chain_circuit`` / ``wide_circuit`` are arithmetic gates only; ``mixed_circuit`` adds blocks of the
gate kinds the reference's gadgets emit through dusk-plonk's composer -- range quads, logic AND / XOR,
fixed-base scalar multiplication rounds and variable-base curve additions on JubJub
(ref:src/zk/gadgets.rs:34,37,40,88-91,211; ref:src/zk/circuits.rs:64-70).

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


def chain_circuit(n: int, seed: int = 1, public_rows=(0,), zero_selectors=()):
    """-> (Circuit, witness [4, n, 4], public_inputs [n, 4]) for a power-of-two n >= 4; PI is non-zero on `public_rows`;
    the selectors named in `zero_selectors` (of q_m q_l q_r q_4 q_c) are identically zero."""
    rng = random.Random(seed)
    rnd = lambda: rng.getrandbits(256) % R_MOD   # noqa: E731
    q = {k: [rnd() for _ in range(n)] for k in ("q_m", "q_l", "q_r", "q_4", "q_c")}
    for k in zero_selectors:
        q[k] = [0] * n
    pi = [0] * n
    for r_ in public_rows:
        pi[r_] = rnd()
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
                      q_arith=np.tile(fr_to_limbs(1), (n, 1)), sigma_index=sigma.reshape(4, n))
    return circuit, witness, fr_vec_to_limbs(pi)


# ------------------------------------------------------------------------------------------------
# JubJub (the curve embedded in Fr that the reference's gadgets work on): -x^2 + y^2 = 1 + d x^2 y^2
EDWARDS_D = (-(10240 * pow(10241, -1, R_MOD))) % R_MOD


def _fr_sqrt(a: int):
    """Tonelli-Shanks in Fr (r - 1 = 2^32 t); None for a non-residue."""
    a %= R_MOD
    if a == 0:
        return 0
    if pow(a, (R_MOD - 1) // 2, R_MOD) != 1:
        return None
    s, q = 32, (R_MOD - 1) >> 32
    z = pow(7, q, R_MOD)                       # 7 generates Fr*: a non-residue
    m, c, t, r = s, z, pow(a, q, R_MOD), pow(a, (q + 1) // 2, R_MOD)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % R_MOD
            i += 1
        b = pow(c, 1 << (m - i - 1), R_MOD)
        m, c = i, b * b % R_MOD
        t, r = t * c % R_MOD, r * b % R_MOD
    return r


def jubjub_add(p, q):
    (x1, y1), (x2, y2) = p, q
    k = EDWARDS_D * x1 % R_MOD * x2 % R_MOD * y1 % R_MOD * y2 % R_MOD
    x3 = (x1 * y2 + y1 * x2) * pow(1 + k, -1, R_MOD) % R_MOD
    y3 = (y1 * y2 + x1 * x2) * pow(1 - k, -1, R_MOD) % R_MOD
    return x3, y3


def jubjub_point(seed: int):
    """Some point of the curve: the first y >= seed with a square (y^2 - 1) / (1 + d y^2)."""
    y = seed % R_MOD
    while True:
        x = _fr_sqrt((y * y - 1) * pow(1 + EDWARDS_D * y * y, -1, R_MOD))
        if x:
            assert (-x * x + y * y - 1 - EDWARDS_D * x * x % R_MOD * y * y) % R_MOD == 0
            return x, y
        y += 1


def mixed_circuit(n: int, seed: int = 1):
    """-> (Circuit, witness [4, n, 4], public_inputs [n, 4]) with every gate kind of dusk-plonk's composer.

    Rows (the widget blocks sit in front, arithmetic chain gates fill the rest; ``*`` = the row only
    carries the accumulators the previous row's "next" terms refer to):
      range      2 rows + *   quads of a 16-bit value, accumulators d -> c -> b -> a -> d_next
      logic      2 rows AND + *, 2 rows XOR + *   quads of two 4-bit operands
      fixed-base 4 rounds + *  bits -1 / 0 / 1 / 1 against table points 2^k B
      var-base   2 additions: (row, * with the sum)
    The widget rows' wires are free variables except where a "next" row is shared; the chain gates are
    tied by copy constraints as in chain_circuit."""
    if n < 32 or n & (n - 1):
        raise ValueError("mixed_circuit needs a power of two >= 32")
    rng = random.Random(seed)
    rnd = lambda: rng.getrandbits(256) % R_MOD   # noqa: E731
    names = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
             "q_variable_group_add")
    rows = []          # (selector dict, [a, b, c, d])

    def row(wires, **sel):
        rows.append((sel, [w % R_MOD for w in wires]))

    # range: value = 8 quads, most significant first; acc_{k+1} = 4 acc_k + quad_k, d of the first row = 0
    quads = [rng.randrange(4) for _ in range(8)]
    acc = [0]
    for q in quads:
        acc.append(4 * acc[-1] + q)
    row([acc[3], acc[2], acc[1], acc[0]], q_range=1)
    row([acc[7], acc[6], acc[5], acc[4]], q_range=1)
    row([0, 0, 0, acc[8]])
    # logic: rows hold accumulators of the a-input (a), b-input (b), output (d) and the product of the quads (c)
    for q_c, op in ((1, lambda x, y: x & y), (R_MOD - 1, lambda x, y: x ^ y)):
        qa, qb = [rng.randrange(4) for _ in range(2)], [rng.randrange(4) for _ in range(2)]
        aa, ab, ad = [0], [0], [0]
        for x, y in zip(qa, qb):
            aa.append(4 * aa[-1] + x)
            ab.append(4 * ab[-1] + y)
            ad.append(4 * ad[-1] + op(x, y))
        for k in range(2):
            row([aa[k], ab[k], qa[k] * qb[k], ad[k]], q_logic=1, q_c=q_c)
        row([aa[2], ab[2], 0, ad[2]])
    # fixed-base scalar multiplication: acc += bit * 2^k B (bits in {-1, 0, 1}); table point in q_l, q_r, q_c
    base = jubjub_point(0x1234567 + seed)
    bits = [R_MOD - 1, 0, 1, 1]
    pt, accp, accb = base, jubjub_point(0x7654321 + seed), 0
    for bit in bits:
        xb, yb = pt
        sb = 1 if bit == 1 else (-1 if bit else 0)
        x_alpha, y_alpha = xb * sb % R_MOD, (sb * sb * (yb - 1) + 1) % R_MOD
        row([accp[0], accp[1], sb * xb * yb, accb], q_fixed_group_add=1, q_l=xb, q_r=yb, q_c=xb * yb)
        accp = jubjub_add(accp, (x_alpha, y_alpha))
        accb = (2 * accb + sb) % R_MOD
        pt = jubjub_add(pt, pt)
    row([accp[0], accp[1], 0, accb])
    # variable-base additions
    for k in range(2):
        p1, p2 = jubjub_point(0xABCDEF + 17 * k + seed), jubjub_point(0xFEDCBA + 31 * k + seed)
        p3 = jubjub_add(p1, p2)
        row([p1[0], p1[1], p2[0], p2[1]], q_variable_group_add=1)
        row([p3[0], p3[1], 0, p1[0] * p2[1]])
    # arithmetic chain for the rest (local recurrence as in chain_circuit, gate g at row w0 + g)
    w0 = len(rows)
    m = n - w0
    pi = [0] * n
    pi[w0] = rnd()
    v = [0] * (m + 1)
    v[0] = rnd()
    for g in range(m):
        qm, ql, qr, q4, qc = rnd(), rnd(), rnd(), rnd(), rnd()
        a, b, d = v[g], v[g - 1 if g else 0], v[g - 2 if g >= 2 else 0]
        v[g + 1] = (qm * a % R_MOD * b + ql * a + qr * b + q4 * d + qc + pi[w0 + g]) % R_MOD
        row([a, b, v[g + 1], d], q_m=qm, q_l=ql, q_r=qr, q_o=R_MOD - 1, q_4=q4, q_c=qc, q_arith=1)
    sel = {k: fr_vec_to_limbs([r[0].get(k, 0) for r in rows]) for k in names}
    witness = np.stack([fr_vec_to_limbs([r[1][j] for r in rows]) for j in range(4)])
    # variables: every widget position is its own variable; chain positions share the chain's variables
    var = np.arange(4 * n, dtype=np.int64) + 4 * n                 # unique ids by default
    g = np.arange(m, dtype=np.int64)
    for j, idx in enumerate((g, np.maximum(g - 1, 0), g + 1, np.maximum(g - 2, 0))):
        var[j * n + w0:j * n + n] = idx
    order = np.argsort(var, kind="stable")
    sv = var[order]
    nxt = np.roll(order, -1)
    starts = np.flatnonzero(np.r_[True, sv[1:] != sv[:-1]])
    ends = np.r_[starts[1:] - 1, var.size - 1]
    nxt[ends] = order[starts]
    sigma = np.empty(4 * n, np.int64)
    sigma[order] = nxt
    return Circuit(sigma_index=sigma.reshape(4, n), **sel), witness, fr_vec_to_limbs(pi)


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
                      q_4=q["q_4"], q_c=q["q_c"], q_arith=np.tile(fr_to_limbs(1), (n, 1)),
                      sigma_index=sigma.reshape(4, n))
    return circuit, wit, None


def wide_mixed_circuit(n: int, ctx: Context, seed: int = 1):
    """wide_circuit on the lower half of the rows; the upper half holds zero rows (every wire 0, copy permutation the
    identity there) on which each of the four widget selectors is switched on over its own block -- zero rows
    satisfy the range, logic, fixed-base and variable-base identities (every quad, bit and product is 0), so the
    circuit is valid and a proof of it runs the full widget arithmetic of the quotient and linearisation at 4n
    resp. n points: the cost of a real dusk-plonk circuit's gate mix at a size no per-gate generator reaches in
    seconds.  The last rows of every block stay plain (the widgets read the NEXT row's wires).
    -> (Circuit, witness DeviceVector [a | b | c | d], None)."""
    if n < 64 or n & (n - 1):
        raise ValueError("n must be a power of two >= 64")
    h = n // 2
    base, wit_h, _ = wide_circuit(h, ctx, seed)
    zero = np.zeros((h, 4), np.uint64)
    up = lambda a: np.concatenate([a, zero])                           # noqa: E731
    one = fr_to_limbs(1)
    blk = h // 4

    def block_selector(k):
        q = np.zeros((n, 4), np.uint64)
        q[h + k * blk: h + (k + 1) * blk - 2] = one
        return q

    sig = np.empty((4, n), np.int64)
    for j in range(4):
        lo = base.sigma_index[j]
        sig[j, :h] = (lo // h) * n + (lo % h)                          # j' h + i' -> j' n + i'
        sig[j, h:] = j * n + np.arange(h, n)
    q_arith = np.concatenate([base.q_arith, zero])
    circuit = Circuit(q_m=up(base.q_m), q_l=up(base.q_l), q_r=up(base.q_r), q_o=up(base.q_o), q_4=up(base.q_4), q_c=up(base.q_c),
                      q_arith=q_arith, q_range=block_selector(0), q_logic=block_selector(1),
                      q_fixed_group_add=block_selector(2), q_variable_group_add=block_selector(3), sigma_index=sig)
    w = wit_h.to_host().reshape(4, h, 4)
    wit_h.free()
    full = np.zeros((4, n, 4), np.uint64)
    full[:, :h] = w
    return circuit, DeviceVector.from_host(ctx, full.reshape(4 * n, 4)), None
