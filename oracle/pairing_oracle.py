"""TEST INFRASTRUCTURE ONLY -- a slow, plain-Python BLS12-381 pairing, so that the tests can run the
verifier's KZG pairing checks on proofs produced by the GPU prover (the reference verifies with
``dusk_bls12_381::multi_miller_loop`` / ``final_exponentiation``; the verifier itself is out of scope
for the product).

Written for obviousness, not speed: F_p12 = F_p[w] / (w^12 - 2 w^6 + 2) as 12 coefficients
(u = w^6 - 1 is the F_p2 generator with u^2 = -1, the sextic twist constant is w^6 = 1 + u), points
of E'(F_p2): y^2 = x^3 + 4 (1 + u) are untwisted to E(F_p12): y^2 = x^3 + 4 by (x / w^2, y / w^3),
the Miller loop runs over |x| = 0xd201000000010000 with affine line functions, and the final
exponentiation is one big ``pow``.  The sign of x is ignored: the map is then the inverse of the
optimal ate pairing, still bilinear and non-degenerate, which is all a pairing-equation check needs.

Not pinned by constants from memory: tests/test_oracle_pairing.py checks that the G2 generator used
here lies on the twist and has order r, and that the map is bilinear and non-degenerate; those
properties are what the KZG verification equation relies on.
"""
from __future__ import annotations

from .bigint_oracle import G1_GEN, P_MOD, R_MOD

P = P_MOD
ATE_LOOP = 0xD201000000010000

# G2 generator (x = x0 + x1 u, y = y0 + y1 u), the standard one of the BLS12-381 specification
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)


# ------------------------------------------------------------------ F_p12 as polynomials in w
def f12(c0=0):
    return [c0 % P] + [0] * 11


F12_ONE = f12(1)


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_neg(a):
    return [(-x) % P for x in a]


def f12_scal(a, k):
    return [x * k % P for x in a]


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for i in range(22, 11, -1):          # w^12 = 2 w^6 - 2
        c = t[i]
        if c:
            t[i - 6] += 2 * c
            t[i - 12] -= 2 * c
    return [x % P for x in t[:12]]


def f12_is_zero(a):
    return not any(a)


def _poly_deg(a):
    d = len(a) - 1
    while d >= 0 and a[d] == 0:
        d -= 1
    return d


def _poly_divmod(num, den):
    """Quotient and remainder of polynomials over F_p (lists, lowest degree first)."""
    num = list(num)
    dd = _poly_deg(den)
    inv_lead = pow(den[dd], -1, P)
    q = [0] * max(len(num) - dd, 1)
    for i in range(_poly_deg(num) - dd, -1, -1):
        c = num[i + dd] * inv_lead % P
        if c:
            q[i] = c
            for j in range(dd + 1):
                num[i + j] = (num[i + j] - c * den[j]) % P
    return q, num


def _poly_mul_sub(s0, q, s1):
    """s0 - q * s1."""
    out = list(s0) + [0] * max(0, len(q) + len(s1) - len(s0))
    for i, x in enumerate(q):
        if x:
            for j, y in enumerate(s1):
                out[i + j] = (out[i + j] - x * y) % P
    return out


def f12_inv(a):
    """Extended Euclid over F_p[w]: a^-1 mod (w^12 - 2 w^6 + 2)."""
    r0, r1 = [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1], list(a)
    s0, s1 = [0], [1]
    if _poly_deg(r1) < 0:
        raise ZeroDivisionError("F_p12 inverse of zero")
    while _poly_deg(r1) >= 0:
        q, rem = _poly_divmod(r0, r1)
        r0, r1 = r1, rem
        s0, s1 = s1, _poly_mul_sub(s0, q, s1)
    k = pow(r0[0], -1, P)                       # the gcd is a non-zero constant: the modulus is irreducible
    out = [x * k % P for x in s0[:12]]
    return out + [0] * (12 - len(out))


def f12_pow(a, e: int):
    r, b = F12_ONE, a
    while e:
        if e & 1:
            r = f12_mul(r, b)
        b = f12_mul(b, b)
        e >>= 1
    return r


W = [0, 1] + [0] * 10
W2_INV = f12_inv(f12_mul(W, W))
W3_INV = f12_inv(f12_mul(f12_mul(W, W), W))


def f2_to_f12(a):
    """a0 + a1 u with u = w^6 - 1."""
    a0, a1 = a
    c = [0] * 12
    c[0], c[6] = (a0 - a1) % P, a1 % P
    return c


# ------------------------------------------------------------------ F_p2 and the twist E'(F_p2)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


TWIST_B = (4, 4)


def g2_is_on_curve(Q) -> bool:
    if Q is None:
        return True
    x, y = Q
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), TWIST_B)) == (0, 0)


def g2_add(A, B):
    if A is None:
        return B
    if B is None:
        return A
    (x1, y1), (x2, y2) = A, B
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(k: int, Q):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, Q)
    return acc


# ------------------------------------------------------------------ E(F_p12) and the Miller loop
def untwist(Q):
    x, y = Q
    return (f12_mul(f2_to_f12(x), W2_INV), f12_mul(f2_to_f12(y), W3_INV))


def _e12_double(R):
    x, y = R
    lam = f12_mul(f12_scal(f12_mul(x, x), 3), f12_inv(f12_scal(y, 2)))
    x3 = f12_sub(f12_mul(lam, lam), f12_scal(x, 2))
    return (x3, f12_sub(f12_mul(lam, f12_sub(x, x3)), y)), lam


def _e12_add(R, Q):
    (x1, y1), (x2, y2) = R, Q
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    x3 = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    return (x3, f12_sub(f12_mul(lam, f12_sub(x1, x3)), y1)), lam


def _line(lam, R, Pt):
    """The line through R with slope lam, evaluated at Pt."""
    return f12_sub(f12_mul(lam, f12_sub(Pt[0], R[0])), f12_sub(Pt[1], R[1]))


def miller_loop(Q2, P1):
    """Q2 on the twist (F_p2 coordinates), P1 in G1 (ints).  Either None -> 1."""
    if Q2 is None or P1 is None:
        return F12_ONE
    Q = untwist(Q2)
    Pt = (f12(P1[0]), f12(P1[1]))
    R, f = Q, F12_ONE
    for bit in bin(ATE_LOOP)[3:]:
        R2, lam = _e12_double(R)
        f = f12_mul(f12_mul(f, f), _line(lam, R, Pt))
        R = R2
        if bit == "1":
            R2, lam = _e12_add(R, Q)
            f = f12_mul(f, _line(lam, R, Pt))
            R = R2
    return f


FINAL_EXP = (P ** 12 - 1) // R_MOD


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(Q2, P1):
    return final_exponentiation(miller_loop(Q2, P1))


def pairing_product_is_one(pairs) -> bool:
    """prod_i e(Q_i, P_i) == 1 with one final exponentiation."""
    f = F12_ONE
    for Q2, P1 in pairs:
        f = f12_mul(f, miller_loop(Q2, P1))
    return final_exponentiation(f) == F12_ONE


G1 = G1_GEN
