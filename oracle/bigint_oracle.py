"""Python big-int oracle for the BLS12-381 Fr NTT / G1 MSM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``plonk-prototype_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use ``oracle/`` -- as the checker, never as the product.

PARITY UNPINNED.  The reference (``/root/reference``, 448 lines of Rust) holds no
NTT/MSM code, no tests, no golden vectors, and its arithmetic lives in crates that
are not vendored (``dusk-plonk = 0.8.2``  ref:Cargo.toml:19, ``dusk-bls12_381 = 0.8``
ref:Cargo.toml:20).  No Rust toolchain exists here, so the reference cannot run.
This oracle therefore restates the *published* algorithms of those crates:

* ``EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}``  (dusk-plonk 0.8.2
  ``fft::domain``; same maths as ark-poly ``Radix2EvaluationDomain``): SURVEY.md CS-3
* ``msm_variable_base`` (dusk-bls12_381 0.8 ``multiscalar_mul``; same as ark-ec 0.2
  ``VariableBaseMSM::multi_scalar_mul``): SURVEY.md CS-4

and is pinned by (i) first-principles re-derivation of every constant (checked in
``tests/test_oracle_constants.py`` against the SURVEY.md section 8c table), and
(ii) algebraic known-answer identities.  Every operation on this path is exact
arithmetic in Fr / on G1, so any correct implementation is bit-identical once the
output is canonical (reduced Montgomery limbs, affine points).

Deliberately naive and *algorithmically different* from both the C restatement
(``oracle/c``) and the HIP path: O(n^2) DFT / recursive NTT here vs iterative
DIT there vs Stockham on the GPU; double-and-add here vs Pippenger there.
"""
from __future__ import annotations

# --------------------------------------------------------------------------- Fr
# dusk_bls12_381::Scalar (= BlsScalar), used in-tree at ref:allocated_scalar.rs:19,30
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FR_BITS = 255
FR_MONT_R = (1 << 256) % R_MOD            # Montgomery radix R = 2^256 mod r
FR_MONT_R2 = (FR_MONT_R * FR_MONT_R) % R_MOD
FR_MONT_RINV = pow(FR_MONT_R, -1, R_MOD)
FR_INV64 = (-pow(R_MOD, -1, 1 << 64)) % (1 << 64)
TWO_ADICITY = 32
FR_ODD_PART = (R_MOD - 1) >> TWO_ADICITY
FR_GENERATOR = 7                          # multiplicative generator == coset shift
ROOT_OF_UNITY = pow(FR_GENERATOR, FR_ODD_PART, R_MOD)   # order 2^32

# --------------------------------------------------------------------------- Fp
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FP_MONT_R = (1 << 384) % P_MOD
FP_MONT_R2 = (FP_MONT_R * FP_MONT_R) % P_MOD
FP_MONT_RINV = pow(FP_MONT_R, -1, P_MOD)
FP_INV64 = (-pow(P_MOD, -1, 1 << 64)) % (1 << 64)
G1_B = 4
G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)


# ----------------------------------------------------------------- limb packing
def to_limbs64(x: int, n: int) -> list[int]:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs64(limbs) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def fr_to_mont(x: int) -> int:
    return (x * FR_MONT_R) % R_MOD


def fr_from_mont(x: int) -> int:
    return (x * FR_MONT_RINV) % R_MOD


def fp_to_mont(x: int) -> int:
    return (x * FP_MONT_R) % P_MOD


def fp_from_mont(x: int) -> int:
    return (x * FP_MONT_RINV) % P_MOD


# ------------------------------------------------------------ EvaluationDomain
class Domain:
    """dusk_plonk::fft::EvaluationDomain::new."""

    def __init__(self, num_coeffs: int):
        size = 1
        while size < num_coeffs:
            size <<= 1
        log_n = size.bit_length() - 1
        if log_n >= TWO_ADICITY:
            raise ValueError("log_size_of_group >= TWO_ADICITY")
        self.size = size
        self.log_n = log_n
        self.group_gen = pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - log_n), R_MOD)
        self.group_gen_inv = pow(self.group_gen, -1, R_MOD)
        self.size_inv = pow(size, -1, R_MOD)
        self.generator_inv = pow(FR_GENERATOR, -1, R_MOD)

    def elements(self):
        w = 1
        for _ in range(self.size):
            yield w
            w = w * self.group_gen % R_MOD

    # the small helpers of dusk_plonk::fft::domain (dusk-plonk 0.8.2, ref:Cargo.toml:19), from
    # their definitions -- no batch inversion, no running products: one modular inverse per element
    def evaluate_vanishing_polynomial(self, tau: int) -> int:
        """Z_H(tau) = tau^size - 1."""
        return (pow(tau, self.size, R_MOD) - 1) % R_MOD

    def compute_vanishing_poly_over_coset(self, poly_degree: int) -> list[int]:
        """[(g w^i)^poly_degree - 1 for i < size], g = GENERATOR; upstream asserts size > poly_degree."""
        if not self.size > poly_degree:
            raise ValueError("domain_size > poly_degree")
        return [(pow(FR_GENERATOR * pow(self.group_gen, i, R_MOD) % R_MOD, poly_degree, R_MOD) - 1) % R_MOD
                for i in range(self.size)]

    def evaluate_all_lagrange_coefficients(self, tau: int) -> list[int]:
        """[L_i(tau)]: L_i(X) = prod_{j != i} (X - w^j) / (w^i - w^j) = (X^n - 1) w^i / (n (X - w^i))."""
        els = list(self.elements())
        if tau % R_MOD in els:
            return [1 if e == tau % R_MOD else 0 for e in els]
        zh = self.evaluate_vanishing_polynomial(tau)
        return [zh * e % R_MOD * pow(self.size * (tau - e) % R_MOD, -1, R_MOD) % R_MOD for e in els]


def naive_dft(a: list[int], omega: int) -> list[int]:
    """O(n^2) definition: out[j] = sum_i a[i] * omega^(i j)."""
    n = len(a)
    out = []
    for j in range(n):
        wj = pow(omega, j, R_MOD)
        acc, w = 0, 1
        for i in range(n):
            acc = (acc + a[i] * w) % R_MOD
            w = w * wj % R_MOD
        out.append(acc)
    return out


def recursive_ntt(a: list[int], omega: int) -> list[int]:
    """Textbook recursive radix-2 split (even/odd); canonical ints in, out."""
    n = len(a)
    if n == 1:
        return [a[0] % R_MOD]
    w2 = omega * omega % R_MOD
    ev = recursive_ntt(a[0::2], w2)
    od = recursive_ntt(a[1::2], w2)
    out = [0] * n
    w = 1
    h = n // 2
    for k in range(h):
        t = od[k] * w % R_MOD
        out[k] = (ev[k] + t) % R_MOD
        out[k + h] = (ev[k] - t) % R_MOD
        w = w * omega % R_MOD
    return out


def _pad(a, size):
    a = list(a)
    if len(a) > size:
        raise ValueError("input longer than domain")
    return a + [0] * (size - len(a))


def fft(a, log_n):
    d = Domain(1 << log_n)
    return recursive_ntt(_pad(a, d.size), d.group_gen)


def ifft(a, log_n):
    d = Domain(1 << log_n)
    out = recursive_ntt(_pad(a, d.size), d.group_gen_inv)
    return [x * d.size_inv % R_MOD for x in out]


def coset_fft(a, log_n):
    """distribute_powers(coeffs, GENERATOR) then fft."""
    d = Domain(1 << log_n)
    a = _pad(a, d.size)
    g, out = 1, []
    for x in a:
        out.append(x * g % R_MOD)
        g = g * FR_GENERATOR % R_MOD
    return recursive_ntt(out, d.group_gen)


def coset_ifft(a, log_n):
    """ifft then distribute_powers(., GENERATOR^-1)."""
    d = Domain(1 << log_n)
    out = ifft(a, log_n)
    g, res = 1, []
    for x in out:
        res.append(x * g % R_MOD)
        g = g * d.generator_inv % R_MOD
    return res


def horner(a, x):
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % R_MOD
    return acc


# ------------------------------------------------------------------------- G1
# Affine points are (x, y) tuples of canonical ints; None is the identity.
def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - G1_B) % P_MOD == 0


def g1_neg(P):
    if P is None:
        return None
    return (P[0], (-P[1]) % P_MOD)


def g1_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % P_MOD == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P_MOD) % P_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    y3 = (lam * (x1 - x3) - y1) % P_MOD
    return (x3, y3)


def g1_mul(k: int, P):
    """Left-to-right double-and-add.  k is reduced mod r (P assumed in G1)."""
    k %= R_MOD
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, P)
    return acc


def naive_msm(points, scalars):
    """sum_i s_i * P_i by independent scalar multiplications."""
    if len(points) != len(scalars):
        raise ValueError("length mismatch")
    acc = None
    for P, s in zip(points, scalars):
        acc = g1_add(acc, g1_mul(s, P))
    return acc


def g1_compress(P) -> bytes:
    """zcash-format 48-byte compressed encoding (used only as a known-answer pin)."""
    if P is None:
        return bytes([0xC0] + [0] * 47)
    x, y = P
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P_MOD - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


# ------------------------------------------------------ deterministic sampling
def splitmix64(state: int):
    """Returns (next_state, output).  Matches oracle/c/prng.h and the HIP-side generator."""
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def sample_fr(seed: int, n: int) -> list[int]:
    """n canonical Fr values: 4 splitmix64 words (LE limbs) -> 256-bit int mod r."""
    out, st = [], seed
    for _ in range(n):
        v = 0
        for j in range(4):
            st, w = splitmix64(st)
            v |= w << (64 * j)
        out.append(v % R_MOD)
    return out
