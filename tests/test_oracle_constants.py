"""Pins the oracles' derived constants against the SURVEY.md section 8c table (hard-coded
here as literals) and against each other.  CPU only."""
import numpy as np

from oracle import bigint_oracle as B
from oracle.cpu_oracle import limbs_to_ints

# SURVEY.md section 8c, verbatim
SURVEY = dict(
    r_limbs=[0xffffffff00000001, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48],
    fr_inv=0xfffffffeffffffff,
    fr_R=[0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f],
    fr_R2=[0xc999e990f3f29c6d, 0x2b6cedcb87925c23, 0x05d314967254398f, 0x0748d9d99f59ff11],
    fr_R3=[0xc62c1807439b73af, 0x1b3e0d188cf06990, 0x73d13c71c7b5f418, 0x6e2a5bb9c8db33e9],
    gen7_mont=[0x0000000efffffff1, 0x17e363d300189c0f, 0xff9c57876f8457b0, 0x351332208fc5a8c4],
    root=0x16a2a19edfe81f20d09b681922c813b4b63683508c2280b93829971f439f0d2b,
    root_mont=[0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec, 0x0af53ae352a31e64, 0x5bf3adda19e9b27b],
    omega={12: 0x564c0a11a0f704f4fc3e8acfe0f8245f0ad1347b378fbf96e206da11a5d36306,
           20: 0x03e1c54bcb947035a57a6e07cb98de4a2f69e02d265e09d9fece7e0e39898d4b,
           22: 0x0abe6a5e5abcaa32f2d38f10fbb8d1bbe08fec7c86389beec6e7a6ffb08e3363,
           24: 0x291cf6d68823e6876e0bcd91ee76273072cf6a8029b7d7bc92cf4deb77bd779c,
           26: 0x0a0a77a3b1980c0d116168bffbedc11d02c8118402867ddc531a11a0d2d75182},
    ninv={20: 0x73eda0144f284aae5b6554d46c21576b363d4ec725be2bff1a400fff00001001,
          24: 0x73eda6df3bf62a1e95bc8fd4cfc9cffbb1e59eaf425a58ff01a400ff00000101},
    fp_inv=0x89f3fffcfffcfffd,
    fp_R=[0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745,
          0x5c071a97a256ec6d, 0x15f65ec3fa80e493],
    g1_compressed_prefix="97f1d3a73197d794", g1_compressed_suffix="fb3af00adb22c6bb",
)


def _int(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def test_bigint_oracle_matches_survey_table():
    assert B.R_MOD == _int(SURVEY["r_limbs"])
    assert B.FR_INV64 == SURVEY["fr_inv"]
    assert B.FR_MONT_R == _int(SURVEY["fr_R"])
    assert B.FR_MONT_R2 == _int(SURVEY["fr_R2"])
    assert B.FR_MONT_R2 * B.FR_MONT_R % B.R_MOD == _int(SURVEY["fr_R3"])
    assert B.fr_to_mont(7) == _int(SURVEY["gen7_mont"])
    assert B.ROOT_OF_UNITY == SURVEY["root"]
    assert B.fr_to_mont(B.ROOT_OF_UNITY) == _int(SURVEY["root_mont"])
    for k, w in SURVEY["omega"].items():
        assert B.Domain(1 << k).group_gen == w
    for k, v in SURVEY["ninv"].items():
        assert B.Domain(1 << k).size_inv == v
    assert B.FP_INV64 == SURVEY["fp_inv"]
    assert B.FP_MONT_R == _int(SURVEY["fp_R"])
    c = B.g1_compress(B.G1_GEN).hex()
    assert c.startswith(SURVEY["g1_compressed_prefix"]) and c.endswith(SURVEY["g1_compressed_suffix"])


def test_first_principles():
    # r, p prime-field structure: x-parametrisation of BLS12-381
    x = -0xd201000000010000
    assert B.R_MOD == x ** 4 - x ** 2 + 1
    assert B.P_MOD == (x - 1) ** 2 * B.R_MOD // 3 + x
    # order of the root of unity is exactly 2^32; 7 is a non-residue
    assert pow(B.ROOT_OF_UNITY, 1 << 32, B.R_MOD) == 1
    assert pow(B.ROOT_OF_UNITY, 1 << 31, B.R_MOD) != 1
    assert pow(7, (B.R_MOD - 1) // 2, B.R_MOD) != 1
    assert (B.R_MOD - 1) % (1 << 32) == 0 and ((B.R_MOD - 1) >> 32) % 2 == 1
    # generator on the curve, in the order-r subgroup
    assert B.g1_is_on_curve(B.G1_GEN)
    assert B.g1_mul(B.R_MOD - 1, B.G1_GEN) == B.g1_neg(B.G1_GEN)
    assert B.g1_add(B.g1_mul(B.R_MOD - 1, B.G1_GEN), B.G1_GEN) is None


def test_c_oracle_constants(oracle, golden):
    c = oracle.constants()
    assert c["fr_inv"] == SURVEY["fr_inv"] and c["fp_inv"] == SURVEY["fp_inv"]
    assert limbs_to_ints(c["fr_one"])[0] == _int(SURVEY["fr_R"])
    assert limbs_to_ints(c["fr_r2"])[0] == _int(SURVEY["fr_R2"])
    assert limbs_to_ints(c["fr_root"])[0] == _int(SURVEY["root_mont"])
    assert limbs_to_ints(c["fr_gen"])[0] == _int(SURVEY["gen7_mont"])
    assert limbs_to_ints(c["fp_one"])[0] == _int(SURVEY["fp_R"])
    g = golden["constants"]
    assert int(g["root_of_unity"], 16) == SURVEY["root"]
    assert g["g1_compressed"].startswith(SURVEY["g1_compressed_prefix"])


def test_msm_window_rule(oracle):
    # c = 3 if n < 32 else ceil(log2 n) * 69 / 100 + 2   (SURVEY.md section 8c last row)
    assert oracle.msm_window_bits(31) == 3
    assert oracle.msm_window_bits(32) == 5
    assert oracle.msm_window_bits(1 << 12) == 10
    assert oracle.msm_window_bits(1 << 20) == 15
    assert oracle.msm_window_bits(1 << 24) == 18
