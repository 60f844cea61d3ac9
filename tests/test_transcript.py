"""CPU: the Fiat-Shamir transcript (plonk-prototype_amd/transcript.py).
Pins: the Keccak permutation against hashlib's SHA3-256, and Merlin's published known-answer vector
("test protocol" / "some label" / "some data" -> 32 challenge bytes)."""
import hashlib

import numpy as np

from oracle import bigint_oracle as B
from plonk_prototype_amd import transcript as T
from plonk_prototype_amd.field import R_MOD, fr_to_limbs


def _sha3_256(msg: bytes, perm=None) -> bytes:
    perm = perm or T.keccak_f1600
    st, rate = bytearray(200), 136
    m = bytearray(msg) + b"\x06"
    m += bytes(-len(m) % rate)
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[off + i]
        perm(st)
    return bytes(st[:32])


def test_keccak_permutation_matches_hashlib():
    for msg in (b"", b"abc", bytes(range(256)) * 3, b"\xff" * 135, b"\x00" * 136):
        assert _sha3_256(msg) == hashlib.sha3_256(msg).digest()                       # pm_keccak_f1600
        assert _sha3_256(msg, T.keccak_f1600_py) == hashlib.sha3_256(msg).digest()    # plain-Python restatement


def test_merlin_known_answer():
    t = T.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == \
        "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_transcript_is_order_and_content_sensitive():
    def run(msgs):
        t = T.Transcript(b"plonk")
        for lab, m in msgs:
            t.append_message(lab, m)
        return t.challenge_scalar(b"beta")
    base = run([(b"a", b"1"), (b"b", b"2")])
    assert base == run([(b"a", b"1"), (b"b", b"2")])
    assert base != run([(b"b", b"2"), (b"a", b"1")])
    assert base != run([(b"a", b"1"), (b"b", b"3")])
    assert base != run([(b"a", b"1b"), (b"", b"2")])
    assert 0 <= base < R_MOD
    # long messages cross the 166-byte STROBE rate
    assert run([(b"x", bytes(1000))]) != run([(b"x", bytes(1001))])


def test_successive_challenges_differ_and_chain():
    t = T.Transcript(b"plonk")
    t.append_scalar(b"s", fr_to_limbs(5))
    c1, c2 = t.challenge_scalar(b"c"), t.challenge_scalar(b"c")
    assert c1 != c2


def test_g1_compress_matches_the_oracle_encoding(oracle):
    from oracle.cpu_oracle import ints_to_limbs, limbs_to_ints
    G = oracle.g1_generator()
    for k in (1, 2, 3, 0xDEADBEEF, R_MOD - 1):
        xy = oracle.g1_mul(G, ints_to_limbs([k], 4)[0])
        can = limbs_to_ints(oracle.fp_from_mont(xy.reshape(2, 6)))
        assert T.g1_compress(xy) == B.g1_compress((can[0], can[1]))
    assert T.g1_compress(np.zeros(12, np.uint64)) == B.g1_compress(None)
    # the generator's compressed encoding starts 0x97f1d3a7... (zcash / dusk_bls12_381 docs)
    assert T.g1_compress(G).hex().startswith("97f1d3a73197d794")


def test_g1_decompress_roundtrip_and_rejects(oracle):
    from oracle.cpu_oracle import ints_to_limbs
    G = oracle.g1_generator()
    for k in (1, 2, 5, 0xC0FFEE, R_MOD - 2):
        xy = oracle.g1_mul(G, ints_to_limbs([k], 4)[0])
        assert np.array_equal(T.g1_decompress(T.g1_compress(xy)), xy)
    assert not T.g1_decompress(bytes([0xC0]) + bytes(47)).any()
    import pytest
    from plonk_prototype_amd.field import P_MOD
    off_curve = next(x for x in range(1, 100) if pow((x ** 3 + 4) % P_MOD, (P_MOD - 1) // 2, P_MOD) != 1)
    for bad in (bytes(48), bytes([0x80]) + bytes(46) + bytes([off_curve]), bytes([0xE0]) + bytes(47),
                bytes([0x9F]) + b"\xff" * 47, bytes(47)):
        with pytest.raises(ValueError):
            T.g1_decompress(bad)


def test_proof_serialisation_roundtrip(oracle):
    from oracle.cpu_oracle import ints_to_limbs
    from plonk_prototype_amd.prover import Proof
    G = oracle.g1_generator()
    p = Proof()
    for i, k in enumerate(Proof.COMMITMENTS):
        p.commitments[k] = oracle.g1_mul(G, ints_to_limbs([i * 977 + 3], 4)[0]) if i != 4 else np.zeros(12, np.uint64)
    for i, k in enumerate(Proof.EVALUATIONS):
        p.evaluations[k] = fr_to_limbs((i + 1) * 0x123456789ABCDEF % R_MOD)
    blob = p.to_bytes()
    assert len(blob) == 11 * 48 + 16 * 32 == 1040
    q = Proof.from_bytes(blob)
    assert all(np.array_equal(q.commitments[k], p.commitments[k]) for k in Proof.COMMITMENTS)
    assert all(np.array_equal(q.evaluations[k], p.evaluations[k]) for k in Proof.EVALUATIONS)
    assert q.to_bytes() == blob
    import pytest
    with pytest.raises(ValueError):
        Proof.from_bytes(blob[:-1])
    with pytest.raises(ValueError):
        Proof.from_bytes(blob[:-32] + b"\xff" * 32)


def test_commit_key_raw_bytes_roundtrip(oracle):
    from oracle.cpu_oracle import ints_to_limbs
    from plonk_prototype_amd import srs
    pts = oracle.g1_bases_arith(ints_to_limbs([5], 4)[0], ints_to_limbs([9], 4)[0], 7, 1)
    pts[3] = 0                                                   # an identity entry
    blob = srs.commit_key_to_raw_bytes(pts)
    assert len(blob) == 8 + 7 * 97 and blob[:8] == (7).to_bytes(8, "little")
    assert blob[8 + 3 * 97 + 96] == 1 and blob[8 + 96] == 0     # infinity flags
    assert blob[8:8 + 8] == int(pts[0, 0]).to_bytes(8, "little")  # raw Montgomery limbs, little-endian
    assert np.array_equal(srs.commit_key_from_raw_bytes(blob), pts)
    assert len(srs.commit_key_to_raw_bytes(np.zeros((0, 12), np.uint64))) == 8
    import pytest
    for bad in (blob[:-1], blob[:5], blob + b"\0"):
        with pytest.raises(ValueError):
            srs.commit_key_from_raw_bytes(bad)


def test_transcript_label_table_is_the_single_source():
    """Both sides of Fiat-Shamir read ONE table (csrc/prover.hip, namespace tl, exported by pm_plonk_transcript_labels):
    every message of a proof has its entry, in order, and the Python verifier side holds no label strings of its own."""
    import inspect
    import plonk_prototype_amd.prover as PR
    L = PR.transcript_labels()
    keys = list(L)
    expect = (["protocol"] + [f"selector_{i}" for i in range(11)] + [f"sigma_{i}" for i in range(4)]
              + ["dom_sep", "dom_sep_value", "circuit_size", "pi_len", "pi_pos", "pi_value"] + [f"wire_{i}" for i in range(4)]
              + ["beta", "gamma", "perm", "alpha", "range_sep", "logic_sep", "fixed_sep", "var_sep"]
              + [f"quotient_{i}" for i in range(4)] + ["z_challenge"] + [f"eval_{i}" for i in range(17)]
              + ["aggregate", "w_z", "w_zw", "batch"])
    assert keys == expect
    assert L["protocol"] == b"plonk" and L["selector_9"] == b"q_variable_group_add" and L["eval_14"] == b"perm_eval"
    src = inspect.getsource(PR.derive_challenges) + inspect.getsource(PR.seeded_transcript)
    assert 'b"' not in src.split('"""', 2)[2].replace('b"abcd"', "")       # no byte-string labels outside the docstring
