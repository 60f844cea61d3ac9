"""CPU: the test-only pairing (oracle/pairing_oracle.py) has the properties the KZG checks rely on,
and the verifier oracle accepts the prover oracle's proofs and rejects tampered ones."""
import random


from oracle import bigint_oracle as B
from oracle import pairing_oracle as PG
from oracle import plonk_rounds_oracle as PO
from oracle import plonk_verifier_oracle as PV

R = B.R_MOD


def test_field_tower():
    rng = random.Random(3)
    a, b, c = ([rng.randrange(PG.P) for _ in range(12)] for _ in range(3))
    assert PG.f12_mul(a, PG.f12_inv(a)) == PG.F12_ONE
    assert PG.f12_mul(PG.f12_mul(a, b), c) == PG.f12_mul(a, PG.f12_mul(b, c))
    assert PG.f12_mul(a, PG.f12_add(b, c)) == PG.f12_add(PG.f12_mul(a, b), PG.f12_mul(a, c))
    u = PG.f2_to_f12((0, 1))
    assert PG.f12_mul(u, u) == PG.f12(-1)                                  # u^2 = -1
    w6 = PG.f12_pow(PG.W, 6)
    assert w6 == PG.f2_to_f12((1, 1))                                      # w^6 = 1 + u
    x, y = (rng.randrange(PG.P), rng.randrange(PG.P)), (rng.randrange(PG.P), rng.randrange(PG.P))
    assert PG.f2_to_f12(PG.f2_mul(x, y)) == PG.f12_mul(PG.f2_to_f12(x), PG.f2_to_f12(y))


def test_g2_generator_and_pairing_properties():
    assert PG.g2_is_on_curve(PG.G2_GEN)
    assert PG.g2_mul(R, PG.G2_GEN) is None and PG.g2_mul(R - 1, PG.G2_GEN) is not None
    x, y = PG.untwist(PG.G2_GEN)
    assert PG.f12_mul(y, y) == PG.f12_add(PG.f12_mul(PG.f12_mul(x, x), x), PG.f12(4))
    e = PG.pairing(PG.G2_GEN, B.G1_GEN)
    assert e != PG.F12_ONE and PG.f12_pow(e, R) == PG.F12_ONE              # non-degenerate, order r
    a, b = 0x1234567, 0x89ABCDE
    assert PG.pairing(PG.g2_mul(b, PG.G2_GEN), B.g1_mul(a, B.G1_GEN)) == PG.f12_pow(e, a * b % R)   # bilinear
    assert PG.pairing_product_is_one([(PG.G2_GEN, B.g1_mul(a, B.G1_GEN)),
                                      (PG.g2_mul(a, PG.G2_GEN), B.g1_neg(B.G1_GEN))])
    assert not PG.pairing_product_is_one([(PG.G2_GEN, B.g1_mul(a, B.G1_GEN)),
                                          (PG.g2_mul(a + 1, PG.G2_GEN), B.g1_neg(B.G1_GEN))])


def _commit(coeffs, tau):
    return B.g1_mul(B.horner(coeffs, tau), B.G1_GEN)


def test_verifier_accepts_oracle_proofs_and_rejects_tampering():
    from plonk_prototype_amd.field import fr_vec_from_limbs
    from plonk_prototype_amd.synthetic import mixed_circuit
    n, tau = 32, 0xABCDEF0123456789 ** 3 % R
    c, w, pi = mixed_circuit(n, 6)                    # every gate kind: all five widgets enter [r]
    sel = {k: fr_vec_from_limbs(getattr(c, k)) for k in PO.SELECTORS}
    sigma, wit, pii = c.sigma_index.tolist(), [fr_vec_from_limbs(w[j]) for j in range(4)], fr_vec_from_limbs(pi)
    ch = {k: pow(11 + 2 * i, 20 + i, R) for i, k in enumerate(PO.CHALLENGES + ("batch",))}
    out = PO.prove(n, sel, sigma, wit, pii, ch)
    vk = {k: _commit(out["sel_coeffs"][k], tau) for k in PO.SELECTORS}
    for j in range(4):
        vk[f"sigma_{j + 1}"] = _commit(out["sigma_coeffs"][j], tau)
    comms = {nm: _commit(out["wire_coeffs"][j], tau) for j, nm in enumerate("abcd")}
    comms["z"] = _commit(out["z_coeffs"], tau)
    for i in range(4):
        comms[f"t_{i + 1}"] = _commit(out["t_coeffs"][i * n:(i + 1) * n], tau)
    comms["w_z"], comms["w_zw"] = _commit(out["w_z"], tau), _commit(out["w_zw"], tau)
    pi_z = B.horner(B.ifft(pii, 5), ch["z"])
    tau_g2 = PG.g2_mul(tau, PG.G2_GEN)
    assert PV.verify(n, vk, comms, out["evals"], ch, pi_z, tau_g2) == (True, True)
    bad_ev = dict(out["evals"], a=(out["evals"]["a"] + 1) % R)
    assert PV.verify(n, vk, comms, bad_ev, ch, pi_z, tau_g2) == (False, False)
    for name in ("a_next", "b_next", "d_next", "q_arith", "q_c", "q_l", "q_r"):      # the openings dusk's Proof adds
        bad_ev = {k: v for k, v in out["evals"].items() if k != "t"}
        bad_ev[name] = (bad_ev[name] + 1) % R
        assert PV.verify(n, vk, comms, bad_ev, ch, pi_z, tau_g2)[1] is False, name
    bad_c = dict(comms, w_z=B.g1_add(comms["w_z"], B.G1_GEN))
    assert PV.verify(n, vk, bad_c, out["evals"], ch, pi_z, tau_g2) == (True, False)
    bad_vk = dict(vk, q_logic=B.g1_add(vk["q_logic"], B.G1_GEN))
    assert PV.verify(n, bad_vk, comms, out["evals"], ch, pi_z, tau_g2) == (True, False)
    assert PV.verify(n, vk, comms, out["evals"], ch, (pi_z + 1) % R, tau_g2)[0] is False
