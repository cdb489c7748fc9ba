"""The oracle pinned to REFERENCE-HELD vectors: fixtures/verification_key.json of the reference checkout (numbers
copied to tests/golden/verification_key_bn254.json by tests/golden/gen_vk_golden.py) carries snarkjs' BN254
points and the golden pairing value vk_alphabeta_12 = e(vk_alpha_1, vk_beta_2).  A wrong q, tower non-residue,
twist constant, G2 generator, loop count or Frobenius constant in oracle/params.py / oracle/curve.py /
oracle/pairing.py makes these tests fail.  The second half restates the reference's own end-to-end assertion,
Groth16::verify_proof on the produced proof (groth16/examples/sha256.rs:389-415), through that pairing."""
import json
import os

import pytest

from oracle import groth16 as g
from oracle import pairing as op
from oracle.curve import g1, g2
from oracle.params import BLS12_381, BN254, CURVES
from oracle.prng import rand_fp

HERE = os.path.dirname(os.path.abspath(__file__))


def _vk():
    with open(os.path.join(HERE, "golden", "verification_key_bn254.json")) as fh:
        d = json.load(fh)
    p1 = lambda v: (int(v[0]), int(v[1]))
    p2 = lambda v: ((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1])))
    gold = tuple(tuple(tuple(int(x) for x in b) for b in c) for c in d["vk_alphabeta_12"])
    return d, p1, p2, gold


def test_golden_pairing_value_of_the_reference_fixture():
    d, p1, p2, gold = _vk()
    assert d["curve"] == "bn128" and d["vk_alpha_1"][2] == "1" and d["vk_beta_2"][2] == ["1", "0"]
    pr = op.pairing_for(BN254)
    assert pr.pairing(p1(d["vk_alpha_1"]), p2(d["vk_beta_2"])) == gold
    # the value is the Fuentes-Castaneda power of the reduced pairing (libff / snarkjs / arkworks), not the exact one
    op.BN_EXACT_HARD_PART = True
    try:
        assert pr.pairing(p1(d["vk_alpha_1"]), p2(d["vk_beta_2"])) != gold
    finally:
        op.BN_EXACT_HARD_PART = False


def test_fixture_points_are_on_curve_and_in_the_r_torsion():
    d, p1, p2, _ = _vk()
    G1, G2 = g1(BN254), g2(BN254)
    assert p2(d["vk_gamma_2"]) == BN254.g2                # snarkjs uses the standard G2 generator for gamma
    for v in [d["vk_alpha_1"]] + d["IC"]:
        assert v[2] == "1"
        pt = p1(v)
        assert G1.on_curve(pt)
        # Group.mul reduces the scalar mod r, so multiply by r-1 and add
        assert G1.is_identity(G1.add(G1.mul(G1.from_affine(pt), BN254.r - 1), G1.from_affine(pt)))
    assert len(d["IC"]) == d["nPublic"] + 1
    for v in (d["vk_beta_2"], d["vk_gamma_2"], d["vk_delta_2"]):
        assert v[2] == ["1", "0"]
        pt = p2(v)
        assert G2.on_curve(pt)
        assert G2.is_identity(G2.add(G2.mul(G2.from_affine(pt), BN254.r - 1), G2.from_affine(pt)))


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_pairing_is_bilinear_and_non_degenerate(name):
    c = CURVES[name]
    G1, G2, pr = g1(c), g2(c), op.pairing_for(c)
    a, b = rand_fp(7, 0, c.r), rand_fp(7, 1, c.r)
    e0 = pr.pairing(c.g1, c.g2)
    assert e0 != pr.T.one12 and pr.T.pow12(e0, c.r) == pr.T.one12
    pa = G1.to_affine(G1.mul(G1.from_affine(c.g1), a))
    qb = G2.to_affine(G2.mul(G2.from_affine(c.g2), b))
    assert pr.pairing(pa, qb) == pr.T.pow12(e0, a * b % c.r)
    assert pr.pairing(pa, c.g2) == pr.pairing(c.g1, G2.to_affine(G2.mul(G2.from_affine(c.g2), a)))
    assert pr.pairing(None, c.g2) == pr.T.one12


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_verify_proof_accepts_the_local_prover_and_rejects_tampering(name):
    """sha256.rs:389-415 on a small circuit: arkworks-style local proof verifies against the vk by pairing."""
    from test_oracle_groth16 import small_r1cs
    c = CURVES[name]
    P = c.r
    r1, w = small_r1cs()
    w = [x % P for x in w]
    if name != "bn254":
        r1, w = _small_r1cs_mod(P)
    assert g.is_satisfied(r1, w, P)
    G1, G2 = g1(c), g2(c)
    key = g.setup_scalars(c, r1, g.Trapdoor.from_seed(42, P))
    pk = g.proving_key_points(key, G1, G2)
    vk = op.verifying_key_from_trapdoor(key, G1, G2)
    r, s = rand_fp(43, 0, P), rand_fp(43, 1, P)
    A, B, Cc = g.create_proof_local(c, r1, pk, G1, G2, w, r, s)
    proof = (G1.to_affine(A), G2.to_affine(B), G1.to_affine(Cc))
    ni = r1.num_instance_variables
    assert op.verify_proof(c, vk, proof, w[1:ni], G1)
    assert not op.verify_proof(c, vk, proof, [(w[1] + 1) % P], G1)
    bad = (proof[0], proof[1], G1.to_affine(G1.add(Cc, G1.from_affine(c.g1))))
    assert not op.verify_proof(c, vk, bad, w[1:ni], G1)
    with pytest.raises(ValueError):
        op.verify_proof(c, vk, proof, [], G1)


def _small_r1cs_mod(P, nc=11):
    """test_oracle_groth16.small_r1cs over another scalar field"""
    w = [1, 0, 7, 5]
    A, B, Cm = [], [], []
    for _ in range(nc - 1):
        k = len(w)
        w.append((w[k - 1] + 3) * (w[k - 1] + w[k - 2]) % P)
        A.append([(1, k - 1), (3, 0)])
        B.append([(1, k - 1), (1, k - 2)])
        Cm.append([(1, k)])
    A.append([(1, len(w) - 1)])
    B.append([(1, 0)])
    Cm.append([(1, 1)])
    w[1] = w[-1]
    return g.R1CS(2, len(w) - 2, A, B, Cm), w
