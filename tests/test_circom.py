"""Circom front end (host side): iden3 .r1cs / .wtns readers against the file written by the REFERENCE's own witness
calculator (tests/golden/sha256_a1_b2.wtns.gz, made by tests/golden/gen_sha256_wtns.py) and round trips; the
ark-compressed wire format against the oracle's restatement of ark-serialize."""
import gzip
import os
import random
import types

import pytest

from oracle import ser as oser
from oracle import curve as ocurve
from oracle.params import BN254, CURVES

HERE = os.path.dirname(os.path.abspath(__file__))


def _wtns_fixture():
    with gzip.open(os.path.join(HERE, "golden", "sha256_a1_b2.wtns.gz"), "rb") as fh:
        return fh.read()


def test_read_wtns_of_reference_witness_calculator():
    from zksaas_amd import circom
    data = _wtns_fixture()
    w, prime = circom.read_wtns(data)
    assert prime == BN254.r and len(w) == 29823
    assert w[0] == 1 and w[2] == 1 and w[3] == 2                    # sha256.rs:164-165 inputs
    assert w[1] == 72587776472194017031617589674261467945970986113287823188107011979
    assert circom.write_wtns(w, prime) == data                      # witness_calculator.js:208-272 byte for byte
    limbs = circom.wtns_to_limbs(data, 4)
    assert limbs.shape == (29823, 4) and int(limbs[3, 0]) == 2 and not limbs[3, 1:].any()


@pytest.mark.parametrize("mut", ["magic", "version", "truncated", "size"])
def test_read_wtns_rejects_malformed(mut):
    from zksaas_amd import circom
    data = bytearray(circom.write_wtns([1, 2, 3], BN254.r))
    if mut == "magic":
        data[0] = ord("x")
    elif mut == "version":
        data[4] = 9
    elif mut == "truncated":
        data = data[:-5]
    else:
        data[12 + 12 + 4 + 32] = 7          # witness size field: 7 elements announced, 3 present
    with pytest.raises(ValueError):
        circom.read_wtns(bytes(data))


def test_r1cs_round_trip_small_and_sha256():
    from zksaas_amd import circom
    from zksaas_amd import sha256_circuit as sc
    rng = random.Random(5)
    p = BN254.r
    nv, ni, nc = 9, 3, 6

    def lc():
        return [(rng.randrange(p), rng.randrange(nv)) for _ in range(rng.randrange(0, 4))]
    small = sc.R1CS(ni, nv - ni, [lc() for _ in range(nc)], [lc() for _ in range(nc)], [lc() for _ in range(nc)])
    blob = circom.write_r1cs(small, p, n_pub_out=1, n_pub_in=1)
    back, prime, info = circom.read_r1cs(blob)
    assert prime == p and (back.a, back.b, back.c) == (small.a, small.b, small.c)
    assert back.num_instance_variables == ni and back.num_witness_variables == nv - ni
    assert info["n_pub_out"] == 1 and info["n_pub_in"] == 1 and info["n_prv_in"] == nv - ni
    assert circom.write_r1cs(back, prime, n_pub_out=1, n_pub_in=1) == blob
    r1, _ = sc.build(1, 2, p)
    blob = circom.write_r1cs(r1, p)
    back, prime, info = circom.read_r1cs(blob)
    assert back.num_constraints == r1.num_constraints == 26493 and back.num_variables == r1.num_variables
    assert [sorted(r) for r in back.a] == [sorted((c % p, i) for c, i in r) for r in r1.a]
    assert info["n_pub_out"] == 1 and info["n_pub_in"] == 0          # sha256.circom: one public output


def test_read_r1cs_rejects_malformed():
    from zksaas_amd import circom
    from zksaas_amd import sha256_circuit as sc
    r = sc.R1CS(1, 2, [[(1, 1)]], [[(1, 2)]], [[(1, 0)]])
    blob = bytearray(circom.write_r1cs(r, BN254.r))
    with pytest.raises(ValueError):
        circom.read_r1cs(bytes(blob[:40]))
    bad = bytearray(blob)
    bad[:4] = b"wtns"
    with pytest.raises(ValueError):
        circom.read_r1cs(bytes(bad))
    # wire index beyond nWires
    off = blob.index(b"\x01\x00\x00\x00\x01\x00\x00\x00", 12 + 12 + 64)      # nA = 1, wire 1
    bad = bytearray(blob)
    bad[off + 4] = 200
    with pytest.raises(ValueError):
        circom.read_r1cs(bytes(bad))


def _stub_pp(curve):
    from zksaas_amd.fields import FQ, MontCodec
    return types.SimpleNamespace(fq=MontCodec(FQ[curve]), curve=curve)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_point_wire_format_matches_oracle_and_round_trips(curve):
    from zksaas_amd import wire
    cv = CURVES[curve]
    pp = _stub_pp(curve)
    G1, G2 = ocurve.g1(cv), ocurve.g2(cv)
    rng = random.Random(11)
    for g2, G, ofn in ((False, G1, oser.g1_compressed), (True, G2, oser.g2_compressed)):
        pts = [None] + [G.to_affine(G.mul(G.from_affine(G.gen), rng.randrange(1, cv.r))) for _ in range(6)]
        for pt in pts:
            blob = wire.point_to_bytes(pp, pt, g2)
            assert blob == ofn(pt, cv.q)
            assert wire.point_from_bytes(pp, blob, g2, curve) == pt
            jac = wire.affine_to_jacobian(pp, pt, g2)
            assert wire.jacobian_to_affine(pp, jac, g2) == pt
    a, c = [G1.to_affine(G1.mul(G1.from_affine(G1.gen), k)) for k in (5, 7)]
    b = G2.to_affine(G2.mul(G2.from_affine(G2.gen), 9))
    blob = wire.proof_to_bytes(pp, wire.affine_to_jacobian(pp, a, False), wire.affine_to_jacobian(pp, b, True),
                               wire.affine_to_jacobian(pp, c, False))
    assert blob == oser.proof_compressed(a, b, c, cv.q)
    assert len(blob) == (128 if curve == "bn254" else 192)
    assert wire.proof_from_bytes(pp, blob, curve) == (a, b, c)


def test_bls12_381_uses_the_zcash_encoding_pinned_on_the_published_generators():
    """ark-bls12-381 0.4 serialises points in the zcash / IETF format (big-endian, flags in the first byte); the
    compressed generators are published constants (draft-irtf-cfrg-pairing-friendly-curves, zcash protocol spec)."""
    from zksaas_amd import wire
    cv = CURVES["bls12_381"]
    pp = _stub_pp("bls12_381")
    g1c = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                        "6c55e83ff97a1aeffb3af00adb22c6bb")
    g2c = bytes.fromhex("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049"
                        "334cf11213945d57e5ac7d055d042b7e024aa2b2f08f0a91260805272dc51051"
                        "c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
    assert wire.point_to_bytes(pp, cv.g1, False) == g1c == oser.g1_compressed(cv.g1, cv.q)
    assert wire.point_to_bytes(pp, cv.g2, True) == g2c == oser.g2_compressed(cv.g2, cv.q)
    assert wire.point_from_bytes(pp, g1c, False, "bls12_381") == cv.g1
    assert wire.point_from_bytes(pp, g2c, True, "bls12_381") == cv.g2
    # the negated generators carry the sort flag (bit 5), infinity is c0 00 .. 00
    G1 = ocurve.g1(cv)
    neg = G1.to_affine(G1.neg(G1.from_affine(cv.g1)))
    assert wire.point_to_bytes(pp, neg, False)[0] == g1c[0] | 0x20 and wire.point_to_bytes(pp, neg, False)[1:] == g1c[1:]
    assert wire.point_to_bytes(pp, None, False) == bytes([0xC0]) + bytes(47)
    assert wire.point_from_bytes(pp, bytes([0xC0]) + bytes(95), True, "bls12_381") is None
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, bytes([0x17]) + g1c[1:], False, "bls12_381")     # compression flag missing
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, bytes([0xE0]) + bytes(47), False, "bls12_381")    # infinity with the sort flag


def test_bls12_377_g1_round_trips_with_tonelli_shanks():
    """the curve the reference's own tests use: q = 1 mod 4, so the square root is Tonelli-Shanks"""
    from zksaas_amd import wire
    cv = CURVES["bls12_377"]
    pp = _stub_pp("bls12_377")
    G1 = ocurve.g1(cv)
    rng = random.Random(12)
    for pt in [None, cv.g1] + [G1.to_affine(G1.mul(G1.from_affine(cv.g1), rng.randrange(1, cv.r))) for _ in range(4)]:
        blob = wire.point_to_bytes(pp, pt, False)
        assert blob == oser.g1_compressed(pt, cv.q) and len(blob) == 48
        assert wire.point_from_bytes(pp, blob, False, "bls12_377") == pt
    with pytest.raises(ValueError):
        wire.point_to_bytes(pp, None, True) and wire.point_from_bytes(pp, bytes(96), True, "bls12_377")


def test_point_from_bytes_rejects_invalid():
    from zksaas_amd import wire
    pp = _stub_pp("bn254")
    q = BN254.q
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, bytes(31), False, "bn254")                       # wrong length
    bad = bytearray(32)
    bad[-1] = 0xC0
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, bytes(bad), False, "bn254")                      # both flags
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, (q + 1).to_bytes(32, "little"), False, "bn254")  # x >= q
    # x = 4: 4^3 + 3 = 67 is not a square mod q
    x = next(x for x in range(2, 50) if pow((x ** 3 + 3) % q, (q - 1) // 2, q) != 1)
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, x.to_bytes(32, "little"), False, "bn254")
    inf = bytearray(32)
    inf[-1] = 0x40
    inf[0] = 1
    with pytest.raises(ValueError):
        wire.point_from_bytes(pp, bytes(inf), False, "bn254")                      # infinity with x != 0
