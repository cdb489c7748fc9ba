"""Oracle pinned against the reference's groth16 tests.

Restates groth16/src/ext_wit.rs:287-538 (libsnark_h m=32, circom_h m=1024 on a=b=[0..m), c=a*b) and the
end-to-end flow of groth16/examples/sha256.rs:159-416 on a small synthetic R1CS, with the Groth16
verification equation checked in the exponent (trapdoor known), SURVEY.md 8c (ii).
"""
import pytest

from oracle import groth16 as g
from oracle import ser
from oracle.curve import g1, g2, GroupOps
from oracle.dist import DegRedMask, FftMask, MsmMask, transpose
from oracle.field import Domain
from oracle.params import BN254
from oracle.prng import rand_fp
from oracle.pss import PackedSharingParams

C = BN254
P = C.r


def _dummy_qap(m):
    a = list(range(m))
    b = list(range(m))
    c = [x * y % P for x, y in zip(a, b)]
    return a, b, c


def test_libsnark_dummy_ext_witness():  # ext_wit.rs:287-417
    m = 32
    pp = PackedSharingParams(C, 2)
    dom = Domain(C, m)
    a, b, c = _dummy_qap(m)
    want = g.libsnark_ref(a, b, c, dom)
    qs = g.QAP(0, 0, a, b, c, dom).pss(pp, 1)
    co = dom.get_coset(C.r_gen)
    masks = ([FftMask.sample(True, co.offset, dom.group_gen_inv, m, pp, 10 + k) for k in range(3)]
             + [FftMask.sample(True, 1, dom.group_gen, m, pp, 20 + k) for k in range(3)]
             + [FftMask.sample(False, co.offset_inv, dom.group_gen_inv, m, pp, 30)])
    hs = g.libsnark_h(qs, masks, pp, dom, seed=2)
    assert [v for ch in transpose(hs) for v in pp.unpack2(ch)] == want


@pytest.mark.parametrize("m", [16, 1024])
def test_circom_dummy_ext_witness(m):  # ext_wit.rs:419-538 (m = 1024)
    pp = PackedSharingParams(C, 2)
    dom = Domain(C, m)
    a, b, c = _dummy_qap(m)
    want = g.circom_ref(a, b, c, dom)
    qs = g.QAP(0, 0, a, b, c, dom).pss(pp, 3)
    w2m = Domain(C, 2 * m).element(1)
    masks = ([FftMask.sample(True, w2m, dom.group_gen_inv, m, pp, 40 + k) for k in range(3)]
             + [FftMask.sample(False, 1, dom.group_gen, m, pp, 50 + k) for k in range(3)])
    dm = DegRedMask.sample(pp, 1, m // pp.l, 60)
    hs = g.circom_h(qs, masks, dm, pp, dom, seed=4)
    assert [v for ch in transpose(hs) for v in pp.unpack2(ch)] == want


def small_r1cs(nc=11):
    """w[k] = (w[k-1] + 3) * (w[k-1] + w[k-2]); public output = last wire."""
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


def test_setup_and_closed_form_verification():
    r1, w = small_r1cs()
    assert g.is_satisfied(r1, w, P)
    key = g.setup_scalars(C, r1, g.Trapdoor.from_seed(42, P))
    r, s = rand_fp(43, 0, P), rand_fp(43, 1, P)
    ps = g.prove_scalars(C, r1, key, w, r, s)
    assert g.verify_scalars(C, r1, key, w, ps)
    bad = list(w)
    bad[1] = (bad[1] + 1) % P              # wrong public input must not verify
    assert not g.verify_scalars(C, r1, key, bad, ps)
    assert len(key.h_query) == key.domain.size   # circom h_query has m entries


@pytest.mark.parametrize("r_zero", [False, True])
def test_distributed_proof_equals_local_proof(r_zero):
    """sha256.rs:159-416 flow: local arkworks-style proof == reconstructed distributed proof == closed form."""
    r1, w = small_r1cs()
    key = g.setup_scalars(C, r1, g.Trapdoor.from_seed(42, P))
    r = 0 if r_zero else rand_fp(43, 0, P)
    s = rand_fp(43, 1, P)
    G1, G2 = g1(C), g2(C)
    o1, o2 = GroupOps(G1), GroupOps(G2)
    pk = g.proving_key_points(key, G1, G2)
    A, B, Cc = g.create_proof_local(C, r1, pk, G1, G2, w, r, s)
    sa, sb, sc = g.prove_scalars(C, r1, key, w, r, s)
    assert G1.eq(A, G1.mul(G1.from_affine(C.g1), sa))
    assert G2.eq(B, G2.mul(G2.from_affine(C.g2), sb))
    assert G1.eq(Cc, G1.mul(G1.from_affine(C.g1), sc))

    pp = PackedSharingParams(C, 2)
    crs = g.pack_proving_key(pk, pp, G1, G2, o1, o2)
    q = g.qap(C, r1, w)
    dom, m = q.domain, q.domain.size
    qs = q.pss(pp, 5)
    ni = r1.num_instance_variables
    ax = g.pack_from_witness(pp, w[ni:], 11)
    a_sh = g.pack_from_witness(pp, w[1:], 12)
    w2m = Domain(C, 2 * m).element(1)
    masks = ([FftMask.sample(True, w2m, dom.group_gen_inv, m, pp, 100 + k) for k in range(3)]
             + [FftMask.sample(False, 1, dom.group_gen, m, pp, 200 + k) for k in range(3)])
    dm = DegRedMask.sample(pp, 1, m // pp.l, 300)
    g1m = [MsmMask.sample(pp, G1, o1, 500 + k) for k in range(4)]
    g2m = MsmMask.sample(pp, G2, o2, 600)
    res = g.dist_prove(C, pp, crs, qs, a_sh, ax, r, s, masks, dm, g1m, g2m, dom, G1, G2, o1, o2, seed=9)
    a = pp.unpack2([x[0] for x in res], o1)[0]
    b = pp.unpack2([x[1] for x in res], o2)[0]
    c = pp.unpack2([x[2] for x in res], o1)[0]
    assert G1.eq(a, A) and G2.eq(b, B) and G1.eq(c, Cc)
    # identical proof bytes (ark-serialize compressed, 128 B for BN254)
    enc = ser.proof_compressed(G1.to_affine(a), G2.to_affine(b), G1.to_affine(c), C.q)
    assert enc == ser.proof_compressed(G1.to_affine(A), G2.to_affine(B), G1.to_affine(Cc), C.q)
    assert len(enc) == 128
