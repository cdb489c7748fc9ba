"""GPU parity for the Groth16 composition (groth16/src/ext_wit.rs, prove.rs, examples/sha256.rs)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from zksaas_amd import sha256_circuit as sc
from oracle import dist as od
from oracle import groth16 as og
from oracle import ser
from oracle.curve import g1, g2, GroupOps
from oracle.field import Domain
from oracle.params import BN254
from oracle.prng import rand_fp

from gpu_util import ctx, opp, down_parties, up_parties, dec_jacobian, enc_jacobian
from test_oracle_groth16 import small_r1cs

P = BN254.r


def _trapdoor(seed):
    return [rand_fp(seed, i, P) for i in range(5)]


def test_circom_h_matches_oracle():  # ext_wit.rs:419-538 (a = b = [0..m), c = a*b), masked
    m, l = 1024, 2
    pp, o = ctx("bn254", l), opp("bn254", l)
    dom = Domain(BN254, m)
    a = list(range(m))
    c = [x * x % P for x in a]
    qs = og.QAP(0, 0, a, a, c, dom).pss(o, 3)
    w2m = Domain(BN254, 2 * m).element(1)
    fm = ([od.FftMask.sample(True, w2m, dom.group_gen_inv, m, o, 40 + k) for k in range(3)]
          + [od.FftMask.sample(False, 1, dom.group_gen, m, o, 50 + k) for k in range(3)])
    dm = od.DegRedMask.sample(o, 1, m // l, 60)
    want = og.circom_h(qs, fm, dm, o, dom, seed=4)
    bufs = [up_parties(pp, [qs[i][k] for i in range(o.n)]) for k in range(3)]
    keep = []
    mk = zg.Masks()
    for k in range(6):
        bi = up_parties(pp, [x.in_mask for x in fm[k]])
        bo = up_parties(pp, [x.out_mask for x in fm[k]])
        keep += [bi, bo]
        mk.fft_in[k], mk.fft_out[k] = bi.ptr, bo.ptr
    di, do = up_parties(pp, [x.in_mask for x in dm]), up_parties(pp, [x.out_mask for x in dm])
    mk.degred_in, mk.degred_out = di.ptr, do.ptr
    h = pp.alloc_fr(o.n * (m // l))
    pp._check(pp.lib.zk_circom_h(pp.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 10, C.byref(mk), 4, h.ptr, None))
    assert down_parties(pp, h, o.n, m // l) == want
    # reconstructs to the in-test reference (ext_wit.rs:239-285) through unpack2
    assert pp.download_fr(pp.unpack2(h, m // l)) == og.circom_ref(a, a, c, dom)


def _reconstruct(o, G, ops, shares):
    return o.unpack2(shares, ops)[0]          # sha256.rs:375-377


@pytest.mark.parametrize("r_zero", [False, True])
def test_small_circuit_proof_equals_oracle(r_zero):
    """sha256.rs flow on a small R1CS: GPU (setup + dealing + prover) == oracle local prover == closed form."""
    r1, w = small_r1cs()
    pp, o = ctx("bn254", 2), opp("bn254", 2)
    td = _trapdoor(42)
    setup = zg.SetupScalars("bn254", r1, *td)
    okey = og.setup_scalars(BN254, r1, og.Trapdoor(*td))
    assert setup.a_query == okey.a_query and setup.b_query == okey.b_query
    assert setup.l_query == okey.l_query and setup.h_query == okey.h_query
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bn254", r1, w, seed=5)
    r = 0 if r_zero else rand_fp(43, 0, P)
    s = rand_fp(43, 1, P)
    pa, pb, pc = zg.prove(pp, crs, wit, r, s, seed=9)
    G1, G2 = g1(BN254), g2(BN254)
    o1, o2 = GroupOps(G1), GroupOps(G2)
    A = _reconstruct(o, G1, o1, [dec_jacobian(pp, pa[i]) for i in range(o.n)])
    B = _reconstruct(o, G2, o2, [dec_jacobian(pp, pb[i], True) for i in range(o.n)])
    Cc = _reconstruct(o, G1, o1, [dec_jacobian(pp, pc[i]) for i in range(o.n)])
    sa, sb, sc_ = og.prove_scalars(BN254, r1, okey, w, r, s)
    assert og.verify_scalars(BN254, r1, okey, w, (sa, sb, sc_))
    assert G1.eq(A, G1.mul(G1.from_affine(BN254.g1), sa))
    assert G2.eq(B, G2.mul(G2.from_affine(BN254.g2), sb))
    assert G1.eq(Cc, G1.mul(G1.from_affine(BN254.g1), sc_))
    # identical compressed proof bytes vs the oracle's arkworks-style local prover
    pk = og.proving_key_points(okey, G1, G2)
    lA, lB, lC = og.create_proof_local(BN254, r1, pk, G1, G2, w, r, s)
    enc = lambda a, b, c: ser.proof_compressed(G1.to_affine(a), G2.to_affine(b), G1.to_affine(c), BN254.q)
    assert enc(A, B, Cc) == enc(lA, lB, lC)


def test_small_circuit_with_all_masks():
    r1, w = small_r1cs()
    pp, o = ctx("bn254", 2), opp("bn254", 2)
    td = _trapdoor(44)
    setup = zg.SetupScalars("bn254", r1, *td)
    okey = og.setup_scalars(BN254, r1, og.Trapdoor(*td))
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bn254", r1, w, seed=6)
    m, l = setup.m, 2
    dom = Domain(BN254, m)
    G1, G2 = g1(BN254), g2(BN254)
    o1, o2 = GroupOps(G1), GroupOps(G2)
    keep = []
    mk = zg.Masks()
    for k in range(6):
        fmk = zk.FftMask.sample(pp, k < 3, Domain(BN254, 2 * m).element(1) if k < 3 else None, 1 if k < 3 else 0,
                                setup.log_m, 100 + k)
        keep.append(fmk)
        mk.fft_in[k], mk.fft_out[k] = fmk.in_mask.ptr, fmk.out_mask.ptr
    dm = zk.DegRedMask.sample(pp, m // l, 200)
    mk.degred_in, mk.degred_out = dm.in_mask.ptr, dm.out_mask.ptr
    for k in range(5):
        G, ops, is2 = (G2, o2, True) if k == 2 else (G1, o1, False)
        om = od.MsmMask.sample(o, G, ops, 300 + k)
        ai = np.stack([enc_jacobian(pp, x.in_mask, is2) for x in om])
        ao = np.stack([enc_jacobian(pp, x.out_mask, is2) for x in om])
        keep += [ai, ao]
        mk.msm_in[k], mk.msm_out[k] = ai.ctypes.data, ao.ctypes.data
    r, s = rand_fp(45, 0, P), rand_fp(45, 1, P)
    pa, pb, pc = zg.prove(pp, crs, wit, r, s, masks=mk, seed=11)
    A = _reconstruct(o, G1, o1, [dec_jacobian(pp, pa[i]) for i in range(o.n)])
    B = _reconstruct(o, G2, o2, [dec_jacobian(pp, pb[i], True) for i in range(o.n)])
    Cc = _reconstruct(o, G1, o1, [dec_jacobian(pp, pc[i]) for i in range(o.n)])
    sa, sb, sc_ = og.prove_scalars(BN254, r1, okey, w, r, s)
    assert G1.eq(A, G1.mul(G1.from_affine(BN254.g1), sa))
    assert G2.eq(B, G2.mul(G2.from_affine(BN254.g2), sb))
    assert G1.eq(Cc, G1.mul(G1.from_affine(BN254.g1), sc_))


def test_sha256_fixture_proof():
    """BASELINE configs 1/4: the SHA-256 circuit (a = 1, b = 2), m = 2^15, l = 2, n = 8."""
    r1, w = sc.build(1, 2, P)
    assert w[1] == 72587776472194017031617589674261467945970986113287823188107011979     # sha256.rs:392-393
    pp, o = ctx("bn254", 2), opp("bn254", 2)
    td = _trapdoor(42)
    setup = zg.SetupScalars("bn254", r1, *td)
    assert setup.log_m == 15
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bn254", r1, w, seed=7)
    assert crs.len_u == 16384 and crs.len_a == (r1.num_variables - 1 + 1) // 2
    r, s = rand_fp(43, 0, P), rand_fp(43, 1, P)
    pa, pb, pc = zg.prove(pp, crs, wit, r, s, seed=13)
    G1, G2 = g1(BN254), g2(BN254)
    o1, o2 = GroupOps(G1), GroupOps(G2)
    A = _reconstruct(o, G1, o1, [dec_jacobian(pp, pa[i]) for i in range(o.n)])
    B = _reconstruct(o, G2, o2, [dec_jacobian(pp, pb[i], True) for i in range(o.n)])
    Cc = _reconstruct(o, G1, o1, [dec_jacobian(pp, pc[i]) for i in range(o.n)])
    R = og.R1CS(2, r1.num_witness_variables, r1.a, r1.b, r1.c)
    okey = og.setup_scalars(BN254, R, og.Trapdoor(*td))
    sa, sb, sc_ = og.prove_scalars(BN254, R, okey, w, r, s)
    assert og.verify_scalars(BN254, R, okey, w, (sa, sb, sc_))          # the Groth16 equation holds
    assert G1.eq(A, G1.mul(G1.from_affine(BN254.g1), sa))
    assert G2.eq(B, G2.mul(G2.from_affine(BN254.g2), sb))
    assert G1.eq(Cc, G1.mul(G1.from_affine(BN254.g1), sc_))


def test_libsnark_h_matches_oracle():  # ext_wit.rs:287-417 (m = 32, masked)
    m, l = 32, 2
    pp, o = ctx("bn254", l), opp("bn254", l)
    dom = Domain(BN254, m)
    a = list(range(m))
    c = [x * x % P for x in a]
    qs = og.QAP(0, 0, a, a, c, dom).pss(o, 3)
    co = dom.get_coset(BN254.r_gen)
    fm = ([od.FftMask.sample(True, co.offset, dom.group_gen_inv, m, o, 10 + k) for k in range(3)]
          + [od.FftMask.sample(True, 1, dom.group_gen, m, o, 20 + k) for k in range(3)]
          + [od.FftMask.sample(False, co.offset_inv, dom.group_gen_inv, m, o, 30)])
    want = og.libsnark_h(qs, fm, o, dom, seed=2)
    bufs = [up_parties(pp, [qs[i][k] for i in range(o.n)]) for k in range(3)]
    masks = [zk.FftMask(up_parties(pp, [x.in_mask for x in f]), up_parties(pp, [x.out_mask for x in f])) for f in fm]
    h = zg.libsnark_h(pp, bufs, masks, 5, seed=2)
    assert down_parties(pp, h, o.n, m // l) == want
    assert pp.download_fr(pp.unpack2(h, m // l)) == og.libsnark_ref(a, a, c, dom)


def test_crs_packing_over_group_elements_equals_trapdoor_path():
    """proving_key.rs:47-123: det_pack over the curve points (zk_pss_pack_points, no trapdoor) must give exactly the
    shares obtained from packing the discrete logs (zk_pss_det_pack + zk_base_mul), point for point."""
    r1, w = small_r1cs()
    pp = ctx("bn254", 2)
    setup = zg.SetupScalars("bn254", r1, *_trapdoor(48))
    crs = zg.Crs(pp, setup, keep_unpacked=True)
    pk = {k: (v, n_) for (k, v), n_ in zip(crs.unpacked.items(), (len(setup.a_query), len(setup.b_query),
                                                                   len(setup.b_query), len(setup.l_query),
                                                                   len(setup.h_query)))}
    crs2 = zg.crs_from_proving_key(pp, pk, crs)
    assert (crs2.len_a, crs2.len_w, crs2.len_u) == (crs.len_a, crs.len_w, crs.len_u)
    for name in ("s", "h", "v", "w", "u"):
        assert np.array_equal(getattr(crs2, name).to_numpy(), getattr(crs, name).to_numpy()), name
    # and the prover accepts it
    wit = zg.Witness(pp, "bn254", r1, w, seed=8)
    r, s = rand_fp(49, 0, P), rand_fp(49, 1, P)
    a1 = zg.prove(pp, crs, wit, r, s, seed=3)
    a2 = zg.prove(pp, crs2, wit, r, s, seed=3)
    G1 = g1(BN254)
    assert G1.eq(dec_jacobian(pp, a1[2][0]), dec_jacobian(pp, a2[2][0]))


def test_pack_points_with_random_points_reconstructs():
    """pss.rs:90-122 on group elements (dmsm/mod.rs:127-137 pack_unpack_test): unpack(pack(points)) == points."""
    pp, o = ctx("bn254", 2), opp("bn254", 2)
    G1 = g1(BN254)
    ops = GroupOps(G1)
    gen = G1.from_affine(BN254.g1)
    nch = 5
    pts = [G1.mul(gen, rand_fp(50, i, P)) for i in range(nch * 4)]        # per chunk: l secrets + t randoms
    aff = G1.batch_to_affine(pts)
    from gpu_util import enc_affine
    sh = zg.pack_points(pp, zk.api.ZK_G1, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, aff)), nch, 4)
    rows = sh.to_numpy().reshape(o.n, nch, 8)
    for j in range(nch):
        want = o.pack(pts[4 * j:4 * j + 2], pts[4 * j + 2:4 * j + 4], ops)
        for p_ in range(o.n):
            v = pp.fq.decode(rows[p_, j].reshape(2, 4))
            assert (v[0], v[1]) == G1.to_affine(want[p_])


@pytest.mark.parametrize("curve,group", [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bls12_381", "g2"),
                                         ("bls12_377", "g1")])
def test_det_pack_over_points_edge_cases_match_oracle(curve, group):
    """proving_key.rs:72-86 (det_pack over curve points, l = 2) chunk by chunk against the oracle's det_pack over GroupOps,
    with the chunks a CRS never holds: an identity in either slot, both, equal points, opposite points.  G2 runs the
    quad-split kernel (one base-field value per lane), G1 the one-lane joint-sparse-form kernel; both walk the scalars split
    by the curve's endomorphism (csrc/glv.hpp), and a second context with pack_glv = 0 must give the same shares."""
    from oracle.curve import g2 as og2
    from oracle.params import CURVES
    from gpu_util import enc_affine
    cv = CURVES[curve]
    pp, o = ctx(curve, 2), opp(curve, 2)
    is2 = group == "g2"
    G = og2(cv) if is2 else g1(cv)
    ops = GroupOps(G)
    gen = G.from_affine(cv.g2 if is2 else cv.g1)
    a, b, c, d = (G.to_affine(G.mul(gen, rand_fp(77, i, cv.r))) for i in range(4))
    neg = lambda pt: (pt[0], G.F.neg(pt[1]))
    chunks = [(a, b), (None, b), (a, None), (None, None), (c, c), (c, neg(c)), (d, a), (b, neg(a))]
    flat = [pt for ch in chunks for pt in ch]
    nch = len(chunks)
    sh = zg.pack_points(pp, zk.api.ZK_G2 if is2 else zk.api.ZK_G1,
                        zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, flat, g2=is2)), nch, 2)
    ncoord = 4 if is2 else 2
    rows = sh.to_numpy().reshape(o.n, nch, ncoord * pp.fq.nl)
    plain = zk.PackedSharingParams(curve, 2)
    plain.set_option("pack_glv", 0)
    sh0 = zg.pack_points(plain, zk.api.ZK_G2 if is2 else zk.api.ZK_G1,
                         zk.DeviceBuffer.from_numpy(plain, enc_affine(plain, flat, g2=is2)), nch, 2)
    assert np.array_equal(sh0.to_numpy(), sh.to_numpy())
    plain.close()
    for j, ch in enumerate(chunks):
        want = o.det_pack([G.from_affine(ch[0]), G.from_affine(ch[1])], ops)
        for p_ in range(o.n):
            v = pp.fq.decode(rows[p_, j].reshape(ncoord, pp.fq.nl))
            got = ((v[0], v[1]), (v[2], v[3])) if is2 else (v[0], v[1])
            w = G.to_affine(want[p_])
            if w is None:
                assert all(x == 0 for x in v), (j, p_)
            else:
                assert got == w, (j, p_)


def test_proof_with_fixed_base_tables_equals_proof_without():
    """Crs.precompute() (zk_msm_precompute on the five query vectors): the SHA-256 circuit proof is the same group
    elements with and without the tables, for r = 0 as well (H skipped, S alone)."""
    import zksaas_amd as zk
    from zksaas_amd import wire
    from bench import build_inputs
    pp = zk.PackedSharingParams("bn254", 2)
    r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)

    def norm(pf):
        return (wire.jacobian_to_affine(pp, pf[0][0], False), wire.jacobian_to_affine(pp, pf[1][0], True),
                wire.jacobian_to_affine(pp, pf[2][0], False))
    plain = [norm(zg.prove(pp, crs, wit, rr, s, seed=5)) for rr in (r, 0)]
    crs.precompute()
    tabled = [norm(zg.prove(pp, crs, wit, rr, s, seed=5)) for rr in (r, 0)]
    assert plain == tabled
