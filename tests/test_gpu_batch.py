"""GPU parity for the batched entry points (zk_msm_batch, zk_groth16_prove_batch) and for the last mile of a proof
(zk_pss_unpack_points / zk_pss_unpack2_points / zk_groth16_reconstruct): secret-sharing/src/pss.rs:125-221 with
T = curve point, groth16/examples/sha256.rs:316-377, dist-primitives/src/dmsm/mod.rs:73."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from zksaas_amd import api
from zksaas_amd import groth16 as zg
from zksaas_amd.api import ZK_G1, ZK_G2, msm
from oracle import dist as od
from oracle import groth16 as og
from oracle import ser
from oracle.curve import g1, g2, GroupOps
from oracle.params import BN254, CURVES
from oracle.prng import rand_fp, rand_vec

from gpu_util import ctx, opp, up, enc_affine, dec_jacobian, enc_jacobian

P = BN254.r


def _circuit(Pm, x=7, y=5, nc=11):
    """test_oracle_groth16.small_r1cs over any scalar field with chosen private inputs (same R1CS, another witness):
    w[k] = (w[k-1] + 3) * (w[k-1] + w[k-2]); public output = last wire."""
    w = [1, 0, x % Pm, y % Pm]
    A, B, Cm = [], [], []
    for _ in range(nc - 1):
        k = len(w)
        w.append((w[k - 1] + 3) * (w[k - 1] + w[k - 2]) % Pm)
        A.append([(1, k - 1), (3, 0)])
        B.append([(1, k - 1), (1, k - 2)])
        Cm.append([(1, k)])
    A.append([(1, len(w) - 1)])
    B.append([(1, 0)])
    Cm.append([(1, 1)])
    w[1] = w[-1]
    return og.R1CS(2, len(w) - 2, A, B, Cm), w


def _points(G, curve, count, seed):
    gen = G.from_affine(G.gen)
    return G.batch_to_affine([G.mul(gen, rand_fp(seed, i, curve.r)) for i in range(count)])


# ------------------------------------------------------------------------------------------------ zk_msm_batch
@pytest.mark.parametrize("curve,group", [("bn254", ZK_G1), ("bn254", ZK_G2), ("bls12_381", ZK_G1)])
@pytest.mark.parametrize("table", [False, True])
def test_msm_batch_matches_oracle_on_both_sort_paths(curve, group, table):
    """One base vector against 1, 3 and 5 scalar vectors (random, edge values, all ones, all zeros): every result
    equals the oracle's MSM and the single-vector zk_msm; both sort paths, with and without a fixed-base table, an
    identity base in the vector, ragged length."""
    c = CURVES[curve]
    pp = zk.PackedSharingParams(curve, 2)
    is2 = group == ZK_G2
    G = g2(c) if is2 else g1(c)
    n = 2048 + 333
    distinct = _points(G, c, 24, 164)
    pts = [distinct[i % 24] for i in range(n)]
    pts[9] = None
    bases = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts, is2))
    vecs = [rand_vec(165, n, c.r), [0, 1, c.r - 1, 2, c.r - 2] * (n // 5) + [1] * (n % 5), [1] * n, [0] * n,
            rand_vec(166, n, c.r)]
    want = []
    for sc in vecs:
        agg = [0] * 24
        for i, s in enumerate(sc):
            if i != 9:
                agg[i % 24] = (agg[i % 24] + s) % c.r
        want.append(G.msm(distinct, agg))
    dev = [up(pp, sc) for sc in vecs]
    if table:
        api.msm_precompute(pp, group, bases, n)
    for big_min in (0, 1 << 40):
        pp.set_option("msm_bigsort_min", big_min)
        for nv in (1, 3, 5):
            got = api.msm_batch(pp, group, bases, dev[:nv], n)
            for b in range(nv):
                assert G.eq(dec_jacobian(pp, got[b], is2), want[b]), (big_min, nv, b)
        single = dec_jacobian(pp, msm(pp, group, bases, dev[0], n), is2)
        assert G.eq(single, want[0])


def test_msm_batch_of_sixteen_and_bad_sizes():
    c = CURVES["bn254"]
    pp = zk.PackedSharingParams("bn254", 2)
    G = g1(c)
    n = 700
    distinct = _points(G, c, 16, 167)
    bases = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, [distinct[i % 16] for i in range(n)]))
    vecs = [rand_vec(170 + b, n, c.r) for b in range(16)]
    dev = [up(pp, sc) for sc in vecs]
    got = api.msm_batch(pp, ZK_G1, bases, dev, n)
    for b in range(16):
        agg = [0] * 16
        for i, s in enumerate(vecs[b]):
            agg[i % 16] = (agg[i % 16] + s) % c.r
        assert G.eq(dec_jacobian(pp, got[b]), G.msm(distinct, agg)), b
    with pytest.raises(zk.ZkError):
        api.msm_batch(pp, ZK_G1, bases, dev + [dev[0]], n)          # 17 vectors
    # empty base vector: identities
    out = api.msm_batch(pp, ZK_G1, bases, dev[:2], 0)
    assert all(dec_jacobian(pp, out[b])[2] == 0 for b in range(2))


# ------------------------------------------------------------------------------------------------ unpack over points
@pytest.mark.parametrize("curve,group", [("bn254", ZK_G1), ("bn254", ZK_G2), ("bls12_381", ZK_G1)])
def test_unpack_points_matches_oracle(curve, group):
    """pack over group elements (oracle) -> zk_pss_unpack_points / zk_pss_unpack2_points == oracle.pss.unpack / unpack2
    with GroupOps, all parties and with one / two parties dropped (lagrange_unpack, pss.rs:170-221)."""
    c = CURVES[curve]
    pp, o = ctx(curve, 2), opp(curve, 2)
    is2 = group == ZK_G2
    G = g2(c) if is2 else g1(c)
    ops = GroupOps(G)
    gen = G.from_affine(G.gen)
    nch = 6
    secrets = [[G.mul(gen, rand_fp(180, j * 4 + i, c.r)) for i in range(2)] for j in range(nch)]
    rnd = [[G.mul(gen, rand_fp(181, j * 4 + i, c.r)) for i in range(2)] for j in range(nch)]
    shares = [o.pack(secrets[j], rnd[j], ops) for j in range(nch)]            # [chunk][party]
    # products of two degree-(l+t-1) sharings are what unpack2 is for: add a second sharing's worth by using the
    # shares themselves (unpack2 of a degree-(l+t-1) sharing returns the secrets as well)
    rows = [G.batch_to_affine([shares[j][p] for j in range(nch)]) for p in range(o.n)]      # [party][chunk]
    width = 4 if is2 else 2
    flat = np.concatenate([enc_affine(pp, rows[p], is2) for p in range(o.n)])
    sh_d = zk.DeviceBuffer.from_numpy(pp, flat)

    def dec(buf, count):
        a = buf.to_numpy().reshape(count, width, pp.fq.nl)
        out = []
        for k in range(count):
            v = pp.fq.decode(a[k])
            if not any(v):
                out.append(None)
            else:
                out.append(((v[0], v[1]), (v[2], v[3])) if is2 else (v[0], v[1]))
        return out

    got1 = dec(api.unpack_points(pp, group, sh_d, nch, two=False), nch * 2)
    got2 = dec(api.unpack_points(pp, group, sh_d, nch), nch * 2)
    for j in range(nch):
        col = [shares[j][p] for p in range(o.n)]
        w1, w2 = o.unpack(col, ops), o.unpack2(col, ops)
        for i in range(2):
            assert got1[2 * j + i] == G.to_affine(w1[i]) == G.to_affine(secrets[j][i])
            assert got2[2 * j + i] == G.to_affine(w2[i])
    for present in ([0, 1, 2, 3, 4, 5, 7], [1, 2, 3, 4, 5, 6, 7]):
        sub = np.concatenate([enc_affine(pp, rows[p], is2) for p in present])
        got = dec(api.unpack_points(pp, group, zk.DeviceBuffer.from_numpy(pp, sub), nch, parties=present), nch * 2)
        for j in range(nch):
            want = o.lagrange_unpack([shares[j][p] for p in present], present, ops)
            for i in range(2):
                assert got[2 * j + i] == G.to_affine(want[i])
    with pytest.raises(zk.ZkError) as e:
        api.unpack_points(pp, group, sh_d, nch, parties=[0, 1, 2, 3, 4, 5])         # 6 <= 2 (t + l - 1)
    assert e.value.code == 2


# ------------------------------------------------------------------------------------------------ batch prover
def _masks(pp, o, setup, seed):
    """all twelve masks of one proof, sampled by the library's dealers (ProofMasks) -- distinct per seed"""
    return zg.ProofMasks(pp, setup.log_m, seed=seed)


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_prove_batch_equals_single_proofs_and_oracle_bytes(curve, masked):
    """Three DIFFERENT witnesses (and (r, s), and mask sets) of one small circuit through zk_groth16_prove_batch: every
    party's share equals the one-at-a-time prover's group element, and the reconstructed proof (zk_groth16_reconstruct:
    unpack2 over the shares, slot 0, compressed) equals the oracle's arkworks-style local prover byte for byte."""
    c = CURVES[curve]
    r1, w0 = _circuit(c.r)
    pp, o = zk.PackedSharingParams(curve, 2), opp(curve, 2)
    pp.set_option("rng_replay", 1)
    td = [rand_fp(190, i, c.r) for i in range(5)]
    setup = zg.SetupScalars(curve, r1, *td)
    okey = og.setup_scalars(c, r1, og.Trapdoor(*td))
    crs = zg.Crs(pp, setup)
    G1, G2 = g1(c), g2(c)
    pk = og.proving_key_points(okey, G1, G2)
    nb = 3
    ws = [_circuit(c.r, 7 + 11 * b, 5 + b)[1] for b in range(nb)]
    assert all(og.is_satisfied(r1, w_, c.r) for w_ in ws) and ws[0] != ws[1]
    wits = [zg.Witness(pp, curve, r1, ws[b], seed=20 + b) for b in range(nb)]
    rs = [rand_fp(191, b, c.r) for b in range(nb)]
    ss = [rand_fp(192, b, c.r) for b in range(nb)]
    rs[1] = 0                                                   # one proof of the batch with r = 0 (H skipped for it)
    mks = [_masks(pp, o, setup, 500 + 40 * b) for b in range(nb)] if masked else None
    batch = zg.prove_batch(pp, crs, wits, rs, ss, masks=mks, seed=77)
    from zksaas_amd import wire
    for b in range(nb):
        single = zg.prove(pp, crs, wits[b], rs[b], ss[b], masks=None if mks is None else mks[b], seed=77 + 16 * b)
        for k, is2 in ((0, False), (1, True), (2, False)):
            for q in range(pp.n):
                assert wire.jacobian_to_affine(pp, batch[b][k][q], is2) == wire.jacobian_to_affine(pp, single[k][q], is2), (b, k, q)
        aff, raw = zg.reconstruct(pp, batch[b])
        lA, lB, lC = og.create_proof_local(c, r1, pk, G1, G2, ws[b], rs[b], ss[b])
        assert raw == ser.proof_compressed(G1.to_affine(lA), G2.to_affine(lB), G1.to_affine(lC), c.q), b
        # one party dropped: lagrange_unpack over the remaining seven gives the same proof
        present = [0, 1, 2, 4, 5, 6, 7]
        aff2, raw2 = zg.reconstruct(pp, tuple(x[present] for x in batch[b]), parties=present)
        assert raw2 == raw and np.array_equal(aff, aff2)
    with pytest.raises(zk.ZkError) as e:
        zg.reconstruct(pp, tuple(x[:6] for x in batch[0]), parties=[0, 1, 2, 3, 4, 5])
    assert e.value.code == 2                                    # "Not enough shares" (pss.rs:183-186)


def test_sha256_batch_with_all_masks_verifies_and_matches_single():
    """BASELINE configs[3] as a batch of 4: the padded SHA-256 circuit (29 823 wires), all twelve masks, tables on; every
    proof of the batch equals the one-at-a-time proof and its reconstruction verifies by pairing."""
    from bench import build_inputs
    from oracle import pairing as opair
    from oracle.curve import g1 as _g1
    pp = zk.PackedSharingParams("bn254", 2)
    r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
    crs.precompute()
    masks = zg.ProofMasks(pp, wit.log_m, seed=77)
    nb = 4
    rs = [r] + [rand_fp(193, b, P) for b in range(1, nb)]
    ss = [s] + [rand_fp(194, b, P) for b in range(1, nb)]
    batch = zg.prove_batch(pp, crs, [wit] * nb, rs, ss, masks=[masks] * nb, seed=5)
    from zksaas_amd import wire
    vk = zg.verifying_key(pp, setup)
    for b in range(nb):
        single = zg.prove(pp, crs, wit, rs[b], ss[b], masks=masks, seed=9)
        for k, is2 in ((0, False), (1, True), (2, False)):
            for q in range(pp.n):
                assert wire.jacobian_to_affine(pp, batch[b][k][q], is2) == wire.jacobian_to_affine(pp, single[k][q], is2)
    aff, raw = zg.reconstruct(pp, batch[1])
    v = pp.fq.decode(aff.reshape(-1, pp.fq.nl))
    A, B, Cc = (v[0], v[1]), ((v[2], v[3]), (v[4], v[5])), (v[6], v[7])
    ovk = opair.VerifyingKey(vk["alpha_g1"], vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"], vk["gamma_abc_g1"])
    assert opair.verify_proof(BN254, ovk, (A, B, Cc), [w[1]], _g1(BN254))
    assert len(raw) == 128
