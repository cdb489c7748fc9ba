"""-m gpu parity of the circom front end and dealer pieces through the C ABI: R1CS -> QAP (qap.rs:42-89), canonical
byte form of Fr vectors (ser_net.rs:24-25), MsmMask::sample (dmsm/mod.rs:21-47)."""
import numpy as np
import pytest

import zksaas_amd as zk
from oracle import dist as od
from oracle import groth16 as og
from oracle import ser as oser
from oracle.curve import GroupOps, g1, g2
from oracle.params import BN254, CURVES
from oracle.prng import rand_vec
from zksaas_amd import circom, wire
from zksaas_amd import groth16 as zg
from zksaas_amd import sha256_circuit as sc
from zksaas_amd.api import ZK_G1, ZK_G2

from gpu_util import ctx, dec_jacobian, enc_affine, opp

pytestmark = pytest.mark.gpu


def test_r1cs_qap_matches_oracle_on_sha256():
    pp = ctx("bn254", 2)
    r1, w = sc.build(1, 2, BN254.r)
    dev = circom.DeviceR1cs(pp, r1)
    assert dev.log_m == 15
    a, b, c = dev.qap(pp.upload_fr(w))
    want = og.qap(BN254, og.R1CS(r1.num_instance_variables, r1.num_witness_variables, r1.a, r1.b, r1.c), w)
    m = 1 << dev.log_m
    assert pp.download_fr(a, m) == want.a and pp.download_fr(b, m) == want.b and pp.download_fr(c, m) == want.c
    # qap.rs:77-80 sanity on the device result: a*b = c on the constraint rows
    assert all(x == 0 for x in want.b[r1.num_constraints:])


def test_r1cs_qap_small_random_and_errors():
    import random
    pp = ctx("bn254", 2)
    p = BN254.r
    rng = random.Random(3)
    nv, ni, nc = 11, 2, 13

    def lc():
        return [(rng.randrange(p), rng.randrange(nv)) for _ in range(rng.randrange(0, 5))]
    r = sc.R1CS(ni, nv - ni, [lc() for _ in range(nc)], [lc() for _ in range(nc)], [[] for _ in range(nc)])
    w = [1] + rand_vec(9, nv - 1, p)
    dev = circom.DeviceR1cs(pp, r)
    assert dev.log_m == 4                                           # nc + ni = 15 -> 16
    a, b, c = dev.qap(pp.upload_fr(w))
    want = og.qap(BN254, og.R1CS(ni, nv - ni, r.a, r.b, r.c), w)
    assert (pp.download_fr(a, 16), pp.download_fr(b, 16), pp.download_fr(c, 16)) == (want.a, want.b, want.c)
    # empty rows and the zero padding (ragged input)
    assert want.a[15] == 0 and pp.download_fr(a, 16)[15] == 0
    dev.log_m = 3                                                   # domain smaller than nc + ni (qap.rs:52-56)
    with pytest.raises(zk.ZkError):
        dev.qap(pp.upload_fr(w))
    dev.log_m = 4
    dev.num_variables = 5                                           # wire indices beyond the assignment
    with pytest.raises(zk.ZkError):
        dev.qap(pp.upload_fr(w))


def test_witness_from_wtns_bytes_equals_witness_from_ints():
    pp = ctx("bn254", 2)
    r1, w = sc.build(1, 2, BN254.r)
    blob = circom.write_wtns(w, BN254.r)
    limbs = circom.wtns_to_limbs(blob, pp.fr.nl)
    w_d = zk.api.fr_from_bytes(pp, limbs.tobytes())
    assert pp.download_fr(w_d, 8) == w[:8]
    dev = circom.DeviceR1cs(pp, r1)
    w1 = zg.Witness(pp, "bn254", r1, w, seed=5, dev_r1cs=dev)
    w2 = zg.Witness(pp, "bn254", r1, w_d, seed=5, dev_r1cs=dev)
    for x, y in zip(w1.qap + [w1.a_share, w1.ax_share], w2.qap + [w2.a_share, w2.ax_share]):
        assert np.array_equal(x.to_numpy(), y.to_numpy())


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_fr_bytes_round_trip_and_invalid(curve):
    pp = ctx(curve, 2)
    p = CURVES[curve].r
    vals = [0, 1, p - 1] + rand_vec(21, 70, p)
    d = pp.upload_fr(vals)
    blob = wire.fr_vec_to_bytes(pp, d, len(vals))
    assert blob == oser.fr_vec(vals, pp.fr.nbytes)
    back, count = wire.fr_vec_from_bytes(pp, blob)
    assert count == len(vals) and pp.download_fr(back, count) == vals
    bad = bytearray(blob)
    bad[8:8 + pp.fr.nbytes] = p.to_bytes(pp.fr.nbytes, "little")   # an element equal to the modulus
    with pytest.raises(zk.ZkError):
        wire.fr_vec_from_bytes(pp, bytes(bad))
    with pytest.raises(ValueError):
        wire.fr_vec_from_bytes(pp, blob[:-1])
    empty, count = wire.fr_vec_from_bytes(pp, (0).to_bytes(8, "little"))
    assert count == 0


@pytest.mark.parametrize("curve,group", [("bn254", ZK_G1), ("bn254", ZK_G2), ("bls12_381", ZK_G1)])
def test_msm_mask_sample_matches_oracle(curve, group):
    pp, o = ctx(curve, 2), opp(curve, 2)
    c = CURVES[curve]
    is2 = group == ZK_G2
    G = g2(c) if is2 else g1(c)
    ops = GroupOps(G)
    gen = enc_affine(pp, [G.gen], is2)[0]
    mask = zk.MsmMask.sample(pp, group, gen, 91)
    want = od.MsmMask.sample(o, G, ops, 91)
    for i in range(pp.n):
        assert G.eq(dec_jacobian(pp, mask.in_mask[i], is2), want[i].in_mask)
        assert G.eq(dec_jacobian(pp, mask.out_mask[i], is2), want[i].out_mask)
    # dmsm/mod.rs:34: the out-mask secret is minus the sum of the in-mask secrets
    ins = o.unpack([dec_jacobian(pp, mask.in_mask[i], is2) for i in range(pp.n)], ops)
    outs = o.unpack([dec_jacobian(pp, mask.out_mask[i], is2) for i in range(pp.n)], ops)
    assert G.eq(G.neg(G.sum(ins)), outs[0]) and G.eq(outs[0], outs[1])
    # and it works as a mask: d_msm output unchanged by it (dmsm_test.rs:50-51 with a sampled mask)
    m, l = 8, 2
    y_pub = rand_vec(92, m, c.r)
    x_pub = [G.mul(G.from_affine(G.gen), k + 3) for k in range(m)]
    x_sh = od.transpose([o.det_pack(x_pub[j:j + l], ops) for j in range(0, m, l)])
    y_sh = od.transpose(od.pack_vec(y_pub, o, 93))
    bases = zk.DeviceBuffer.from_numpy(pp, np.concatenate([enc_affine(pp, G.batch_to_affine(v), is2) for v in x_sh]))
    scal = pp.upload_fr([v for vec in y_sh for v in vec])
    out = zk.d_msm(pp, group, bases, scal, m // l, mask)
    got = [dec_jacobian(pp, out[i], is2) for i in range(pp.n)]
    assert G.eq(o.unpack2(got, ops)[0], G.msm(G.batch_to_affine(x_pub), y_pub))


def test_bls12_377_points_decompress_on_the_device_with_tonelli_shanks():
    """BLS12-377 (q = 1 mod 4: q - 1 = 2^46 t), the curve the reference's own dist-primitives tests run on
    (dmsm/mod.rs:114-119): zk_points_decompress takes its square roots by Tonelli-Shanks.  The arkworks generator
    (ark-bls12-377 G1_GENERATOR_X / _Y, restated as literals here) compresses to its x with the sort flag the oracle's
    ark-serialize restatement gives and decompresses back; 60 random multiples, their negatives and the identity
    round-trip; an x off the curve is rejected with its index."""
    import random
    from zksaas_amd import wire
    from zksaas_amd.api import ZK_G1
    from oracle import curve as ocurve, ser as oser
    from oracle.params import CURVES
    from gpu_util import enc_affine
    cv = CURVES["bls12_377"]
    gx = 0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF
    gy = 0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6
    assert cv.g1 == (gx, gy) and (gy * gy - gx ** 3 - 1) % cv.q == 0
    pp = ctx("bls12_377", 2)
    G = ocurve.g1(cv)
    rng = random.Random(6)
    pts = [None, cv.g1] + [G.to_affine(G.mul(G.from_affine(cv.g1), rng.randrange(1, cv.r))) for _ in range(60)]
    pts += [G.to_affine(G.neg(G.from_affine(p))) for p in pts[1:8]]
    want = b"".join(oser.g1_compressed(p, cv.q) for p in pts)
    dev = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts, False))
    assert wire.points_to_bytes(pp, ZK_G1, dev, len(pts)) == want
    back, count = wire.points_from_bytes(pp, ZK_G1, want)
    assert count == len(pts) and np.array_equal(back.to_numpy()[: dev.nbytes // 8], dev.to_numpy())
    size = 48
    for k in range(1, 200):
        bad = bytearray(want)
        bad[5 * size + 7] ^= k
        try:
            wire.point_from_bytes(pp, bytes(bad[5 * size:6 * size]), False, "bls12_377")
        except ValueError:
            with pytest.raises(zk.ZkError) as e:
                wire.points_from_bytes(pp, ZK_G1, bytes(bad))
            assert e.value.code == 1 and "index 5" in e.value.msg
            break
    else:
        pytest.fail("no invalid mutation found")


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("is2", [False, True])
def test_point_vectors_compressed_on_the_device_match_the_host_codec(curve, is2):
    """zk_points_compress / zk_points_decompress (ser_net.rs:111-120 payloads) against wire.py / the oracle's
    ark-serialize restatement, including the identity, both roots, and rejection of invalid encodings."""
    import random
    from zksaas_amd import wire
    from zksaas_amd.api import ZK_G1, ZK_G2
    from oracle import curve as ocurve, ser as oser
    from oracle.params import CURVES
    from gpu_util import enc_affine
    cv = CURVES[curve]
    pp = ctx(curve, 2)
    G = ocurve.g2(cv) if is2 else ocurve.g1(cv)
    grp = ZK_G2 if is2 else ZK_G1
    rng = random.Random(5)
    pts = [None] + [G.to_affine(G.mul(G.from_affine(G.gen), rng.randrange(1, cv.r))) for _ in range(40)]
    pts += [G.to_affine(G.neg(G.from_affine(p))) for p in pts[1:6]]
    ofn = oser.g2_compressed if is2 else oser.g1_compressed
    want = b"".join(ofn(p, cv.q) for p in pts)
    dev = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts, is2))
    assert wire.points_to_bytes(pp, grp, dev, len(pts)) == want
    back, count = wire.points_from_bytes(pp, grp, want)
    assert count == len(pts)
    assert np.array_equal(back.to_numpy()[: dev.nbytes // 8], dev.to_numpy())
    # an x that is not on the curve, and a non-canonical x, are rejected with the index
    size = len(want) // len(pts)
    for k in range(1, 200):
        bad = bytearray(want)
        bad[3 * size + size // 2] ^= k
        try:
            wire.point_from_bytes(pp, bytes(bad[3 * size:4 * size]), is2, curve)
        except ValueError:
            with pytest.raises(zk.ZkError) as e:
                wire.points_from_bytes(pp, grp, bytes(bad))
            assert e.value.code == 1 and "index 3" in e.value.msg
            break
    else:
        pytest.fail("no invalid mutation found")
    # a WHOLE mpc-net frame (ser_net.rs:24-25, 111-112): Vec<G::Affine>::serialize_compressed = u64 length prefix + items
    frame = wire.point_vec_to_bytes(pp, grp, dev, len(pts))
    assert frame == len(pts).to_bytes(8, "little") + want
    back2, count2 = wire.point_vec_from_bytes(pp, grp, frame)
    assert count2 == len(pts) and np.array_equal(back2.to_numpy()[: dev.nbytes // 8], dev.to_numpy())
    with pytest.raises(ValueError):
        wire.point_vec_from_bytes(pp, grp, frame[:-1])                  # truncated payload
    with pytest.raises(ValueError):
        wire.point_vec_from_bytes(pp, grp, (len(pts) + 1).to_bytes(8, "little") + want)     # prefix says one more
    empty, zero = wire.point_vec_from_bytes(pp, grp, (0).to_bytes(8, "little"))
    assert zero == 0


def test_deg_red_over_group_elements_matches_oracle():
    """deg_red.rs:80-126 with T = G1 (and DegRedMask::sample with the group generator, :40-66) vs oracle/dist.py with
    group ops: every output share point equal, masks equal, and the result reconstructs to the secrets."""
    from zksaas_amd import groth16 as zg
    from zksaas_amd.api import ZK_G1
    from oracle import dist as od
    from oracle.curve import g1, GroupOps
    from oracle.params import BN254
    from oracle.prng import rand_fp
    from gpu_util import enc_affine, opp
    import ctypes as C
    pp, o = ctx("bn254", 2), opp("bn254", 2)
    G = g1(BN254)
    ops = GroupOps(G)
    gen = G.from_affine(BN254.g1)
    nch = 3
    # shares of degree 2(l+t-1): products of a degree-(l+t-1) scalar sharing with ... simply: random points per party
    # that ARE a valid higher-degree sharing: pack scalars twice and multiply share-wise, then lift to the group
    a = [rand_fp(90, i, o.p) for i in range(nch * o.l)]
    b = [rand_fp(91, i, o.p) for i in range(nch * o.l)]
    sa, sb = od.transpose(od.pack_vec(a, o, 92)), od.transpose(od.pack_vec(b, o, 93))
    prod = [[x * y % o.p for x, y in zip(sa[p], sb[p])] for p in range(o.n)]
    xs = [[G.mul(gen, v) for v in row] for row in prod]
    masks = od.DegRedMask.sample(o, gen, nch, 94, ops)
    want = od.deg_red(xs, masks, o, seed=95, ops=ops, gen=gen)
    aff = lambda rows: zk.DeviceBuffer.from_numpy(pp, np.concatenate([enc_affine(pp, G.batch_to_affine(r)) for r in rows]))
    g_aff = pp.fq.encode([1, 2]).reshape(-1)
    im, om = zk.DeviceBuffer(pp, o.n * nch * 64), zk.DeviceBuffer(pp, o.n * nch * 64)
    pp._check(pp.lib.zk_degred_mask_sample_points(pp.h, ZK_G1, g_aff.ctypes.data, nch, 94, im.ptr, om.ptr, None))
    assert np.array_equal(im.to_numpy(), aff([m.in_mask for m in masks]).to_numpy())
    assert np.array_equal(om.to_numpy(), aff([m.out_mask for m in masks]).to_numpy())
    out = zk.DeviceBuffer(pp, o.n * nch * 64)
    xs_d = aff(xs)                      # keep the buffer alive across the call
    pp._check(pp.lib.zk_deg_red_points(pp.h, ZK_G1, xs_d.ptr, im.ptr, om.ptr, nch, g_aff.ctypes.data, 95, out.ptr, None))
    assert np.array_equal(out.to_numpy(), aff(want).to_numpy())
    # zero masks, and the in-place call is refused
    out0 = zk.DeviceBuffer(pp, o.n * nch * 64)
    pp._check(pp.lib.zk_deg_red_points(pp.h, ZK_G1, xs_d.ptr, None, None, nch, g_aff.ctypes.data, 95, out0.ptr, None))
    zero = [od.DegRedMask.zero(nch, G.identity)] * o.n
    assert np.array_equal(out0.to_numpy(), aff(od.deg_red(xs, zero, o, seed=95, ops=ops, gen=gen)).to_numpy())
    assert pp.lib.zk_deg_red_points(pp.h, ZK_G1, xs_d.ptr, None, None, nch, g_aff.ctypes.data, 95, xs_d.ptr, None) == 4
    # the re-shared points reconstruct (degree l+t-1 now) to a_i * b_i * G
    for j in range(nch):
        rec = o.unpack([want[p][j] for p in range(o.n)], ops)
        for i in range(o.l):
            assert G.eq(rec[i], G.mul(gen, a[j * o.l + i] * b[j * o.l + i] % o.p))



@pytest.mark.parametrize("curve,is2", [("bn254", True), ("bls12_381", False), ("bls12_381", True)])
def test_deg_red_over_group_elements_other_groups_match_oracle(curve, is2):
    """deg_red.rs:80-126 with T = G2 and on BLS12-381 (round 2 covered G1 / BN254 only): output share points and the
    sampled masks equal oracle/dist.py with group operations, 3 chunks."""
    from zksaas_amd.api import ZK_G1, ZK_G2
    from oracle import dist as od
    from oracle.curve import g1, g2, GroupOps
    from oracle.params import CURVES
    from oracle.prng import rand_fp
    from gpu_util import enc_affine, opp
    cv = CURVES[curve]
    pp, o = ctx(curve, 2), opp(curve, 2)
    G = g2(cv) if is2 else g1(cv)
    grp = ZK_G2 if is2 else ZK_G1
    ops = GroupOps(G)
    gen = G.from_affine(G.gen)
    nch = 3
    a = [rand_fp(190, i, o.p) for i in range(nch * o.l)]
    b = [rand_fp(191, i, o.p) for i in range(nch * o.l)]
    sa, sb = od.transpose(od.pack_vec(a, o, 192)), od.transpose(od.pack_vec(b, o, 193))
    xs = [[G.mul(gen, x * y % o.p) for x, y in zip(sa[p], sb[p])] for p in range(o.n)]
    masks = od.DegRedMask.sample(o, gen, nch, 194, ops)
    want = od.deg_red(xs, masks, o, seed=195, ops=ops, gen=gen)
    width = pp.fq.nbytes * (4 if is2 else 2)
    aff = lambda rows: zk.DeviceBuffer.from_numpy(pp, np.concatenate([enc_affine(pp, G.batch_to_affine(r), is2) for r in rows]))
    g_aff = enc_affine(pp, [G.gen], is2).reshape(-1)
    im, om = zk.DeviceBuffer(pp, o.n * nch * width), zk.DeviceBuffer(pp, o.n * nch * width)
    pp._check(pp.lib.zk_degred_mask_sample_points(pp.h, grp, g_aff.ctypes.data, nch, 194, im.ptr, om.ptr, None))
    assert np.array_equal(im.to_numpy(), aff([m.in_mask for m in masks]).to_numpy())
    assert np.array_equal(om.to_numpy(), aff([m.out_mask for m in masks]).to_numpy())
    out = zk.DeviceBuffer(pp, o.n * nch * width)
    xs_d = aff(xs)
    pp._check(pp.lib.zk_deg_red_points(pp.h, grp, xs_d.ptr, im.ptr, om.ptr, nch, g_aff.ctypes.data, 195, out.ptr, None))
    assert np.array_equal(out.to_numpy(), aff(want).to_numpy())


@pytest.mark.parametrize("curve,is2", [("bn254", False), ("bls12_381", False), ("bn254", True)])
def test_deg_red_over_group_elements_1024_chunks(curve, is2):
    """The same with 1024 chunks (2048 secrets): the product sharing is dealt in the scalar field and lifted with
    zk_base_mul; after deg_red with sampled masks zk_pss_unpack_points (degree l + t - 1 now) must give a_i * b_i * G for
    every secret, and four chunks chosen across the vector equal the oracle's deg_red point for point."""
    from zksaas_amd import api, groth16 as zg
    from zksaas_amd.api import ZK_G1, ZK_G2
    from oracle import dist as od
    from oracle.curve import g1, g2, GroupOps
    from oracle.params import CURVES
    from oracle.prng import rand_vec
    from gpu_util import enc_affine, opp
    cv = CURVES[curve]
    pp, o = ctx(curve, 2), opp(curve, 2)
    G = g2(cv) if is2 else g1(cv)
    grp = ZK_G2 if is2 else ZK_G1
    ops = GroupOps(G)
    gen = G.from_affine(G.gen)
    nch = 1024
    a, b = rand_vec(290, nch * o.l, o.p), rand_vec(291, nch * o.l, o.p)
    sa, sb = od.transpose(od.pack_vec(a, o, 292)), od.transpose(od.pack_vec(b, o, 293))
    prod = [x * y % o.p for p in range(o.n) for x, y in zip(sa[p], sb[p])]                     # [n][nch] flat
    width = pp.fq.nbytes * (4 if is2 else 2)
    xs_d = zg.base_points(pp, grp, pp.upload_fr(prod), o.n * nch)
    g_aff = enc_affine(pp, [G.gen], is2).reshape(-1)
    im, om = zk.DeviceBuffer(pp, o.n * nch * width), zk.DeviceBuffer(pp, o.n * nch * width)
    pp._check(pp.lib.zk_degred_mask_sample_points(pp.h, grp, g_aff.ctypes.data, nch, 294, im.ptr, om.ptr, None))
    out = zk.DeviceBuffer(pp, o.n * nch * width)
    pp._check(pp.lib.zk_deg_red_points(pp.h, grp, xs_d.ptr, im.ptr, om.ptr, nch, g_aff.ctypes.data, 295, out.ptr, None))
    rec = api.unpack_points(pp, grp, out, nch, two=False)
    want = zg.base_points(pp, grp, pp.upload_fr([x * y % o.p for x, y in zip(a, b)]), nch * o.l)
    assert np.array_equal(rec.to_numpy(), want.to_numpy())
    # four chunks against the oracle (the king step is chunk-local; share randomness is indexed by the chunk)
    words = width // 8
    rows = lambda buf: buf.to_numpy().reshape(o.n, nch, words)
    x_np, im_np, om_np, out_np = rows(xs_d), rows(im), rows(om), rows(out)

    def dec(v):
        c = pp.fq.decode(v.reshape(-1, pp.fq.nl))
        if not any(c):
            return G.identity
        return G.from_affine(((c[0], c[1]), (c[2], c[3])) if is2 else (c[0], c[1]))
    for j in (0, 1, 517, 1023):
        xm = [ops.add(dec(x_np[p, j]), dec(im_np[p, j])) for p in range(o.n)]
        sec = o.unpack2(xm, ops)
        from oracle.prng import rand_fp
        rnd = [G.mul(gen, rand_fp(295, j * o.t + i, o.p)) for i in range(o.t)]
        sh = o.pack(sec, rnd, ops)
        for p in range(o.n):
            assert G.eq(ops.add(sh[p], dec(om_np[p, j])), dec(out_np[p, j])), (j, p)
