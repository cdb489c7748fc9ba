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
