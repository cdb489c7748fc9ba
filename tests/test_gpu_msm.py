"""GPU parity for MSM / d_msm (dist-primitives/src/dmsm) and deg_red / d_pp, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from zksaas_amd.api import ZK_G1, ZK_G2, msm
from oracle import dist as od
from oracle.curve import g1, g2, GroupOps
from oracle.params import CURVES
from oracle.prng import rand_fp, rand_vec

from gpu_util import ctx, opp, up, up_parties, down_parties, enc_affine, dec_jacobian, enc_jacobian


def _points(G, curve, count, seed):
    gen = G.from_affine(G.gen)
    return G.batch_to_affine([G.mul(gen, rand_fp(seed, i, curve.r)) for i in range(count)])


@pytest.mark.parametrize("curve,count", [("bn254", 1), ("bn254", 7), ("bn254", 300), ("bls12_377", 65),
                                          ("bls12_381", 40)])
def test_msm_g1_matches_oracle(curve, count):
    c = CURVES[curve]
    pp = ctx(curve, 2)
    G = g1(c)
    pts = _points(G, c, count, 60)
    sc = rand_vec(61, count, c.r)
    got = msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts)), up(pp, sc), count)
    assert G.eq(dec_jacobian(pp, got), G.msm(pts, sc))


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_msm_g2_matches_oracle(curve):
    c = CURVES[curve]
    pp = ctx(curve, 2)
    G = g2(c)
    pts = _points(G, c, 33, 62)
    sc = rand_vec(63, 33, c.r)
    got = msm(pp, ZK_G2, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts, True)), up(pp, sc), 33)
    assert G.eq(dec_jacobian(pp, got, True), G.msm(pts, sc))


def test_msm_edge_cases():
    """all-ones scalars (dmsm/mod.rs:147, one bucket), zeros, r-1, identity bases, repeated and opposite points."""
    c = CURVES["bn254"]
    pp = ctx("bn254", 2)
    G = g1(c)
    pts = _points(G, c, 200, 64)
    bases = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts))
    for sc in ([1] * 200, [0] * 200, [c.r - 1] * 200, [1 << 253 if i % 2 else 3 for i in range(200)]):
        got = msm(pp, ZK_G1, bases, up(pp, sc), 200)
        assert G.eq(dec_jacobian(pp, got), G.msm(pts, sc))
    P = pts[0]
    negP = (P[0], (-P[1]) % c.q)
    special = [P, P, negP, None, P, None, negP, negP]
    sc = [5, 5, 5, 9, 1, 0, 6, 2]          # 5P + 5P - 5P + P - 6P - 2P = -2P
    got = msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, special)), up(pp, sc), 8)
    want = G.neg(G.double(G.from_affine(P)))
    assert G.eq(dec_jacobian(pp, got), want)
    # empty msm is the identity
    got = msm(pp, ZK_G1, pp.alloc_fr(0), pp.alloc_fr(0), 0)
    assert G.is_identity(dec_jacobian(pp, got))


def test_msm_length_mismatch_is_generic_error():  # dmsm/mod.rs:73 `G::msm(..)?` -> MpcNetError::Generic
    pp = ctx("bn254", 2)
    with pytest.raises(zk.ZkError) as e:
        msm(pp, ZK_G1, pp.alloc_fr(4), pp.alloc_fr(1), 2, len_scalars=1)
    assert e.value.code == 1 and e.value.msg == "1"


def test_msm_window_sizes():
    """Exercise several window widths (context option "msm_c"), including multi-workgroup bucket reduction (c = 14 -> 8192
    buckets); the plan reports the forced width."""
    from zksaas_amd.api import msm_plan
    c = CURVES["bn254"]
    pp = ctx("bn254", 2)
    G = g1(c)
    pts = _points(G, c, 50, 65)
    sc = rand_vec(66, 50, c.r)
    want = G.msm(pts, sc)
    bases = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts))
    try:
        for cc in (2, 5, 9, 13, 14):
            pp.set_option("msm_c", cc)
            assert G.eq(dec_jacobian(pp, msm(pp, ZK_G1, bases, up(pp, sc), 50)), want)
            assert msm_plan(pp, ZK_G1, 50)["window_bits"] <= cc
    finally:
        pp.set_option("msm_c", 0)


def test_context_options_are_validated():
    """zk_ctx_set_option: every tunable the library keeps is a named, range-checked context option (nothing is read from the
    environment); an unknown name or a value out of range is BAD_INPUT."""
    pp = zk.PackedSharingParams("bn254", 2)
    try:
        for name, good, bad in (("msm_c", 12, 21), ("msm_c_g2", 10, 1), ("msm_table_c", 16, 5), ("msm_table_c_g2", 0, 30),
                                ("msm_bigsort_min", 1 << 14, -1), ("h_first_log_m", 20, 0), ("host_threads", 8, 2),
                                ("king_alltoall", 1, None), ("dist_deadline", 1, None), ("rng_replay", 0, None)):
            pp.set_option(name, good)
            if bad is not None:
                with pytest.raises(zk.ZkError) as e:
                    pp.set_option(name, bad)
                assert e.value.code == 4, name
        with pytest.raises(zk.ZkError) as e:
            pp.set_option("no_such_option", 1)
        assert e.value.code == 4
    finally:
        pp.close()


@pytest.mark.parametrize("group", [ZK_G1, ZK_G2])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_d_msm_matches_oracle(group, masked, curve):  # dmsm_test.rs:13-93
    c = CURVES[curve]
    m, l = 16, 2
    pp, o = ctx(curve, l), opp(curve, l)
    is2 = group == ZK_G2
    G = g2(c) if is2 else g1(c)
    ops = GroupOps(G)
    gen = G.from_affine(G.gen)
    y_pub = rand_vec(70, m, c.r)
    x_pub = [G.mul(gen, rand_fp(71, i, c.r)) for i in range(m)]
    want = G.msm(G.batch_to_affine(x_pub), y_pub)
    x_sh = od.transpose([o.det_pack(x_pub[j:j + l], ops) for j in range(0, m, l)])
    y_sh = od.transpose(od.pack_vec(y_pub, o, 72))
    bases = zk.DeviceBuffer.from_numpy(pp, np.concatenate([enc_affine(pp, G.batch_to_affine(v), is2) for v in x_sh]))
    scal = up_parties(pp, y_sh)
    mask = zk.MsmMask.zero()
    omasks = [od.MsmMask.zero(G)] * pp.n
    if masked:
        omasks = od.MsmMask.sample(o, G, ops, 73)
        mask = zk.MsmMask(np.stack([enc_jacobian(pp, mk.in_mask, is2) for mk in omasks]),
                          np.stack([enc_jacobian(pp, mk.out_mask, is2) for mk in omasks]))
    out = zk.d_msm(pp, group, bases, scal, m // l, mask)
    got = [dec_jacobian(pp, out[i], is2) for i in range(pp.n)]
    ref = od.d_msm([G.batch_to_affine(v) for v in x_sh], y_sh, omasks, o, G, ops)
    assert all(G.eq(a, b) for a, b in zip(got, ref))
    assert G.eq(o.unpack2(got, ops)[0], want)        # dmsm_test.rs:50-51


@pytest.mark.parametrize("curve,l", [("bls12_377", 2), ("bls12_377", 4), ("bls12_381", 2), ("bn254", 2)])
def test_deg_red_matches_oracle(curve, l):  # deg_red.rs:142-191 (BLS12-377 as the reference's tests; BLS12-381 = config 5)
    pp, o = ctx(curve, l), opp(curve, l)
    nch = 21
    secrets = rand_vec(80, nch * l, o.p)
    shares = od.transpose(od.pack_vec(secrets, o, 81))
    mul = [[x * x % o.p for x in v] for v in shares]
    masks = od.DegRedMask.sample(o, 1, nch, 82)
    want = od.deg_red(mul, masks, o, seed=83)
    buf = up_parties(pp, mul)
    zk.deg_red(pp, buf, zk.DegRedMask(up_parties(pp, [m.in_mask for m in masks]),
                                      up_parties(pp, [m.out_mask for m in masks])), nch, seed=83)
    assert down_parties(pp, buf, pp.n, nch) == want
    assert pp.download_fr(pp.unpack(buf, nch)) == [x * x % o.p for x in secrets]
    # device-side mask sampling equals the oracle's
    dm = zk.DegRedMask.sample(pp, nch, 82)
    assert down_parties(pp, dm.in_mask, pp.n, nch) == [m.in_mask for m in masks]
    assert down_parties(pp, dm.out_mask, pp.n, nch) == [m.out_mask for m in masks]


@pytest.mark.parametrize("curve,m", [("bls12_377", 32), ("bls12_377", 4096 + 64), ("bls12_381", 32),
                                     ("bls12_381", 4096 + 64), ("bn254", 32)])
def test_d_pp_matches_oracle(curve, m):  # dpp_test.rs:16-91 and a multi-block scan; BLS12-381 = BASELINE config 5
    l = 2
    pp, o = ctx(curve, l), opp(curve, l)
    m -= m % l
    num, den = rand_vec(90, m, o.p), rand_vec(91, m, o.p)
    ns, ds = od.transpose(od.pack_vec(num, o, 92)), od.transpose(od.pack_vec(den, o, 93))
    masks = od.DegRedMask.sample(o, 1, m // l, 94)
    want = od.d_pp(ns, ds, masks, o, seed=95)
    out = zk.d_pp(pp, up_parties(pp, ns), up_parties(pp, ds),
                  zk.DegRedMask(up_parties(pp, [k.in_mask for k in masks]), up_parties(pp, [k.out_mask for k in masks])),
                  m // l, seed=95)
    assert down_parties(pp, out, pp.n, m // l) == want
    # x = 1..m with num = den reconstructs to all ones (dpp_test.rs:51,62-65)
    x = list(range(1, m + 1))
    xs = up_parties(pp, od.transpose(od.pack_vec(x, o, 96)))
    out = zk.d_pp(pp, xs, xs, zk.DegRedMask.zero(), m // l, seed=97)
    assert pp.download_fr(pp.unpack(out, m // l)) == [1] * m


def test_d_pp_zero_denominator_is_error():  # dpp/mod.rs:55 inverse().unwrap()
    curve, l, m = "bls12_377", 2, 8
    pp, o = ctx(curve, l), opp(curve, l)
    num = rand_vec(98, m, o.p)
    den = list(num)
    den[5] = 0
    with pytest.raises(zk.ZkError) as e:
        zk.d_pp(pp, up_parties(pp, od.transpose(od.pack_vec(num, o, 1))),
                up_parties(pp, od.transpose(od.pack_vec(den, o, 2))), zk.DegRedMask.zero(), m // l)
    assert e.value.code == 1


def test_deg_red_and_d_msm_with_dropout():
    """ser_net.rs:57-94: the king got only n-1 contributions; reconstruction goes through lagrange_unpack."""
    from zksaas_amd.api import d_msm_parties, deg_red_parties
    curve = "bn254"
    c = CURVES[curve]
    l = 2
    pp, o = ctx(curve, l), opp(curve, l)
    parties = [0, 1, 2, 3, 5, 6, 7]
    # deg_red on squared shares
    nch = 13
    secrets = rand_vec(110, nch * l, o.p)
    shares = od.transpose(od.pack_vec(secrets, o, 111))
    mul = [[x * x % o.p for x in v] for v in shares]
    want = od.deg_red(mul, [od.DegRedMask.zero(nch)] * o.n, o, seed=112, parties=parties)
    out = deg_red_parties(pp, up_parties(pp, [mul[i] for i in parties]), parties, zk.DegRedMask.zero(), nch, seed=112)
    assert down_parties(pp, out, pp.n, nch) == want
    assert pp.download_fr(pp.unpack(out, nch)) == [x * x % o.p for x in secrets]
    # d_msm
    m = 16
    G = g1(c)
    ops = GroupOps(G)
    gen = G.from_affine(G.gen)
    y_pub = rand_vec(113, m, c.r)
    x_pub = [G.mul(gen, rand_fp(114, i, c.r)) for i in range(m)]
    x_sh = od.transpose([o.det_pack(x_pub[j:j + l], ops) for j in range(0, m, l)])
    y_sh = od.transpose(od.pack_vec(y_pub, o, 115))
    bases = zk.DeviceBuffer.from_numpy(pp, np.concatenate([enc_affine(pp, G.batch_to_affine(x_sh[i])) for i in parties]))
    scal = up_parties(pp, [y_sh[i] for i in parties])
    out = d_msm_parties(pp, ZK_G1, bases, scal, m // l, parties)
    want = G.msm(G.batch_to_affine(x_pub), y_pub)
    ref = od.d_msm([G.batch_to_affine(v) for v in x_sh], y_sh, [od.MsmMask.zero(G)] * o.n, o, G, ops, parties=parties)
    for i in range(pp.n):
        assert G.eq(dec_jacobian(pp, out[i]), ref[i]) and G.eq(dec_jacobian(pp, out[i]), want)


@pytest.mark.parametrize("curve,group", [("bn254", ZK_G1), ("bn254", ZK_G2), ("bls12_381", ZK_G1)])
def test_msm_both_sort_paths_agree_and_match_oracle(curve, group):
    """The two-level LDS counting sort (default from 2^16 points) and the global-atomics sort are forced in turn on the
    same inputs through zk_ctx_set_option: random scalars (exact against the oracle), ragged length (not a multiple
    of the 4096-point tile), zeros, r-1, all ones (one heavy bucket per window) and a duplicated base."""
    c = CURVES[curve]
    pp = zk.PackedSharingParams(curve, 2)          # own context: the option must not leak into other tests
    is2 = group == ZK_G2
    G = g2(c) if is2 else g1(c)
    n = 4096 + 777
    distinct = _points(G, c, 48, 64)
    pts = [distinct[i % 48] for i in range(n)]
    ident = {5} | {i for i in range(n) if i % 11 == 3}     # identity bases: one early, then every eleventh (a CRS is full of them)
    for i in ident:
        pts[i] = None
    bases = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts, is2))
    cases = {
        "random": rand_vec(65, n, c.r),
        "edge": [0, 1, c.r - 1, 2, c.r - 2] * (n // 5) + [1] * (n % 5),
        "ones": [1] * n,
    }
    for name, sc in cases.items():
        sc_d = up(pp, sc)
        outs = []
        # both sorts, and the identity bases found by the histogram kernel itself (default) or by round 5's mask kernel
        for big_min, skip_kernel in ((0, 0), (1 << 40, 0), (0, 1), (1 << 40, 1)):
            pp.set_option("msm_bigsort_min", big_min)
            pp.set_option("msm_skip_kernel", skip_kernel)
            outs.append(dec_jacobian(pp, msm(pp, group, bases, sc_d, n), is2))
        assert all(G.eq(outs[0], o) for o in outs[1:]), name
        agg = [0] * 48                                   # sum_i s_i * P_(i mod 48), identity bases skipped
        for i, s in enumerate(sc):
            if i not in ident:
                agg[i % 48] = (agg[i % 48] + s) % c.r
        assert G.eq(outs[0], G.msm(distinct, agg)), name
    with pytest.raises(zk.ZkError):
        pp.set_option("no_such_option", 1)


def test_d_msm_big_sort_path_with_party_coefficients():
    """The fused d_msm (scalars pre-multiplied by the per-party unpack2 coefficients inside the sort kernels) through
    the two-level sort, against the global-atomics path."""
    pp = zk.PackedSharingParams("bn254", 2)
    c = CURVES["bn254"]
    G = g1(c)
    ln = 2500
    distinct = _points(G, c, 32, 66)
    bases = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, [distinct[i % 32] for i in range(pp.n * ln)]))
    sc = up(pp, rand_vec(67, pp.n * ln, c.r))
    res = []
    for big_min in (0, 1 << 40):
        pp.set_option("msm_bigsort_min", big_min)
        res.append(zk.d_msm(pp, ZK_G1, bases, sc, ln))
    for p in range(pp.n):
        assert G.eq(dec_jacobian(pp, res[0][p]), dec_jacobian(pp, res[1][p]))


@pytest.mark.parametrize("curve,group", [("bn254", ZK_G1), ("bn254", ZK_G2), ("bls12_381", ZK_G1),
                                         ("bls12_381", ZK_G2)])
def test_fixed_base_table_gives_the_same_msm(curve, group):
    """zk_msm_precompute: MSMs over a registered base vector (whole vector, a sub-range starting inside it, the fused
    d_msm over all parties) equal the table-free results and the oracle; zk_msm_forget restores the plain path."""
    from zksaas_amd import api
    c = CURVES[curve]
    pp = zk.PackedSharingParams(curve, 2)
    is2 = group == ZK_G2
    G = g2(c) if is2 else g1(c)
    n = 1500
    distinct = _points(G, c, 40, 68)
    pts = [distinct[i % 40] for i in range(n)]
    pts[7] = None
    rows = enc_affine(pp, pts, is2)
    bases = zk.DeviceBuffer.from_numpy(pp, rows)
    sc = rand_vec(69, n, c.r)
    sc[3], sc[4], sc[11] = 0, c.r - 1, 1
    sc_d = up(pp, sc)
    plain = dec_jacobian(pp, msm(pp, group, bases, sc_d, n), is2)
    assert api.msm_table_info(pp, group, bases)["windows"] == 0
    api.msm_precompute(pp, group, bases, n)
    # default window of a short vector: 15 bits in both groups (half the buckets of 16 bits in the reduction a chain ends with)
    bits = 15
    assert api.msm_table_info(pp, group, bases) == {"window_bits": bits, "windows": -(-(c.r.bit_length() + 1) // bits)}
    agg = [0] * 40
    for i, s in enumerate(sc):
        if i != 7:
            agg[i % 40] = (agg[i % 40] + s) % c.r
    want = G.msm(distinct, agg)
    with_table = dec_jacobian(pp, msm(pp, group, bases, sc_d, n), is2)
    assert G.eq(with_table, plain) and G.eq(with_table, want)
    # a sub-range that starts inside the registered vector (the prover's party halves do this)
    off, cnt = 520, 700
    sub = bases.view(off * rows.shape[1] * 8)
    sub_sc = up(pp, sc[off:off + cnt])
    got = dec_jacobian(pp, msm(pp, group, sub, sub_sc, cnt), is2)
    agg = [0] * 40
    for i in range(off, off + cnt):
        if i != 7:
            agg[i % 40] = (agg[i % 40] + sc[i]) % c.r
    assert G.eq(got, G.msm(distinct, agg))
    # fused d_msm over all parties: [n_parties][len] is one registered vector
    ln = 150
    d_with = zk.d_msm(pp, group, bases, sc_d, ln)
    api.msm_forget(pp, bases)
    assert api.msm_table_info(pp, group, bases)["windows"] == 0
    d_plain = zk.d_msm(pp, group, bases, sc_d, ln)
    for p in range(pp.n):
        assert G.eq(dec_jacobian(pp, d_with[p], is2), dec_jacobian(pp, d_plain[p], is2))
    with pytest.raises(zk.ZkError):
        api.msm_forget(pp, bases)


def test_fixed_base_table_is_dropped_with_its_vector_and_errors():
    """Tables are found by address: freeing the base vector through zk_free must drop its table (a later allocation at
    the same address would otherwise be multiplied through a stale table); argument errors of the table API."""
    from zksaas_amd import api
    c = CURVES["bn254"]
    pp = zk.PackedSharingParams("bn254", 2)
    G = g1(c)
    n = 300
    pts_a, pts_b = _points(G, c, n, 70), _points(G, c, n, 71)
    sc = rand_vec(72, n, c.r)
    sc_d = up(pp, sc)
    a = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts_a))
    api.msm_precompute(pp, ZK_G1, a, n)
    addr = a.ptr
    a.free()
    b = zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, pts_b))        # very likely the same address again
    if b.ptr == addr:
        assert api.msm_table_info(pp, ZK_G1, b)["windows"] == 0
    assert G.eq(dec_jacobian(pp, msm(pp, ZK_G1, b, sc_d, n)), G.msm(pts_b, sc))
    with pytest.raises(zk.ZkError):
        api.msm_precompute(pp, ZK_G1, None, n)
    with pytest.raises(zk.ZkError):
        api.msm_precompute(pp, 7, b, n)
    with pytest.raises(zk.ZkError):
        pp.set_option("msm_table_c", 40)
    pp.set_option("msm_table_c", 12)                                  # another window width: 22 digit windows
    api.msm_precompute(pp, ZK_G1, b, n)
    assert api.msm_table_info(pp, ZK_G1, b) == {"window_bits": 12, "windows": 22}
    assert G.eq(dec_jacobian(pp, msm(pp, ZK_G1, b, sc_d, n)), G.msm(pts_b, sc))


def test_table_registry_is_process_wide_across_contexts():
    """a table built through context B for a buffer owned by context A must die with the buffer when A frees it
    (a reallocation at the same address must not be multiplied through the stale table), must be visible to A's MSMs
    while it lives, must survive B's sibling contexts, and must die with the context that built it."""
    from zksaas_amd import api
    c = CURVES["bn254"]
    A = zk.PackedSharingParams("bn254", 2)
    B = zk.PackedSharingParams("bn254", 2)
    G = g1(c)
    n = 300
    pts_a, pts_b = _points(G, c, n, 80), _points(G, c, n, 81)
    sc = rand_vec(82, n, c.r)
    sc_d = up(A, sc)
    buf = zk.DeviceBuffer.from_numpy(A, enc_affine(A, pts_a))             # owned by A
    api.msm_precompute(B, ZK_G1, buf, n)                                   # table built through B
    assert api.msm_table_info(A, ZK_G1, buf)["windows"] == 17             # ... is found by A's MSMs
    assert G.eq(dec_jacobian(A, msm(A, ZK_G1, buf, sc_d, n)), G.msm(pts_a, sc))
    addr = buf.ptr
    buf.free()                                                             # freed through A
    again = zk.DeviceBuffer.from_numpy(A, enc_affine(A, pts_b))
    if again.ptr == addr:
        assert api.msm_table_info(B, ZK_G1, again)["windows"] == 0
    assert G.eq(dec_jacobian(B, msm(B, ZK_G1, again, up(B, sc), n)), G.msm(pts_b, sc))
    # a table dies with the context that built it
    api.msm_precompute(B, ZK_G1, again, n)
    assert api.msm_table_info(A, ZK_G1, again)["windows"] == 17
    B.close()
    assert api.msm_table_info(A, ZK_G1, again)["windows"] == 0
    assert G.eq(dec_jacobian(A, msm(A, ZK_G1, again, sc_d, n)), G.msm(pts_b, sc))
    # and freeing with no context at all (the owner is gone) still releases the memory and any table over it
    api.msm_precompute(A, ZK_G1, again, n)
    again.ctx = B                                                          # B.h is None now: zk_free(NULL, p)
    again.free()
    A.close()


def test_concurrent_sorts_stress():
    """msm_hist_kernel's last-workgroup scan of the bin totals rests on relaxed device-scope atomics + a barrier + a relaxed
    ticket (csrc/msm.hpp; ADVICE r4 asked for acq_rel, which costs every concurrent sort an L2 write-back: measured and
    kept behind -DZK_HIST_TICKET_ACQ_REL).  Stress: four host threads, each with its own context (own streams and
    workspaces), run 160 MSMs of assorted two-level-sort sizes AT THE SAME TIME on the one GPU; every result must equal the
    one the same context computed alone beforehand (a stale bin total would misplace sorted entries and change the sum)."""
    import threading
    from zksaas_amd import groth16 as zg
    sizes = [20000, 33000, 70001, 150000]
    rng = np.random.default_rng(9)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 58) - 1)
        return a
    ctxs = [zk.PackedSharingParams("bn254", 2) for _ in range(4)]
    G = g1(CURVES["bn254"])
    try:
        work = []
        for pp in ctxs:
            mx = max(sizes)
            pts = zg.base_points(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, rand(mx)), mx)
            sc = zk.DeviceBuffer.from_numpy(pp, rand(mx))
            want = {n_: msm(pp, ZK_G1, pts, sc, n_).copy() for n_ in sizes}
            work.append((pp, pts, sc, want))
        bad = []

        def run(t):
            pp, pts, sc, want = work[t]
            for i in range(40):
                n_ = sizes[(i + t) % len(sizes)]
                got = msm(pp, ZK_G1, pts, sc, n_)
                # compared as group elements (the order of a bucket's entries, hence the Jacobian coordinates of the sum,
                # may differ between runs)
                if not G.eq(dec_jacobian(pp, got), dec_jacobian(pp, want[n_])):
                    bad.append((t, i, n_))
        ths = [threading.Thread(target=run, args=(t,)) for t in range(4)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert not bad, bad[:5]
    finally:
        for pp in ctxs:
            pp.close()
