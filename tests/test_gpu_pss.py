"""GPU parity (through the C ABI) for the PSS layer: secret-sharing/src/pss.rs, utils/pack.rs, dfft bitrev."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.dist import pack_vec, stride_pack, transpose
from oracle.field import bitrev_permute
from oracle.prng import rand_vec

from gpu_util import ctx, opp, up, up_parties, down_parties


@pytest.mark.parametrize("curve", ["bn254", "bls12_377", "bls12_381"])
@pytest.mark.parametrize("l", [2, 4])
@pytest.mark.parametrize("order", [0, 1])
def test_pack_matches_oracle_bit_for_bit(curve, l, order):
    pp, o = ctx(curve, l), opp(curve, l)
    nch = 37
    secrets = rand_vec(11, nch * l, o.p)
    got = down_parties(pp, pp.pack(up(pp, secrets), nch, seed=77, order=order), pp.n, nch)
    want = transpose(pack_vec(secrets, o, 77) if order == 0 else stride_pack(secrets, o, 77))
    assert got == want


@pytest.mark.parametrize("l", [1, 2, 4, 8])
def test_det_pack_unpack_roundtrip(l):
    pp, o = ctx("bls12_377", l), opp("bls12_377", l)
    nch = 19
    secrets = rand_vec(12, nch * l, o.p)
    sh = pp.det_pack(up(pp, secrets), nch)
    got = down_parties(pp, sh, pp.n, nch)
    want = transpose([o.det_pack(secrets[j * l:(j + 1) * l]) for j in range(nch)])
    assert got == want
    assert pp.download_fr(pp.unpack(sh, nch)) == secrets
    assert pp.download_fr(pp.unpack2(sh, nch)) == secrets


@pytest.mark.parametrize("l", [2, 4])
def test_unpack2_of_products_and_dropout(l):  # pss.rs:288-310
    pp, o = ctx("bls12_377", l), opp("bls12_377", l)
    nch = 9
    secrets = rand_vec(13, nch * l, o.p)
    shares = transpose(pack_vec(secrets, o, 5))
    mul = [[x * x % o.p for x in v] for v in shares]
    expected = [x * x % o.p for x in secrets]
    assert pp.download_fr(pp.unpack2(up_parties(pp, mul), nch)) == expected
    for drop in (0, 3, pp.n - 1):
        parties = [i for i in range(pp.n) if i != drop]
        buf = up_parties(pp, [mul[i] for i in parties])
        assert pp.download_fr(pp.lagrange_unpack(buf, nch, parties)) == expected
    # `unpack` truncates to l+t coefficients (pss.rs:131-135): compare with the oracle on degree-2(l+t-1) shares
    want = [v for j in range(nch) for v in o.unpack([mul[i][j] for i in range(pp.n)])]
    assert pp.download_fr(pp.unpack(up_parties(pp, mul), nch)) == want


def test_not_enough_shares_is_protocol_error():
    import zksaas_amd as zk
    pp = ctx("bls12_377", 2)
    buf = pp.alloc_fr(6 * 4)
    with pytest.raises(zk.ZkError) as e:
        pp.lagrange_unpack(buf, 4, [0, 1, 2, 3, 4, 5])     # needs > 2(t+l-1) = 6 shares
    assert e.value.code == 2


def test_empty_inputs():
    pp = ctx("bn254", 2)
    assert pp.download_fr(pp.pack(pp.alloc_fr(0), 0, seed=1)) == []


@pytest.mark.parametrize("logn", [0, 1, 5, 12])
def test_bitrev(logn):
    pp = ctx("bn254", 2)
    x = list(range(1, (1 << logn) + 1))
    buf = up(pp, x)
    pp._check(pp.lib.zk_bitrev(pp.h, buf.ptr, logn, None))
    want = list(x)
    bitrev_permute(want)
    assert pp.download_fr(buf) == want


def test_production_share_randomness_is_fresh_per_call_and_ignores_the_seed():
    """Outside of replay mode the t random points come from the context's ChaCha20 stream: the same secrets packed
    twice with the SAME seed give different shares (fresh nonces), different contexts differ too (fresh keys), and
    everything still reconstructs; d_fft and deg_red king steps likewise."""
    import zksaas_amd as zk
    from oracle.prng import rand_vec
    from oracle.params import BN254
    pp = zk.PackedSharingParams("bn254", 2)
    pp2 = zk.PackedSharingParams("bn254", 2)
    try:
        for c in (pp, pp2):
            c.set_option("rng_replay", 0)
        nch = 200
        sec = rand_vec(11, 2 * nch, BN254.r)
        a = pp.pack(up(pp, sec), nch, seed=5)
        b = pp.pack(up(pp, sec), nch, seed=5)
        c = pp2.pack(up(pp2, sec), nch, seed=5)
        sa, sb, sc_ = (x.to_numpy() for x in (a, b, c))
        assert not np.array_equal(sa, sb) and not np.array_equal(sa, sc_) and not np.array_equal(sb, sc_)
        # no share row repeats between the two packs (the randomness is fresh everywhere, not just somewhere)
        assert (sa.reshape(-1, 4) == sb.reshape(-1, 4)).all(axis=1).sum() == 0
        for ctx_, x in ((pp, a), (pp, b), (pp2, c)):
            assert ctx_.download_fr(ctx_.unpack(x, nch)) == sec
        # king steps: two d_fft of the same input differ share-wise, agree after reconstruction
        m = 256
        x = rand_vec(12, pp.n * m // 2, BN254.r)
        u, v = up(pp, x), up(pp, x)
        zk.d_fft(pp, u, zk.FftMask.zero(), False, 8, seed=3)
        zk.d_fft(pp, v, zk.FftMask.zero(), False, 8, seed=3)
        assert not np.array_equal(u.to_numpy(), v.to_numpy())
        assert pp.download_fr(pp.unpack(u, m // 2)) == pp.download_fr(pp.unpack(v, m // 2))
        pp.set_option("rng_replay", 1)          # and replay mode is deterministic again
        u, v = up(pp, x), up(pp, x)
        zk.d_fft(pp, u, zk.FftMask.zero(), False, 8, seed=3)
        zk.d_fft(pp, v, zk.FftMask.zero(), False, 8, seed=3)
        assert np.array_equal(u.to_numpy(), v.to_numpy())
    finally:
        pp.close()
        pp2.close()


@pytest.mark.parametrize("curve", ["bn254", "bls12_377", "bls12_381"])
def test_field_arithmetic_edge_values(curve):
    """out = a*b - c (zk_vec_mul_sub), x += y (zk_vec_add) and x *= k (zk_vec_scale) on the values a Montgomery multiplier
    and the carry chains get wrong first: 0, 1, p-1, p-2, 2^k +- 1, all-ones limbs below p, R mod p, plus random ones --
    against Python integers (the device multiplier is the product-scanning form of csrc/field.hpp, the adder / subtracter
    the carry-builtin chains)."""
    import zksaas_amd as zk
    pp, o = ctx(curve, 2), opp(curve, 2)
    p = o.p
    nb = p.bit_length()
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, 1 << 64,
            (1 << 128) - 1, (1 << (nb - 1)) - 1, (1 << (nb - 1)) % p, (1 << 256) % p, ((1 << 256) - 1) % p,
            pow(2, 256, p) * pow(2, 256, p) % p, p - ((1 << 256) % p)]
    edge = [e % p for e in edge]
    rnd = rand_vec(91, 400, p)
    a = [x for x in edge for _ in edge] + rnd
    b = [y for _ in edge for y in edge] + rand_vec(92, 400, p)
    c = [edge[(i * 7) % len(edge)] for i in range(len(edge) ** 2)] + rand_vec(93, 400, p)
    n = len(a)
    out = pp.alloc_fr(n)
    zk.api.vec_mul_sub(pp, out, up(pp, a), up(pp, b), up(pp, c), n)
    assert pp.download_fr(out) == [(x * y - z) % p for x, y, z in zip(a, b, c)]
    xs = up(pp, a)
    pp._check(pp.lib.zk_vec_add(pp.h, xs.ptr, up(pp, b).ptr, n, None))
    assert pp.download_fr(xs) == [(x + y) % p for x, y in zip(a, b)]
    for k in (0, 1, p - 1, (1 << 256) % p, rnd[0]):
        xs = up(pp, a)
        zk.api.vec_scale(pp, xs, k, n)
        assert pp.download_fr(xs) == [x * k % p for x in a]



@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_base_field_primitives_of_the_group_kernels_on_edge_values(curve):
    """zk_fq_selftest: a*b - c*d with ONE Montgomery reduction (Fp::mul_sub_mul, Y3 of every XYZZ formula) and the Fq2
    product of the G2 kernels (three unreduced products, two reductions on 8-limb curves) on the values that break a
    multiplier first -- 0, 1, q-1, q-2, powers of two +- 1, R mod q, R^2 mod q -- in every operand position, plus random
    ones, against Python integers.  (The MSM parity tests exercise these with the coordinates of real points only.)"""
    import zksaas_amd as zk
    from oracle.params import CURVES
    pp = ctx(curve, 2)
    q = CURVES[curve].q
    nb = q.bit_length()
    R = 1 << (32 * pp.fq.nl * 2)
    edge = [0, 1, 2, q - 1, q - 2, (q - 1) // 2, (q + 1) // 2, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, (1 << 128) - 1,
            (1 << (nb - 1)) - 1, (1 << (nb - 1)) % q, R % q, (R - 1) % q, R * R % q, q - (R % q)]
    edge = [e % q for e in edge]
    ne = len(edge)
    a = [edge[i % ne] for i in range(ne ** 2)] + rand_vec(191, 300, q)
    b = [edge[(i // ne) % ne] for i in range(ne ** 2)] + rand_vec(192, 300, q)
    c = [edge[(i * 5 + 1) % ne] for i in range(ne ** 2)] + rand_vec(193, 300, q)
    d = [edge[(i * 11 + 3) % ne] for i in range(ne ** 2)] + rand_vec(194, 300, q)
    n = len(a)

    def upq(vals):
        return zk.DeviceBuffer.from_numpy(pp, pp.fq.encode(vals))
    da, db, dc, dd = upq(a), upq(b), upq(c), upq(d)
    out = zk.DeviceBuffer(pp, 2 * n * pp.fq.nbytes)
    pp._check(pp.lib.zk_fq_selftest(pp.h, 0, da.ptr, db.ptr, dc.ptr, dd.ptr, n, out.ptr, None))
    pp.sync()
    got = pp.fq.decode(out.to_numpy().reshape(-1, pp.fq.nl)[:n])
    assert got == [(w * x - y * z) % q for w, x, y, z in zip(a, b, c, d)]
    # the lazy residues of the G1 accumulate kernel ([0, 2p): field.hpp mul_lazy / sub_lazy / dbl_lazy / mul_sub_mul_lazy /
    # is_zero_lazy), every operand entered both as x and as x + p
    out5 = zk.DeviceBuffer(pp, 5 * n * pp.fq.nbytes)
    for k in range(16):
        pp._check(pp.lib.zk_fq_selftest(pp.h, 2 + k, da.ptr, db.ptr, dc.ptr, dd.ptr, n, out5.ptr, None))
        pp.sync()
        raw = out5.to_numpy().reshape(-1, 5, pp.fq.nl)
        assert not (raw[:, :4] == 0xffffffff).all(axis=2).any(), "a lazy result left [0, 2p)"
        for j, f in enumerate((lambda w, x, y, z: w * x, lambda w, x, y, z: w - x, lambda w, x, y, z: 2 * w,
                               lambda w, x, y, z: w * x - y * z)):
            got = pp.fq.decode(np.ascontiguousarray(raw[:, j]))
            assert got == [f(w, x, y, z) % q for w, x, y, z in zip(a, b, c, d)], (k, j)
        flags = raw[:, 4, 0].tolist()
        assert flags == [(1 if w == y else 0) | (2 if w == 0 else 0) for w, y in zip(a, c)], k
    if curve == "bls12_377":
        with pytest.raises(zk.ZkError):
            pp._check(pp.lib.zk_fq_selftest(pp.h, 1, da.ptr, db.ptr, dc.ptr, dd.ptr, n, out.ptr, None))
        return
    pp._check(pp.lib.zk_fq_selftest(pp.h, 1, da.ptr, db.ptr, dc.ptr, dd.ptr, n, out.ptr, None))
    pp.sync()
    got = pp.fq.decode(out.to_numpy().reshape(-1, pp.fq.nl))
    want = []
    for w, x, y, z in zip(a, b, c, d):                   # (w + x u)(y + z u), u^2 = -1
        want += [(w * y - x * z) % q, (w * z + x * y) % q]
    assert got == want


def test_host_pointer_forms_equal_the_device_forms():
    """zk_d_fft_host / zk_msm_host / zk_d_msm_host (operands and results in HOST memory, as the reference's own signatures)
    give exactly what the device-pointer entry points give."""
    import zksaas_amd as zk
    from zksaas_amd.api import ZK_G1, msm
    from zksaas_amd import groth16 as zg, wire
    pp = ctx("bn254", 2)
    log_m = 12
    m = 1 << log_m
    rng = np.random.default_rng(9)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    sh = rand(pp.n * (m // 2))
    mask = zk.FftMask.sample(pp, False, None, 0, log_m, 31)
    dev = zk.DeviceBuffer.from_numpy(pp, sh)
    zk.d_fft(pp, dev, mask, False, log_m, seed=4)
    host = sh.copy()
    im, om = mask.in_mask.to_numpy().copy(), mask.out_mask.to_numpy().copy()
    pp._check(pp.lib.zk_d_fft_host(pp.h, host.ctypes.data, im.ctypes.data, om.ctypes.data, 0, log_m, 0, None, 4, None))
    assert np.array_equal(host.reshape(-1), dev.to_numpy().reshape(-1))
    # d_ifft with the coset element, zero masks
    g = zg._root_of_unity("bn254", log_m + 1)
    dev = zk.DeviceBuffer.from_numpy(pp, sh)
    zk.d_ifft(pp, dev, zk.FftMask.zero(), True, log_m, g=g, seed=5)
    host = sh.copy()
    garr = pp.fr.encode_one(g)
    pp._check(pp.lib.zk_d_fft_host(pp.h, host.ctypes.data, None, None, 1, log_m, 1, garr.ctypes.data, 5, None))
    assert np.array_equal(host.reshape(-1), dev.to_numpy().reshape(-1))
    # msm / d_msm
    ln = 300
    bases_d = zg.base_points(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, rand(pp.n * ln)), pp.n * ln)
    sc = rand(pp.n * ln)
    bases_h = bases_d.to_numpy().copy()
    want = msm(pp, ZK_G1, bases_d, zk.DeviceBuffer.from_numpy(pp, sc), pp.n * ln)
    got = np.zeros_like(want)
    pp._check(pp.lib.zk_msm_host(pp.h, ZK_G1, bases_h.ctypes.data, pp.n * ln, sc.ctypes.data, pp.n * ln, got.ctypes.data, None))
    # the same group element; its Jacobian coordinates depend on the order buckets were summed in
    assert wire.jacobian_to_affine(pp, got, False) == wire.jacobian_to_affine(pp, want, False)
    want = zk.d_msm(pp, ZK_G1, bases_d, zk.DeviceBuffer.from_numpy(pp, sc), ln)
    got = np.zeros_like(want)
    pp._check(pp.lib.zk_d_msm_host(pp.h, ZK_G1, bases_h.ctypes.data, sc.ctypes.data, ln, None, None, got.ctypes.data, None))
    got, want = got.reshape(pp.n, -1), np.asarray(want).reshape(pp.n, -1)
    for i in range(pp.n):
        assert wire.jacobian_to_affine(pp, got[i], False) == wire.jacobian_to_affine(pp, want[i], False)


def test_host_pointer_forms_of_deg_red_d_pp_circom_h_and_the_prover():
    """zk_deg_red_host / zk_d_pp_host / zk_circom_h_host / zk_groth16_prove_host (deg_red.rs:80, dpp/mod.rs:15,
    ext_wit.rs:104 and sha256.rs:32 take and return host vectors): shares bit for bit what the device-pointer entry points
    give on the replay stream; the proof as group elements."""
    import ctypes as C
    import zksaas_amd as zk
    from zksaas_amd import groth16 as zg, wire
    from test_oracle_groth16 import small_r1cs
    from oracle.params import BN254
    pp = ctx("bn254", 2)
    rng = np.random.default_rng(19)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    ln = 700
    x = rand(pp.n * ln)
    mk = zk.DegRedMask.sample(pp, ln, 41)
    dev = zk.DeviceBuffer.from_numpy(pp, x)
    zk.deg_red(pp, dev, mk, ln, seed=9)
    host = x.copy()
    im, om = mk.in_mask.to_numpy().copy(), mk.out_mask.to_numpy().copy()
    pp._check(pp.lib.zk_deg_red_host(pp.h, host.ctypes.data, im.ctypes.data, om.ctypes.data, ln, 9, None))
    assert np.array_equal(host.reshape(-1), dev.to_numpy().reshape(-1))
    # d_pp: denominators must be shares of non-zero values: pack ones
    num = rand(pp.n * ln)
    ones = pp.pack(pp.upload_fr([1 + (i % 7) for i in range(ln * pp.l)]), ln, 3)
    want = zk.d_pp(pp, zk.DeviceBuffer.from_numpy(pp, num), ones, mk, ln, seed=10)
    got = np.zeros_like(num)
    den_h = ones.to_numpy().copy()
    pp._check(pp.lib.zk_d_pp_host(pp.h, num.ctypes.data, den_h.ctypes.data, im.ctypes.data, om.ctypes.data, ln, 10,
                                  got.ctypes.data, None))
    assert np.array_equal(got.reshape(-1), want.to_numpy().reshape(-1))
    # circom_h and the whole prover on a small circuit, all masks
    r1, w = small_r1cs(37)
    td = [int.from_bytes(rng.bytes(32), "little") % BN254.r for _ in range(5)]
    setup = zg.SetupScalars("bn254", r1, *td)
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bn254", r1, w, seed=5)
    masks = zg.ProofMasks(pp, wit.log_m, seed=70)
    Lc = (1 << wit.log_m) // pp.l
    h_dev = pp.alloc_fr(pp.n * Lc)
    pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, wit.log_m, C.byref(masks.ct), 4,
                                 h_dev.ptr, None))
    keep = []

    def hp(buf):
        a = buf.to_numpy().copy()
        keep.append(a)
        return a.ctypes.data
    mh = zg.Masks()
    C.memmove(C.byref(mh), C.byref(masks.ct), C.sizeof(zg.Masks))
    for i in range(6):
        mh.fft_in[i], mh.fft_out[i] = hp(masks.fft[i].in_mask), hp(masks.fft[i].out_mask)
    mh.degred_in, mh.degred_out = hp(masks.degred.in_mask), hp(masks.degred.out_mask)
    h_host = np.zeros((pp.n * Lc, 4), dtype=np.uint64)
    pp._check(pp.lib.zk_circom_h_host(pp.h, hp(wit.qap[0]), hp(wit.qap[1]), hp(wit.qap[2]), wit.log_m, C.byref(mh), 4,
                                      h_host.ctypes.data, None))
    assert np.array_equal(h_host.reshape(-1), h_dev.to_numpy().reshape(-1))
    r, s = 123456789, 987654321
    ref = zg.prove(pp, crs, wit, r, s, masks=masks, seed=6)
    ch = zg.CrsShare()
    C.memmove(C.byref(ch), C.byref(crs.ct), C.sizeof(zg.CrsShare))
    ch.s_d, ch.h_d, ch.v_d, ch.w_d, ch.u_d = hp(crs.s), hp(crs.h), hp(crs.v), hp(crs.w), hp(crs.u)
    nl = pp.fq.nl
    pa, pb, pc = (np.zeros((pp.n, c_ * nl), dtype=np.uint64) for c_ in (3, 6, 3))
    rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
    pp._check(pp.lib.zk_groth16_prove_host(pp.h, C.byref(ch), hp(wit.qap[0]), hp(wit.qap[1]), hp(wit.qap[2]), hp(wit.a_share),
                                           hp(wit.ax_share), rr.ctypes.data, ss.ctypes.data, wit.log_m, C.byref(mh), 6,
                                           pa.ctypes.data, pb.ctypes.data, pc.ctypes.data, None))
    for got_, ref_, g2 in ((pa, ref[0], False), (pb, ref[1], True), (pc, ref[2], False)):
        for i in range(pp.n):
            assert wire.jacobian_to_affine(pp, got_[i], g2) == wire.jacobian_to_affine(pp, ref_[i], g2)
