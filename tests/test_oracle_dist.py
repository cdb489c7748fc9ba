"""Oracle pinned against the reference's own tests for dist-primitives.

Restates dist-primitives/examples/local_dfft_test.rs, dfft_test.rs, dpp_test.rs,
dmsm_test.rs, src/dfft/tests.rs:20-357, src/dmsm/mod.rs:139-180 and
src/utils/deg_red.rs:142-191.  The reference runs these over BLS12-377.
"""
import pytest

from oracle.params import BLS12_377, BN254
from oracle.curve import g1, GroupOps
from oracle.dist import (FftMask, MsmMask, DegRedMask, d_fft, d_ifft, d_msm, d_pp, deg_red, fft1_in_place,
                         fft2_in_place, pack_vec, stride_pack, transpose)
from oracle.field import Domain, bitrev_permute, log2_ceil
from oracle.prng import rand_fp, rand_vec
from oracle.pss import PackedSharingParams

C = BLS12_377
L = 2
M = 8


def _deal_bitrev(x, pp, seed):
    y = list(x)
    bitrev_permute(y)
    return transpose(stride_pack(y, pp, seed))


def _reconstruct(out, pp):
    return [v for ch in transpose(out) for v in pp.unpack(ch)]


@pytest.mark.parametrize("m,l", [(8, 2), (64, 2), (64, 4), (128, 8)])
def test_local_dfft(m, l):
    """local_dfft_test.rs:10-84: fft1 on the l plaintext lanes + fft2 + rotate == DFT (no network)."""
    pp = PackedSharingParams(C, l)
    dom = Domain(C, m)
    x = list(range(m))
    want = dom.fft(x)
    y = list(x)
    bitrev_permute(y)
    mbyl = m // l
    lanes = [y[ii * mbyl:(ii + 1) * mbyl] for ii in range(l)]   # px[i][ii] = y[i + ii*mbyl]
    lanes = [fft1_in_place(v, pp, dom.group_gen) for v in lanes]
    sx = [lanes[ii][i] for i in range(mbyl) for ii in range(l)]
    assert fft2_in_place(sx, pp, dom.group_gen) == want


@pytest.mark.parametrize("m", [8, 1024])
def test_dfft_example(m):
    """dfft_test.rs:13-94 (x[i] = i, zero masks, m = 1024 in the reference)."""
    pp = PackedSharingParams(C, L)
    dom = Domain(C, m)
    x = list(range(m))
    shares = _deal_bitrev(x, pp, 1)
    out = d_fft(shares, [FftMask.zero(m // L)] * pp.n, False, dom, pp, seed=2)
    assert _reconstruct(out, pp) == dom.fft(x)


def test_d_ifft_works():  # dfft/tests.rs:20-77
    pp = PackedSharingParams(C, L)
    dom = Domain(C, M)
    evals = rand_vec(3, M, C.r)
    shares = _deal_bitrev(evals, pp, 4)
    mask = FftMask.sample(False, 1, dom.group_gen_inv, M, pp, 5)
    out = d_ifft(shares, mask, False, dom, 1, pp, seed=6)
    assert _reconstruct(out, pp) == dom.ifft(evals)


def test_d_fft_works():  # dfft/tests.rs:79-135
    pp = PackedSharingParams(C, L)
    dom = Domain(C, M)
    coeffs = rand_vec(7, M, C.r)
    shares = _deal_bitrev(coeffs, pp, 8)
    mask = FftMask.sample(False, 1, dom.group_gen, M, pp, 9)
    out = d_fft(shares, mask, False, dom, pp, seed=10)
    assert _reconstruct(out, pp) == dom.fft(coeffs)


@pytest.mark.parametrize("m,l", [(8, 2), (256, 2), (64, 4)])
def test_d_ifft_x_d_fft(m, l):  # dfft/tests.rs:142-220
    pp = PackedSharingParams(C, l)
    dom = Domain(C, m)
    evals = rand_vec(11, m, C.r)
    shares = _deal_bitrev(evals, pp, 12)
    im = FftMask.sample(True, 1, dom.group_gen_inv, m, pp, 13)
    fm = FftMask.sample(False, 1, dom.group_gen, m, pp, 14)
    co = d_ifft(shares, im, True, dom, 1, pp, seed=15)
    out = d_fft(co, fm, False, dom, pp, seed=16)
    assert _reconstruct(out, pp) == evals


def test_coset_chain():  # dfft/tests.rs:222-357
    pp = PackedSharingParams(C, L)
    dom = Domain(C, M)
    cos = dom.get_coset(C.r_gen)
    evals = rand_vec(17, M, C.r)
    shares = _deal_bitrev(evals, pp, 18)
    masks = [
        FftMask.sample(True, cos.coset_offset(), dom.group_gen_inv, M, pp, 19),
        FftMask.sample(True, 1, cos.group_gen, M, pp, 20),
        FftMask.sample(True, cos.coset_offset_inv(), dom.group_gen_inv, M, pp, 21),
        FftMask.sample(False, 1, cos.group_gen, M, pp, 22),
    ]
    x = d_ifft(shares, masks[0], True, dom, cos.coset_offset(), pp, seed=23)
    x = d_fft(x, masks[1], True, dom, pp, seed=24)
    # after step 2 the shares hold evaluations over the coset
    x = d_ifft(x, masks[2], True, dom, cos.coset_offset_inv(), pp, seed=25)
    x = d_fft(x, masks[3], False, dom, pp, seed=26)
    assert _reconstruct(x, pp) == evals


def test_dfft_with_dropout():
    """ser_net.rs:57-94 + pss.rs:210-221: king reconstructs through lagrange_unpack when a party is missing."""
    pp = PackedSharingParams(C, L)
    dom = Domain(C, 16)
    coeffs = rand_vec(27, 16, C.r)
    shares = _deal_bitrev(coeffs, pp, 28)
    mask = FftMask.sample(False, 1, dom.group_gen, 16, pp, 29)
    out = d_fft(shares, mask, False, dom, pp, seed=30, parties=[0, 1, 2, 4, 5, 6, 7])
    assert _reconstruct(out, pp) == dom.fft(coeffs)


def test_pack_unpack2_msm():  # dmsm/mod.rs:139-180 (M = 256 there; 32 keeps pure Python quick)
    m = 32
    pp = PackedSharingParams(C, L)
    G = g1(C)
    ops = GroupOps(G)
    gen = G.from_affine(C.g1)
    gsec = [G.mul(gen, rand_fp(31, 0, C.r))] * m          # [G1P::rand(rng); M] repeats one point
    fsec = [1] * m
    aff = G.batch_to_affine(gsec)
    expected = G.msm(aff, fsec)
    gshares = transpose([pp.pack(gsec[j:j + L], [G.mul(gen, rand_fp(32, j * L + i, C.r)) for i in range(L)], ops)
                         for j in range(0, m, L)])
    fshares = transpose(pack_vec(fsec, pp, 33))
    res = [G.msm(G.batch_to_affine(gshares[i]), fshares[i]) for i in range(pp.n)]
    assert G.eq(G.sum(pp.unpack2(res, ops)), expected)


@pytest.mark.parametrize("masked", [False, True])
def test_dmsm_example(masked):  # dmsm_test.rs:13-93 (m = 256 there)
    m = 16
    pp = PackedSharingParams(C, L)
    G = g1(C)
    ops = GroupOps(G)
    gen = G.from_affine(C.g1)
    y_pub = rand_vec(34, m, C.r)
    x_pub = [G.mul(gen, rand_fp(35, i, C.r)) for i in range(m)]
    want = G.msm(G.batch_to_affine(x_pub), y_pub)
    assert G.eq(want, G.msm_naive(G.batch_to_affine(x_pub), y_pub))
    x_sh = transpose([pp.pack(x_pub[j:j + L], [G.mul(gen, rand_fp(36, j + i, C.r)) for i in range(L)], ops)
                      for j in range(0, m, L)])
    y_sh = transpose(pack_vec(y_pub, pp, 37))
    masks = MsmMask.sample(pp, G, ops, 38) if masked else [MsmMask.zero(G)] * pp.n
    out = d_msm([G.batch_to_affine(v) for v in x_sh], y_sh, masks, pp, G, ops)
    assert G.eq(pp.unpack_missing_shares(out, list(range(pp.n)), ops)[0], want)
    # the output is the repeated-secret packing: every slot holds the result
    assert all(G.eq(v, want) for v in pp.unpack2(out, ops))


def test_msm_length_mismatch_is_error():  # dmsm/mod.rs:73 `G::msm(..)?`
    G = g1(C)
    with pytest.raises(ValueError):
        G.msm([C.g1, C.g1], [1])


def test_deg_red_with_dropout():  # deg_red.rs:142-191 (L = 4, last party's result dropped)
    l = 4
    pp = PackedSharingParams(C, l)
    secrets = rand_vec(39, l, C.r)
    expected = [x * x % C.r for x in secrets]
    shares = pp.pack(secrets, rand_vec(40, l, C.r))
    mul = [x * x % C.r for x in shares]
    masks = DegRedMask.sample(pp, 1, 1, 41)
    out = deg_red([[v] for v in mul], masks, pp, seed=42)
    full = [v for ch in transpose(out) for v in pp.unpack(ch)]
    assert full == expected
    parties = list(range(pp.n - 1))                          # simulate_lossy_network_round, multi.rs:359-362
    lossy = [v for ch in transpose(out[:-1]) for v in pp.lagrange_unpack(ch, parties)]
    assert lossy == expected


def test_dpp_example():  # dpp_test.rs:16-91: x = 1..32, num = den => all ones
    m = 32
    pp = PackedSharingParams(C, L)
    x = list(range(1, m + 1))
    px = transpose(pack_vec(x, pp, 43))
    masks = DegRedMask.sample(pp, 1, m // L, 44)
    out = d_pp(px, px, masks, pp, seed=45)
    assert [v for ch in transpose(out) for v in pp.unpack(ch)] == [1] * m


def test_dpp_partial_products():
    m = 16
    p = BN254.r
    pp = PackedSharingParams(BN254, L)
    num, den = rand_vec(46, m, p), rand_vec(47, m, p)
    want, acc = [], 1
    for a, b in zip(num, den):
        acc = acc * a % p * pow(b, p - 2, p) % p
        want.append(acc)
    out = d_pp(transpose(pack_vec(num, pp, 48)), transpose(pack_vec(den, pp, 49)),
               DegRedMask.sample(pp, 1, m // L, 50), pp, seed=51)
    assert [v for ch in transpose(out) for v in pp.unpack(ch)] == want
    with pytest.raises(ZeroDivisionError):
        den0 = list(den)
        den0[3] = 0
        d_pp(transpose(pack_vec(num, pp, 48)), transpose(pack_vec(den0, pp, 49)),
             DegRedMask.sample(pp, 1, m // L, 50), pp, seed=51)
