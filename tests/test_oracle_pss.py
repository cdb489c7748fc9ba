"""Oracle pinned against the reference's own tests for the PSS layer.

Restates secret-sharing/src/pss.rs:238-323 (BLS12-377 Fr, L=2) and
dist-primitives/src/utils/pack.rs:41-48, plus SURVEY.md 8c constants.
"""
import pytest

from oracle.params import CURVES, SURVEY_TWO_ADIC, BLS12_377, BN254, BLS12_381
from oracle.curve import g1, g2, GroupOps
from oracle.dist import transpose
from oracle.prng import rand_vec, rand_fp
from oracle.pss import PackedSharingParams, lagrange_interpolate, poly_eval
from oracle.field import FieldOps, Domain, bitrev_permute

L = 2


@pytest.mark.parametrize("name", sorted(CURVES))
def test_field_constants(name):
    c = CURVES[name]
    assert (c.two_adicity, c.two_adic_root) == SURVEY_TWO_ADIC[name]
    # generator is a quadratic non-residue, root has exact order 2^s
    assert pow(c.r_gen, (c.r - 1) // 2, c.r) == c.r - 1
    assert pow(c.two_adic_root, 1 << (c.two_adicity - 1), c.r) == c.r - 1
    G = g1(c)
    assert G.on_curve(c.g1)
    P = G.from_affine(c.g1)
    assert G.is_identity(G.add(G.mul(P, c.r - 1), P))
    if c.g2 is not None:
        H = g2(c)
        assert H.on_curve(c.g2)
        Q = H.from_affine(c.g2)
        assert H.is_identity(H.add(H.mul(Q, c.r - 1), Q))


def test_initialize():  # pss.rs:238-247
    pp = PackedSharingParams(BLS12_377, L)
    assert (pp.t, pp.l, pp.n) == (L, L, 4 * L)
    assert pp.share.size == 4 * L and pp.secret.size == 2 * L and pp.secret2.size == 4 * L


@pytest.mark.parametrize("curve", [BLS12_377, BN254, BLS12_381])
@pytest.mark.parametrize("l", [2, 4])
def test_packing(curve, l):  # pss.rs:249-270
    pp = PackedSharingParams(curve, l)
    secrets = rand_vec(1, l, curve.r)
    shares = pp.pack(secrets, rand_vec(2, l, curve.r))
    assert pp.unpack(shares) == secrets
    k = max(pp.n - pp.t + 1, 2 * (pp.t + pp.l - 1) + 1)   # pss.rs:183-186 needs > 2(t+l-1) shares
    assert pp.lagrange_unpack(shares[:k], list(range(k))) == secrets


def test_det_packing():  # pss.rs:272-286
    pp = PackedSharingParams(BLS12_377, L)
    secrets = rand_vec(3, L, pp.p)
    assert pp.unpack(pp.det_pack(secrets)) == secrets


def test_multiplication():  # pss.rs:288-310
    pp = PackedSharingParams(BLS12_377, L)
    secrets = rand_vec(4, L, pp.p)
    expected = [x * x % pp.p for x in secrets]
    shares = pp.pack(secrets, rand_vec(5, L, pp.p))
    mul = [x * x % pp.p for x in shares]
    assert pp.unpack2(mul) == expected
    assert pp.lagrange_unpack(mul[: pp.n - 1], list(range(pp.n - 1))) == expected
    # any single dropout, not only the last party
    for drop in range(pp.n):
        parties = [i for i in range(pp.n) if i != drop]
        assert pp.unpack_missing_shares([mul[i] for i in parties], parties) == expected


def test_eval_interpolate():  # pss.rs:312-323
    p = BLS12_377.r
    degree = 32
    poly = rand_vec(6, degree, p)
    xs = list(range(1, 2 * degree + 1))
    ys = [poly_eval(poly, x, p) for x in xs]
    assert lagrange_interpolate(xs, ys, FieldOps(p), p) == poly


def test_transpose():  # pack.rs:41-48
    assert transpose([[1, 2, 3], [4, 5, 6], [7, 8, 9]]) == [[1, 4, 7], [2, 5, 8], [3, 6, 9]]


def test_pack_of_n_copies_is_constant_share():
    """sha256.rs:203-204: pp.pack(vec![r; n]) truncates to l+t copies => every share equals r."""
    pp = PackedSharingParams(BN254, 2)
    r = rand_fp(9, 0, pp.p)
    secrets_and_rand = [r] * pp.n
    shares = pp.share.fft(pp.secret.ifft(secrets_and_rand))
    assert shares == [r] * pp.n


def test_group_pack_unpack():  # dmsm/mod.rs:127-137
    c = BLS12_377
    pp = PackedSharingParams(c, L)
    G = g1(c)
    ops = GroupOps(G)
    gen = G.from_affine(c.g1)
    secrets = [G.mul(gen, rand_fp(7, i, c.r)) for i in range(L)]
    rnd = [G.mul(gen, rand_fp(8, i, c.r)) for i in range(L)]
    shares = pp.pack(secrets, rnd, ops)
    out = pp.unpack(shares, ops)
    assert all(G.eq(a, b) for a, b in zip(out, secrets))


def test_bitrev_matches_definition():  # dfft/mod.rs:322-335
    for logn in range(0, 8):
        n = 1 << logn
        x = list(range(n))
        bitrev_permute(x)
        assert x == [int(format(i, "0%db" % logn)[::-1], 2) if logn else 0 for i in range(n)]


def test_domain_fft_is_dft():
    c = BN254
    d = Domain(c, 8).get_coset(c.r_gen)
    coeffs = rand_vec(10, 8, c.r)
    ev = d.fft(coeffs)
    for i in range(8):
        assert ev[i] == poly_eval(coeffs, d.element(i), c.r)
    assert d.ifft(ev) == coeffs
    # resize semantics: truncation and zero padding
    assert Domain(c, 4).fft(coeffs) == Domain(c, 4).fft(coeffs[:4])
    assert Domain(c, 8).fft(coeffs[:3]) == Domain(c, 8).fft(coeffs[:3] + [0] * 5)
