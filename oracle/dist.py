"""dist-primitives restated: d_fft / d_ifft, d_msm, deg_red, d_pp and their masks.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).

The reference runs n parties as tokio tasks that meet at two collectives per
primitive (mpc-net/src/lib.rs:89-176).  Here the n parties are lists indexed by
party id and each primitive is "local step for every party -> king step ->
local step for every party", which is the same data flow.  ``parties`` lets a
caller drop parties from the gather (mpc-net/src/ser_net.rs:57-94 ->
``unpack_missing_shares``).

Share randomness: the reference uses ark_std::test_rng()/thread_rng(); this
repo uses oracle/prng.py (see its header).  ``seed`` arguments select the
stream; chunk ``j`` uses indices ``j*t .. j*t+t-1``.
"""

from .field import Domain, FieldOps, bitrev_permute, log2_ceil
from .prng import rand_fp


def transpose(matrix):
    """dist-primitives/src/utils/pack.rs:22-35."""
    assert len(matrix) > 0
    return [list(col) for col in zip(*matrix)]


def _rand_points(pp, seed, j, ops=None, gen=None):
    vals = [rand_fp(seed, j * pp.t + i, pp.p) for i in range(pp.t)]
    if gen is not None:          # T::rand for group elements: random multiple of the generator
        return [ops.mul(gen, v) for v in vals]
    return vals


def pack_vec(secrets, pp, seed, ops=None, gen=None):
    """dist-primitives/src/utils/pack.rs:8-20: chunk j = secrets[j*l:(j+1)*l]."""
    assert len(secrets) % pp.l == 0
    return [
        pp.pack(secrets[j * pp.l:(j + 1) * pp.l], _rand_points(pp, seed, j, ops, gen), ops)
        for j in range(len(secrets) // pp.l)
    ]


def stride_pack(x, pp, seed):
    """Chunk j packs (x[j], x[j+L], ...): dfft/mod.rs:286-299, qap.rs:103-112."""
    L = len(x) // pp.l
    return [pp.pack(x[j::L], _rand_points(pp, seed, j)) for j in range(L)]


# ---------------------------------------------------------------------------- d_fft
def fft1_in_place(px, pp, gen):
    """dist-primitives/src/dfft/mod.rs:178-208."""
    p = pp.p
    dom_size = len(px) * pp.l
    for i in range(log2_ceil(dom_size), log2_ceil(pp.l), -1):
        poly_size = dom_size >> i
        factor_stride = pow(gen, 1 << (i - 1), p)
        factor = factor_stride
        for k in range(poly_size):
            for j in range((1 << (i - 1)) // pp.l):
                x = px[(2 * j) * poly_size + k]
                y = px[(2 * j + 1) * poly_size + k] * factor % p
                px[j * (2 * poly_size) + k] = (x + y) % p
                px[j * (2 * poly_size) + k + poly_size] = (x - y) % p
            factor = factor * factor_stride % p
    return px


def fft2_in_place(s1, pp, gen):
    """dist-primitives/src/dfft/mod.rs:210-237 (returns the new vector)."""
    p = pp.p
    dom_size = len(s1)
    s1 = list(s1)
    s2 = [0] * dom_size
    for i in range(log2_ceil(pp.l), 0, -1):
        poly_size = dom_size >> i
        factor_stride = pow(gen, 1 << (i - 1), p)
        factor = factor_stride
        for k in range(poly_size):
            for j in range(1 << (i - 1)):
                x = s1[k * (1 << i) + 2 * j]
                y = s1[k * (1 << i) + 2 * j + 1] * factor % p
                s2[k * (1 << (i - 1)) + j] = (x + y) % p
                s2[(k + poly_size) * (1 << (i - 1)) + j] = (x - y) % p
            factor = factor * factor_stride % p
        s1, s2 = s2, s1
    return s1[-1:] + s1[:-1]          # rotate_right(1)


def king_fft2(rs_shares, parties, rearrange, g, pp, gen, seed):
    """King closure of fft2_with_rearrange, dfft/mod.rs:264-304.

    rs_shares: list (one per surviving party, in ``parties`` order) of vectors of
    length m/l.  Returns n vectors of length m/l (one per party).
    """
    p = pp.p
    all_shares = transpose(rs_shares)
    mbyl = len(all_shares)
    s1 = [0] * (mbyl * pp.l)
    for i, share in enumerate(all_shares):
        tmp = pp.unpack_missing_shares(share, parties)
        for j in range(pp.l):
            s1[i * pp.l + j] = tmp[j]
    s1 = fft2_in_place(s1, pp, gen)
    if g % p != 1:
        s1 = Domain.distribute_powers(s1, g, FieldOps(p), p)
    if rearrange:
        bitrev_permute(s1)
        out_shares = stride_pack(s1, pp, seed)
    else:
        out_shares = pack_vec(s1, pp, seed)
    return transpose(out_shares)


class FftMask:
    """dist-primitives/src/dfft/mod.rs:16-95 (one party's share of the mask)."""

    def __init__(self, in_mask, out_mask):
        self.in_mask = in_mask
        self.out_mask = out_mask

    @staticmethod
    def sample(rearrange, g, gen, m, pp, seed):
        p = pp.p
        mask_values = [rand_fp(seed, i, p) for i in range(m)]
        in_shares = transpose(pack_vec(mask_values, pp, seed ^ 0x1111))
        mv = fft2_in_place(mask_values, pp, gen)
        if g % p != 1:
            mv = Domain.distribute_powers(mv, g, FieldOps(p), p)
        mv = [(-x) % p for x in mv]
        if rearrange:
            bitrev_permute(mv)
            out_shares = transpose(stride_pack(mv, pp, seed ^ 0x2222))
        else:
            out_shares = transpose(pack_vec(mv, pp, seed ^ 0x2222))
        return [FftMask(i, o) for i, o in zip(in_shares, out_shares)]

    @staticmethod
    def zero(mbyl):
        return FftMask([0] * mbyl, [0] * mbyl)


def _fft2_with_rearrange(pxs, masks, rearrange, g, pp, gen, seed, parties):
    """dfft/mod.rs:240-320 for all parties at once."""
    p = pp.p
    n = pp.n
    parties = list(range(n)) if parties is None else list(parties)
    outs = [[(x + mk) % p for x, mk in zip(pxs[i], masks[i].in_mask)] for i in range(n)]
    king_answer = king_fft2([outs[i] for i in parties], parties, rearrange, g, pp, gen, seed)
    return [[(x + mk) % p for x, mk in zip(king_answer[i], masks[i].out_mask)] for i in range(n)]


def d_fft(pcoeff_shares, masks, rearrange, dom, pp, seed=0, parties=None):
    """dfft/mod.rs:99-134; ``pcoeff_shares[i]`` is party i's vector of length m/l."""
    assert all(len(s) * pp.l == dom.size for s in pcoeff_shares)
    pxs = [fft1_in_place(list(s), pp, dom.group_gen) for s in pcoeff_shares]
    return _fft2_with_rearrange(pxs, masks, rearrange, 1, pp, dom.group_gen, seed, parties)


def d_ifft(peval_shares, masks, rearrange, dom, g, pp, seed=0, parties=None):
    """dfft/mod.rs:137-175."""
    assert all(len(s) * pp.l == dom.size for s in peval_shares)
    p = pp.p
    pxs = [[x * dom.size_inv % p for x in s] for s in peval_shares]
    pxs = [fft1_in_place(s, pp, dom.group_gen_inv) for s in pxs]
    return _fft2_with_rearrange(pxs, masks, rearrange, g, pp, dom.group_gen_inv, seed, parties)


# ---------------------------------------------------------------------------- d_msm
class MsmMask:
    """dist-primitives/src/dmsm/mod.rs:10-57."""

    def __init__(self, in_mask, out_mask):
        self.in_mask = in_mask
        self.out_mask = out_mask

    @staticmethod
    def sample(pp, group, gops, seed):
        gen = group.from_affine(group.gen)
        vals = [rand_fp(seed, i, pp.p) for i in range(pp.l)]
        mask_values = [group.mul(gen, v) for v in vals]
        out_value = group.neg(group.sum(mask_values))
        in_shares = pp.pack(mask_values, _rand_points(pp, seed ^ 0x1111, 0, gops, gen), gops)
        out_shares = pp.pack([out_value] * pp.l, _rand_points(pp, seed ^ 0x2222, 0, gops, gen), gops)
        return [MsmMask(i, o) for i, o in zip(in_shares, out_shares)]

    @staticmethod
    def zero(group):
        return MsmMask(group.identity, group.identity)


def d_msm(bases, scalars, masks, pp, group, gops, parties=None):
    """dmsm/mod.rs:59-102 for all parties; bases[i] affine, scalars[i] ints."""
    n = pp.n
    parties = list(range(n)) if parties is None else list(parties)
    c_shares = [group.add(group.msm(bases[i], scalars[i]), masks[i].in_mask) for i in range(n)]
    result = pp.unpack_missing_shares([c_shares[i] for i in parties], parties, gops)
    output = group.sum(result)
    return [group.add(output, masks[i].out_mask) for i in range(n)]


# ---------------------------------------------------------------------------- deg_red
class DegRedMask:
    """dist-primitives/src/utils/deg_red.rs:14-77."""

    def __init__(self, in_mask, out_mask):
        self.in_mask = in_mask
        self.out_mask = out_mask

    @staticmethod
    def sample(pp, gen, num, seed, ops=None):
        ops = ops or FieldOps(pp.p)
        in_vals, out_vals = [], []
        for i in range(num * pp.l):
            mv = ops.mul(gen, rand_fp(seed, i, pp.p))
            in_vals.append(mv)
            out_vals.append(ops.sub(ops.zero, mv))
        is_group = not isinstance(gen, int)
        g = gen if is_group else None
        in_shares = transpose(pack_vec(in_vals, pp, seed ^ 0x1111, ops, g))
        out_shares = transpose(pack_vec(out_vals, pp, seed ^ 0x2222, ops, g))
        return [DegRedMask(i, o) for i, o in zip(in_shares, out_shares)]

    @staticmethod
    def zero(num, zero=0):
        return DegRedMask([zero] * num, [zero] * num)


def king_deg_red(rs_shares, parties, pp, seed, ops=None, gen=None):
    """King closure of deg_red, deg_red.rs:103-111."""
    x_shares = transpose(rs_shares)
    for j in range(len(x_shares)):
        xi = pp.unpack_missing_shares(x_shares[j], parties, ops)
        x_shares[j] = pp.pack(xi, _rand_points(pp, seed, j, ops, gen), ops)
    return transpose(x_shares)


def deg_red(x_shares, masks, pp, seed=0, ops=None, gen=None, parties=None):
    """deg_red.rs:80-126 for all parties."""
    ops = ops or FieldOps(pp.p)
    n = pp.n
    parties = list(range(n)) if parties is None else list(parties)
    for i in range(n):
        assert len(x_shares[i]) == len(masks[i].in_mask) == len(masks[i].out_mask)
    x_mask = [[ops.add(x, mk) for x, mk in zip(x_shares[i], masks[i].in_mask)] for i in range(n)]
    ans = king_deg_red([x_mask[i] for i in parties], parties, pp, seed, ops, gen)
    return [[ops.add(x, mk) for x, mk in zip(ans[i], masks[i].out_mask)] for i in range(n)]


# ---------------------------------------------------------------------------- d_pp
def king_d_pp(rs_shares, parties, pp, seed):
    """King closure of d_pp, dpp/mod.rs:41-76."""
    p = pp.p
    numden_shares = transpose(rs_shares)
    numden = []
    for x in numden_shares:
        numden.extend(pp.unpack_missing_shares(x, parties))
    half = len(numden) // 2
    for i in range(half):
        if numden[i + half] % p == 0:
            raise ZeroDivisionError("d_pp: zero denominator")  # inverse().unwrap() panics, dpp/mod.rs:55
        numden[i] = numden[i] * pow(numden[i + half], p - 2, p) % p
    numden = numden[:half]
    for i in range(1, len(numden)):
        numden[i] = numden[i] * numden[i - 1] % p
    return transpose(pack_vec(numden, pp, seed))


def d_pp(num, den, degred_masks, pp, seed=0, parties=None):
    """dpp/mod.rs:15-87 for all parties (s = 1 as in the reference, :25-26)."""
    n = pp.n
    parties = list(range(n)) if parties is None else list(parties)
    numden = [list(num[i]) + list(den[i]) for i in range(n)]
    ans = king_d_pp([numden[i] for i in parties], parties, pp, seed)
    return deg_red(ans, degred_masks, pp, seed ^ 0x3333, parties=parties)
