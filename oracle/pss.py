"""Packed Shamir secret sharing: restatement of secret-sharing/src/pss.rs and utils.rs.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).
"""

from .field import Domain, FieldOps, inv_mod


def poly_eval(p, x, mod):
    """secret-sharing/src/utils.rs:7-15 (Horner)."""
    acc = 0
    for c in reversed(p):
        acc = (acc * x + c) % mod
    return acc


def get_zero_roots(xs, mod):
    """secret-sharing/src/utils.rs:120-136: coefficients of prod (x - xs[i])."""
    result = [0] * (len(xs) + 1)
    n = len(result) - 1
    result[n] = 1
    for i in range(len(xs)):
        n -= 1
        result[n] = 0
        for j in range(n, len(xs)):
            result[j] = (result[j] - result[j + 1] * xs[i]) % mod
    return result


def syn_div(p, a, b, mod):
    """secret-sharing/src/utils.rs:36-74 synthetic division by (x^a - b)."""
    p = list(p)
    assert a != 0 and b % mod != 0 and len(p) > a
    if a == 1:
        c = 0
        for i in reversed(range(len(p))):
            p[i] = (p[i] + b * c) % mod
            p[i], c = c, p[i]
    else:
        deg_off = len(p) - a
        for i in reversed(range(deg_off)):
            p[i] = (p[i] + p[i + a] * b) % mod
        p = p[a:] + [0] * a
    return p


def lagrange_interpolate(xs, ys, ops, mod):
    """secret-sharing/src/utils.rs:78-116 (generic over DomainCoeff ys)."""
    roots = get_zero_roots(xs, mod)
    numerators = [syn_div(roots, 1, x, mod) for x in xs]
    denominators = [inv_mod(poly_eval(f, x, mod), mod) for f, x in zip(numerators, xs)]
    result = [ops.zero] * len(numerators)
    for i in range(len(ys)):
        y_slice = ops.mul(ys[i], denominators[i])
        for j in range(len(result)):
            result[j] = ops.add(result[j], ops.mul(y_slice, numerators[i][j]))
    # remove leading zeros (utils.rs:104-112)
    pos = len(result)
    for i in reversed(range(len(result))):
        if not ops.eq(result[i], ops.zero):
            pos = i + 1
            break
    return result[:pos]


class PackedSharingParams:
    """secret-sharing/src/pss.rs:19-66: n = 4l parties, threshold t = l."""

    def __init__(self, curve, l):
        self.curve = curve
        self.p = curve.r
        self.l = l
        self.t = l
        self.n = 4 * l
        self.share = Domain(curve, self.n)
        self.secret = Domain(curve, self.l + self.t).get_coset(curve.r_gen)
        self.secret2 = Domain(curve, 2 * (self.l + self.t)).get_coset(curve.r_gen)
        assert self.share.size == self.n
        assert self.secret.size == self.l + self.t
        assert self.secret2.size == 2 * (self.l + self.t)
        self.fops = FieldOps(self.p)

    def det_pack(self, secrets, ops=None):
        """pss.rs:69-87."""
        ops = ops or self.fops
        result = list(secrets)
        # result.resize(self.t, zero): truncates/pads to t (== l) entries (pss.rs:78)
        result = result[: self.t] + [ops.zero] * max(0, self.t - len(result))
        result = self.secret.ifft(result, ops)
        return self.share.fft(result, ops)

    def pack(self, secrets, rand_points, ops=None):
        """pss.rs:90-122; ``rand_points`` are the t values the reference draws from rng."""
        ops = ops or self.fops
        assert len(rand_points) == self.t
        result = list(secrets) + list(rand_points)
        result = self.secret.ifft(result, ops)   # resize => truncate to l+t (sha256.rs:203 relies on it)
        return self.share.fft(result, ops)

    def unpack(self, shares, ops=None):
        """pss.rs:125-138."""
        ops = ops or self.fops
        result = self.share.ifft(shares, ops)
        result = self.secret.fft(result, ops)    # truncates to l+t coefficients
        return result[: self.l]

    def unpack2(self, shares, ops=None):
        """pss.rs:141-166."""
        ops = ops or self.fops
        result = self.share.ifft(shares, ops)
        result = self.secret2.fft(result, ops)
        return result[0 : 2 * self.l : 2]

    def lagrange_unpack(self, shares, parties, ops=None):
        """pss.rs:170-205."""
        ops = ops or self.fops
        assert len(shares) == len(parties)
        assert len(parties) > 2 * (self.t + self.l - 1), "Not enough shares to reconstruct"
        elems = self.share.elements()
        xs = [elems[i] for i in parties]
        result = lagrange_interpolate(xs, shares, ops, self.p)
        result = self.secret2.fft(result, ops)
        return result[0 : 2 * self.l : 2]

    def unpack_missing_shares(self, shares, parties, ops=None):
        """pss.rs:210-221."""
        assert len(shares) == len(parties)
        if len(shares) == self.n:
            return self.unpack2(shares, ops)
        return self.lagrange_unpack(shares, parties, ops)
