"""Short-Weierstrass (a = 0) group arithmetic over Fq and Fq2 with Python ints.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).

Restates what the reference reaches through ark-ec ^0.4.2 (third-party, not
vendored): ``CurveGroup`` add / neg / scalar-mul,
``VariableBaseMSM::msm`` (dist-primitives/src/dmsm/mod.rs:73) and the use of
group elements as ``DomainCoeff`` in PSS (secret-sharing/src/pss.rs:69-166).
Group elements are mathematically unique, so the formulas used here (textbook
Jacobian) need not match arkworks' for results to be bit-identical after
normalisation to affine.

Points are handled in Jacobian form ``(X, Y, Z)``; ``Z == zero`` is the identity.
Affine points are ``(x, y)`` or ``None`` for the identity.
"""

from .field import inv_mod


class Fq:
    def __init__(self, q):
        self.q = q
        self.zero = 0
        self.one = 1

    def add(self, a, b):
        return (a + b) % self.q

    def sub(self, a, b):
        return (a - b) % self.q

    def neg(self, a):
        return (-a) % self.q

    def mul(self, a, b):
        return a * b % self.q

    def sqr(self, a):
        return a * a % self.q

    def inv(self, a):
        return inv_mod(a, self.q)

    def is_zero(self, a):
        return a % self.q == 0

    def small(self, k):
        return k % self.q

    def sqrt(self, a):
        q = self.q
        if a % q == 0:
            return 0
        if pow(a, (q - 1) // 2, q) != 1:
            return None
        if q % 4 == 3:
            return pow(a, (q + 1) // 4, q)
        # Tonelli-Shanks
        s, t = 0, q - 1
        while t % 2 == 0:
            t //= 2
            s += 1
        z = 2
        while pow(z, (q - 1) // 2, q) != q - 1:
            z += 1
        c = pow(z, t, q)
        x = pow(a, (t + 1) // 2, q)
        b = pow(a, t, q)
        m = s
        while b != 1:
            i, b2 = 0, b
            while b2 != 1:
                b2 = b2 * b2 % q
                i += 1
            e = pow(c, 1 << (m - i - 1), q)
            x = x * e % q
            c = e * e % q
            b = b * c % q
            m = i
        return x


class Fq2:
    """Fq[u]/(u^2 - nonres); elements are (c0, c1)."""

    def __init__(self, q, nonres):
        self.q = q
        self.nr = nonres % q
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b):
        return ((a[0] + b[0]) % self.q, (a[1] + b[1]) % self.q)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.q, (a[1] - b[1]) % self.q)

    def neg(self, a):
        return ((-a[0]) % self.q, (-a[1]) % self.q)

    def mul(self, a, b):
        q = self.q
        return ((a[0] * b[0] + self.nr * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)

    def sqr(self, a):
        return self.mul(a, a)

    def inv(self, a):
        q = self.q
        n = (a[0] * a[0] - self.nr * a[1] * a[1]) % q
        ni = inv_mod(n, q)
        return (a[0] * ni % q, (-a[1]) * ni % q)

    def is_zero(self, a):
        return a[0] % self.q == 0 and a[1] % self.q == 0

    def small(self, k):
        return (k % self.q, 0)


class Group:
    """y^2 = x^3 + b over field ``F`` (Fq or Fq2)."""

    def __init__(self, F, b, gen, r):
        self.F = F
        self.b = b
        self.gen = gen
        self.r = r
        self.identity = (F.one, F.one, F.zero)

    # ---- conversions ---------------------------------------------------------
    def from_affine(self, p):
        if p is None:
            return self.identity
        return (p[0], p[1], self.F.one)

    def to_affine(self, P):
        F = self.F
        if F.is_zero(P[2]):
            return None
        zi = F.inv(P[2])
        zi2 = F.sqr(zi)
        return (F.mul(P[0], zi2), F.mul(P[1], F.mul(zi2, zi)))

    def batch_to_affine(self, Ps):
        return [self.to_affine(P) for P in Ps]

    def is_identity(self, P):
        return self.F.is_zero(P[2])

    def on_curve(self, p):
        if p is None:
            return True
        F = self.F
        return F.sub(F.sqr(p[1]), F.add(F.mul(F.sqr(p[0]), p[0]), self.b)) == F.zero

    # ---- group law (Jacobian, a = 0) ----------------------------------------------
    def double(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return P
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        t = F.sub(F.sqr(F.add(X, B)), F.add(A, C))
        D = F.add(t, t)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        YZ = F.mul(Y, Z)
        return (X3, Y3, F.add(YZ, YZ))

    def add(self, P, Q):
        F = self.F
        if F.is_zero(P[2]):
            return Q
        if F.is_zero(Q[2]):
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if U1 == U2:
            if S1 == S2:
                return self.double(P)
            return self.identity
        H = F.sub(U2, U1)
        R = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(R), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def neg(self, P):
        return (P[0], self.F.neg(P[1]), P[2])

    def sub(self, P, Q):
        return self.add(P, self.neg(Q))

    def mul(self, P, k):
        k %= self.r
        R = self.identity
        if k == 0 or self.is_identity(P):
            return R
        for bit in bin(k)[2:]:
            R = self.double(R)
            if bit == "1":
                R = self.add(R, P)
        return R

    def eq(self, P, Q):
        F = self.F
        if F.is_zero(P[2]) or F.is_zero(Q[2]):
            return F.is_zero(P[2]) and F.is_zero(Q[2])
        Z1Z1 = F.sqr(P[2])
        Z2Z2 = F.sqr(Q[2])
        if F.mul(P[0], Z2Z2) != F.mul(Q[0], Z1Z1):
            return False
        return F.mul(P[1], F.mul(Q[2], Z2Z2)) == F.mul(Q[1], F.mul(P[2], Z1Z1))

    def sum(self, Ps):
        acc = self.identity
        for P in Ps:
            acc = self.add(acc, P)
        return acc

    def msm_naive(self, bases_affine, scalars):
        """G::msm restated as the definition sum_i s_i * B_i."""
        if len(bases_affine) != len(scalars):
            raise ValueError(min(len(bases_affine), len(scalars)))  # dmsm/mod.rs:73 `?` on usize
        acc = self.identity
        for b, s in zip(bases_affine, scalars):
            acc = self.add(acc, self.mul(self.from_affine(b), s))
        return acc

    def msm(self, bases_affine, scalars, c=None):
        """Bucket (Pippenger) MSM; same value as ``msm_naive``."""
        n = len(bases_affine)
        if n != len(scalars):
            raise ValueError(min(n, len(scalars)))
        if n == 0:
            return self.identity
        if c is None:
            c = 3 if n < 32 else max(3, (n.bit_length() * 69) // 100 + 2)
        nbits = self.r.bit_length()
        nwin = (nbits + c - 1) // c
        pts = [self.from_affine(b) for b in bases_affine]
        sc = [s % self.r for s in scalars]
        total = self.identity
        for w in reversed(range(nwin)):
            for _ in range(c):
                total = self.double(total)
            buckets = [None] * ((1 << c) - 1)
            for P, s in zip(pts, sc):
                d = (s >> (w * c)) & ((1 << c) - 1)
                if d:
                    buckets[d - 1] = P if buckets[d - 1] is None else self.add(buckets[d - 1], P)
            run = self.identity
            acc = self.identity
            for bkt in reversed(buckets):
                if bkt is not None:
                    run = self.add(run, bkt)
                acc = self.add(acc, run)
            total = self.add(total, acc)
        return total


class GroupOps:
    """DomainCoeff adapter: group elements (Jacobian) scaled by Fr scalars."""

    def __init__(self, group):
        self.g = group
        self.zero = group.identity

    def add(self, a, b):
        return self.g.add(a, b)

    def sub(self, a, b):
        return self.g.sub(a, b)

    def mul(self, a, k):
        return self.g.mul(a, k)

    def eq(self, a, b):
        return self.g.eq(a, b)


_cache = {}


def g1(curve):
    key = (curve.name, 1)
    if key not in _cache:
        _cache[key] = Group(Fq(curve.q), curve.b1 % curve.q, curve.g1, curve.r)
    return _cache[key]


def g2(curve):
    key = (curve.name, 2)
    if key not in _cache:
        if curve.g2 is None:
            raise ValueError("no G2 parameters for " + curve.name)
        _cache[key] = Group(Fq2(curve.q, curve.nonres), curve.b2, curve.g2, curve.r)
    return _cache[key]
