"""Optimal-ate pairing for BN254 and BLS12-381 and the Groth16 verifier, with Python ints.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).

Why it exists: the reference's only end-to-end assertion is ``Groth16::verify_proof`` on the reconstructed proof
(groth16/examples/sha256.rs:389-415), i.e. the pairing equation
    e(A, B) == e(alpha_g1, beta_g2) * e(sum_i x_i * gamma_abc_g1[i], gamma_g2) * e(C, delta_g2)
(ark-groth16 ^0.4 ``verify_proof_with_prepared_inputs``; third-party, not vendored -- SURVEY.md F2), and the
reference checkout HOLDS a golden pairing value: fixtures/verification_key.json carries snarkjs' ``vk_alphabeta_12 =
e(vk_alpha_1, vk_beta_2)`` next to the points.  That value pins every constant this module and oracle/params.py /
oracle/curve.py use for BN254: q, the Fq2 / Fq6 / Fq12 tower (u^2 = -1, v^3 = 9 + u, w^2 = v), the twist
y^2 = x^3 + 3/(9+u), the G2 generator (= vk_gamma_2), the loop count 6x+2 and the Frobenius constants
(tests/test_pairing.py).

Conventions (both libraries the fixture can come from agree on them):
* Fq12 element = (c0, c1) over Fq6, c_i = (b0, b1, b2) over Fq2, b_j = (a0, a1): the nesting of snarkjs'
  ``vk_alphabeta_12`` ([2][3][2] decimal strings) and of arkworks' Fp12 = Fp6[w]/(w^2 - v).
* BN254 final exponentiation: easy part (q^6-1)(q^2+1), hard part by the Fuentes-Castaneda et al. addition chain that
  libff / ffjavascript (snarkjs) and arkworks' ``Bn::final_exponentiation`` all use; that chain raises to
  lambda' = 2x(6x^2+3x+1) * (q^4-q^2+1)/r -- a fixed power of the reduced pairing coprime to r, still bilinear and
  non-degenerate.  ``BN_EXACT_HARD_PART = False`` selects it (the golden value decides: see the test).
* BLS12-381: exact exponent (q^12-1)/r (no golden value exists in the reference for this curve; only bilinearity
  and the verification equation are used).

Miller loop: affine line functions on the twist (one Fq2 inversion per step -- speed is irrelevant here), the
G2 point untwisted on the fly: D-type twist (BN254) psi(x', y') = (x' w^2, y' w^3); M-type (BLS12-381)
psi(x', y') = (x' / w^2, y' / w^3), the line scaled by w^3, which lies in the proper subfield Fq4 and is killed by the
final exponentiation.
"""

from .field import inv_mod
from .params import BLS12_381, BN254

BN_EXACT_HARD_PART = False

_BN_X = 4965661367192848881
_BLS_X = 0xD201000000010000        # |x|; the BLS12-381 parameter is negative


class Tower:
    """Fq2 = Fq[u]/(u^2 - nr), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v)."""

    def __init__(self, q, nr, xi):
        self.q, self.nr, self.xi = q, nr % q, xi
        self.zero2, self.one2 = (0, 0), (1, 0)
        self.zero6 = (self.zero2,) * 3
        self.one6 = (self.one2, self.zero2, self.zero2)
        self.one12 = (self.one6, self.zero6)

    # ---- Fq2
    def add2(self, a, b):
        return ((a[0] + b[0]) % self.q, (a[1] + b[1]) % self.q)

    def sub2(self, a, b):
        return ((a[0] - b[0]) % self.q, (a[1] - b[1]) % self.q)

    def neg2(self, a):
        return ((-a[0]) % self.q, (-a[1]) % self.q)

    def mul2(self, a, b):
        q = self.q
        return ((a[0] * b[0] + self.nr * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)

    def inv2(self, a):
        q = self.q
        ni = inv_mod((a[0] * a[0] - self.nr * a[1] * a[1]) % q, q)
        return (a[0] * ni % q, (-a[1]) * ni % q)

    def conj2(self, a):
        return (a[0], (-a[1]) % self.q)

    def pow2(self, a, e):
        r, b = self.one2, a
        while e:
            if e & 1:
                r = self.mul2(r, b)
            b = self.mul2(b, b)
            e >>= 1
        return r

    def mulxi(self, a):
        return self.mul2(a, self.xi)

    # ---- Fq6
    def add6(self, a, b):
        return tuple(self.add2(x, y) for x, y in zip(a, b))

    def sub6(self, a, b):
        return tuple(self.sub2(x, y) for x, y in zip(a, b))

    def neg6(self, a):
        return tuple(self.neg2(x) for x in a)

    def mul6(self, a, b):
        m, ad, xi = self.mul2, self.add2, self.mulxi
        a0, a1, a2 = a
        b0, b1, b2 = b
        c0 = ad(m(a0, b0), xi(ad(m(a1, b2), m(a2, b1))))
        c1 = ad(ad(m(a0, b1), m(a1, b0)), xi(m(a2, b2)))
        c2 = ad(ad(m(a0, b2), m(a1, b1)), m(a2, b0))
        return (c0, c1, c2)

    def mulv(self, a):
        """a * v"""
        return (self.mulxi(a[2]), a[0], a[1])

    def inv6(self, a):
        m, sb, ad, xi = self.mul2, self.sub2, self.add2, self.mulxi
        a0, a1, a2 = a
        t0 = sb(m(a0, a0), xi(m(a1, a2)))
        t1 = sb(xi(m(a2, a2)), m(a0, a1))
        t2 = sb(m(a1, a1), m(a0, a2))
        d = ad(m(a0, t0), xi(ad(m(a2, t1), m(a1, t2))))
        di = self.inv2(d)
        return (m(t0, di), m(t1, di), m(t2, di))

    # ---- Fq12
    def mul12(self, a, b):
        a0, a1 = a
        b0, b1 = b
        t0, t1 = self.mul6(a0, b0), self.mul6(a1, b1)
        c0 = self.add6(t0, self.mulv(t1))
        c1 = self.sub6(self.sub6(self.mul6(self.add6(a0, a1), self.add6(b0, b1)), t0), t1)
        return (c0, c1)

    def conj12(self, a):
        return (a[0], self.neg6(a[1]))

    def inv12(self, a):
        a0, a1 = a
        d = self.sub6(self.mul6(a0, a0), self.mulv(self.mul6(a1, a1)))
        di = self.inv6(d)
        return (self.mul6(a0, di), self.neg6(self.mul6(a1, di)))

    def pow12(self, a, e):
        r, b = self.one12, a
        while e:
            if e & 1:
                r = self.mul12(r, b)
            b = self.mul12(b, b)
            e >>= 1
        return r


class Pairing:
    def __init__(self, curve, xi, twist, loop, loop_neg, bn_x=None):
        self.curve = curve
        self.q, self.r = curve.q, curve.r
        self.T = Tower(curve.q, curve.nonres, xi)
        self.twist = twist
        self.loop, self.loop_neg, self.bn_x = loop, loop_neg, bn_x
        T = self.T
        if bn_x is not None:
            q = self.q
            self.g12 = T.pow2(xi, (q - 1) // 3)
            self.g13 = T.pow2(xi, (q - 1) // 2)
            self.g22 = T.pow2(xi, (q * q - 1) // 3)
            self.g23 = T.pow2(xi, (q * q - 1) // 2)

    # line through T (slope lam on the twist) evaluated at P = (xp, yp) in G1, as an Fq12 element
    def _line(self, lam, xt, yt, xp, yp):
        T = self.T
        a = T.sub2(T.mul2(lam, xt), yt)                     # lam*xT - yT
        b = T.neg2((lam[0] * xp % self.q, lam[1] * xp % self.q))   # -lam*xP
        y = (yp % self.q, 0)
        z = T.zero2
        if self.twist == "D":      # yP - lam xP w + (lam xT - yT) w^3,  w^3 = v w
            return ((y, z, z), (b, a, z))
        # M: (lam xT - yT) - lam xP w^2 + yP w^3  (the line times w^3), w^2 = v
        return ((a, b, z), (z, y, z))

    def _dbl(self, pt):
        T = self.T
        x, y = pt
        x2 = T.mul2(x, x)
        lam = T.mul2(T.add2(T.add2(x2, x2), x2), T.inv2(T.add2(y, y)))
        x3 = T.sub2(T.mul2(lam, lam), T.add2(x, x))
        return lam, (x3, T.sub2(T.mul2(lam, T.sub2(x, x3)), y))

    def _add(self, pt, q):
        T = self.T
        lam = T.mul2(T.sub2(q[1], pt[1]), T.inv2(T.sub2(q[0], pt[0])))
        x3 = T.sub2(T.sub2(T.mul2(lam, lam), pt[0]), q[0])
        return lam, (x3, T.sub2(T.mul2(lam, T.sub2(pt[0], x3)), pt[1]))

    def miller(self, P, Q):
        """f_{loop,Q}(P) (+ the two Frobenius lines for BN); P affine in G1, Q affine on the twist, both != identity."""
        T = self.T
        xp, yp = P
        f = T.one12
        t = Q
        for bit in bin(self.loop)[3:]:
            lam, t2 = self._dbl(t)
            f = T.mul12(T.mul12(f, f), self._line(lam, t[0], t[1], xp, yp))
            t = t2
            if bit == "1":
                lam, t2 = self._add(t, Q)
                f = T.mul12(f, self._line(lam, t[0], t[1], xp, yp))
                t = t2
        if self.bn_x is not None:
            q1 = (T.mul2(T.conj2(Q[0]), self.g12), T.mul2(T.conj2(Q[1]), self.g13))
            nq2 = (T.mul2(Q[0], self.g22), T.neg2(T.mul2(Q[1], self.g23)))
            lam, t2 = self._add(t, q1)
            f = T.mul12(f, self._line(lam, t[0], t[1], xp, yp))
            t = t2
            lam, _ = self._add(t, nq2)
            f = T.mul12(f, self._line(lam, t[0], t[1], xp, yp))
        if self.loop_neg:
            f = T.conj12(f)       # f_{-s} = 1/f_s up to subfield factors; conj == inverse after the easy part
        return f

    def final_exp(self, f):
        T, q, r = self.T, self.q, self.r
        f = T.mul12(T.conj12(f), T.inv12(f))                 # ^(q^6 - 1)
        f = T.pow12(f, q * q + 1)                            # ^(q^2 + 1)
        if self.bn_x is not None and not BN_EXACT_HARD_PART:
            x = self.bn_x
            l0 = 1 + 6 * x + 12 * x * x + 12 * x ** 3
            l1 = 4 * x + 6 * x * x + 12 * x ** 3
            l2 = 6 * x + 6 * x * x + 12 * x ** 3
            l3 = -1 + 4 * x + 6 * x * x + 12 * x ** 3
            e = l0 + l1 * q + l2 * q * q + l3 * q ** 3
            assert e == 2 * x * (6 * x * x + 3 * x + 1) * ((q ** 4 - q * q + 1) // r)
            return T.pow12(f, e)
        return T.pow12(f, (q ** 4 - q * q + 1) // r)

    def pairing(self, P, Q):
        """e(P, Q); identity in either slot gives 1."""
        if P is None or Q is None:
            return self.T.one12
        return self.final_exp(self.miller(P, Q))

    def multi_pairing(self, pairs):
        """prod e(P_i, Q_i) with one final exponentiation."""
        T = self.T
        f = T.one12
        for P, Q in pairs:
            if P is None or Q is None:
                continue
            f = T.mul12(f, self.miller(P, Q))
        return self.final_exp(f)


_cache = {}


def pairing_for(curve):
    if curve.name not in _cache:
        if curve.name == "bn254":
            _cache[curve.name] = Pairing(BN254, (9, 1), "D", 6 * _BN_X + 2, False, bn_x=_BN_X)
        elif curve.name == "bls12_381":
            _cache[curve.name] = Pairing(BLS12_381, (1, 1), "M", _BLS_X, True)
        else:
            raise ValueError("no pairing parameters for " + curve.name)
    return _cache[curve.name]


class VerifyingKey:
    """ark_groth16::VerifyingKey: alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (affine)."""

    def __init__(self, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1):
        self.alpha_g1, self.beta_g2, self.gamma_g2, self.delta_g2 = alpha_g1, beta_g2, gamma_g2, delta_g2
        self.gamma_abc_g1 = list(gamma_abc_g1)


def verifying_key_from_trapdoor(key, G1, G2):
    """The vk of oracle.groth16.setup_scalars' key (generate_parameters: vk.gamma_abc_g1 = gamma_abc * G1)."""
    td = key.td

    def m1(s):
        return G1.to_affine(G1.mul(G1.from_affine(G1.gen), s))

    def m2(s):
        return G2.to_affine(G2.mul(G2.from_affine(G2.gen), s))

    return VerifyingKey(m1(td.alpha), m2(td.beta), m2(td.gamma), m2(td.delta), [m1(s) for s in key.gamma_abc])


def verify_proof(curve, vk, proof, public_inputs, G1):
    """ark_groth16 Groth16::verify_proof (sha256.rs:400-415): proof = (A, B, C) affine (A, C in G1, B in G2),
    public_inputs WITHOUT the leading constant 1 (prepare_inputs adds gamma_abc_g1[0])."""
    if len(public_inputs) + 1 != len(vk.gamma_abc_g1):
        raise ValueError("malformed verifying key")        # SynthesisError::MalformedVerifyingKey
    pr = pairing_for(curve)
    acc = G1.from_affine(vk.gamma_abc_g1[0])
    for x, b in zip(public_inputs, vk.gamma_abc_g1[1:]):
        acc = G1.add(acc, G1.mul(G1.from_affine(b), x))
    acc = G1.to_affine(acc)
    A, B, C = proof
    lhs = pr.pairing(A, B)
    rhs = pr.multi_pairing([(vk.alpha_g1, vk.beta_g2), (acc, vk.gamma_g2), (C, vk.delta_g2)])
    return lhs == rhs
