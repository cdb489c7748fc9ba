#!/usr/bin/env python3
"""Generate zk-saas_amd/csrc/glv_params.hpp: constants of the j = 0 endomorphism phi(x, y) = (beta x, y) = lambda (x, y) of
BN254 / BLS12-381 / BLS12-377 (G1, and G2 where the curve has one on this path) and of the two-dimensional scalar
decomposition k = k1 + lambda k2 (mod r) with |k1|, |k2| ~ sqrt(r), as the dealer's point-packing kernels use them
(csrc/groth16.hpp, csrc/pack_split.hpp: det_pack over curve points, groth16/src/proving_key.rs:72-86).

Build-time tool of the product; independent of oracle/ (published curve parameters + its own few lines of curve arithmetic
to pick the beta that belongs to lambda).  Run:  python tools/gen_glv.py > zk-saas_amd/csrc/glv_params.hpp

Decomposition (Gallant-Lambert-Vanstone with a reduced lattice basis (a1, b1), (a2, b2) of {(x, y): x + lambda y = 0 mod r}):
    c1 = round(b2 k / r),  c2 = round(-b1 k / r),  k1 = k - c1 a1 - c2 a2,  k2 = -c1 b1 - c2 b2
The host code takes c_i = (k * G_i) >> SHIFT with G_i = round(2^SHIFT |.| / r) (an error of at most one unit, which only moves
k1, k2 by one basis vector), and then forms k1, k2 MODULO r with the four constants N11, N12, N21, N22 below (signs absorbed
mod r); a value above r / 2 is the negative r - value.  The host verifies k1 + lambda k2 = k for every scalar it decomposes and
falls back to the plain digits when that fails or a part exceeds the digit length.
"""
import sys

SHIFT = 384

CURVES = [
    # name, FrP, FqP, r, q, b1, G1 generator, G2: (nonresidue, b2, generator) or None
    ("bn254", "Bn254Fr", "Bn254Fq",
     21888242871839275222246405745257275088548364400416034343698204186575808495617,
     21888242871839275222246405745257275088696311157297823662689037894645226208583, 3, (1, 2),
     (-1, None,
      ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634),
       (8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531)))),
    ("bls12_381", "Bls381Fr", "Bls381Fq",
     0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
     0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 4,
     (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
      0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
     (-1, None,
      ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
       (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)))),
    ("bls12_377", "Bls377Fr", "Bls377Fq",
     8444461749428370424248824938781546531375899335154063827935233455917409239041,
     0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001, 1,
     (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
      241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030),
     None),
]


def limbs(v, n):
    return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))


def cube_root_of_unity(p):
    a = 2
    while True:
        w = pow(a, (p - 1) // 3, p)
        if w != 1:
            return w
        a += 1


class Fq2:
    """Fq[u] / (u^2 - nonres) on pairs; Fq itself is handled as nonres = None on ints."""

    def __init__(self, q, nonres):
        self.q, self.nr = q, nonres

    def add(self, a, b):
        if self.nr is None:
            return (a + b) % self.q
        return ((a[0] + b[0]) % self.q, (a[1] + b[1]) % self.q)

    def sub(self, a, b):
        if self.nr is None:
            return (a - b) % self.q
        return ((a[0] - b[0]) % self.q, (a[1] - b[1]) % self.q)

    def mul(self, a, b):
        if self.nr is None:
            return a * b % self.q
        return ((a[0] * b[0] + self.nr * a[1] * b[1]) % self.q, (a[0] * b[1] + a[1] * b[0]) % self.q)

    def inv(self, a):
        if self.nr is None:
            return pow(a, self.q - 2, self.q)
        n = pow((a[0] * a[0] - self.nr * a[1] * a[1]) % self.q, self.q - 2, self.q)
        return (a[0] * n % self.q, -a[1] * n % self.q)

    def scale(self, a, k):      # by an element of Fq
        if self.nr is None:
            return a * k % self.q
        return (a[0] * k % self.q, a[1] * k % self.q)


def ec_add(F, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if P[1] != Q[1]:
            return None
        x2 = F.mul(P[0], P[0])
        lam = F.mul(F.add(F.add(x2, x2), x2), F.inv(F.add(P[1], P[1])))
    else:
        lam = F.mul(F.sub(Q[1], P[1]), F.inv(F.sub(Q[0], P[0])))
    x3 = F.sub(F.sub(F.mul(lam, lam), P[0]), Q[0])
    return (x3, F.sub(F.mul(lam, F.sub(P[0], x3)), P[1]))


def ec_mul(F, P, k):
    R = None
    for bit in bin(k)[2:]:
        R = ec_add(F, R, R)
        if bit == "1":
            R = ec_add(F, R, P)
    return R


def reduced_basis(r, lam):
    """Extended Euclid on (r, lambda) stopped around sqrt(r) (GLV, section 4): two short vectors (a, b) with a + lambda b = 0."""
    import math
    rows = [(r, 0), (lam, 1)]           # (remainder, t): remainder = s r + t lambda
    sq = math.isqrt(r)
    while rows[-1][0] >= sq:
        q = rows[-2][0] // rows[-1][0]
        rows.append((rows[-2][0] - q * rows[-1][0], rows[-2][1] - q * rows[-1][1]))
    q = rows[-2][0] // rows[-1][0]
    nxt = (rows[-2][0] - q * rows[-1][0], rows[-2][1] - q * rows[-1][1])
    v1 = (rows[-1][0], -rows[-1][1])
    cand = [(rows[-2][0], -rows[-2][1]), (nxt[0], -nxt[1])]
    v2 = min(cand, key=lambda v: v[0] * v[0] + v[1] * v[1])
    for a, b in (v1, v2):
        assert (a + lam * b) % r == 0
    return v1, v2


def main():
    out = []
    out.append("// GENERATED by tools/gen_glv.py -- do not edit.")
    out.append("// phi(x, y) = (beta x, y) = lambda (x, y) on the j = 0 curves of this path, and the constants of the scalar decomposition")
    out.append("// k = k1 + lambda k2 (mod r), |k1|, |k2| < 2^GLV_BITS (tools/gen_glv.py says how they are used; groth16.hpp glv_split).")
    out.append("#pragma once")
    out.append('#include "params.hpp"')
    out.append("namespace zk {")
    out.append("template <class FrP>")
    out.append("struct Glv {")
    out.append("  static constexpr bool OK = false;")
    out.append("};")
    for name, frp, fqp, r, q, b1, g1, g2 in CURVES:
        nq = (q.bit_length() + 31) // 32
        Rq = 1 << (32 * nq)
        lam = cube_root_of_unity(r)
        beta = cube_root_of_unity(q)
        F1 = Fq2(q, None)
        want = ec_mul(F1, g1, lam)
        b_g1 = next(b for b in (beta, beta * beta % q) if (b * g1[0] % q, g1[1]) == want)
        b_g2 = 0
        if g2 is not None:
            F2 = Fq2(q, g2[0] % q)
            G = g2[2]
            want2 = ec_mul(F2, G, lam)
            b_g2 = next(b for b in (beta, beta * beta % q) if (F2.scale(G[0], b), G[1]) == want2)
        (a1, bb1), (a2, bb2) = reduced_basis(r, lam)
        det = a1 * bb2 - a2 * bb1
        if det < 0:
            (a1, bb1), (a2, bb2) = (a2, bb2), (a1, bb1)
            det = -det
        assert det == r, "basis does not span the lattice"
        # c1 = round(b2 k / r), c2 = round(-b1 k / r); fold the signs of b2 / -b1 into the N constants
        s1 = 1 if bb2 >= 0 else -1
        s2 = 1 if -bb1 >= 0 else -1
        G1c = ((abs(bb2) << SHIFT) + r // 2) // r
        G2c = ((abs(bb1) << SHIFT) + r // 2) // r
        N11, N12 = (-s1 * a1) % r, (-s2 * a2) % r
        N21, N22 = (-s1 * bb1) % r, (-s2 * bb2) % r
        # self-check on a few scalars, with the truncating multiply-high the host code uses
        import random
        rnd = random.Random(1)
        worst = 0
        for _ in range(2000):
            k = rnd.randrange(r)
            c1, c2 = (k * G1c) >> SHIFT, (k * G2c) >> SHIFT
            k1 = (k + c1 * N11 + c2 * N12) % r
            k2 = (c1 * N21 + c2 * N22) % r
            assert (k1 + lam * k2 - k) % r == 0
            for v in (k1, k2):
                worst = max(worst, min(v, r - v).bit_length())
        bits = worst + 1
        assert bits <= 131, bits
        ng = (max(G1c, G2c).bit_length() + 31) // 32
        out.append("template <>")
        out.append("struct Glv<%s> {   // %s" % (frp, name))
        out.append("  static constexpr bool OK = true;")
        out.append("  static constexpr int SHIFT = %d, NG = %d, BITS = %d;   // parts below 2^BITS in magnitude (2000 random scalars + 1 bit)" % (SHIFT, ng, bits))
        out.append("  static constexpr uint32_t G1C[%d] = {%s};" % (ng, limbs(G1c, ng)))
        out.append("  static constexpr uint32_t G2C[%d] = {%s};" % (ng, limbs(G2c, ng)))
        for nm, v in (("N11", N11), ("N12", N12), ("N21", N21), ("N22", N22), ("LAMBDA", lam)):
            out.append("  static constexpr uint32_t %s[8] = {%s};   // canonical" % (nm, limbs(v, 8)))
        out.append("  static constexpr uint32_t BETA_G1[%d] = {%s};   // Montgomery form (%s)" % (nq, limbs(b_g1 * Rq % q, nq), fqp))
        out.append("  static constexpr uint32_t BETA_G2[%d] = {%s};   // Montgomery form; zero: no G2 on this path" % (nq, limbs(b_g2 * Rq % q, nq)))
        out.append("};")
    out.append("}  // namespace zk")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
