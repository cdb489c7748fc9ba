// Lane-cooperative group addition: one point per QUAD of lanes.
//
// The bucket finalize / reduction steps of Pippenger (msm.hpp) are chains of dependent XYZZ additions with few
// points in flight, so a lane that computes a whole addition alone (14 coordinate-field multiplications one after the
// other, ~1 us each for a lone wave; 40 base-field multiplications for G2) sets the duration of every step while most
// of the chip idles.  Here the four coordinates (X, Y, ZZ, ZZZ) of a point live in the four lanes of a quad
// (lane & 3 = coordinate index), the 14 multiplications of add-2008-s are scheduled as FOUR rounds of one
// multiplication per lane, and operands move between the lanes of a quad with DPP quad_perm moves (full-rate VALU,
// no LDS).  A wave therefore carries 16 points instead of 64 and a step costs 4 multiplication times instead of 14;
// the launch uses 4x the waves, which the otherwise idle SIMDs absorb.
//
//   round 1   U1 = X1 ZZ2      U2 = X2 ZZ1      S1 = Y1 ZZZ2      S2 = Y2 ZZZ1
//             P = U2 - U1 (lanes 0,1)           R = S2 - S1 (lanes 2,3)
//   round 2   PP = P^2         ZZ12 = ZZ1 ZZ2   RR = R^2          ZZZ12 = ZZZ1 ZZZ2
//   round 3   Q = U1 PP        PPP = P PP       ZZ3 = ZZ12 PP     -
//             X3 = RR - PPP - 2Q
//   round 4   S1 PPP           R (Q - X3)       -                 ZZZ3 = ZZZ12 PPP
//             Y3 = R (Q - X3) - S1 PPP
// Control flow is quad-uniform everywhere (all four lanes of a quad take the same branches), which is what makes the
// DPP reads well defined under divergence between quads.
#pragma once
#include "ec.hpp"
#include "ntt.hpp"

namespace zk {
#if defined(__HIPCC__)

template <int P0, int P1, int P2, int P3>
ZK_D uint32_t qperm_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
}
// lane q of every quad receives v from lane P_q of the same quad
template <int P0, int P1, int P2, int P3, class P>
ZK_D Fp<P> qperm(const Fp<P>& v) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < P::N; i++) r.v[i] = qperm_u32<P0, P1, P2, P3>(v.v[i]);
  return r;
}
template <int P0, int P1, int P2, int P3, class P, bool I>
ZK_D Fp2T<P, I> qperm(const Fp2T<P, I>& v) {
  return {qperm<P0, P1, P2, P3>(v.c0), qperm<P0, P1, P2, P3>(v.c1)};
}
template <class P>
ZK_D Fp<P> qsel(bool a, const Fp<P>& x, const Fp<P>& y) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < P::N; i++) r.v[i] = a ? x.v[i] : y.v[i];
  return r;
}
template <class P, bool I>
ZK_D Fp2T<P, I> qsel(bool a, const Fp2T<P, I>& x, const Fp2T<P, I>& y) {
  return {qsel(a, x.c0, y.c0), qsel(a, x.c1, y.c1)};
}

// coordinate q of the identity (X = Y = 1, ZZ = ZZZ = 0)
template <class Fld>
ZK_D Fld qidentity(int q) {
  return q < 2 ? Fld::one() : Fld::zero();
}
// my coordinate of a point stored as XYZZ<Fld> (X, Y, ZZ, ZZZ contiguous: a quad reads one whole point)
template <class Fld>
ZK_D Fld qload(const XYZZ<Fld>* p, int q) {
  return load_elem(reinterpret_cast<const Fld*>(p) + q);
}
template <class Fld>
ZK_D void qstore(XYZZ<Fld>* p, int q, const Fld& c) {
  store_elem(reinterpret_cast<Fld*>(p) + q, c);
}

// a + b for points spread over quads (ca, cb: this lane's coordinate of a and of b); every lane returns its
// coordinate of the sum.  q = lane & 3.
//
// Equal points (P = R = 0 after round 1) are doubled IN THE SAME three remaining rounds by switching operands
// (dbl-2008-s-1: U = 2Y, V = U^2, W = U V, S = X V, M = 3 X^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W Y, ZZ3 = V ZZ,
// ZZZ3 = W ZZZ), so the rare case costs a few selects instead of a second code path:
//   round 2   XX = X X         V = U U          -                 -
//   round 3   S = X V          W = U V          ZZ3 = ZZ V        MM = M M
//   round 4   M (S - X3)       W Y              -                 ZZZ3 = W ZZZ
template <class Fld>
ZK_D Fld qadd(const Fld& ca, const Fld& cb, int q) {
  const bool even = (q & 1) == 0;
  const bool za = qperm<2, 2, 2, 2>(ca).is_zero();
  const bool zb = qperm<2, 2, 2, 2>(cb).is_zero();
  // round 1
  Fld A = qsel(even, qperm<0, 0, 1, 1>(ca), qperm<0, 0, 1, 1>(cb));   // X1  X2  Y1   Y2
  Fld B = qsel(even, qperm<2, 2, 3, 3>(cb), qperm<2, 2, 3, 3>(ca));   // ZZ2 ZZ1 ZZZ2 ZZZ1
  const Fld m1 = A * B;                                                // U1  U2  S1   S2
  const Fld d = qperm<1, 1, 3, 3>(m1) - qperm<0, 0, 2, 2>(m1);         // P   P   R    R
  const bool pz = qperm<0, 0, 0, 0>(d).is_zero();
  const bool rz = qperm<2, 2, 2, 2>(d).is_zero();
  const bool D = pz && rz && !za && !zb;                               // doubling (quad-uniform)
  const Fld cu = qsel(q == 1, ca.dbl(), ca);                           // X   U   ZZ   ZZZ
  // round 2
  A = qsel(D, cu, qsel(even, d, qperm<0, 2, 2, 3>(ca)));               // P   ZZ1 R    ZZZ1   | X  U  -  -
  B = qsel(D, cu, qsel(even, d, qperm<0, 2, 2, 3>(cb)));               // P   ZZ2 R    ZZZ2   | X  U  -  -
  const Fld m2 = A * B;                                                // PP  ZZ12 RR  ZZZ12  | XX V  -  -
  const Fld XX = qperm<0, 0, 0, 0>(m2);
  const Fld M = XX.dbl() + XX;                                         // used by the doubling only
  // round 3
  A = qsel(D, qsel(q == 3, M, cu), qsel(q == 2, qperm<0, 1, 1, 3>(m2), qsel(q == 1, d, m1)));   // U1 P ZZ12 - | X U ZZ M
  B = qsel(D, qsel(q == 3, M, qperm<1, 1, 1, 1>(m2)), XX);                                       // PP          | V V V  M
  const Fld m3 = A * B;                                                // Q   PPP ZZ3  -      | S  W  ZZ3 MM
  const Fld m3_1 = qperm<1, 1, 1, 1>(m3);                              // PPP                 | W
  const Fld m3_0 = qperm<0, 0, 0, 0>(m3);                              // Q                   | S
  const Fld T = qsel(D, qperm<3, 3, 3, 3>(m3), qperm<2, 2, 2, 2>(m2)) - m3_0.dbl();     // RR - 2Q | MM - 2S
  const Fld X3f = qsel(D, T, T - m3_1);                                              // RR - 2Q - PPP | MM - 2S
  // round 4
  A = qsel(D, qsel(q == 0, M, m3_1),                                                 // M  W  -  W
           qsel(q == 1, qperm<2, 2, 2, 2>(d), qsel(q == 0, qperm<2, 2, 2, 2>(m1), m2)));   // S1 R (RR) ZZZ12
  B = qsel(D, qsel(q == 0, m3_0 - X3f, ca),                                          // S-X3  Y  -  ZZZ
           qsel(q == 1, m3_0 - X3f, m3_1));                                          // PPP Q-X3 PPP PPP
  const Fld m4 = A * B;                                                // S1PPP R(Q-X3) - ZZZ3 | M(S-X3) WY - ZZZ3
  const Fld m4_0 = qperm<0, 0, 0, 0>(m4), m4_1 = qperm<1, 1, 1, 1>(m4);
  const Fld Y3 = qsel(D, m4_0 - m4_1, m4_1 - m4_0);
  Fld r = qsel(q == 0, X3f, qsel(q == 1, Y3, qsel(q == 2, m3, m4)));
  if (pz && !rz) r = qidentity<Fld>(q);                                // inverse points
  if (zb) r = ca;
  if (za) r = cb;
  return r;
}

// ---- mixed addition of an EXTENSION-field point shared by a QUAD of lanes, ONE BASE-FIELD VALUE PER LANE (round 4)
// The pair form above keeps whole Fq2 values in a lane; for 12-limb base fields that is 24 limbs per value and the
// kernel ends at the 256-register cap with 99 spilled dwords (BLS12-381 G2: 34 % of the multiplier's peak, a third of a
// 2^24-constraint proof).  Here lane q of a quad holds component q & 1 (c0 / c1) of the values of half q >> 1 (half 0:
// X, ZZ, x; half 1: Y, ZZZ, y): every lane works on single base-field values -- the register footprint of a G1 kernel --
// and an Fq2 product is two base-field products per lane with ONE reduction (mul_pm_mul):
//   c0 = a0 b0 - a1 b1 (lane 0),   c1 = a1 b0 + a0 b1 (lane 1);     squaring: (a0 + a1)(a0 - a1) | 2 a1 a0
// i.e. schoolbook (4 products, 2 reductions per Fq2 product over the two lanes) instead of Karatsuba with lazy reduction
// (3 products, 2 reductions in one lane): 18 % more multiply instructions per addition, none of them spilled.
// The ten multiplications of madd-2008-s run as five rounds of one Fq2 product per half:
//   round 1   half 0: U2 = x2 ZZ1        half 1: S2 = y2 ZZZ1         P = U2 - X1, R = S2 - Y1
//   round 2   half 0: PP = P^2           half 1: RR = R^2
//   round 3   half 0: Q = X1 PP          half 1: PPP = P PP
//   round 4   half 0: ZZ3 = ZZ1 PP       half 1: ZZZ3 = ZZZ1 PPP      X3 = RR - PPP - 2Q
//   round 5   half 0: R (Q - X3)         half 1: Y1 PPP               Y3 = R (Q - X3) - Y1 PPP
// (the pair form of rounds 2-3, with whole Fq2 values per lane); equal points are doubled from the AFFINE addend
// (mdbl-2008-s-1: U = 2y, V = U^2, W = U V, S = x V, M = 3 x^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W y, ZZ3 = V, ZZZ3 = W).
// The other component of a value comes by DPP (cswap), the other half's value by DPP (hswap).
template <class F>
ZK_D F cswap(const F& v) {
  return qperm<1, 0, 3, 2>(v);
}
template <class F>
ZK_D F hswap(const F& v) {
  return qperm<2, 3, 0, 1>(v);
}
template <class P>
ZK_D Fp<P> s2_mul(const Fp<P>& a, const Fp<P>& b, bool comp) {
  const Fp<P> ao = cswap(a), bo = cswap(b);
  return Fp<P>::mul_pm_mul(a, qsel(comp, bo, b), ao, qsel(comp, b, bo), comp);
}
template <class P>
ZK_D Fp<P> s2_sqr(const Fp<P>& a, bool comp) {
  const Fp<P> ao = cswap(a);
  return (a + qsel(comp, a, ao)) * qsel(comp, ao, a - ao);
}
// both components zero (the same answer in the two lanes of a value)
template <class P>
ZK_D bool s2_is_zero(const Fp<P>& a) {
  const uint32_t z = a.is_zero() ? 1u : 0u;
  return (z & qperm_u32<1, 0, 3, 2>(z)) != 0;
}
template <class P>
struct SplitAcc {      // half 0: c0 = X, c1 = ZZ; half 1: c0 = Y, c1 = ZZZ -- this lane's component of each
  Fp<P> c0, c1;
};
template <class P>
ZK_D SplitAcc<P> split_identity(bool comp) {        // X = Y = 1, ZZ = ZZZ = 0
  return SplitAcc<P>{qsel(comp, Fp<P>::zero(), Fp<P>::one()), Fp<P>::zero()};
}
// acc += (x2, y2); `in` = this lane's component of x2 (half 0) or y2 (half 1).  Control flow is quad-uniform.
template <class P>
ZK_D SplitAcc<P> split_madd(const SplitAcc<P>& a, const Fp<P>& in, bool half, bool comp) {
  using F = Fp<P>;
  const uint32_t zz = s2_is_zero(a.c1) ? 1u : 0u;
  SplitAcc<P> r;
  if (qperm_u32<0, 0, 0, 0>(zz)) {                     // running sum is the identity: the affine point itself
    r.c0 = in;
    r.c1 = qsel(comp, F::zero(), F::one());
    return r;
  }
  const F m1 = s2_mul(in, a.c1, comp);                 // 0: U2 = x2 ZZ1        1: S2 = y2 ZZZ1
  const F d = m1 - a.c0;                               // 0: P                  1: R
  const F od = hswap(d);                               // 0: R                  1: P
  const uint32_t dz = s2_is_zero(d) ? 1u : 0u;
  const bool pz = qperm_u32<0, 0, 0, 0>(dz) != 0, rz = qperm_u32<2, 2, 2, 2>(dz) != 0;
  if (pz) {
    if (!rz) return split_identity<P>(comp);           // inverse points
    // equal points: double the affine addend (mdbl-2008-s-1)
    const F u = qsel(half, in.dbl(), in);                 // 0: x                  1: U = 2 y
    const F m2 = s2_sqr(u, comp);                      // 0: XX                 1: V
    const F o2 = hswap(m2);                            // 0: V                  1: XX
    const F m3 = s2_mul(u, qsel(half, m2, o2), comp);    // 0: S = x V            1: W = U V
    const F M = m2.dbl() + m2;                         // 0: 3 XX
    const F m4 = s2_sqr(M, comp);                      // 0: MM
    const F X3 = m4 - m3.dbl();                        // 0: MM - 2 S
    const F m5 = s2_mul(qsel(half, m3, M), qsel(half, in, m3 - X3), comp);     // 0: M (S - X3)   1: W y
    const F o5 = hswap(m5);
    r.c0 = qsel(half, o5 - m5, X3);
    r.c1 = qsel(half, m3, o2);                           // ZZ3 = V, ZZZ3 = W
    return r;
  }
  const F m2 = s2_sqr(d, comp);                        // 0: PP                 1: RR
  const F o2 = hswap(m2);                              // 0: RR                 1: PP
  const F m3 = s2_mul(qsel(half, od, a.c0), qsel(half, o2, m2), comp);       // 0: Q = X1 PP    1: PPP = P PP
  const F o3 = hswap(m3);                              // 0: PPP                1: Q
  const F m4 = s2_mul(a.c1, qsel(half, m3, m2), comp);    // 0: ZZ3 = ZZ1 PP       1: ZZZ3 = ZZZ1 PPP
  const F X3 = o2 - o3 - m3.dbl();                     // 0: RR - PPP - 2 Q
  const F m5 = s2_mul(qsel(half, a.c0, od), qsel(half, m3, m3 - X3), comp);    // 0: R (Q - X3)   1: Y1 PPP
  const F o5 = hswap(m5);
  r.c0 = qsel(half, o5 - m5, X3);
  r.c1 = m4;
  return r;
}

// 2 acc in the same split form (dbl-2008-s-1, a = 0): nine Fq2 products as five rounds of one product per half
//   round 1   half 0: XX = X^2           half 1: V = U^2 (U = 2 Y)
//   round 2   half 0: S = X V            half 1: W = U V
//   round 3   half 0: MM = M^2 (M = 3 XX) half 1: W Y                     X3 = MM - 2 S
//   round 4   half 0: M (S - X3)         half 1: ZZZ3 = W ZZZ            Y3 = M (S - X3) - W Y
//   round 5   half 0: ZZ3 = V ZZ         half 1: -
template <class P>
ZK_D SplitAcc<P> split_dbl(const SplitAcc<P>& a, bool half, bool comp) {
  using F = Fp<P>;
  const uint32_t zz = s2_is_zero(a.c1) ? 1u : 0u;
  if (qperm_u32<0, 0, 0, 0>(zz)) return a;               // identity (quad-uniform)
  const F u = qsel(half, a.c0.dbl(), a.c0);              // 0: X                  1: U
  const F m1 = s2_sqr(u, comp);                          // 0: XX                 1: V
  const F o1 = hswap(m1);                                // 0: V                  1: XX
  const F m2 = s2_mul(u, qsel(half, m1, o1), comp);      // 0: S                  1: W
  const F M = m1.dbl() + m1;                             // 0: 3 XX
  const F m3 = s2_mul(qsel(half, m2, M), qsel(half, a.c0, M), comp);          // 0: MM           1: W Y
  const F X3 = m3 - m2.dbl();                            // 0: MM - 2 S
  const F m4 = s2_mul(qsel(half, m2, M), qsel(half, a.c1, m2 - X3), comp);    // 0: M (S - X3)   1: ZZZ3
  const F m5 = s2_mul(o1, a.c1, comp);                   // 0: ZZ3 = V ZZ
  SplitAcc<P> r;
  r.c0 = qsel(half, hswap(m4) - m3, X3);
  r.c1 = qsel(half, m4, m5);
  return r;
}

// ---- affine extension-field points in the same lane layout (lane q of a quad holds base-field value q of x.c0 x.c1 y.c0 y.c1)
// 1 / z for an Fq2 value whose component `comp` this lane holds (the same value in both halves of the quad)
template <class P>
ZK_D Fp<P> split_inv(const Fp<P>& z, bool comp) {
  const Fp<P> sq = z * z;
  const Fp<P> ninv = (sq + cswap(sq)).inverse_fast();    // 1 / (c0^2 + c1^2)
  const Fp<P> r = z * ninv;
  return qsel(comp, r.neg(), r);
}
ZK_D bool quad_all(bool b) {
  uint32_t z = b ? 1u : 0u;
  z &= qperm_u32<1, 0, 3, 2>(z);
  z &= qperm_u32<2, 3, 0, 1>(z);
  return z != 0;
}
// a + b for two affine points with every special case (identities, equal, opposite); `ida` / `idb` / `idr`: the point is
// the identity (quad-uniform flags; the coordinates are then ignored).  Out of line: setup code of the pack kernel.
template <class P>
__device__ __noinline__ Fp<P> split_affine_add(const Fp<P>& a, bool ida, const Fp<P>& b, bool idb, bool half, bool comp,
                                               bool* idr) {
  using F = Fp<P>;
  *idr = false;
  if (ida) {
    *idr = idb;
    return b;
  }
  if (idb) return a;
  const F d = b - a;                                     // 0: x2 - x1            1: y2 - y1
  const uint32_t dz = s2_is_zero(d) ? 1u : 0u;
  const bool xz = qperm_u32<0, 0, 0, 0>(dz) != 0, yz = qperm_u32<2, 2, 2, 2>(dz) != 0;
  const F xa = qperm<0, 1, 0, 1>(a), ya = qperm<2, 3, 2, 3>(a);        // every lane: its component of x1, of y1
  F lam;
  if (xz) {
    if (!yz) {                                           // opposite points
      *idr = true;
      return F::zero();
    }
    const F xx = s2_sqr(xa, comp);                       // tangent: 3 x^2 / (2 y)
    lam = s2_mul(xx.dbl() + xx, split_inv(ya.dbl(), comp), comp);
  } else {
    lam = s2_mul(qperm<2, 3, 2, 3>(d), split_inv(qperm<0, 1, 0, 1>(d), comp), comp);
  }
  const F x3 = s2_sqr(lam, comp) - xa - qperm<0, 1, 0, 1>(b);
  const F y3 = s2_mul(lam, xa - x3, comp) - ya;
  return qsel(half, y3, x3);
}

// Normalise a split running sum and store it as Affine<Fq2> (dst: this lane's base-field element of the point, i.e.
// reinterpret_cast<Fp<P>*>(point) + q): 1 / ZZZ through the norm -- ONE base-field inversion, computed by the four lanes
// alike -- then x = X (ZZ / ZZZ)^2, y = Y / ZZZ.  The identity is stored as the (0, 0) sentinel.  Quad-uniform.
template <class P>
ZK_D void split_store_affine(const SplitAcc<P>& acc, Fp<P>* dst, bool half, bool comp) {
  using F = Fp<P>;
  const uint32_t zz = s2_is_zero(acc.c1) ? 1u : 0u;
  if (qperm_u32<0, 0, 0, 0>(zz)) {
    store_elem(dst, F::zero());
    return;
  }
  const F zi = split_inv(qperm<2, 3, 2, 3>(acc.c1), comp);                      // 1 / ZZZ, every lane its component
  const F zi2 = s2_sqr(s2_mul(zi, qperm<0, 1, 0, 1>(acc.c1), comp), comp);      // (ZZ / ZZZ)^2 = 1 / Z^2
  store_elem(dst, s2_mul(acc.c0, qsel(half, zi, zi2), comp));                   // 0: X / Z^2        1: Y / Z^3
}

// Tree sum inside aligned sub-blocks of `nvl` (power of two) virtual lanes of a workgroup: every quad contributes the
// point whose coordinates its lanes hold; afterwards the first quad of each sub-block holds the sub-block's total.
// `sh` is LDS for blockDim.x / 4 points.  log2(nvl) dependent additions.
template <class Fld>
ZK_D Fld wg_quad_sum(Fld c, XYZZ<Fld>* sh, int vl, int q, int nvl) {
  qstore(sh + vl, q, c);
  __syncthreads();
  for (int off = nvl >> 1; off >= 1; off >>= 1) {
    const bool act = (vl & (nvl - 1)) < off;         // quad-uniform
    Fld r = c;
    if (act) r = qadd(c, qload(sh + vl + off, q), q);
    __syncthreads();
    if (act) {
      c = r;
      qstore(sh + vl, q, c);
    }
    __syncthreads();
  }
  return c;
}

#endif  // __HIPCC__
}  // namespace zk
