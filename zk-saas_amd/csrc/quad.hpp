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
