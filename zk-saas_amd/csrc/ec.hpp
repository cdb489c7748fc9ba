// Short-Weierstrass (a = 0) group arithmetic, generic over the coordinate field (Fq for G1, Fq2 for G2).
//
// Replaces, on this path, ark-ec's `CurveGroup` add / double / mixed add reached from
// `VariableBaseMSM::msm` (dist-primitives/src/dmsm/mod.rs:73) and from prove.rs:40-56.
// Group elements are unique, so the choice of coordinates (XYZZ for bucket work, Jacobian at the ABI)
// does not affect results once normalised.
#pragma once
#include "field.hpp"

namespace zk {

template <class Fld>
struct Affine {   // (0, 0) is the identity sentinel (not on any supported curve)
  Fld x, y;
  ZK_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

template <class Fld>
struct Jacobian {  // Z = 0 is the identity
  Fld X, Y, Z;
};

// Extended Jacobian: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 is the identity.
template <class Fld>
struct XYZZ {
  Fld X, Y, ZZ, ZZZ;
  ZK_HD static XYZZ identity() { return {Fld::one(), Fld::one(), Fld::zero(), Fld::zero()}; }
  ZK_HD bool is_identity() const { return ZZ.is_zero(); }
  ZK_HD static XYZZ from_affine(const Affine<Fld>& p) {
    if (p.is_identity()) return identity();
    return {p.x, p.y, Fld::one(), Fld::one()};
  }
  ZK_HD XYZZ neg() const { return {X, Y.neg(), ZZ, ZZZ}; }
};

// 2 * (affine p)  [mdbl-2008-s-1, a = 0]
template <class Fld>
ZK_HD XYZZ<Fld> xyzz_dbl_affine(const Fld& x, const Fld& y) {
  Fld U = y.dbl();
  Fld V = U.sqr();
  Fld W = U * V;
  Fld S = x * V;
  Fld xx = x.sqr();
  Fld M = xx.dbl() + xx;
  Fld X3 = M.sqr() - S.dbl();
  Fld Y3 = Fld::mul_sub_mul(M, S - X3, W, y);          // one reduction for the two products (field.hpp)
  return {X3, Y3, V, W};
}

// 2 * p  [dbl-2008-s-1, a = 0]
template <class Fld>
ZK_HD XYZZ<Fld> xyzz_dbl(const XYZZ<Fld>& p) {
  if (p.is_identity()) return p;
  Fld U = p.Y.dbl();
  Fld V = U.sqr();
  Fld W = U * V;
  Fld S = p.X * V;
  Fld xx = p.X.sqr();
  Fld M = xx.dbl() + xx;
  Fld X3 = M.sqr() - S.dbl();
  Fld Y3 = Fld::mul_sub_mul(M, S - X3, W, p.Y);
  return {X3, Y3, V * p.ZZ, W * p.ZZZ};
}

// acc + (x, +-y)  [madd-2008-s]; the affine point must not be the identity.
template <class Fld>
ZK_HD XYZZ<Fld> xyzz_madd(const XYZZ<Fld>& a, const Fld& x2, const Fld& y2) {
  if (a.is_identity()) return {x2, y2, Fld::one(), Fld::one()};
  Fld U2 = x2 * a.ZZ;
  Fld S2 = y2 * a.ZZZ;
  Fld P = U2 - a.X;
  Fld R = S2 - a.Y;
  if (P.is_zero()) {
    if (R.is_zero()) return xyzz_dbl_affine(x2, y2);
    return XYZZ<Fld>::identity();
  }
  Fld PP = P.sqr();
  Fld PPP = P * PP;
  Fld Q = a.X * PP;
  Fld X3 = R.sqr() - PPP - Q.dbl();
  Fld Y3 = Fld::mul_sub_mul(R, Q - X3, a.Y, PPP);
  return {X3, Y3, a.ZZ * PP, a.ZZZ * PPP};
}

// a + b  [add-2008-s]
template <class Fld>
ZK_HD XYZZ<Fld> xyzz_add(const XYZZ<Fld>& a, const XYZZ<Fld>& b) {
  if (a.is_identity()) return b;
  if (b.is_identity()) return a;
  Fld U1 = a.X * b.ZZ;
  Fld U2 = b.X * a.ZZ;
  Fld S1 = a.Y * b.ZZZ;
  Fld S2 = b.Y * a.ZZZ;
  Fld P = U2 - U1;
  Fld R = S2 - S1;
  if (P.is_zero()) {
    if (R.is_zero()) return xyzz_dbl(a);
    return XYZZ<Fld>::identity();
  }
  Fld PP = P.sqr();
  Fld PPP = P * PP;
  Fld Q = U1 * PP;
  Fld X3 = R.sqr() - PPP - Q.dbl();
  Fld Y3 = Fld::mul_sub_mul(R, Q - X3, S1, PPP);
  return {X3, Y3, a.ZZ * b.ZZ * PP, a.ZZZ * b.ZZZ * PPP};
}

#if defined(__HIPCC__)
// parity-test kernel for the base-field primitives of the group kernels (engine fq_selftest)
template <class Fq, bool INL>
__global__ void fq_selftest_kernel(int op, const Fq* __restrict__ a, const Fq* __restrict__ b, const Fq* __restrict__ c,
                                   const Fq* __restrict__ d, size_t n, Fq* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (op == 0) {
    out[i] = Fq::mul_sub_mul(a[i], b[i], c[i], d[i]);
  } else if (op >= 2) {
    // the lazy-residue operations of the accumulate kernel (field.hpp): operand k enters as x + p when bit k of op - 2 is
    // set; every result must stay below 2p (all-ones is written otherwise) and is compared in canonical form
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Fq::LAZY_OK) {
      auto lift = [&](const Fq& x, int k) {
        if (!(((op - 2) >> k) & 1)) return x;
        Fq r;
        unsigned cy = 0;
        for (int j = 0; j < Fq::N; j++) r.v[j] = __builtin_addc(x.v[j], Fq::Params::MOD[j], cy, &cy);
        return r;
      };
      auto checked = [&](const Fq& x) {
        const Fq w = Fq::wrap2p(x);                  // unchanged when x < 2p
        bool same = true;
        for (int j = 0; j < Fq::N; j++) same = same && w.v[j] == x.v[j];
        Fq r = x.canon();
        if (!same)
          for (int j = 0; j < Fq::N; j++) r.v[j] = 0xffffffffu;
        return r;
      };
      const Fq A = lift(a[i], 0), B = lift(b[i], 1), C = lift(c[i], 2), D = lift(d[i], 3);
      out[5 * i] = checked(Fq::mul_lazy(A, B));
      out[5 * i + 1] = checked(Fq::sub_lazy(A, B));
      out[5 * i + 2] = checked(Fq::dbl_lazy(A));
      out[5 * i + 3] = checked(Fq::mul_sub_mul_lazy(A, B, C, D));
      Fq z = Fq::zero();
      z.v[0] = (Fq::sub_lazy(A, C).is_zero_lazy() ? 1u : 0u) | (A.is_zero_lazy() ? 2u : 0u);
      out[5 * i + 4] = z;
    }
#endif
  } else {
    using F2 = Fp2T<typename Fq::Params, INL>;
    const F2 z = F2{a[i], b[i]} * F2{c[i], d[i]};
    out[2 * i] = z.c0;
    out[2 * i + 1] = z.c1;
  }
}
#endif

// Out-of-line forms for cold code (bucket reduction, finalize, host folding).
template <class Fld>
ZK_HD_NOINLINE XYZZ<Fld> xyzz_add_ni(const XYZZ<Fld>& a, const XYZZ<Fld>& b) {
  return xyzz_add(a, b);
}
template <class Fld>
ZK_HD_NOINLINE XYZZ<Fld> xyzz_dbl_ni(const XYZZ<Fld>& p) {
  return xyzz_dbl(p);
}

// XYZZ -> Jacobian without inversion: Z' = ZZ*ZZZ (= Z^5), X' = X*ZZ*ZZZ^2, Y' = Y*ZZ^3*ZZZ^2.
template <class Fld>
ZK_HD Jacobian<Fld> xyzz_to_jacobian(const XYZZ<Fld>& p) {
  if (p.is_identity()) return {Fld::one(), Fld::one(), Fld::zero()};
  Fld z2 = p.ZZZ.sqr();
  Fld zz2 = p.ZZ.sqr();
  return {p.X * p.ZZ * z2, p.Y * zz2 * p.ZZ * z2, p.ZZ * p.ZZZ};
}

template <class Fld>
ZK_HD XYZZ<Fld> jacobian_to_xyzz(const Jacobian<Fld>& p) {
  if (p.Z.is_zero()) return XYZZ<Fld>::identity();
  Fld zz = p.Z.sqr();
  return {p.X, p.Y, zz, zz * p.Z};
}

// Normalisation (one inversion: batched divsteps, field.hpp inverse_fast -- round 5 ran the Fermat ladder here, 380 / 570
// products per point for 8 / 12 limbs, more than the 32 mixed additions of a fixed-base multiplication).
template <class Fld>
ZK_HD Affine<Fld> xyzz_to_affine(const XYZZ<Fld>& p) {
  if (p.is_identity()) return {Fld::zero(), Fld::zero()};
  Fld zi = p.ZZZ.inverse_fast();     // 1/Z^3
  Fld zi2 = (zi * p.ZZ).sqr();       // (Z^2/Z^3)^2 = 1/Z^2
  return {p.X * zi2, p.Y * zi};
}

// k * p for a small non-negative integer k (double-and-add, MSB first).
template <class Fld>
ZK_HD XYZZ<Fld> xyzz_mul_small(const XYZZ<Fld>& p, uint64_t k) {
  XYZZ<Fld> r = XYZZ<Fld>::identity();
  for (int b = 63; b >= 0; b--) {
    r = xyzz_dbl_ni(r);
    if ((k >> b) & 1) r = xyzz_add_ni(r, p);
  }
  return r;
}

}  // namespace zk

