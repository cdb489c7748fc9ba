// Prime-field arithmetic in Montgomery form for gfx950 (and the host side of the same library).
//
// Layout: N little-endian 32-bit limbs, value*R mod p with R = 2^(32N) -- bit-identical to
// arkworks' in-memory `Fp<MontBackend, N/2>` ([u64; N/2] little-endian, R = 2^(64*N/2)), which is
// what crosses the C ABI.  All values are kept fully reduced (< p).
//
// Replaces, on this path, ark-ff's `Fp::{mul,add,sub,neg,inverse,pow}` (reached from
// secret-sharing/src/pss.rs, dist-primitives/src/dfft/mod.rs:159,196-206,222-233, dpp/mod.rs:55).
//
// The multiply is an operand-scanning CIOS written so that hipcc lowers every 32x32+64 step to one
// v_mad_u64_u32 (D.u64 = S0.u32*S1.u32 + S2.u64).  There is no MFMA use: nothing here is a dense
// contraction.
#pragma once
#include <stdint.h>
#include "params.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#define ZK_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define ZK_HD inline
#define ZK_D inline
#define ZK_HD_NOINLINE __attribute__((noinline))
#endif

// Fields of up to this many 32-bit limbs inline the Montgomery product at every call site; wider ones go through the
// out-of-line copy.  A translation unit whose kernels are dominated by 12-limb products (the BLS12 G1 Pippenger kernels)
// may raise it before including this header.
#ifndef ZK_MUL_INLINE_LIMBS
#define ZK_MUL_INLINE_LIMBS 8
#endif

namespace zk {

template <class P>
struct Fp {
  using Params = P;
  static constexpr int N = P::N;
  uint32_t v[N];

  ZK_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  ZK_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::R1[i];
    return r;
  }
  ZK_HD static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::R2[i];
    return r;
  }
  ZK_HD bool is_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= v[i];
    return acc == 0;
  }
  ZK_HD bool operator==(const Fp& o) const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= v[i] ^ o.v[i];
    return acc == 0;
  }
  ZK_HD bool operator!=(const Fp& o) const { return !(*this == o); }
  // limbs, read as a plain integer, are below the modulus
  ZK_HD bool is_canonical() const {
    uint32_t borrow = 0;
    for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)v[i] - P::MOD[i] - borrow;
      borrow = (uint32_t)(t >> 63);
    }
    return borrow != 0;
  }

  // The carry chains below are written with clang's __builtin_addc / __builtin_subc: on the device they become one
  // v_addc_co / v_subb_co per limb.  The portable form ((uint64_t)a - b - borrow, borrow = t >> 63) compiled to two
  // 64-bit adds, a shift and a move per limb -- 60 instructions for a conditional subtraction that needs 18; measured
  // with tools/mulbench.hip and on the NTT (one addition and one subtraction per multiplication).

  // r = a - p if a >= p (a < 2p assumed); `carry` is the bit above limb N-1 of a.
  ZK_HD static Fp reduce_once(const Fp& a, uint32_t carry = 0) {
    Fp d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d.v[i] = __builtin_subc(a.v[i], P::MOD[i], borrow, &borrow);
    // a >= p  <=>  carry set, or no borrow
    const bool ge = (carry | (borrow ^ 1u)) != 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = ge ? d.v[i] : a.v[i];
    return r;
  }

  ZK_HD friend Fp operator+(const Fp& a, const Fp& b) {
    Fp s;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    return reduce_once(s, c);
  }

  ZK_HD friend Fp operator-(const Fp& a, const Fp& b) {
    Fp d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], borrow, &borrow);
    // add p back if we borrowed
    const uint32_t mask = 0u - borrow;
    unsigned c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = __builtin_addc(d.v[i], P::MOD[i] & mask, c, &c);
    return r;
  }

  ZK_HD Fp neg() const { return zero() - *this; }
  ZK_HD Fp dbl() const { return *this + *this; }

  // Montgomery product a*b*R^-1 mod p (CIOS, operand scanning, one row of a per outer step).
  // 8-limb fields inline it at every call site; 12-limb fields (BLS12 base fields) always go through the
  // out-of-line copy, which keeps kernels that chain dozens of them at a size the compiler handles well.
  ZK_HD friend Fp operator*(const Fp& a, const Fp& b) {
    if constexpr (N > ZK_MUL_INLINE_LIMBS) return mul_ni(a, b);
    else return mul_inline(a, b);
  }
  ZK_HD static Fp mul_inline(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // host: same CIOS on 64-bit limbs with 128-bit products (the king-side folds and the dealer run here)
    constexpr int M = N / 2;
    static_assert(N % 2 == 0, "even limb count expected");
    uint64_t A[M], B[M], Pm[M], t[M + 2];
    for (int i = 0; i < M; i++) {
      A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
      B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
      Pm[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: x' = x (2 + p x)  (x = -p^-1 mod 2^32)
    uint64_t x = P::N0INV;
    x = x * (2 + Pm[0] * x);
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
      uint64_t c = 0;
      for (int j = 0; j < M; j++) {
        unsigned __int128 z = (unsigned __int128)A[i] * B[j] + t[j] + c;
        t[j] = (uint64_t)z;
        c = (uint64_t)(z >> 64);
      }
      unsigned __int128 z = (unsigned __int128)t[M] + c;
      t[M] = (uint64_t)z;
      t[M + 1] = (uint64_t)(z >> 64);
      uint64_t m = t[0] * x;
      z = (unsigned __int128)m * Pm[0] + t[0];
      c = (uint64_t)(z >> 64);
      for (int j = 1; j < M; j++) {
        z = (unsigned __int128)m * Pm[j] + t[j] + c;
        t[j - 1] = (uint64_t)z;
        c = (uint64_t)(z >> 64);
      }
      z = (unsigned __int128)t[M] + c;
      t[M - 1] = (uint64_t)z;
      t[M] = t[M + 1] + (uint64_t)(z >> 64);
    }
    Fp r;
    for (int i = 0; i < M; i++) {
      r.v[2 * i] = (uint32_t)t[i];
      r.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    return reduce_once(r, (uint32_t)t[M]);
#else
    return mul_fips(a, b);
#endif
  }

  // Reference device form (plain operand-scanning CIOS); kept for cross-checks (tools/mulbench.hip).
  ZK_HD static Fp mul_ref(const Fp& a, const Fp& b) {
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < N; j++) {
        uint64_t x = (uint64_t)a.v[i] * b.v[j] + t[j] + c;
        t[j] = (uint32_t)x;
        c = x >> 32;
      }
      uint64_t x = (uint64_t)t[N] + c;
      t[N] = (uint32_t)x;
      t[N + 1] = (uint32_t)(x >> 32);

      uint32_t m = t[0] * P::N0INV;
      x = (uint64_t)m * P::MOD[0] + t[0];
      c = x >> 32;
#pragma unroll
      for (int j = 1; j < N; j++) {
        x = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)x;
        c = x >> 32;
      }
      x = (uint64_t)t[N] + c;
      t[N - 1] = (uint32_t)x;
      t[N] = t[N + 1] + (uint32_t)(x >> 32);
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return reduce_once(r, t[N]);
  }

#if defined(__HIP_DEVICE_COMPILE__)
  // X = a*b + add as one v_mad_u64_u32; the carry out of bit 63 is returned as a lane mask (SGPR pair).
  static ZK_D uint64_t madc(uint32_t a, uint32_t b, uint64_t add, uint64_t* cy) {
    uint64_t x, c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(x), "=s"(c) : "v"(a), "v"(b), "v"(add));
    *cy = c;
    return x;
  }
  static ZK_D uint64_t madc_k(uint32_t a, uint32_t k, uint64_t add, uint64_t* cy) {   // k: wave-uniform constant
    uint64_t x, c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(x), "=s"(c) : "v"(a), "s"(k), "v"(add));
    *cy = c;
    return x;
  }
  // v + carry bit from a lane mask; the caller guarantees v <= 2^32 - 2
  static ZK_D uint32_t add_cy(uint32_t v, uint64_t cy) {
    uint32_t r;
    asm("v_addc_co_u32_e64 %0, vcc, %1, 0, %2" : "=v"(r) : "v"(v), "s"(cy) : "vcc");
    return r;
  }
#endif

  // Device multiply.  Per row of a: the even-indexed partial products take the limb PAIR (t[j+1]:t[j]) as the
  // 64-bit addend of v_mad_u64_u32 (carry-out kept as a lane mask), the odd-indexed ones are plain products
  // whose high word (<= 2^32-2) absorbs that carry bit; one add-with-carry chain recombines the two
  // interleaved rows.  The Montgomery reduction row has the same shape.  Compared with the plain CIOS this
  // removes the zero-extension moves and 64-bit adds the compiler otherwise emits (8 limbs: 128 v_mad_u64_u32
  // + ~240 32-bit adds instead of + ~440 other VALU instructions).
  ZK_HD static Fp mul_pairs(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(N % 2 == 0, "even limb count expected");
    uint32_t t[N + 1];
#pragma unroll
    for (int j = 0; j <= N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t X[N], cy[N / 2];
#pragma unroll
      for (int j = 0; j < N; j += 2) {
        uint64_t add = ((uint64_t)t[j + 1] << 32) | t[j];
        X[j] = madc(a.v[i], b.v[j], add, &cy[j / 2]);
        X[j + 1] = (uint64_t)a.v[i] * b.v[j + 1];
      }
      uint32_t c, u[N + 1];
      u[0] = (uint32_t)X[0];
      u[1] = __builtin_addc((uint32_t)(X[0] >> 32), (uint32_t)X[1], 0u, &c);
#pragma unroll
      for (int k = 2; k < N; k++) {
        uint32_t e = (k & 1) ? (uint32_t)(X[k - 1] >> 32) : (uint32_t)X[k];
        uint32_t o = (k & 1) ? (uint32_t)X[k] : add_cy((uint32_t)(X[k - 1] >> 32), cy[k / 2 - 1]);
        u[k] = __builtin_addc(e, o, c, &c);
      }
      u[N] = __builtin_addc(t[N], add_cy((uint32_t)(X[N - 1] >> 32), cy[N / 2 - 1]), c, &c);
      uint32_t top = c;
      uint32_t m = u[0] * P::N0INV;
      uint64_t Y[N], dy[N / 2];
#pragma unroll
      for (int j = 0; j < N; j += 2) {
        uint64_t add = ((uint64_t)u[j + 1] << 32) | u[j];
        Y[j] = madc_k(m, P::MOD[j], add, &dy[j / 2]);
        Y[j + 1] = (uint64_t)m * P::MOD[j + 1];
      }
      t[0] = __builtin_addc((uint32_t)(Y[0] >> 32), (uint32_t)Y[1], 0u, &c);
#pragma unroll
      for (int k = 2; k < N; k++) {
        uint32_t e = (k & 1) ? (uint32_t)(Y[k - 1] >> 32) : (uint32_t)Y[k];
        uint32_t o = (k & 1) ? (uint32_t)Y[k] : add_cy((uint32_t)(Y[k - 1] >> 32), dy[k / 2 - 1]);
        t[k - 1] = __builtin_addc(e, o, c, &c);
      }
      t[N - 1] = __builtin_addc(u[N], add_cy((uint32_t)(Y[N - 1] >> 32), dy[N / 2 - 1]), c, &c);
      t[N] = top + c;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return reduce_once(r, t[N]);
#else
    return mul_ref(a, b);
#endif
  }

  // THE device multiply.  Product-scanning (FIPS) Montgomery with one 96-bit column accumulator: each v_mad_u64_u32 adds
  // its product into the 64-bit pair and the carry-out bit (an SGPR lane mask) is counted in the third word -- one carry
  // instruction per mad (128 for 8 limbs) where the row-wise form below (mul_pairs, round 1's multiplier, kept for the
  // cross-check in tools/mulbench.hip) needs ~190 plus the moves that pack limb pairs, and no carry instruction feeds
  // the next one (gfx950 wants two wait states between a VALU that writes a carry and the VALU that reads it).
  // Measured (profiles/r02_mulbench.txt): 123.8 vs 98.1 G products/s on BN254 Fq (80 % of the v_mad_u64_u32 issue
  // bound), 59.6 vs 47.4 on the 12-limb BLS12-381 Fq, 0.76 vs 1.00 us per dependent product for a lone wave.
  ZK_HD static Fp mul_fips(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m[N], r[N];
    uint64_t acc = 0, cy;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc = madc(a.v[i], b.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
#pragma unroll
      for (int i = 0; i < k; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      m[k] = (uint32_t)acc * P::N0INV;
      acc = madc_k(m[k], P::MOD[0], acc, &cy);
      acc2 = add_cy(acc2, cy);
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc = madc(a.v[i], b.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      r[k - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
    r[N - 1] = (uint32_t)acc;
    Fp o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = r[i];
    return reduce_once(o, (uint32_t)(acc >> 32));
#else
    return mul_ref(a, b);
#endif
  }

  static constexpr bool LAZY_OK = (P::MOD[N - 1] >> 30) == 0;
#if defined(__HIP_DEVICE_COMPILE__)
  // ---- LAZY residues in [0, 2p) (round 4; moduli with two spare bits, 4p < R: BN254 Fq / Fr, the BLS12 base fields).
  // The Montgomery product of two such values, (a b + m p) / R < 4 p^2 / R + p < 2p, is again below 2p WITHOUT the final
  // conditional subtraction; subtraction adds 2p back instead of p, doubling wraps at 2p.  The accumulate kernel of the
  // Pippenger MSM keeps its running sum in this form (eight of the ten products of a mixed addition lose their
  // compare-and-select tail) and stores canonical values (canon()).
  struct Mod2 {
    uint32_t v[N];
  };
  static constexpr Mod2 mod2() {
    Mod2 r{};
    uint32_t c = 0;
    for (int i = 0; i < N; i++) {
      r.v[i] = (P::MOD[i] << 1) | c;
      c = P::MOD[i] >> 31;
    }
    return r;
  }
  static ZK_D Fp mul_lazy(const Fp& a, const Fp& b) {
    uint32_t m[N], r[N];
    uint64_t acc = 0, cy;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc = madc(a.v[i], b.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
#pragma unroll
      for (int i = 0; i < k; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      m[k] = (uint32_t)acc * P::N0INV;
      acc = madc_k(m[k], P::MOD[0], acc, &cy);
      acc2 = add_cy(acc2, cy);
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc = madc(a.v[i], b.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      r[k - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
    r[N - 1] = (uint32_t)acc;               // (the word above it is zero: the value is below 2p < 2^(32 N))
    Fp o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = r[i];
    return o;
  }
  // a - b for a, b in [0, 2p): in [0, 2p)
  static ZK_D Fp sub_lazy(const Fp& a, const Fp& b) {
    constexpr Mod2 M2 = mod2();
    Fp d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], borrow, &borrow);
    const uint32_t mask = 0u - borrow;
    unsigned c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = __builtin_addc(d.v[i], M2.v[i] & mask, c, &c);
    return r;
  }
  // x - 2p when x >= 2p (x < 4p)
  static ZK_D Fp wrap2p(const Fp& x) {
    constexpr Mod2 M2 = mod2();
    Fp d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d.v[i] = __builtin_subc(x.v[i], M2.v[i], borrow, &borrow);
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = borrow ? x.v[i] : d.v[i];
    return r;
  }
  static ZK_D Fp dbl_lazy(const Fp& a) {          // 2a mod 2p-range: a < 2p, 2a < 4p < 2^(32 N)
    Fp s;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s.v[i] = __builtin_addc(a.v[i], a.v[i], c, &c);
    return wrap2p(s);
  }
  ZK_D Fp canon() const { return reduce_once(*this, 0); }          // [0, 2p) -> [0, p)
  ZK_D bool is_zero_lazy() const {                                  // 0 or p
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      z |= v[i];
      e |= v[i] ^ P::MOD[i];
    }
    return z == 0 || e == 0;
  }
  // a*b - c*d on lazy residues with one reduction: a b + (2p - c) d < 8 p^2, reduced to below 2p + p, wrapped into [0, 2p)
  static ZK_D Fp mul_sub_mul_lazy(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
    constexpr Mod2 M2 = mod2();
    uint32_t nc[N];
    {
      unsigned bw = 0;
#pragma unroll
      for (int i = 0; i < N; i++) nc[i] = __builtin_subc(M2.v[i], c.v[i], bw, &bw);
    }
    uint32_t m[N], r[N];
    uint64_t acc = 0, cy;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc = madc(a.v[i], b.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
        acc = madc(nc[i], d.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
#pragma unroll
      for (int i = 0; i < k; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      m[k] = (uint32_t)acc * P::N0INV;
      acc = madc_k(m[k], P::MOD[0], acc, &cy);
      acc2 = add_cy(acc2, cy);
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc = madc(a.v[i], b.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
        acc = madc(nc[i], d.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      r[k - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
    r[N - 1] = (uint32_t)acc;               // below 3p < 2^(32 N)
    Fp o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = r[i];
    return wrap2p(o);
  }
#endif

  // a*b - c*d (all Montgomery) in ONE product-scanning pass: a*b + (p - c)*d < 2 p^2 < p R is reduced once -- 3 N^2 + N
  // multiply instructions instead of 2 (2 N^2 + N): 200 vs 272 for N = 8.  Same field element as a*b - c*d.  Used for
  // Y3 = R (Q - X3) - Y1 PPP of the mixed addition (msm.hpp).  The modulus must leave two spare bits.
  ZK_HD static Fp mul_sub_mul(const Fp& a, const Fp& b, const Fp& c, const Fp& d) { return mul_pm_mul(a, b, c, d, false); }
  // a*b + c*d (plus) or a*b - c*d, chosen per lane at run time: the two component products of a quadratic-extension
  // multiplication held one component per lane (quad.hpp s2_mul) differ only in this sign
  ZK_HD static Fp mul_pm_mul(const Fp& a, const Fp& b, const Fp& c, const Fp& d, bool plus) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_MUL_SUB_MUL)
    if constexpr (N <= ZK_MUL_INLINE_LIMBS && (P::MOD[N - 1] >> 30) == 0) {
      uint32_t nc[N];                               // minus: p - c, in (0, p]; plus: c itself
      {
        unsigned bw = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
          const uint32_t t = __builtin_subc(P::MOD[i], c.v[i], bw, &bw);
          nc[i] = plus ? c.v[i] : t;
        }
      }
      uint32_t m[N], r[N];
      uint64_t acc = 0, cy;
      uint32_t acc2 = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
          acc = madc(a.v[i], b.v[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
          acc = madc(nc[i], d.v[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
          acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
        m[k] = (uint32_t)acc * P::N0INV;
        acc = madc_k(m[k], P::MOD[0], acc, &cy);
        acc2 = add_cy(acc2, cy);
        acc = (acc >> 32) | ((uint64_t)acc2 << 32);
        acc2 = 0;
      }
#pragma unroll
      for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
          acc = madc(a.v[i], b.v[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
          acc = madc(nc[i], d.v[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
          acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
        r[k - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)acc2 << 32);
        acc2 = 0;
      }
      r[N - 1] = (uint32_t)acc;
      Fp o;
#pragma unroll
      for (int i = 0; i < N; i++) o.v[i] = r[i];
      return reduce_once(o, (uint32_t)(acc >> 32));
    }
#endif
    return plus ? a * b + c * d : a * b - c * d;
  }

#if defined(__HIP_DEVICE_COMPILE__)
  // ---- unreduced (double-width) arithmetic for the lazy-reduction product of the quadratic extension (Fp2T::mul_lazy)
  // T = a * b as 2N limbs, no reduction (product scanning; N^2 multiply instructions); a, b < 2^(32 N)
  static ZK_D void mul_wide(const uint32_t* a, const uint32_t* b, uint32_t* T) {
    uint64_t acc = 0, cy;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); i++) {
        acc = madc(a[i], b[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      T[k] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
    T[2 * N - 1] = (uint32_t)acc;
  }
  // Montgomery reduction of a 2N-limb value T < p * 2^(32 N): T / R mod p, fully reduced (N (N + 1) / 2 + N (N - 1) / 2
  // multiply instructions by the modulus + N for the quotient digits)
  static ZK_D Fp redc_wide(const uint32_t* T) {
    uint32_t m[N], r[N];
    uint64_t acc = 0, cy;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
      acc += T[k];                                   // acc < 2^40 here: no carry out
#pragma unroll
      for (int i = 0; i < k; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      m[k] = (uint32_t)acc * P::N0INV;
      acc = madc_k(m[k], P::MOD[0], acc, &cy);
      acc2 = add_cy(acc2, cy);
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
      acc += T[k];
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      r[k - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
    acc += T[2 * N - 1];
    r[N - 1] = (uint32_t)acc;
    Fp o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = r[i];
    return reduce_once(o, (uint32_t)(acc >> 32));
  }
#endif

  // ---- dot product with ONE Montgomery reduction (round 5): sum_{s < NT} x_s * k_s for canonical x_s (VGPRs) and
  // wave-uniform canonical k_s (SGPRs: the rows of the PSS unpack matrices).  Product scanning over all NT products at once:
  // NT N^2 + N^2 + N multiply instructions instead of NT (2 N^2 + N) -- 584 instead of 1088 for the eight-party unpack2 of
  // an 8-limb field.  The accumulated value T < NT p^2 reduces to V = (T + m p) / R < (NT p / R + 1) p <= (NT / 2 + 1) p
  // (every modulus here has its top bit clear), which a ladder of conditional subtractions of 2^j p brings below p:
  // the same field element as the sum of the NT reduced products.
  struct ModSh {
    uint32_t v[N + 1];
  };
  static constexpr ModSh mod_shl(int sh) {
    ModSh r{};
    uint64_t c = 0;
    for (int i = 0; i < N; i++) {
      const uint64_t t = ((uint64_t)P::MOD[i] << sh) | c;
      r.v[i] = (uint32_t)t;
      c = t >> 32;
    }
    r.v[N] = (uint32_t)c;
    return r;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  template <bool KS>
  static ZK_D uint64_t mad_sel(uint32_t a, uint32_t b, uint64_t add, uint64_t* cy) {
    if constexpr (KS) return madc_k(a, b, add, cy);
    else return madc(a, b, add, cy);
  }
#endif
  template <int NT>
  ZK_HD static Fp dot_k(const Fp* const* x /* [NT] */, const Fp* __restrict__ k /* [NT], wave-uniform */) {
    return dot_impl<NT, true>(x, k);
  }
  // the same with per-lane coefficients (VGPR operands): the lane-cooperative king kernels, where lane p applies row p of
  // the pack matrix
  template <int NT>
  ZK_HD static Fp dot_v(const Fp* const* x /* [NT] */, const Fp* k /* [NT] */) {
    return dot_impl<NT, false>(x, k);
  }
  template <int NT, bool KS>
  ZK_HD static Fp dot_impl(const Fp* const* x, const Fp* k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m[N], r[N + 1];
    uint64_t acc = 0, cy;
    uint32_t acc2 = 0;
#pragma unroll
    for (int c = 0; c < N; c++) {
#pragma unroll
      for (int i = 0; i <= c; i++) {
#pragma unroll
        for (int s = 0; s < NT; s++) {
          acc = mad_sel<KS>(x[s]->v[i], k[s].v[c - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
      }
#pragma unroll
      for (int i = 0; i < c; i++) {
        acc = madc_k(m[i], P::MOD[c - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      m[c] = (uint32_t)acc * P::N0INV;
      acc = madc_k(m[c], P::MOD[0], acc, &cy);
      acc2 = add_cy(acc2, cy);
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
#pragma unroll
    for (int c = N; c < 2 * N - 1; c++) {
#pragma unroll
      for (int i = c - N + 1; i < N; i++) {
#pragma unroll
        for (int s = 0; s < NT; s++) {
          acc = mad_sel<KS>(x[s]->v[i], k[s].v[c - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
        acc = madc_k(m[i], P::MOD[c - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      r[c - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
    r[N - 1] = (uint32_t)acc;
    r[N] = (uint32_t)(acc >> 32);
    // V < (NT / 2 + 1) p: subtract 2^j p while it fits, j = J-1 .. 0 with 2^J >= NT / 2 + 1
    constexpr int B = NT / 2 + 1;
    constexpr int J = B <= 2 ? 1 : B <= 4 ? 2 : B <= 8 ? 3 : B <= 16 ? 4 : 5;
    // (odd NT: the accumulated value can reach (NT + 1) / 2 + 1/2 times p, above B p with the integer division -- ADVICE r5;
    // only even term counts are instantiated, and only they are accepted)
    static_assert(NT % 2 == 0 && NT <= 32 && (P::MOD[N - 1] >> 31) == 0, "dot_k: bound");
#pragma unroll
    for (int j = J - 1; j >= 0; j--) {
      const ModSh M = mod_shl(j);
      uint32_t d[N + 1];
      unsigned bw = 0;
#pragma unroll
      for (int i = 0; i <= N; i++) d[i] = __builtin_subc(r[i], M.v[i], bw, &bw);
#pragma unroll
      for (int i = 0; i <= N; i++) r[i] = bw ? r[i] : d[i];
    }
    Fp o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = r[i];
    return o;
#else
    Fp acc = zero();                                   // host side of the same name: the plain sum
    for (int s = 0; s < NT; s++) acc = acc + *x[s] * k[s];
    return acc;
#endif
  }

  // Dedicated product-scanning SQUARING, measured and NOT adopted (tools/mulbench.hip variant 3,
  // profiles/r03_mulbench_sqr.txt): the off-diagonal products a_i a_j (i < j) of a column are accumulated once and
  // doubled, the diagonal and the Montgomery terms follow -- 36 + 64 + 8 multiply instructions instead of 128 + 8.  The
  // doubling of the 96-bit column accumulator and the re-insertion of the carried column cost ~9 full-rate instructions
  // per column, about what the 28 saved quarter-rate multiplies (and their carry instructions) are worth.
  ZK_HD static Fp sqr_fips(const Fp& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m[N], r[N];
    uint64_t carry = 0;                // column value carried in from the column below (not to be doubled)
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
      uint64_t acc = 0, cy;
      uint32_t acc2 = 0;
      // off-diagonal products of this column, once
#pragma unroll
      for (int i = (k < N ? 0 : k - N + 1); 2 * i < k; i++) {
        acc = madc(a.v[i], a.v[k - i], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      // ... doubled
      acc2 = (acc2 << 1) | (uint32_t)(acc >> 63);
      acc <<= 1;
      // + the carried column
      {
        const uint64_t t = acc + carry;
        acc2 += t < acc ? 1u : 0u;
        acc = t;
      }
      if (!(k & 1)) {
        acc = madc(a.v[k / 2], a.v[k / 2], acc, &cy);
        acc2 = add_cy(acc2, cy);
      }
      if (k < N) {
#pragma unroll
        for (int i = 0; i < k; i++) {
          acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
        m[k] = (uint32_t)acc * P::N0INV;
        acc = madc_k(m[k], P::MOD[0], acc, &cy);
        acc2 = add_cy(acc2, cy);
      } else {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
          acc = madc_k(m[i], P::MOD[k - i], acc, &cy);
          acc2 = add_cy(acc2, cy);
        }
        r[k - N] = (uint32_t)acc;
      }
      carry = (acc >> 32) | ((uint64_t)acc2 << 32);
    }
    r[N - 1] = (uint32_t)carry;
    Fp o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = r[i];
    return reduce_once(o, (uint32_t)(carry >> 32));
#else
    return mul_ref(a, a);
#endif
  }

  // Out-of-line multiply for cold code (G2 tower, bucket reduction, l > 2 kernels): one copy per field
  // instead of one per call site keeps code size and compile time bounded.
  static ZK_HD_NOINLINE Fp mul_ni(const Fp& a, const Fp& b) { return mul_inline(a, b); }

  ZK_HD Fp sqr() const { return *this * *this; }

  ZK_HD Fp to_mont() const { return *this * r2(); }      // canonical integer -> Montgomery
  ZK_HD Fp from_mont() const {                            // Montgomery -> canonical integer
    Fp o = zero();
    o.v[0] = 1;
    return *this * o;
  }

  // this^e for a little-endian exponent of `nl` 32-bit limbs (square-and-multiply, MSB first).
  ZK_HD Fp pow(const uint32_t* e, int nl) const {
    Fp r = one();
    bool started = false;
    for (int i = nl - 1; i >= 0; i--) {
      for (int b = 31; b >= 0; b--) {
        if (started) r = mul_ni(r, r);
        if ((e[i] >> b) & 1u) {
          r = started ? mul_ni(r, *this) : *this;
          started = true;
        }
      }
    }
    return r;
  }
  ZK_HD Fp pow_u64(uint64_t e) const {
    uint32_t ee[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
    return pow(ee, 2);
  }
  // Fermat inverse (this != 0): this^(p-2).
  ZK_HD Fp inverse() const {
    uint32_t e[N];
    uint32_t borrow = 2;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)P::MOD[i] - borrow;
      e[i] = (uint32_t)t;
      borrow = (uint32_t)(t >> 63);
    }
    return pow(e, N);
  }
  // a >>= z, a <<= z for 1 <= z <= 31
  ZK_HD static void limbs_shr(uint32_t* a, int z) {
#pragma unroll
    for (int i = 0; i < N - 1; i++) a[i] = (a[i] >> z) | (a[i + 1] << (32 - z));
    a[N - 1] >>= z;
  }
  ZK_HD static void limbs_shl(uint32_t* a, int z) {
#pragma unroll
    for (int i = N - 1; i > 0; i--) a[i] = (a[i] << z) | (a[i - 1] >> (32 - z));
    a[0] <<= z;
  }
  ZK_HD static void limbs_add(uint32_t* a, const uint32_t* b) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = __builtin_addc(a[i], b[i], c, &c);
  }
  ZK_HD static bool limbs_nonzero(const uint32_t* a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= a[i];
    return acc != 0;
  }
  // trailing zero bits of a non-zero multi-limb value, capped at 31 (the caller comes back for the rest)
  ZK_HD static int limbs_ctz31(const uint32_t* a) {
    const int z = a[0] ? __builtin_ctz(a[0]) : 31;
    return z > 31 ? 31 : z;
  }
  // Inverse by the binary extended Euclid (Kaliski's "almost Montgomery inverse"), for ONE lane that a whole kernel waits
  // on (d_pp's single inversion, csrc/dpp.hpp): shifts, one subtraction and one addition on N limbs per step, none of
  // them a multiply -- about a tenth of the instructions of the Fermat ladder above.  The textbook loop halves once per
  // iteration (u even: u /= 2, s *= 2; v even: v /= 2, r *= 2; else the larger becomes (larger - smaller) / 2 with
  // r += s, s *= 2 or s += r, r *= 2; k counts the halvings); here every run of halvings is taken in one shift by the
  // number of trailing zero bits, the same sequence of states in about half the iterations.  Phase 1 leaves
  // x = a^-1 2^k (mod p), BITS <= k <= 2 BITS, for the integer a held in the limbs; the value wanted is the Montgomery form
  // of (a / R)^-1 = a^-1 R^2, i.e. x 2^(2 * 32 N - k): one product with the Montgomery form of that power of two.
  // this != 0.  Same field element as inverse().
  ZK_HD Fp inverse_gcd() const {
    uint32_t u[N], w[N], r[N], s[N];
#pragma unroll
    for (int i = 0; i < N; i++) u[i] = P::MOD[i], w[i] = v[i], r[i] = 0, s[i] = 0;
    s[0] = 1;
    int k = 0;
    for (;;) {                                   // u is odd here (p is odd; u only changes below, ending odd)
      if (!(w[0] & 1)) {                         // w even and non-zero
        const int z = limbs_ctz31(w);
        limbs_shr(w, z);
        limbs_shl(r, z);
        k += z;
        continue;
      }
      uint32_t d[N];
      unsigned bw = 0;
#pragma unroll
      for (int i = 0; i < N; i++) d[i] = __builtin_subc(u[i], w[i], bw, &bw);
      if (!bw && limbs_nonzero(d)) {             // u > w: u = u - w (even), r += s, then the halvings of u
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = d[i];
        limbs_add(r, s);
        do {
          const int z = limbs_ctz31(u);
          limbs_shr(u, z);
          limbs_shl(s, z);
          k += z;
        } while (!(u[0] & 1));
      } else {                                   // w >= u: w = w - u, s += r
        unsigned c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) w[i] = __builtin_subc(0u, d[i], c, &c);     // -(u - w), exact in N limbs
        limbs_add(s, r);
        if (!limbs_nonzero(w)) {                 // u == w (== 1): the textbook loop's last step, r *= 2
          limbs_shl(r, 1);
          k += 1;
          break;
        }
      }
    }
    // r < 2p: x = p - (r mod p)
    Fp x;
    {
      uint32_t d[N];
      unsigned bw = 0;
#pragma unroll
      for (int i = 0; i < N; i++) d[i] = __builtin_subc(r[i], P::MOD[i], bw, &bw);
      if (!bw) {
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = d[i];
      }
      bw = 0;
#pragma unroll
      for (int i = 0; i < N; i++) x.v[i] = __builtin_subc(P::MOD[i], r[i], bw, &bw);
    }
    // Montgomery form of 2^e, e = 64 N - k in [2, 32 N + 2]
    Fp two = one() + one();
    return mul_ni(x, two.pow_u64((uint64_t)(64 * N - k)));
  }
  // ---- inverse by BATCHED DIVSTEPS (Bernstein-Yang "safegcd", the half-delta variant with zeta = -(delta + 1/2); the
  // formulation libsecp256k1's modinv32 made standard), for the one lane a whole kernel waits on (d_pp's carry kernel).
  // The binary Euclid above decides every step on the full numbers -- a subtraction, a comparison and two shifts over N
  // limbs per bit.  Here 30 steps at a time are decided on the LOW 32 bits of f and g alone (a divstep only ever looks
  // at the parity of g and the sign of zeta) and collected in a 2 x 2 matrix of 31-bit integers; the matrix is then applied
  // to the full f, g (exact division by 2^30) and to the cofactors d, e modulo p (division by 2^30 through p^-1 mod 2^30):
  // per 30 steps ~360 one-word operations + 8 K multiply-adds for K = ceil((32 N + 2) / 30) signed 30-bit limbs,
  // against 30 x ~100 for the binary Euclid.  At most ceil((45907 BITS + 26313) / 19929) steps (590 for 255 bits) are
  // needed; the loop leaves as soon as g = 0.  Invariants: d x = f, e x = g (mod p) for the input x; at the end f = +-1,
  // so x^-1 = +-d.  The value wanted is the Montgomery form of (x / R)^-1 = x^-1 R^2: two products with R^2.
  // this != 0.  Same field element as inverse() (tests/test_native_field.py: six fields against the Fermat ladder).
  static constexpr int K30 = (32 * N + 2 + 29) / 30;
  static constexpr uint32_t modinv30() {             // p^-1 mod 2^30 (Newton, p odd)
    uint32_t x = P::MOD[0];                          // correct to 3 bits
    for (int i = 0; i < 5; i++) x *= 2u - P::MOD[0] * x;
    return x & 0x3fffffffu;
  }
  ZK_HD static void to_s30(const uint32_t* a, int32_t* r) {        // N x 32 bits -> K30 x 30 bits, non-negative
#pragma unroll
    for (int i = 0; i < K30; i++) {
      const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
      uint64_t lo = w < N ? a[w] : 0u, hi = w + 1 < N ? a[w + 1] : 0u;
      r[i] = (int32_t)((((hi << 32) | lo) >> sh) & 0x3fffffffu);
    }
  }
  ZK_HD Fp inverse_safegcd() const {
    constexpr int32_t M30 = 0x3fffffff;
    int32_t f[K30], g[K30], d[K30], e[K30], m[K30];
    to_s30(P::MOD, m);
    to_s30(P::MOD, f);
    to_s30(v, g);
#pragma unroll
    for (int i = 0; i < K30; i++) d[i] = 0, e[i] = 0;
    e[0] = 1;
    constexpr uint32_t minv = modinv30();
    constexpr int MAX_STEPS = (int)((45907ll * P::BITS + 26313) / 19929) + 1;
    constexpr int MAX_IT = (MAX_STEPS + 29) / 30;
    int32_t zeta = -1;
    for (int it = 0; it < MAX_IT; it++) {
      // ---- 30 divsteps on the low words: [f', g'] = t [f, g] / 2^30, t = [[u, v], [q, r]]
      uint32_t u = 1, vv = 0, q = 0, r = 1;
      uint32_t fl = (uint32_t)f[0] | ((uint32_t)f[1] << 30), gl = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
#pragma unroll 1
      for (int i = 0; i < 30; i++) {
        uint32_t mask1 = (uint32_t)(zeta >> 31);                  // zeta < 0
        const uint32_t mask2 = 0u - (gl & 1u);                    // g odd
        const uint32_t x = (fl ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (vv ^ mask1) - mask1;
        gl += x & mask2;
        q += y & mask2;
        r += z & mask2;
        mask1 &= mask2;
        zeta = (int32_t)((uint32_t)zeta ^ mask1) - 1;
        fl += gl & mask1;
        u += q & mask1;
        vv += r & mask1;
        gl >>= 1;
        u <<= 1;
        vv <<= 1;
      }
      const int64_t tu = (int32_t)u, tv = (int32_t)vv, tq = (int32_t)q, tr = (int32_t)r;
      // ---- d, e <- t [d, e] / 2^30 mod p   (d, e in (-2p, p))
      {
        const int32_t sd = d[K30 - 1] >> 31, se = e[K30 - 1] >> 31;
        int32_t md = ((int32_t)tu & sd) + ((int32_t)tv & se), me = ((int32_t)tq & sd) + ((int32_t)tr & se);
        int64_t cd = tu * d[0] + tv * e[0], ce = tq * d[0] + tr * e[0];
        md -= (int32_t)((minv * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((minv * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
        cd += (int64_t)m[0] * md;
        ce += (int64_t)m[0] * me;
        cd >>= 30;
        ce >>= 30;
#pragma unroll
        for (int i = 1; i < K30; i++) {
          cd += tu * d[i] + tv * e[i] + (int64_t)m[i] * md;
          ce += tq * d[i] + tr * e[i] + (int64_t)m[i] * me;
          d[i - 1] = (int32_t)cd & M30;
          e[i - 1] = (int32_t)ce & M30;
          cd >>= 30;
          ce >>= 30;
        }
        d[K30 - 1] = (int32_t)cd;
        e[K30 - 1] = (int32_t)ce;
      }
      // ---- f, g <- t [f, g] / 2^30 (exact)
      {
        int64_t cf = tu * f[0] + tv * g[0], cg = tq * f[0] + tr * g[0];
        cf >>= 30;
        cg >>= 30;
#pragma unroll
        for (int i = 1; i < K30; i++) {
          cf += tu * f[i] + tv * g[i];
          cg += tq * f[i] + tr * g[i];
          f[i - 1] = (int32_t)cf & M30;
          g[i - 1] = (int32_t)cg & M30;
          cf >>= 30;
          cg >>= 30;
        }
        f[K30 - 1] = (int32_t)cf;
        g[K30 - 1] = (int32_t)cg;
      }
      int32_t nz = 0;
#pragma unroll
      for (int i = 0; i < K30; i++) nz |= g[i];
      if (nz == 0) break;
    }
    // ---- x^-1 = sign(f) d, brought into [0, p)
    {
      const int32_t neg_d = d[K30 - 1] >> 31;                     // d in (-2p, p): add p when negative
#pragma unroll
      for (int i = 0; i < K30; i++) d[i] += m[i] & neg_d;
      const int32_t sf = f[K30 - 1] >> 31;                        // f = -1: negate
#pragma unroll
      for (int i = 0; i < K30; i++) d[i] = (d[i] ^ sf) - sf;
#pragma unroll
      for (int i = 0; i < K30 - 1; i++) {                         // carries
        d[i + 1] += d[i] >> 30;
        d[i] &= M30;
      }
      const int32_t neg2 = d[K30 - 1] >> 31;                      // now in (-p, p)
#pragma unroll
      for (int i = 0; i < K30; i++) d[i] += m[i] & neg2;
#pragma unroll
      for (int i = 0; i < K30 - 1; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= M30;
      }
    }
    Fp y;
#pragma unroll
    for (int w = 0; w < N; w++) {                                 // K30 x 30 bits -> N x 32 bits
      const int bit = 32 * w, i = bit / 30, sh = bit % 30;
      uint64_t acc = (uint64_t)(uint32_t)d[i] >> sh;
      if (i + 1 < K30) acc |= (uint64_t)(uint32_t)d[i + 1] << (30 - sh);
      if (i + 2 < K30) acc |= (uint64_t)(uint32_t)d[i + 2] << (60 - sh);
      y.v[w] = (uint32_t)acc;
    }
    return mul_ni(mul_ni(y, r2()), r2());
  }
  // The inversion of code that inverts per lane or per point (affine normalisation, ec.hpp xyzz_to_affine): batched
  // divsteps -- about 35 (8 limbs) / 25 (12 limbs) product equivalents of issue against the Fermat ladder's 380 / 570.
  ZK_HD Fp inverse_fast() const { return inverse_safegcd(); }
  ZK_HD static Fp from_u64(uint64_t x) {
    Fp r = zero();
    r.v[0] = (uint32_t)x;
    r.v[1] = (uint32_t)(x >> 32);
    return r.to_mont();
  }
  ZK_HD static Fp from_limbs(const uint32_t* p) {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = p[i];
    return r;
  }
};

// Quadratic extension Fq2 = Fq[u]/(u^2 - NONRES) for G2 arithmetic; NONRES = -1 for BN254 and BLS12-381.
// INL selects how the tower reaches the base-field multiply: the MSM hot kernels of 8-limb curves use the inlined
// form (Fp2I; measured 126 vs 107 proofs/s on the SHA-256 circuit), everything else the out-of-line copy, which
// keeps dealer / host code small (inlining it everywhere doubles the build time).  Both have the same layout.
template <class P, bool INL>
struct Fp2T;
template <class P>
using Fp2 = Fp2T<P, false>;
template <class P>
using Fp2I = Fp2T<P, true>;

template <class P, bool INL>
struct Fp2T {
  using Fp2 = Fp2T;
  using B = Fp<P>;
  B c0, c1;
  static ZK_HD B bmul(const B& a, const B& b) {
    if constexpr (INL) return a * b;
    else return B::mul_ni(a, b);
  }
  ZK_HD static Fp2 zero() { return {B::zero(), B::zero()}; }
  ZK_HD static Fp2 one() { return {B::one(), B::zero()}; }
  ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZK_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
  ZK_HD bool operator!=(const Fp2& o) const { return !(*this == o); }
  ZK_HD friend Fp2 operator+(const Fp2& a, const Fp2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  ZK_HD friend Fp2 operator-(const Fp2& a, const Fp2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  ZK_HD Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  ZK_HD Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
#if defined(__HIP_DEVICE_COMPILE__)
  // p^2 as 2N limbs (compile-time), the offset that keeps a0 b0 - a1 b1 non-negative before its reduction
  struct ModSq {
    uint32_t v[2 * B::N];
  };
  static constexpr ModSq mod_sq() {
    ModSq r{};
    for (int i = 0; i < B::N; i++) {
      uint64_t c = 0;
      for (int j = 0; j < B::N; j++) {
        uint64_t t = (uint64_t)r.v[i + j] + (uint64_t)P::MOD[i] * P::MOD[j] + c;
        r.v[i + j] = (uint32_t)t;
        c = t >> 32;
      }
      r.v[i + B::N] = (uint32_t)c;
    }
    return r;
  }
  // Karatsuba with LAZY reduction (u^2 = -1): three unreduced 2N-limb products and TWO Montgomery reductions instead of
  // three full products -- 3 N^2 + 2 (N^2 + N) multiply instructions instead of 3 (2 N^2 + N): 336 vs 408 for N = 8.
  //   c0 = redc(a0 b0 - a1 b1 + p^2),   c1 = redc((a0 + a1)(b0 + b1) - a0 b0 - a1 b1)
  // Bounds (p < 2^(32 N - 2)): the sums a0 + a1, b0 + b1 < 2 p fit N limbs; both reduction inputs are in [0, 2 p^2) and
  // 2 p^2 < p R, which is what redc_wide needs; its output is fully reduced.  Same field elements as the three-product
  // form, hence bit-identical results.
  static ZK_D Fp2 mul_lazy(const Fp2& a, const Fp2& b) {
    constexpr int N = B::N;
    constexpr ModSq P2 = mod_sq();
    uint32_t T0[2 * N], T1[2 * N];
    B::mul_wide(a.c0.v, b.c0.v, T0);
    B::mul_wide(a.c1.v, b.c1.v, T1);
    // in place, three interleaved carry chains: T0 <- a0 b0 + p^2 - a1 b1 (c0 before reduction), T1 <- a0 b0 + a1 b1
    {
      unsigned c1 = 0, c2 = 0, bw = 0;
#pragma unroll
      for (int i = 0; i < 2 * N; i++) {
        const uint32_t x = T0[i], y = T1[i];
        T1[i] = __builtin_addc(x, y, c1, &c1);
        T0[i] = __builtin_subc(__builtin_addc(x, P2.v[i], c2, &c2), y, bw, &bw);
      }
    }
    Fp2 r;
    r.c0 = B::redc_wide(T0);
    uint32_t sa[N], sb[N];
    {
      unsigned c = 0;
#pragma unroll
      for (int i = 0; i < N; i++) sa[i] = __builtin_addc(a.c0.v[i], a.c1.v[i], c, &c);
      c = 0;
#pragma unroll
      for (int i = 0; i < N; i++) sb[i] = __builtin_addc(b.c0.v[i], b.c1.v[i], c, &c);
    }
    B::mul_wide(sa, sb, T0);
    {
      unsigned bw = 0;
#pragma unroll
      for (int i = 0; i < 2 * N; i++) T0[i] = __builtin_subc(T0[i], T1[i], bw, &bw);
    }
    r.c1 = B::redc_wide(T0);
    return r;
  }
#endif
  // (a0 + a1 u)(b0 + b1 u) with u^2 = -1, Karatsuba: 3 base multiplications.
  ZK_HD friend Fp2 operator*(const Fp2& a, const Fp2& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_FQ2_NO_LAZY)
    // the modulus must leave two spare bits
    // the MSM hot kernels of 8-limb curves (Fp2I).  The 12-limb form, as one out-of-line copy for the BLS12-381 G2
    // kernels, was measured and left out: C5 (2^24) 1.587-1.593 s per proof against 1.570-1.579 s.
    if constexpr (INL && B::N == 8 && (P::MOD[B::N - 1] >> 30) == 0) return mul_lazy(a, b);
#endif
    B v0 = bmul(a.c0, b.c0);
    B v1 = bmul(a.c1, b.c1);
    B s = bmul(a.c0 + a.c1, b.c0 + b.c1);
    return {v0 - v1, s - v0 - v1};
  }
  // (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u : 2 base multiplications.
  ZK_HD Fp2 sqr() const {
    B t = bmul(c0, c1);
    return {bmul(c0 + c1, c0 - c1), t.dbl()};
  }
  ZK_HD Fp2 inverse() const {
    B n = (B::mul_ni(c0, c0) + B::mul_ni(c1, c1)).inverse();
    return {B::mul_ni(c0, n), B::mul_ni(c1, n).neg()};
  }
  ZK_HD Fp2 inverse_fast() const {
    B n = (B::mul_ni(c0, c0) + B::mul_ni(c1, c1)).inverse_fast();
    return {B::mul_ni(c0, n), B::mul_ni(c1, n).neg()};
  }
  static ZK_HD Fp2 mul_ni(const Fp2& a, const Fp2& b) { return a * b; }
  ZK_HD static Fp2 mul_sub_mul(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) { return a * b - c * d; }
};

// true for the quadratic extension (G2 coordinates)
template <class F>
struct IsExtField {
  static constexpr bool value = false;
};
template <class P, bool INL>
struct IsExtField<Fp2T<P, INL>> {
  static constexpr bool value = true;
};
// parameters of the base prime field of a coordinate field (Fp<P> -> P, Fp2T<P, I> -> P)
template <class T>
struct BaseParams;
template <class P>
struct BaseParams<Fp<P>> {
  using type = P;
};
template <class P, bool INL>
struct BaseParams<Fp2T<P, INL>> {
  using type = P;
};

}  // namespace zk
