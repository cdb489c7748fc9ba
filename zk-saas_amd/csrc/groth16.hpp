// Groth16 prover composition on the device (groth16/src/ext_wit.rs:104-181 circom_h, groth16/src/prove.rs,
// groth16/examples/sha256.rs:32-129 dsha256) and the fixed-base multiplication used by the dealer to
// produce CRS elements / packed CRS shares (groth16/src/proving_key.rs:47-123).
#pragma once
#include <thread>
#include <vector>

#include "ec.hpp"
#include "engine.hpp"
#include "glv.hpp"
#include "ntt.hpp"

namespace zk {
#if defined(__HIPCC__)

// out[i] = scalars[i] * Base, Base given by a table of affine multiples: table[w][d-1] = d * 2^(WB w) * Base, d < 2^WB.
// One lane per scalar: `nwin` mixed additions and one inversion (dealer-side, one-off per circuit).  WB = 8: 32 windows
// of 255 entries (built on the host); WB = 16: 16 windows of 65 535 entries (fixed_base_widen_kernel), taken from 2^19
// scalars up -- half the mixed additions; the table (64-201 MB) sits in the Infinity Cache.
template <class FrP, class Fld, int WB>
__global__ __launch_bounds__(128) void fixed_base_mul_kernel(const Fp<FrP>* __restrict__ scalars, size_t len,
                                                            const Affine<Fld>* __restrict__ table, int nwin,
                                                            Affine<Fld>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  Fp<FrP> s = load_elem(scalars + i).from_mont();
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  constexpr int N = FrP::N;
  constexpr uint32_t PER = (1u << WB) - 1u;
  for (int w = 0; w < nwin; w++) {
    uint32_t d = s.v[0] & PER;
#pragma unroll
    for (int q = 0; q < N - 1; q++) s.v[q] = (s.v[q] >> WB) | (s.v[q + 1] << (32 - WB));
    s.v[N - 1] >>= WB;
    if (d) {
      Affine<Fld> t = load_elem(table + (size_t)w * PER + (d - 1));
      acc = xyzz_madd(acc, t.x, t.y);
    }
  }
  store_elem(out + i, xyzz_to_affine(acc));
}

// 16-bit windows from the 8-bit table: wide[w][d-1] = t8[2w][lo-1] + t8[2w+1][hi-1], d = hi 2^8 + lo (one mixed addition and
// one inversion per entry; entries of a window that does not exist in the 8-bit table (odd window count) have hi = 0).
template <class Fld>
__global__ __launch_bounds__(128) void fixed_base_widen_kernel(const Affine<Fld>* __restrict__ t8, int nwin8, int nwin16,
                                                              Affine<Fld>* __restrict__ wide) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nwin16 * 65535u) return;
  const int w = (int)(i / 65535u);
  const uint32_t d = (uint32_t)(i % 65535u) + 1u, lo = d & 0xffu, hi = d >> 8;
  Affine<Fld> r{Fld::zero(), Fld::zero()};
  const bool has_hi = hi && 2 * w + 1 < nwin8;
  if (lo) r = load_elem(t8 + (size_t)(2 * w) * 255 + (lo - 1));
  if (has_hi) {
    Affine<Fld> h = load_elem(t8 + (size_t)(2 * w + 1) * 255 + (hi - 1));
    if (lo) r = xyzz_to_affine(xyzz_madd(XYZZ<Fld>::from_affine(r), h.x, h.y));
    else r = h;
  }
  store_elem(wide + i, r);
}

// PSS pack / det_pack over GROUP elements (secret-sharing/src/pss.rs:69-122 with T = curve point, as used by
// PackedProvingKeyShare::pack_from_arkworks_proving_key, groth16/src/proving_key.rs:72-86, and MsmMask::sample,
// dist-primitives/src/dmsm/mod.rs:34-38): share_p = sum_i P[p][i] * point_i.  One lane per (chunk, party):
// interleaved double-and-add over the NV scalars (canonical integers, `coef` = [n][NV]), affine output.
//   points: [nchunks][NV] affine (order 0)     shares: [n][nchunks] affine
template <class FrP, class Fld, int NV>
__global__ __launch_bounds__(128) void pss_pack_points_kernel(const Affine<Fld>* __restrict__ points, size_t nchunks,
                                                             int n, const Fp<FrP>* __restrict__ coef,
                                                             Affine<Fld>* __restrict__ shares) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks * (size_t)n) return;
  size_t j = t % nchunks;
  int p = (int)(t / nchunks);
  Affine<Fld> pt[NV];
  Fp<FrP> k[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    pt[i] = load_elem(points + j * NV + i);
    k[i] = load_elem(coef + (size_t)p * NV + i);
  }
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  constexpr int N = FrP::N;
  for (int w = N - 1; w >= 0; w--) {
    for (int b = 31; b >= 0; b--) {
      acc = xyzz_dbl(acc);
      // one inlined mixed addition per loop body (NV > 2 is not unrolled: code size)
#pragma unroll(NV <= 2 ? NV : 1)
      for (int i = 0; i < NV; i++)
        if (((k[i].v[w] >> b) & 1u) && !pt[i].is_identity()) acc = xyzz_madd(acc, pt[i].x, pt[i].y);
    }
  }
  store_elem(shares + t, xyzz_to_affine(acc));
}

// a + b over affine points with every special case (identities, equal, opposite): the cold path of the table below
template <class Fld>
ZK_HD_NOINLINE Affine<Fld> affine_add_any(const Affine<Fld>& a, const Affine<Fld>& b) {
  if (a.is_identity()) return b;
  if (b.is_identity()) return a;
  return xyzz_to_affine(xyzz_madd(XYZZ<Fld>::from_affine(a), b.x, b.y));
}

// det_pack over group elements at l = 2 (two points per chunk; the CRS packing of proving_key.rs:72-86), round 6:
// share_p = k0 P0 + k1 P1 with the party's two FIXED scalars recoded on the host in Solinas' joint sparse form
// (digits in {-1, 0, 1}, on average every second column non-zero) -- one shared doubling chain and ~128 mixed additions
// of +-P0, +-P1, +-(P0 + P1), +-(P0 - P1) instead of ~256 of P0 / P1; the two sums come from ONE inversion (both slopes
// divide by x1 - x0); the result is normalised with the divstep inversion.  4 755 -> ~3 500 products per share on 8
// limbs (5 100 with round 5's Fermat ladder).  Lanes of a wave share the party, so the digit branches are wave-uniform.
//   dig: [n][jlen] bytes, MOST significant column first, (u0 + 1) | (u1 + 1) << 2 for pair A, the same << 4 for pair B
template <class FrP, class Fld>
__global__ __launch_bounds__(128) void pss_pack_points_jsf_kernel(const Affine<Fld>* __restrict__ points, size_t nchunks,
                                                                 int n, const uint8_t* __restrict__ dig, int jlen,
                                                                 const Fld beta, Affine<Fld>* __restrict__ shares) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // grid.y = party: a wave never straddles two parties
  const int p = (int)blockIdx.y;
  if (j >= nchunks || p >= n) return;
  const size_t t = (size_t)p * nchunks + j;
  const Affine<Fld> P0 = load_elem(points + 2 * j), P1 = load_elem(points + 2 * j + 1);
  Affine<Fld> S, D;                                   // P0 + P1, P0 - P1
  const Fld dx = P1.x - P0.x;
  if (!P0.is_identity() && !P1.is_identity() && !dx.is_zero()) {
    const Fld inv = dx.inverse_fast();
    const Fld sx = P0.x + P1.x;
    const Fld l1 = (P1.y - P0.y) * inv;
    S.x = l1.sqr() - sx;
    S.y = l1 * (P0.x - S.x) - P0.y;
    const Fld l2 = (P1.y + P0.y).neg() * inv;         // slope through P0 and -P1
    D.x = l2.sqr() - sx;
    D.y = l2 * (P0.x - D.x) - P0.y;
  } else {
    S = affine_add_any(P0, P1);
    D = affine_add_any(P0, Affine<Fld>{P1.x, P1.y.neg()});
  }
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  const uint8_t* dg = dig + (size_t)p * jlen;
  for (int b = 0; b < jlen; b++) {
    acc = xyzz_dbl(acc);
    const uint32_t cc = dg[b];
    // pair A (low nibble) over the points themselves, pair B (high nibble) over their images under the endomorphism
    // phi(x, y) = (beta x, y): the scalars were split k = k1 + lambda k2 on the host (glv_split) and phi(P) = lambda P, so
    // the chain is half as long.  Without the split every B nibble is (0, 0).  One inlined addition site for both.
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const uint32_t c = h ? cc >> 4 : cc & 15u;
      const int u0 = (int)(c & 3u) - 1, u1 = (int)((c >> 2) & 3u) - 1;
      if (u0 | u1) {
        const bool two = u0 != 0 && u1 != 0;
        const bool neg = u0 ? u0 < 0 : u1 < 0;        // the column is +-(entry): sign of its first non-zero digit
        Affine<Fld> q = two ? (u0 == u1 ? S : D) : (u0 ? P0 : P1);
        if (!q.is_identity()) {
          if (neg) q.y = q.y.neg();
          if (h) q.x = q.x * beta;
          acc = xyzz_madd(acc, q.x, q.y);
        }
      }
    }
  }
  store_elem(shares + t, xyzz_to_affine(acc));
}

#endif  // __HIPCC__

// Host-side scalar multiplication k * P (k in Montgomery form), plain double-and-add over XYZZ.
template <class FrP, class Fld>
inline XYZZ<Fld> host_scalar_mul(const XYZZ<Fld>& p, const Fp<FrP>& k_mont) {
  Fp<FrP> k = k_mont.from_mont();
  XYZZ<Fld> r = XYZZ<Fld>::identity();
  bool started = false;
  for (int i = FrP::N - 1; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      if (started) r = xyzz_dbl_ni(r);
      if ((k.v[i] >> b) & 1) {
        r = xyzz_add_ni(r, p);
        started = true;
      }
    }
  return r;
}

}  // namespace zk
