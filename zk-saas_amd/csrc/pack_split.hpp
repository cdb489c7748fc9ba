// det_pack over EXTENSION-field points in the quad-split lane layout (dealer: CRS share packing of b_g2_query,
// groth16/src/proving_key.rs:72-86).  Own header because the kernel is instantiated in the msm_<curve>_g2.hip translation
// units (pack_points_split_launch, msm_impl.hpp): those compile the 12-limb base field with INLINE products
// (ZK_MUL_INLINE_LIMBS = 12), which is what gives s2_mul its single-reduction form (field.hpp mul_pm_mul) -- compiled with
// the engine's out-of-line 12-limb product the kernel was 1.6x SLOWER than the one-lane form on BLS12-381 (78.9 vs 49.7 ms).
#pragma once
#include "ec.hpp"
#include "quad.hpp"

namespace zk {
#if defined(__HIPCC__)

// det_pack at l = 2 (groth16.hpp pss_pack_points_jsf_kernel) for EXTENSION-field points (b_g2_query), a QUAD of lanes per (chunk, party) with one base-field
// value per lane (quad.hpp split_dbl / split_madd: the lane layout of the G2 accumulate kernel).  The one-lane form
// holds whole Fq2 values -- four affine points and the running sum are 24 base-field values in a lane -- and runs at
// 0.30 of the multiplier's issue bound with spills (profiles/r06_dealer.json before this kernel); here a lane holds six.
// P0 + P1 and P0 - P1 come from split_affine_add (two inversions per quad instead of one shared: 1 % of the chain), the
// result is normalised in the split form too (1 / ZZZ through the norm: one base-field inversion, computed by all four
// lanes alike).  Digits as above; blockIdx.y = party, so every digit branch is wave-uniform.
template <class FrP, class P>
__global__ __launch_bounds__(128, (P::N > 8 ? 2 : 3)) void pss_pack_points_jsf_split_kernel(const Affine<Fp2<P>>* __restrict__ points,
                                                                       size_t nchunks, int n,
                                                                       const uint8_t* __restrict__ dig, int jlen,
                                                                       const Fp<P> beta,
                                                                       Affine<Fp2<P>>* __restrict__ shares) {
  using F = Fp<P>;
  const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;        // chunk of this quad
  const int p = (int)blockIdx.y;
  if (j >= nchunks || p >= n) return;                                           // quad-uniform
  const int q = threadIdx.x & 3;
  const bool half = (q >> 1) != 0, comp = (q & 1) != 0;
  const F* src = reinterpret_cast<const F*>(points + 2 * j);                    // P0: x.c0 x.c1 y.c0 y.c1, then P1
  const F in0 = load_elem(src + q), in1 = load_elem(src + 4 + q);
  const bool id0 = quad_all(in0.is_zero()), id1 = quad_all(in1.is_zero());
  bool idS, idD;
  const F inS = split_affine_add<P>(in0, id0, in1, id1, half, comp, &idS);
  const F inD = split_affine_add<P>(in0, id0, qsel(half, in1.neg(), in1), id1, half, comp, &idD);
  SplitAcc<P> acc = split_identity<P>(comp);
  const uint8_t* dg = dig + (size_t)p * jlen;
  for (int b = 0; b < jlen; b++) {
    acc = split_dbl(acc, half, comp);
    const uint32_t cc = dg[b];
    // pair A (low nibble) over the points, pair B (high nibble) over their images under phi(x, y) = (beta x, y), beta in the
    // BASE field: the x lanes (half 0) multiply their component by it (groth16.hpp pss_pack_points_jsf_kernel, glv_split)
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const uint32_t c = h ? cc >> 4 : cc & 15u;
      const int u0 = (int)(c & 3u) - 1, u1 = (int)((c >> 2) & 3u) - 1;
      if (u0 | u1) {
        const bool two = u0 != 0 && u1 != 0;
        const bool neg = u0 ? u0 < 0 : u1 < 0;
        const bool same = u0 == u1;
        F v = two ? (same ? inS : inD) : (u0 ? in0 : in1);
        const bool idv = two ? (same ? idS : idD) : (u0 ? id0 : id1);
        if (!idv) {
          if (neg) v = qsel(half, v.neg(), v);
          if (h) v = qsel(half, v, v * beta);
          acc = split_madd(acc, v, half, comp);
        }
      }
    }
  }
  split_store_affine<P>(acc, reinterpret_cast<F*>(shares + (size_t)p * nchunks + j) + q, half, comp);
}

// out[i] = scalars[i] * Base for EXTENSION-field points (groth16.hpp fixed_base_mul_kernel: table[w][d-1] = d 2^(WB w) Base) in
// the same lane layout: a quad of lanes per scalar, one base-field value per lane.  The one-lane form holds whole Fq2
// values and runs ONE wave per SIMD on 12-limb fields (profiles/r06_c5_sq_counters.json: 0.99 waves in flight, 52 % of
// the wave's cycles issuing, 0.12 VALU instructions per SIMD cycle against 0.23-0.25 of the saturated kernels).
template <class FrP, class P, int WB>
__global__ __launch_bounds__(128, (P::N > 8 ? 2 : 3)) void fixed_base_mul_split_kernel(const Fp<FrP>* __restrict__ scalars, size_t len,
                                                                                      const Affine<Fp2<P>>* __restrict__ table,
                                                                                      int nwin, Affine<Fp2<P>>* __restrict__ out) {
  using F = Fp<P>;
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= len) return;                                                         // quad-uniform
  const int q = threadIdx.x & 3;
  const bool half = (q >> 1) != 0, comp = (q & 1) != 0;
  Fp<FrP> s = load_elem(scalars + i).from_mont();                               // the four lanes alike
  SplitAcc<P> acc = split_identity<P>(comp);
  constexpr int N = FrP::N;
  constexpr uint32_t PER = (1u << WB) - 1u;
  for (int w = 0; w < nwin; w++) {
    const uint32_t d = s.v[0] & PER;
#pragma unroll
    for (int k = 0; k < N - 1; k++) s.v[k] = (s.v[k] >> WB) | (s.v[k + 1] << (32 - WB));
    s.v[N - 1] >>= WB;
    if (d) {
      const F v = load_elem(reinterpret_cast<const F*>(table + (size_t)w * PER + (d - 1)) + q);
      acc = split_madd(acc, v, half, comp);
    }
  }
  split_store_affine<P>(acc, reinterpret_cast<F*>(out + i) + q, half, comp);
}

#endif  // __HIPCC__
}  // namespace zk
