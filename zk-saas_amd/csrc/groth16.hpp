// Groth16 prover composition on the device (groth16/src/ext_wit.rs:104-181 circom_h, groth16/src/prove.rs,
// groth16/examples/sha256.rs:32-129 dsha256) and the fixed-base multiplication used by the dealer to
// produce CRS elements / packed CRS shares (groth16/src/proving_key.rs:47-123).
#pragma once
#include <thread>

#include "ec.hpp"
#include "engine.hpp"
#include "ntt.hpp"

namespace zk {
#if defined(__HIPCC__)

// out[i] = scalars[i] * Base, Base given by a table of affine multiples: table[w][d-1] = d * 256^w * Base.
// One lane per scalar: 32 mixed additions and one Fermat inversion (dealer-side, one-off per circuit).
template <class FrP, class Fld>
__global__ __launch_bounds__(128) void fixed_base_mul_kernel(const Fp<FrP>* __restrict__ scalars, size_t len,
                                                            const Affine<Fld>* __restrict__ table, int nwin,
                                                            Affine<Fld>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  Fp<FrP> s = load_elem(scalars + i).from_mont();
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  constexpr int N = FrP::N;
  for (int w = 0; w < nwin; w++) {
    uint32_t d = s.v[0] & 0xffu;
#pragma unroll
    for (int q = 0; q < N - 1; q++) s.v[q] = (s.v[q] >> 8) | (s.v[q + 1] << 24);
    s.v[N - 1] >>= 8;
    if (d) {
      Affine<Fld> t = load_elem(table + (size_t)w * 255 + (d - 1));
      acc = xyzz_madd(acc, t.x, t.y);
    }
  }
  store_elem(out + i, xyzz_to_affine(acc));
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
