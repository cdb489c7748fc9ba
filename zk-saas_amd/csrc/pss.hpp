// Packed-Shamir pack / unpack and the king-side kernels of d_fft, deg_red and d_pp.
//
// Reference: secret-sharing/src/pss.rs:69-221 (det_pack, pack, unpack, unpack2, lagrange_unpack),
// dist-primitives/src/dfft/mod.rs:210-304 (fft2_in_place + king closure of fft2_with_rearrange),
// dist-primitives/src/utils/deg_red.rs:103-111, dist-primitives/src/dpp/mod.rs:41-76.
//
// pack and unpack are fixed linear maps for a given l (DESIGN.md "PSS as matrices"):
//   shares  = P (n x (l+t)) * [secrets ; randoms]          -- coset-IFFT on g*H_{l+t}, FFT on H_n
//   secrets = U (l x n')    * shares                        -- IFFT on H_n, (coset-)FFT, truncation;
//                                                              for a party subset the Lagrange form.
// The matrices are built on the host with the same field code and live in HBM; their entries are
// wave-uniform, so the compiler fetches them with scalar loads.
#pragma once
#include "field.hpp"
#include "ntt.hpp"
#include "prng.hpp"

namespace zk {
#if defined(__HIPCC__)

constexpr int KING_THREADS = 256;
// The king kernels of d_fft / deg_red reach unpack2 through the one-reduction dot product (Fp::dot_k) or through one
// product per term: build-time choice for same-box A/B runs (tools/ab.sh), see DESIGN.md "unpack2 as dot products".
#ifndef ZK_KING_DOT
#define ZK_KING_DOT 1
#endif
constexpr bool KING_DOT = ZK_KING_DOT != 0;
// One-wave workgroups for the short kernels of a Groth16 proof (NTT passes on 2^8-element tiles, king / deg_red / vec
// kernels, workspace zeroing), so that a workgroup fits any single wave slot the concurrent MSM accumulate kernels free:
// a 512-thread / 64 KB NTT workgroup needs a whole CU to drain and waited 0.7-1.9 ms for a 0.07 ms pass.  While the proof
// was bound by the chip's multiplier throughput this changed nothing (the earlier U-MSM only displaced the other
// accumulates: 290 vs 290 proofs/s, profiles/r02_small_groups_timeline.txt); with the faster multiplier and the
// identity-free sorts the circom_h -> U chain became the last to finish and it is worth +8.6 % (500 vs 460 proofs/s with
// tables, 366 vs 360 without).
constexpr bool small_groups() { return true; }
inline int king_block(size_t chunks) { return (small_groups() && chunks <= ((size_t)1 << 16)) ? 64 : KING_THREADS; }

// The l = 2 kernels (the configuration every reference example runs) inline their multiplies; larger
// packing factors call the out-of-line one to keep code size bounded.
template <int L, class F>
ZK_D F mulsel(const F& a, const F& b) {
  if constexpr (L <= 2) return a * b;
  else return F::mul_ni(a, b);
}

// Constants of the FFT-structured pack for l = 2 (n = 8, secret domain g*H_4), built on the host:
//   w4inv = w_4^-1, kc[i] = g^-i / 4, w8 = w_8, w4 = w_4, w8_3 = w_8^3
template <class F>
struct PackL2 {
  F w4inv, kc[4], w8, w4, w8_3;
};

// pack (secret-sharing/src/pss.rs:90-122) for one chunk: v = [secrets ; randoms] -> n shares.
// l = 2 follows the reference's own structure -- 4-point coset IFFT on the secret domain, zero-padded 8-point
// FFT on the share domain -- which costs 10 multiplications instead of the 32 of the dense 8 x 4 matrix
// (same linear map, so the shares are bit-identical).  Other packing factors use the dense matrix.
template <class P, int L, int NV>
ZK_D void pack_chunk(const Fp<P>* v, const Fp<P>* __restrict__ Pm, const PackL2<Fp<P>>* __restrict__ k2,
                     Fp<P>* out /* [4L] */) {
  using F = Fp<P>;
  if constexpr (L == 2) {
    F v2 = NV > 2 ? v[2] : F::zero(), v3 = NV > 2 ? v[3] : F::zero();
    F w4inv = k2->w4inv;
    // 4-point inverse DFT (unscaled)
    F e0 = v[0] + v2, e1 = v[0] - v2, o0 = v[1] + v3, o1 = (v[1] - v3) * w4inv;
    F c0 = (e0 + o0) * k2->kc[0], c1 = (e1 + o1) * k2->kc[1], c2 = (e0 - o0) * k2->kc[2], c3 = (e1 - o1) * k2->kc[3];
    // 8-point DFT of (c0, c1, c2, c3, 0, 0, 0, 0): even outputs = DFT_4(c), odd outputs = DFT_4(c_i w_8^i)
    F w4 = k2->w4;
    F b1 = c1 * k2->w8, b2 = c2 * w4, b3 = c3 * k2->w8_3;
    {
      F t0 = c0 + c2, t1 = c0 - c2, u0 = c1 + c3, u1 = (c1 - c3) * w4;
      out[0] = t0 + u0;
      out[4] = t0 - u0;
      out[2] = t1 + u1;
      out[6] = t1 - u1;
    }
    {
      F t0 = c0 + b2, t1 = c0 - b2, u0 = b1 + b3, u1 = (b1 - b3) * w4;
      out[1] = t0 + u0;
      out[5] = t0 - u0;
      out[3] = t1 + u1;
      out[7] = t1 - u1;
    }
  } else {
#pragma unroll 1
    for (int p = 0; p < 4 * L; p++) {
      F acc = F::zero();
#pragma unroll
      for (int i = 0; i < NV; i++) acc = acc + mulsel<L>(Pm[p * (2 * L) + i], v[i]);
      out[p] = acc;
    }
  }
}

// sec[i] += sum_s U[i][s] * row(s) over the np party rows of one chunk (unpack / unpack2 / lagrange_unpack as a matrix,
// pss.rs:125-221).  The rows are fetched in groups of up to 8 whose loads are all in flight before the first
// multiplication (one load -> multiply -> next load chain per party made the king kernels latency-bound: d_fft 2^20
// king 270 us for 30 multiplications per chunk); party counts that are not a multiple of the group (dropouts) take the
// row-at-a-time loop.
template <class F, int L>
ZK_D void unpack_term(F* sec, const F* __restrict__ U, int np, int s, const F& x) {
#pragma unroll
  for (int i = 0; i < L; i++) sec[i] = sec[i] + mulsel<L>(U[i * np + s], x);
}
// sec[i] += sum over one group of G rows, as L dot products with one reduction each (Fp::dot_k): the rows of U are
// wave-uniform, so its limbs are the scalar operands of the multiply instructions
template <class F, int L, int G>
ZK_D void unpack_group(F* sec, const F* __restrict__ U, int np, int s0, const F* const* x) {
#pragma unroll
  for (int i = 0; i < L; i++) sec[i] = sec[i] + F::template dot_k<G>(x, U + i * np + s0);
}
template <class F, int L, bool DOT = true, class RowFn>
ZK_D void unpack_accumulate(F* sec, const F* __restrict__ U, int np, RowFn row) {
  constexpr int N = 4 * L, G = N < 8 ? N : 8;
  if (np % G == 0) {
#pragma unroll 1
    for (int s0 = 0; s0 < np; s0 += G) {
      // named values, not an array: an array of loaded elements indexed in an unrolled loop ended up in scratch memory
      const F x0 = row(s0), x1 = row(s0 + 1), x2 = row(s0 + 2), x3 = row(s0 + 3);
      if constexpr (G == 8) {
        const F x4 = row(s0 + 4), x5 = row(s0 + 5), x6 = row(s0 + 6), x7 = row(s0 + 7);
        if constexpr (L <= 2 && DOT) {
          const F* const xs[8] = {&x0, &x1, &x2, &x3, &x4, &x5, &x6, &x7};
          unpack_group<F, L, 8>(sec, U, np, s0, xs);
        } else {
          unpack_term<F, L>(sec, U, np, s0, x0);
          unpack_term<F, L>(sec, U, np, s0 + 1, x1);
          unpack_term<F, L>(sec, U, np, s0 + 2, x2);
          unpack_term<F, L>(sec, U, np, s0 + 3, x3);
          unpack_term<F, L>(sec, U, np, s0 + 4, x4);
          unpack_term<F, L>(sec, U, np, s0 + 5, x5);
          unpack_term<F, L>(sec, U, np, s0 + 6, x6);
          unpack_term<F, L>(sec, U, np, s0 + 7, x7);
        }
      } else if constexpr (DOT) {
        const F* const xs[4] = {&x0, &x1, &x2, &x3};
        unpack_group<F, L, 4>(sec, U, np, s0, xs);
      } else {
        unpack_term<F, L>(sec, U, np, s0, x0);
        unpack_term<F, L>(sec, U, np, s0 + 1, x1);
        unpack_term<F, L>(sec, U, np, s0 + 2, x2);
        unpack_term<F, L>(sec, U, np, s0 + 3, x3);
      }
    }
    return;
  }
#pragma unroll 1
  for (int s = 0; s < np; s++) unpack_term<F, L>(sec, U, np, s, row(s));
}

// shares[p][j] for p < n from l secrets + t randoms.  One thread per chunk.
//   order 0: secrets[j*l + i];  order 1: secrets[j + i*nchunks].
template <class P, int L, bool DET>
__global__ __launch_bounds__(KING_THREADS) void pss_pack_kernel(const Fp<P>* __restrict__ secrets, size_t nchunks,
                                                               int order, RngSeed seed,
                                                               const Fp<P>* __restrict__ Pm /* [n][l+t] */,
                                                               const PackL2<Fp<P>>* __restrict__ k2,
                                                               Fp<P>* __restrict__ shares /* [n][nchunks] */) {
  using F = Fp<P>;
  constexpr int T = L, N = 4 * L;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nchunks) return;
  F v[L + T];
#pragma unroll
  for (int i = 0; i < L; i++) v[i] = load_elem(secrets + (order ? j + (size_t)i * nchunks : j * L + i));
  if (!DET) {
    if constexpr (T == 2) {
      rand_fp_pair<P>(seed, (uint64_t)j * T, &v[L], &v[L + 1]);
    } else {
#pragma unroll
      for (int i = 0; i < T; i++) v[L + i] = rand_fp<P>(seed, j * T + i);
    }
  }
  F sh[N];
  pack_chunk<P, L, (DET ? L : L + T)>(v, Pm, k2, sh);
#pragma unroll
  for (int p = 0; p < N; p++) store_elem(shares + (size_t)p * nchunks + j, sh[p]);
}

// secrets[j*l + i] = sum_s U[i][s] * shares[s][j].
template <class P, int L>
__global__ __launch_bounds__(KING_THREADS) void pss_unpack_kernel(const Fp<P>* __restrict__ shares /* [np][nchunks] */,
                                                                 int np, size_t nchunks,
                                                                 const Fp<P>* __restrict__ U /* [l][np] */,
                                                                 Fp<P>* __restrict__ secrets) {
  using F = Fp<P>;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nchunks) return;
  F acc[L];
#pragma unroll
  for (int i = 0; i < L; i++) acc[i] = F::zero();
#pragma unroll 1
  for (int s = 0; s < np; s++) {
    F x = load_elem(shares + (size_t)s * nchunks + j);
#pragma unroll
    for (int i = 0; i < L; i++) acc[i] = acc[i] + mulsel<L>(U[i * np + s], x);
  }
#pragma unroll
  for (int i = 0; i < L; i++) store_elem(secrets + j * L + i, acc[i]);
}

// King closure of fft2_with_rearrange (dfft/mod.rs:264-304), fused:
//   unpack_missing_shares per chunk -> fft2_in_place (log2 l stages inside the chunk, then the
//   rotate_right(1)) -> *g^pos (and *1/m when `gtab` was built with it) -> [bit-reverse] -> pack.
//
// Index facts (DESIGN.md "king"): chunk k's in-place radix-l butterfly leaves element e at position
// pos = (k + 1 + e*Lc) mod m, Lc = m/l; the stage pairing elements that differ in bit s' of e uses the
// twiddle gen^((l >> (s'+1)) * (k + 1 + Lc*(e & (2^s'-1)))).  Output chunk q packs positions q*l..q*l+l-1
// (rearrange: written to chunk bitrev(q), slot i taking position q*l + bitrev_l(i)).
//
// One workgroup handles Wc = min(blockDim.x, Lc) input chunks k0-1 .. k0+Wc-2 (mod Lc), which produce exactly
// the positions {h*Lc + k0 .. h*Lc + k0 + Wc - 1 : h < l}; these are exchanged through LDS and packed.
//   in     : [np][Lc] (+ in_mask, optional)   out: [n][Lc] (+ out_mask, optional)
//   gentab : gen^e, e in [0, m]               gtab: c*g^e, e in [0, Lc] or nullptr (g = 1, c = 1)
//   gstep  : g^(Lc*e), e < l                  in_scale: optional factor applied to every input share
// Batched form: blockIdx.y selects one of up to KING_BATCH independent vectors (the a, b, c polynomials of
// circom_h): input / output at y * stride, masks per item, share randomness stream seed + y.
// A batch of proofs (zk_groth16_prove_batch) makes that 3 vectors per proof: item y = 3 b + k; its share randomness is
// stream seed + (y / items_per) * seed_step + y % items_per, so that proof b of a batch draws what a single proof with
// seed + seed_step * b draws (items_per == 0: stream seed + y).
constexpr int KING_BATCH = 48;
template <class F>
struct KingBatch {
  const F* in_mask[KING_BATCH];
  const F* out_mask[KING_BATCH];
  size_t stride;
  uint32_t items_per, seed_step;
  // distance between the rows of two parties, in and out (0: the vector length m/l).  A king round that carries the
  // vectors of a whole batch of proofs has party rows [items][m/l]: stride = m/l, row_pitch = items * m/l
  size_t row_pitch;
  ZK_HD uint64_t seed_off(uint32_t y) const {
    return items_per ? (uint64_t)(y / items_per) * seed_step + y % items_per : (uint64_t)y;
  }
};

template <class P, int L, bool NEGATE>
__global__ __launch_bounds__(KING_THREADS, 4) void king_fft2_kernel(
    const Fp<P>* __restrict__ in0, KingBatch<Fp<P>> kb, int np, uint32_t log_lc,
    const Fp<P>* __restrict__ U, const Fp<P>* __restrict__ Pm, const Fp<P>* __restrict__ gentab,
    const Fp<P>* __restrict__ gtab, const Fp<P>* __restrict__ gstep, const Fp<P>* __restrict__ in_scale,
    const PackL2<Fp<P>>* __restrict__ k2, int rearrange, RngSeed seed0, Fp<P>* __restrict__ out0, uint32_t rs,
    uint32_t seg) {
  // seg != 0: the all-to-all king.  This launch covers the workgroups k0 = rs + blockIdx.x * Wc of ONE rank's chunk
  // range; `in` is [np][seg] holding input chunks rs-1 .. (column c = chunk rs - 1 + c), `out` is [n][seg] in range-local
  // order: the l runs of seg/l output chunks the range produces, one after the other (a2a_unpack_fft_kernel places them).
  // Twiddles, g^pos and the share randomness use the GLOBAL chunk indices, so the shares equal the star king's.
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  using F = Fp<P>;
  const F* __restrict__ in = in0 + blockIdx.y * kb.stride;
  F* __restrict__ out = out0 + blockIdx.y * kb.stride;
  const F* __restrict__ in_mask = kb.in_mask[blockIdx.y];
  const F* __restrict__ out_mask = kb.out_mask[blockIdx.y];
  const RngSeed seed = seed0.plus(kb.seed_off(blockIdx.y));
  constexpr int T = L, N = 4 * L;
  constexpr int LOGL = (L == 1) ? 0 : (L == 2) ? 1 : (L == 4) ? 2 : (L == 8) ? 3 : 4;
  extern __shared__ uint4 smem[];
  const uint32_t Lc = 1u << log_lc;
  const uint32_t Wc = Lc < blockDim.x ? Lc : blockDim.x;
  const uint32_t log_m = log_lc + LOGL;
  LdsVec<F> lds{smem, (int)(L * Wc)};
  const uint32_t tid = threadIdx.x;
  const uint32_t k0 = rs + blockIdx.x * Wc;

  if (tid < Wc) {
    // ---- phase 1: chunk k = k0 - 1 + tid (mod Lc), i.e. k + 1 = k0 + tid (or Lc when that is 0)
    uint32_t kp1 = k0 + tid;                 // k + 1 in [0, Lc)
    bool wrapped = (kp1 == 0);               // k = Lc - 1
    uint32_t k = wrapped ? Lc - 1 : kp1 - 1;
    if (wrapped) kp1 = Lc;
    F v[L];
#pragma unroll
    for (int i = 0; i < L; i++) v[i] = F::zero();
    // d_ifft scales the share by 1/m BEFORE the mask is added (dfft/mod.rs:159 then :254-258); without a mask the
    // factor is folded into gtab instead.  One branch-free row function per case: a uniform branch between the loads
    // of a row keeps the compiler from putting the loads of the whole group in flight together.
    const F* __restrict__ col = seg ? in + (blockIdx.x * Wc + tid) : in + k;
    const size_t pitch = seg ? (size_t)seg : (kb.row_pitch ? kb.row_pitch : (size_t)1 << log_lc);
    if (!in_mask) {
      unpack_accumulate<F, L, KING_DOT>(v, U, np, [&](int s) { return load_elem(col + (size_t)s * pitch); });
    } else if (!in_scale) {
      unpack_accumulate<F, L, KING_DOT>(v, U, np, [&](int s) {
        return load_elem(col + (size_t)s * pitch) + load_elem(in_mask + ((size_t)s << log_lc) + k);
      });
    } else {
      const F sc = load_elem(in_scale);
      unpack_accumulate<F, L, KING_DOT>(v, U, np, [&](int s) {
        return mulsel<L>(load_elem(col + (size_t)s * pitch), sc) + load_elem(in_mask + ((size_t)s << log_lc) + k);
      });
    }
    // fft2 inside the chunk.  tk[s'] = gen^((l >> (s'+1)) * (k+1)); last stage uses gen^(k+1).
    if (L > 1) {
      F tk[LOGL > 0 ? LOGL : 1];
      tk[LOGL - 1] = load_elem(gentab + kp1);
#pragma unroll
      for (int s = LOGL - 2; s >= 0; s--) tk[s] = tk[s + 1].sqr();
#pragma unroll
      for (int s = 0; s < LOGL; s++) {
#pragma unroll
        for (int e = 0; e < L; e++) {
          if (e & (1 << s)) continue;
          int elow = e & ((1 << s) - 1);
          // gen^((l >> (s+1)) * Lc * elow) = gen^(m * elow / 2^(s+1))
          F tw = tk[s];
          if (elow) tw = mulsel<L>(tw, load_elem(gentab + ((size_t)elow << (log_m - s - 1))));
          F y = mulsel<L>(v[e | (1 << s)], tw);
          F x = v[e];
          v[e] = x + y;
          v[e | (1 << s)] = x - y;
        }
      }
    }
    // scale by g^pos and park in LDS at local position (region, offset)
    F gk = gtab ? load_elem(gtab + kp1) : F::one();
#pragma unroll
    for (int e = 0; e < L; e++) {
      F val = v[e];
      uint32_t region = e;
      if (wrapped) {
        region = (e + 1) % L;
        // pos = (e+1)*Lc mod m ; g^pos = gstep[(e+1) % L] (times c, folded into gtab[0])
        if (gtab) val = val * load_elem(gtab + 0) * load_elem(gstep + region);
      } else if (gtab) {
        val = val * gk;
        if (e) val = val * load_elem(gstep + e);
      }
      if (NEGATE) val = val.neg();   // FftMask::sample negates the mask values (dfft/mod.rs:56)
      uint32_t off = wrapped ? 0 : tid;
      lds.put(region * Wc + off, val);
    }
  }
  __syncthreads();
  if (tid < Wc) {
    // ---- phase 2: one output chunk per thread
    uint32_t q, ql = 0;
    if (Wc == Lc) {
      q = tid;
    } else {
      uint32_t per_region = Wc / L;
      uint32_t h = tid / per_region, jj = tid % per_region;
      q = (h * Lc + k0) / L + jj;
      ql = h * (seg / L) + (blockIdx.x * Wc) / L + jj;
    }
    F sec[L + T];
#pragma unroll
    for (int i = 0; i < L; i++) {
      uint32_t src = rearrange ? bitrev32(i, LOGL) : i;
      uint32_t pos = q * L + src;
      uint32_t lp = (pos >> log_lc) * Wc + ((pos & (Lc - 1)) - k0);
      sec[i] = lds.get(lp);
    }
    uint32_t j = rearrange ? bitrev32(q, log_lc) : q;
    if constexpr (T == 2) {
      rand_fp_pair<P>(seed, (uint64_t)j * T, &sec[L], &sec[L + 1]);
    } else {
#pragma unroll
      for (int i = 0; i < T; i++) sec[L + i] = rand_fp<P>(seed, (uint64_t)j * T + i);
    }
    F sh[N];
    pack_chunk<P, L, L + T>(sec, Pm, k2, sh);
#pragma unroll
    for (int p = 0; p < N; p++) {
      size_t o = seg ? (size_t)p * seg + ql : (kb.row_pitch ? (size_t)p * kb.row_pitch : (size_t)p << log_lc) + j;
      F acc = sh[p];
      if (out_mask) acc = acc + load_elem(out_mask + o);
      store_elem(out + o, acc);
    }
  }
}

// ---- exchange layouts of the all-to-all king (engine_impl.hpp king_round_a2a).  Rank index i of the PRESENT ranks owns the
// king workgroups k0 in [i*seg, (i+1)*seg) (d_fft) or the chunks [i*seg, (i+1)*seg) (deg_red); `shift` = 1 for d_fft,
// whose workgroup k0 reads the input chunks k0-1 .. k0+Wc-2.
//   pack  : send[r][p][c] = local[p][(idx(r)*seg + c - shift) mod len]       (block of rank r at r*k*seg; c < count(r))
//   unpack: local[p][dest] = recv[i][p][c]                                    (blocks compacted over present ranks)
// `idx_of_rank[r]` = index of rank r among the present ranks or -1.
struct KingRange {      // a rank's share of the king workgroup columns: [rs, rs + cnt) of the [i*seg, (i+1)*seg) it owns
  uint32_t rs, seg, cnt;
};
struct A2aMap {
  int idx_of_rank[16];
  int nranks, npresent;
};
template <class F>
__global__ void a2a_pack_kernel(const F* __restrict__ local, int k, size_t len, uint32_t seg, uint32_t shift, A2aMap map,
                                F* __restrict__ send) {
  const size_t per = (size_t)k * seg;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * map.nranks) return;
  const int r = (int)(t / per);
  const int i = map.idx_of_rank[r];
  if (i < 0) return;
  const uint32_t p = (uint32_t)((t % per) / seg), c = (uint32_t)(t % seg);
  const size_t g = (size_t)i * seg + c;          // d_fft: king workgroup column k0 + tid; deg_red: chunk
  if (g >= len) return;                          // beyond the last range
  const size_t src = (g + len - shift) % len;
  store_elem(send + t, load_elem(local + (size_t)p * len + src));
}
// d_fft: the range of present rank i produced, for h < l, the output chunks q = (h*len + i*seg)/l + e, e < seg/l, at
// range-local position h*(seg/l) + e; they land at chunk q (or bitrev(q) when rearranged)
template <class F>
__global__ void a2a_unpack_fft_kernel(const F* __restrict__ recv, int k, uint32_t log_lc, uint32_t seg, int l_, int log_l,
                                      int rearrange, int npresent, F* __restrict__ local) {
  const size_t per = (size_t)k * seg;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * npresent) return;
  const uint32_t i = (uint32_t)(t / per), p = (uint32_t)((t % per) / seg), c = (uint32_t)(t % seg);
  const size_t len = (size_t)1 << log_lc;
  const uint32_t run = seg / l_;
  const uint32_t h = c / run, e = c % run;
  if ((size_t)i * seg + (size_t)e * l_ >= len) return;         // workgroups beyond the last range were not launched
  const uint32_t q = (uint32_t)(((size_t)h * len + (size_t)i * seg) >> log_l) + e;
  const uint32_t j = rearrange ? bitrev32(q, log_lc) : q;
  store_elem(local + ((size_t)p << log_lc) + j, load_elem(recv + t));
}
template <class F>
__global__ void a2a_unpack_rows_kernel(const F* __restrict__ recv, int k, size_t len, uint32_t seg, int npresent,
                                       F* __restrict__ local) {
  const size_t per = (size_t)k * seg;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * npresent) return;
  const uint32_t i = (uint32_t)(t / per), p = (uint32_t)((t % per) / seg), c = (uint32_t)(t % seg);
  const size_t g = (size_t)i * seg + c;
  if (g >= len) return;
  store_elem(local + (size_t)p * len + g, load_elem(recv + t));
}

// King closure of deg_red (deg_red.rs:103-111): unpack_missing_shares then pack, per chunk.
// Batched form (blockIdx.y = item, the proofs of zk_groth16_prove_batch): item y reads in / mul_b / sub_c at
// + y * in_step, writes out at + y * out_step, uses masks[y] and the randomness stream seed + y * seed_step.
constexpr int DEGRED_BATCH = 16;
template <class F>
struct DegredBatch {
  const F* in_mask[DEGRED_BATCH];
  const F* out_mask[DEGRED_BATCH];
  size_t in_step, out_step;
  uint32_t seed_step;
};
template <class P, int L>
__global__ __launch_bounds__(KING_THREADS, 4) void king_degred_kernel(
    const Fp<P>* in /* may alias out: a thread reads and writes only its own column */, DegredBatch<Fp<P>> db, int np,
    size_t len, const Fp<P>* __restrict__ U, const Fp<P>* __restrict__ Pm, const PackL2<Fp<P>>* __restrict__ k2,
    RngSeed seed, Fp<P>* out, size_t stride, size_t j0,
    const Fp<P>* __restrict__ mul_b, const Fp<P>* __restrict__ sub_c) {
  const Fp<P>* __restrict__ in_mask = db.in_mask[blockIdx.y];
  const Fp<P>* __restrict__ out_mask = db.out_mask[blockIdx.y];
  in += blockIdx.y * db.in_step;
  out += blockIdx.y * db.out_step;
  if (mul_b) {
    mul_b += blockIdx.y * db.in_step;
    sub_c += blockIdx.y * db.in_step;
  }
  seed = seed.plus((uint64_t)blockIdx.y * db.seed_step);
  // stride = row pitch of in / out (len for whole vectors); j0 = global index of column 0 (share randomness of a chunk
  // range of the all-to-all king must be the one the star king would draw).  mul_b / sub_c (optional, same layout as
  // `in`): the input share is in * mul_b - sub_c, i.e. circom_h's a*b - c (ext_wit.rs:173-177) computed at the load
  // instead of by a kernel of its own on the proof's critical chain.
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  using F = Fp<P>;
  constexpr int T = L, N = 4 * L;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= len) return;
  F sec[L + T];
#pragma unroll
  for (int i = 0; i < L; i++) sec[i] = F::zero();
  // one branch-free row function per case (see king_fft2_kernel)
  if (!mul_b && !in_mask) {
    unpack_accumulate<F, L, KING_DOT>(sec, U, np, [&](int s) { return load_elem(in + (size_t)s * stride + j); });
  } else if (!mul_b) {
    unpack_accumulate<F, L, KING_DOT>(sec, U, np, [&](int s) {
      return load_elem(in + (size_t)s * stride + j) + load_elem(in_mask + (size_t)s * stride + j);
    });
  } else if (!in_mask) {
    unpack_accumulate<F, L, KING_DOT>(sec, U, np, [&](int s) {
      return load_elem(in + (size_t)s * stride + j) * load_elem(mul_b + (size_t)s * stride + j) -
             load_elem(sub_c + (size_t)s * stride + j);
    });
  } else {
    unpack_accumulate<F, L, KING_DOT>(sec, U, np, [&](int s) {
      return load_elem(in + (size_t)s * stride + j) * load_elem(mul_b + (size_t)s * stride + j) -
             load_elem(sub_c + (size_t)s * stride + j) + load_elem(in_mask + (size_t)s * stride + j);
    });
  }
  if constexpr (T == 2) {
    rand_fp_pair<P>(seed, (uint64_t)(j0 + j) * T, &sec[L], &sec[L + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < T; i++) sec[L + i] = rand_fp<P>(seed, (uint64_t)(j0 + j) * T + i);
  }
  F sh[N];
  pack_chunk<P, L, L + T>(sec, Pm, k2, sh);
#pragma unroll
  for (int p = 0; p < N; p++) {
    size_t o = (size_t)p * stride + j;
    F acc = sh[p];
    if (out_mask) acc = acc + load_elem(out_mask + o);
    store_elem(out + o, acc);
  }
}

// ---- lane-cooperative deg_red for SHORT vectors (l = 2, all n = 8 parties present) -----------------------------------
// king_degred_kernel gives a lane a whole chunk: eight share loads' worth of products, a ChaCha block and the pack, ~17
// multiplication times in a row.  On the 2^14-chunk vector of a SHA-256-sized proof that launch is 256 waves on 1024 SIMDs --
// a quarter of the chip, pure latency: 43 us ALONE (profiles/r05_circom_h_solo_kernel_stats.csv), 75-79 us next to the
// proof's accumulate kernels, on the proof's critical chain.  Here a chunk is worked on by EIGHT lanes, one per party: lane p
// loads party p's share (forms a*b - c for its party), multiplies it by column p of the unpack2 matrix, the partial sums are
// added across the eight lanes (three xor-shuffle steps), the random draws are computed by all eight lanes alike (same
// instruction stream: no extra time), and lane p forms share p as ONE four-term dot product with row p of the dense pack
// matrix (Fp::dot_v).  ~6 multiplication times per lane, two waves per SIMD: 24 us alone, 56-62 us inside a proof (same-box
// A/B).  Same linear maps, exact arithmetic: the shares are bit-identical.
// The same form was built for king_fft2_kernel and REMOVED: its launch covers three vectors (49 152 chunks), eight lanes per
// chunk make six waves per SIMD with 1.8x the multiply instructions per chunk -- issue-bound at 54 us alone against 42 us for
// the lane-per-chunk kernel, 126-140 against 112-121 us inside a proof.
constexpr int KING_COOP_LANES = 8;
#ifndef ZK_KING_COOP
#define ZK_KING_COOP 1                                 // build-time switch for same-box A/B runs (tools/ab.sh)
#endif
constexpr size_t KING_COOP_MAX = ZK_KING_COOP ? (size_t)1 << 16 : 0;      // chunks per vector up to which the cooperative kernels are used
template <class F>
ZK_D F lanes8_sum(F v) {
#pragma unroll
  for (int m = 1; m < KING_COOP_LANES; m <<= 1) {
    F o;
#pragma unroll
    for (int i = 0; i < F::N; i++) o.v[i] = __shfl_xor(v.v[i], m, 64);
    v = v + o;
  }
  return v;
}
// share p of pack([s0, s1, r0, r1]) = row p of the dense pack matrix Pm [n][l + t] times the vector
template <class F>
ZK_D F pack_row(const F* __restrict__ Pm, int p, const F& s0, const F& s1, const F& r0, const F& r1) {
  const F k[4] = {load_elem(Pm + p * 4), load_elem(Pm + p * 4 + 1), load_elem(Pm + p * 4 + 2), load_elem(Pm + p * 4 + 3)};
  const F* const xs[4] = {&s0, &s1, &r0, &r1};
  return F::template dot_v<4>(xs, k);
}

// king_degred_kernel for l = 2, np = n = 8: blockDim.x / 8 chunks per workgroup
template <class P>
__global__ __launch_bounds__(64) void king_degred_coop_kernel(
    const Fp<P>* in, DegredBatch<Fp<P>> db, size_t len, const Fp<P>* __restrict__ U, const Fp<P>* __restrict__ Pm,
    RngSeed seed, Fp<P>* out, size_t stride, size_t j0, const Fp<P>* __restrict__ mul_b,
    const Fp<P>* __restrict__ sub_c) {
  __builtin_amdgcn_s_setprio(3);
  using F = Fp<P>;
  constexpr int NP = 8;
  const F* __restrict__ in_mask = db.in_mask[blockIdx.y];
  const F* __restrict__ out_mask = db.out_mask[blockIdx.y];
  in += blockIdx.y * db.in_step;
  out += blockIdx.y * db.out_step;
  if (mul_b) {
    mul_b += blockIdx.y * db.in_step;
    sub_c += blockIdx.y * db.in_step;
  }
  seed = seed.plus((uint64_t)blockIdx.y * db.seed_step);
  const uint32_t p = threadIdx.x % NP;
  const size_t j = (size_t)blockIdx.x * (blockDim.x / NP) + threadIdx.x / NP;
  const bool live = j < len;                    // (the eight lanes of a chunk agree; dead lanes still join the shuffles)
  const size_t jj = live ? j : 0;
  const size_t o = (size_t)p * stride + jj;
  F x = load_elem(in + o);
  if (mul_b) x = x * load_elem(mul_b + o) - load_elem(sub_c + o);      // circom_h's a*b - c at the load (ext_wit.rs:173-177)
  if (in_mask) x = x + load_elem(in_mask + o);
  const F s0 = lanes8_sum(load_elem(U + p) * x);
  const F s1 = lanes8_sum(load_elem(U + NP + p) * x);
  F r0, r1;
  rand_fp_pair<P>(seed, (uint64_t)(j0 + jj) * 2, &r0, &r1);
  F sh = pack_row<F>(Pm, (int)p, s0, s1, r0, r1);
  if (out_mask) sh = sh + load_elem(out_mask + o);
  if (live) store_elem(out + o, sh);
}

// (the king-side kernels of d_pp live in dpp.hpp)

#endif  // __HIPCC__
}  // namespace zk
