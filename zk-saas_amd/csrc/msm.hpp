// Variable-base MSM (Pippenger bucket method) for G1 and G2, and d_msm on top of it.
//
// Reference: `G::msm(bases, scalars)` (dist-primitives/src/dmsm/mod.rs:73, ark-ec VariableBaseMSM) and
// d_msm (dmsm/mod.rs:59-102).  The result of an MSM is a unique group element, so the algorithm below is
// free to differ from arkworks' (SURVEY.md F6); what is reproduced is the value.
//
// Pipeline (all on one stream; DESIGN.md "MSM"):
//   1. digits+count : scalar -> canonical integer -> signed c-bit digits; histogram per (window, |digit|)
//   2. scan         : exclusive scan of {count, #segments} pairs
//   3. scatter      : counting sort of point indices by bucket (sign in bit 31)
//   4. expand       : segment descriptors (a bucket longer than SEG points is cut into segments, so that a
//                     degenerate scalar distribution cannot serialise on one lane)
//   5. accumulate   : one lane per segment, mixed XYZZ additions                     <- dominant kernel
//   6. finalize     : one lane per bucket sums its segments
//   7. reduce       : sum_b (b+1) * bucket_b per window with per-lane suffix sums + an LDS tree across the
//                     workgroup ("wavefront-level bucket reduction"); one (S, A) pair per workgroup
//   8. host         : combines the few (S, A) pairs per window and folds windows high -> low
//                     (254 sequential doublings: latency-bound on any one lane, cheap on a CPU core).
#pragma once
#include <atomic>
#include <future>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "ec.hpp"
#include "engine.hpp"
#include "ntt.hpp"
#include "quad.hpp"

namespace zk {
#if defined(__HIPCC__)

constexpr int MSM_SEG_MAX = 16;       // max points per accumulate lane (smaller for small MSMs, see pick_seg)
constexpr int MSM_WS = 24;            // independent workspaces: 6 per proof in flight (concurrent MSMs on separate
                                      // streams) + 6 per batch of proofs in flight (zk_groth16_prove_batch)

// Sort-stage arrays of a launch over TWO base vectors with per-vector sorts (their identity bases differ) live in two
// copies of one workspace region; blockIdx.y picks the copy: every sort-stage kernel shifts its array pointers by
// blockIdx.y * ys bytes (ys = 0: one sort shared by both vectors).
#define ZK_YSHIFT(p)                                                                                        \
  do {                                                                                                      \
    if (p) p = reinterpret_cast<decltype(p)>(reinterpret_cast<uintptr_t>(p) + (size_t)blockIdx.y * ys);      \
  } while (0)

struct SegDesc {
  uint32_t bucket, start, end;
};

// A launch multiplies ONE base vector (or two) by a BATCH of scalar vectors -- the same query of a CRS against the
// witnesses of several proofs (zk_groth16_prove_batch).  Vector b of the batch gets its own bucket sets: everything
// downstream of the sort sees `nb * sets_per` bucket sets where a single MSM has `sets_per` (1 with a fixed-base table,
// one per window without), so the sort, accumulate, finalize and reduce launches are paid once per batch and every
// launch carries nb times the lanes.
constexpr int MSM_MAXB = 16;
template <class F>
struct MsmScalars {
  const F* p[MSM_MAXB];   // scalar vector b of the batch, npts elements each
  uint32_t npts;          // points per vector (= length of the base vector)
  uint32_t nb;            // vectors in the batch
  uint32_t sets_per;      // bucket sets per vector
};

// -------------------------------------------------------------------------------------------------- digits
// pass 0: histogram; pass 1: scatter.  coef (optional): per-part multiplier, part = i / part_len.
template <class FrP, int PASS>
__global__ void msm_digits_kernel(MsmScalars<Fp<FrP>> sc, const Fp<FrP>* __restrict__ coef,
                                  size_t part_len, int c, int nwin, int wide /* windows [0, wide) have c bits, the rest
                                  c-1 */, uint32_t pre_stride /* fixed-base table: rows of this many points, one per
                                  window; all windows share ONE bucket set; 0 = no table */, uint32_t pre_off,
                                  uint32_t* __restrict__ counts /* [nsets*B] */,
                                  uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted,
                                  const uint32_t* __restrict__ skip /* bit i: base i is the identity */, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(counts);
  ZK_YSHIFT(cursor);
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(skip);
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)sc.npts * sc.nb) return;
  const uint32_t vb = (uint32_t)(g / sc.npts), i = (uint32_t)(g % sc.npts);
  if (skip && ((skip[i >> 5] >> (i & 31)) & 1u)) return;
  Fp<FrP> s = load_elem(sc.p[vb] + i);
  if (coef) s = s * coef[i / part_len];
  s = s.from_mont();
  const uint32_t set0 = vb * sc.sets_per;
  uint32_t carry = 0;
  constexpr int N = FrP::N;
  for (int w = 0; w < nwin; w++) {
    // low cw bits, then shift the whole scalar right by cw (static limb indices: stays in registers)
    const int cw = w < wide ? c : c - 1;
    uint32_t val = s.v[0] & ((1u << cw) - 1);
#pragma unroll
    for (int q = 0; q < N - 1; q++) s.v[q] = (s.v[q] >> cw) | (s.v[q + 1] << (32 - cw));
    s.v[N - 1] >>= cw;
    int32_t d = (int32_t)(val + carry);
    if ((uint32_t)d > (1u << (cw - 1))) {
      d -= (int32_t)(1u << cw);
      carry = 1;
    } else {
      carry = 0;
    }
    if (d == 0) continue;
    uint32_t neg = d < 0 ? 1u : 0u;
    uint32_t b = (uint32_t)(neg ? -d : d) - 1;
    uint32_t key = ((set0 + (pre_stride ? 0u : (uint32_t)w)) << (c - 1)) + b;
    if (PASS == 0) {
      atomicAdd(counts + key, 1u);
    } else {
      uint32_t pos = atomicAdd(cursor + key, 1u);
      uint32_t idx = pre_stride ? (uint32_t)w * pre_stride + pre_off + i : i;
      sorted[pos] = idx | (neg << 31);
    }
  }
}

// Workspace zeroing by one-wave workgroups: hipMemsetAsync's 256-thread fill kernel waits for four free wave slots on
// one CU, which the concurrent accumulate kernels of a proof rarely leave (measured 1.0 ms for a 20 us fill).
static __global__ __launch_bounds__(64) void msm_zero_kernel(uint32_t* __restrict__ p, size_t n_words, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(p);
  size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
  if (i + 4 <= n_words) {
    *reinterpret_cast<uint4*>(p + i) = make_uint4(0, 0, 0, 0);
  } else {
    for (; i < n_words; i++) p[i] = 0;
  }
}
inline hipError_t msm_zero(void* p, size_t bytes, hipStream_t st, unsigned ny = 1, size_t ys = 0) {   // p 16-byte aligned
  const size_t words = bytes / 4;
  if (!words) return hipSuccess;
  msm_zero_kernel<<<dim3((unsigned)((words + 255) / 256), ny), dim3(64), 0, st>>>((uint32_t*)p, words, ys);
  return hipGetLastError();
}

// Identity bases contribute nothing: bit i of `skip` is set when base i is the identity in EVERY base vector of the
// launch, and the sort then emits no entry for point i.  A Groth16 CRS is full of them -- b_query holds the identity for
// every wire that no B-row mentions (59 % of the SHA-256 circuit's wires; half of the PACKED shares) -- and a lane
// that loads an identity idles while its wave adds (measured: 49 % of the lanes of the G2 accumulate active).
template <class Fld>
__global__ __launch_bounds__(256) void msm_skip_mask_kernel(const Affine<Fld>* __restrict__ bases0,
                                                            const Affine<Fld>* __restrict__ bases1, size_t npts,
                                                            uint32_t* __restrict__ skip, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  // ys != 0 (grid.y = 2): one mask per base vector; ys == 0: one mask, set where EVERY vector holds the identity
  ZK_YSHIFT(skip);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool id = false;
  if (i < npts) {
    if (ys) {
      id = load_elem((blockIdx.y ? bases1 : bases0) + i).is_identity();
    } else {
      id = load_elem(bases0 + i).is_identity();
      if (id && bases1) id = load_elem(bases1 + i).is_identity();
    }
  }
  const uint64_t m = __ballot(id);
  const int lane = threadIdx.x & 63;
  if (lane == 0 && i < npts) skip[i >> 5] = (uint32_t)m;
  if (lane == 32 && i < npts) skip[i >> 5] = (uint32_t)(m >> 32);
}

// -------------------------------------------------------------------------------------------------- big sort
// For multi-million-point MSMs the two atomics-per-(point, window) passes above dominate (measured at 8 x 2^20
// points: 4.7 ms histogram + 12.2 ms scatter against 15.6 ms of accumulate).  The big-sort path is a two-level
// counting sort whose global atomics are per (workgroup tile, bin) instead of per entry:
//   bin = (window, top BIG_HI bits of the bucket index)         nbins = nwin * 2^BIG_HI  (a few thousand)
//   part_hist   : per tile of BIG_TILE points, LDS histogram over the bins -> one global add per non-empty bin
//   (scan of the bin totals)
//   part_scatter: same tile, reserves its range in every bin with one global add, LDS ranks inside the range,
//                 writes {index|sign, low bucket bits} to tmp[]          (runs of ~16 entries per bin and tile)
//   bin_sort    : one workgroup per bin: counts of the 2^lo low-bit buckets (written out as the per-key counts the
//                 rest of the pipeline scans), LDS scan, second sweep places index|sign into sorted[]
// The order inside a bucket is arbitrary, as before; bucket sums do not depend on it.
constexpr int BIG_HI = 8;                 // top bucket bits of a bin (fewer when a batch has many bucket sets: msm_big_hi)
constexpr int BIG_MAX_BINS = 8192;        // LDS: 4 B per bin in part_hist, 8 B in part_scatter
constexpr int BIG_THREADS = 256;
constexpr int BIG_PTS_PER_THREAD = 16;   // points per thread for multi-million-point MSMs; fewer for small ones so that
                                         // the tiles (BIG_THREADS * points-per-thread points each) still fill the chip
inline int msm_big_hi(size_t nsets) {
  int hi = BIG_HI;
  while (hi > 0 && (nsets << hi) > (size_t)BIG_MAX_BINS) hi--;
  return hi;
}

// signed-digit walk over one scalar (shared by all sort kernels): fn(window, bucket index, negative)
template <class FrP, class Fn>
__device__ __forceinline__ void msm_for_each_digit(Fp<FrP> s, int c, int nwin, int wide, Fn fn) {
  uint32_t carry = 0;
  constexpr int N = FrP::N;
  for (int w = 0; w < nwin; w++) {
    const int cw = w < wide ? c : c - 1;
    uint32_t val = s.v[0] & ((1u << cw) - 1);
#pragma unroll
    for (int q = 0; q < N - 1; q++) s.v[q] = (s.v[q] >> cw) | (s.v[q + 1] << (32 - cw));
    s.v[N - 1] >>= cw;
    int32_t d = (int32_t)(val + carry);
    if ((uint32_t)d > (1u << (cw - 1))) {
      d -= (int32_t)(1u << cw);
      carry = 1;
    } else {
      carry = 0;
    }
    if (d == 0) continue;
    uint32_t neg = d < 0 ? 1u : 0u;
    fn(w, (uint32_t)(neg ? -d : d) - 1, neg);
  }
}

// scalar of global entry g = (vector vb of the batch, point i); identity bases and out-of-range entries give zero
template <class FrP>
__device__ __forceinline__ Fp<FrP> msm_load_scalar(const MsmScalars<Fp<FrP>>& sc, const Fp<FrP>* coef, size_t part_len,
                                                   size_t g, const uint32_t* skip, uint32_t* vb_out, uint32_t* i_out) {
  const uint32_t vb = (uint32_t)(g / sc.npts), i = (uint32_t)(g % sc.npts);
  *vb_out = vb;
  *i_out = i;
  if (skip && ((skip[i >> 5] >> (i & 31)) & 1u)) return Fp<FrP>::zero();      // identity base: no digit, no entry
  Fp<FrP> s = load_elem(sc.p[vb] + i);
  if (coef) s = s * coef[i / part_len];
  return s.from_mont();
}

template <class FrP>
__global__ __launch_bounds__(BIG_THREADS) void msm_part_hist_kernel(MsmScalars<Fp<FrP>> sc,
                                                                    const Fp<FrP>* __restrict__ coef, size_t part_len,
                                                                    int c, int nwin, int wide, int hi_bits, int lo_bits,
                                                                    int ppt /* points per thread */,
                                                                    uint32_t wmask /* 0: fixed-base table, all windows
                                                                    share one bucket set; ~0: one set per window */,
                                                                    uint32_t* __restrict__ bin_counts,
                                                                    const uint32_t* __restrict__ skip, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bin_counts);
  ZK_YSHIFT(skip);
  extern __shared__ uint32_t big_lds[];
  const uint32_t nbins = (sc.nb * sc.sets_per) << hi_bits;
  const size_t total = (size_t)sc.npts * sc.nb;
  for (uint32_t b = threadIdx.x; b < nbins; b += BIG_THREADS) big_lds[b] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * BIG_THREADS * ppt;
  for (int k = 0; k < ppt; k++) {
    size_t g = base + (size_t)k * BIG_THREADS + threadIdx.x;
    if (g >= total) break;
    uint32_t vb, i;
    Fp<FrP> s = msm_load_scalar<FrP>(sc, coef, part_len, g, skip, &vb, &i);
    const uint32_t set0 = vb * sc.sets_per;
    msm_for_each_digit<FrP>(s, c, nwin, wide,
                            [&](int w, uint32_t b, uint32_t) { atomicAdd(&big_lds[((set0 + ((uint32_t)w & wmask)) << hi_bits) | (b >> lo_bits)], 1u); });
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nbins; b += BIG_THREADS)
    if (big_lds[b]) atomicAdd(&bin_counts[b], big_lds[b]);
}

// exclusive scan of the bin totals by one workgroup: bin_base[0..nbins], bin_cursor = copy of bin_base
static __global__ __launch_bounds__(BIG_THREADS) void msm_bin_scan_kernel(const uint32_t* __restrict__ bin_counts,
                                                                          uint32_t nbins,
                                                                          uint32_t* __restrict__ bin_base,
                                                                          uint32_t* __restrict__ bin_cursor, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bin_counts);
  ZK_YSHIFT(bin_base);
  ZK_YSHIFT(bin_cursor);
  __shared__ uint32_t sh[BIG_THREADS];
  const uint32_t per = (nbins + BIG_THREADS - 1) / BIG_THREADS;
  const uint32_t b0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; j++)
    if (b0 + j < nbins) sum += bin_counts[b0 + j];
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < BIG_THREADS; off <<= 1) {
    uint32_t t = sh[threadIdx.x];
    if ((int)threadIdx.x >= off) t += sh[threadIdx.x - off];
    __syncthreads();
    sh[threadIdx.x] = t;
    __syncthreads();
  }
  uint32_t run = threadIdx.x ? sh[threadIdx.x - 1] : 0u;
  for (uint32_t j = 0; j < per; j++)
    if (b0 + j < nbins) {
      bin_base[b0 + j] = run;
      bin_cursor[b0 + j] = run;
      run += bin_counts[b0 + j];
    }
  if (threadIdx.x == BIG_THREADS - 1) bin_base[nbins] = sh[BIG_THREADS - 1];
}

template <class FrP>
__global__ __launch_bounds__(BIG_THREADS) void msm_part_scatter_kernel(MsmScalars<Fp<FrP>> sc,
                                                                       const Fp<FrP>* __restrict__ coef,
                                                                       size_t part_len, int c, int nwin, int wide,
                                                                       int hi_bits, int lo_bits, int ppt, uint32_t wmask,
                                                                       uint32_t pre_stride, uint32_t pre_off,
                                                                       uint32_t* __restrict__ bin_cursor,
                                                                       uint2* __restrict__ tmp,
                                                                       const uint32_t* __restrict__ skip, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bin_cursor);
  ZK_YSHIFT(tmp);
  ZK_YSHIFT(skip);
  extern __shared__ uint32_t big_lds[];
  const uint32_t nbins = (sc.nb * sc.sets_per) << hi_bits;
  const size_t total = (size_t)sc.npts * sc.nb;
  uint32_t* cnt = big_lds;            // per-bin count of this tile, then the running local rank
  uint32_t* gbase = big_lds + nbins;  // start of this tile's range inside the bin
  for (uint32_t b = threadIdx.x; b < nbins; b += BIG_THREADS) cnt[b] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * BIG_THREADS * ppt;
  for (int k = 0; k < ppt; k++) {
    size_t g = base + (size_t)k * BIG_THREADS + threadIdx.x;
    if (g >= total) break;
    uint32_t vb, i;
    Fp<FrP> s = msm_load_scalar<FrP>(sc, coef, part_len, g, skip, &vb, &i);
    const uint32_t set0 = vb * sc.sets_per;
    msm_for_each_digit<FrP>(s, c, nwin, wide,
                            [&](int w, uint32_t b, uint32_t) { atomicAdd(&cnt[((set0 + ((uint32_t)w & wmask)) << hi_bits) | (b >> lo_bits)], 1u); });
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nbins; b += BIG_THREADS) {
    uint32_t n = cnt[b];
    gbase[b] = n ? atomicAdd(&bin_cursor[b], n) : 0u;
    cnt[b] = 0;
  }
  __syncthreads();
  const uint32_t lo_mask = (1u << lo_bits) - 1;
  for (int k = 0; k < ppt; k++) {
    size_t g = base + (size_t)k * BIG_THREADS + threadIdx.x;
    if (g >= total) break;
    uint32_t vb, i;
    Fp<FrP> s = msm_load_scalar<FrP>(sc, coef, part_len, g, skip, &vb, &i);
    const uint32_t set0 = vb * sc.sets_per;
    msm_for_each_digit<FrP>(s, c, nwin, wide,
                            [&](int w, uint32_t b, uint32_t neg) {
                              uint32_t bin = ((set0 + ((uint32_t)w & wmask)) << hi_bits) | (b >> lo_bits);
                              uint32_t r = atomicAdd(&cnt[bin], 1u);
                              uint32_t idx = pre_stride ? (uint32_t)w * pre_stride + pre_off + i : i;
                              tmp[gbase[bin] + r] = make_uint2(idx | (neg << 31), b & lo_mask);
                            });
  }
}

// one workgroup per bin; the bin's keys are [bin << lo_bits, (bin + 1) << lo_bits) in the (set-major) key order
static __global__ __launch_bounds__(BIG_THREADS) void msm_bin_sort_kernel(const uint2* __restrict__ tmp,
                                                                          const uint32_t* __restrict__ bin_base,
                                                                          int hi_bits, int lo_bits,
                                                                          uint32_t keys_per_set_log2,
                                                                          uint32_t* __restrict__ counts,
                                                                          uint32_t* __restrict__ sorted, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(tmp);
  ZK_YSHIFT(bin_base);
  ZK_YSHIFT(counts);
  ZK_YSHIFT(sorted);
  __shared__ uint32_t cur[1 << 12];
  const uint32_t nlo = 1u << lo_bits;
  const uint32_t bin = blockIdx.x;
  const uint32_t w = bin >> hi_bits, hi = bin & ((1u << hi_bits) - 1);
  const size_t key0 = ((size_t)w << keys_per_set_log2) + ((size_t)hi << lo_bits);
  const uint32_t e0 = bin_base[bin], e1 = bin_base[bin + 1];
  for (uint32_t j = threadIdx.x; j < nlo; j += BIG_THREADS) cur[j] = 0;
  __syncthreads();
  for (uint32_t e = e0 + threadIdx.x; e < e1; e += BIG_THREADS) atomicAdd(&cur[tmp[e].y], 1u);
  __syncthreads();
  // counts out, exclusive scan in place (nlo <= 4096: one lane per entry, Hillis-Steele over BIG_THREADS-wide strips)
  __shared__ uint32_t strip[BIG_THREADS];
  uint32_t carry = 0;
  for (uint32_t s0 = 0; s0 < nlo; s0 += BIG_THREADS) {
    uint32_t j = s0 + threadIdx.x;
    uint32_t v = j < nlo ? cur[j] : 0u;
    if (j < nlo) counts[key0 + j] = v;
    strip[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < BIG_THREADS; off <<= 1) {
      uint32_t t = strip[threadIdx.x];
      if ((int)threadIdx.x >= off) t += strip[threadIdx.x - off];
      __syncthreads();
      strip[threadIdx.x] = t;
      __syncthreads();
    }
    uint32_t incl = strip[threadIdx.x], tot = strip[BIG_THREADS - 1];
    __syncthreads();
    if (j < nlo) cur[j] = e0 + carry + incl - v;
    carry += tot;
    __syncthreads();
  }
  for (uint32_t e = e0 + threadIdx.x; e < e1; e += BIG_THREADS) {
    uint2 t = tmp[e];
    sorted[atomicAdd(&cur[t.y], 1u)] = t.x;
  }
}

// -------------------------------------------------------------------------------------------------- scan
// Exclusive scan of pairs {count, nseg(count)} over `len` keys in three launches.
constexpr int ISCAN_THREADS = 256;
constexpr int ISCAN_PER = 8;
constexpr int ISCAN_BLOCK = ISCAN_THREADS * ISCAN_PER;

ZK_D uint32_t nseg_of(uint32_t cnt, uint32_t seg) { return (cnt + seg - 1) / seg; }

ZK_D uint2 block_scan_u2(uint2 v, uint2* sh, uint2* total) {
  int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int off = 1; off < ISCAN_THREADS; off <<= 1) {
    uint2 t = sh[tid];
    if (tid >= off) {
      t.x += sh[tid - off].x;
      t.y += sh[tid - off].y;
    }
    __syncthreads();
    sh[tid] = t;
    __syncthreads();
  }
  uint2 ex = tid ? sh[tid - 1] : make_uint2(0, 0);
  *total = sh[ISCAN_THREADS - 1];
  return ex;
}

// mode 0: write block totals; mode 1: write exclusive scans (offsets.x = point offset, offsets.y = seg offset)
static __global__ __launch_bounds__(ISCAN_THREADS) void iscan_block_kernel(const uint32_t* __restrict__ counts, size_t len,
                                                                   uint2* __restrict__ block_tot,
                                                                   const uint2* __restrict__ carry,
                                                                   uint2* __restrict__ offsets, int mode,
                                                                   uint32_t seg, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(counts);
  ZK_YSHIFT(block_tot);
  ZK_YSHIFT(carry);
  ZK_YSHIFT(offsets);
  __shared__ uint2 sh[ISCAN_THREADS];
  size_t base = (size_t)blockIdx.x * ISCAN_BLOCK + (size_t)threadIdx.x * ISCAN_PER;
  uint2 loc[ISCAN_PER];
  uint2 acc = make_uint2(0, 0);
#pragma unroll
  for (int i = 0; i < ISCAN_PER; i++) {
    uint32_t cnt = base + i < len ? counts[base + i] : 0u;
    loc[i] = acc;
    acc.x += cnt;
    acc.y += nseg_of(cnt, seg);
  }
  uint2 tot;
  uint2 ex = block_scan_u2(acc, sh, &tot);
  if (mode == 0) {
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
    return;
  }
  uint2 cr = carry[blockIdx.x];
#pragma unroll
  for (int i = 0; i < ISCAN_PER; i++)
    if (base + i < len) offsets[base + i] = make_uint2(cr.x + ex.x + loc[i].x, cr.y + ex.y + loc[i].y);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == ISCAN_THREADS - 1)
    offsets[len] = make_uint2(cr.x + tot.x, cr.y + tot.y);
}

static __global__ __launch_bounds__(ISCAN_THREADS) void iscan_carry_kernel(uint2* __restrict__ bt, size_t nblocks,
                                                                           size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bt);
  __shared__ uint2 sh[ISCAN_THREADS];
  uint2 running = make_uint2(0, 0);
  for (size_t b0 = 0; b0 < nblocks; b0 += ISCAN_BLOCK) {
    size_t base = b0 + (size_t)threadIdx.x * ISCAN_PER;
    uint2 loc[ISCAN_PER];
    uint2 acc = make_uint2(0, 0);
#pragma unroll
    for (int i = 0; i < ISCAN_PER; i++) {
      uint2 v = base + i < nblocks ? bt[base + i] : make_uint2(0, 0);
      loc[i] = acc;
      acc.x += v.x;
      acc.y += v.y;
    }
    uint2 tot;
    uint2 ex = block_scan_u2(acc, sh, &tot);
#pragma unroll
    for (int i = 0; i < ISCAN_PER; i++)
      if (base + i < nblocks)
        bt[base + i] = make_uint2(running.x + ex.x + loc[i].x, running.y + ex.y + loc[i].y);
    running.x += tot.x;
    running.y += tot.y;
    __syncthreads();
  }
}

// cursor[key] = offsets[key].x ; segment descriptors for every bucket; histogram of the segment lengths
constexpr int SEG_BINS = 65;
__device__ __forceinline__ uint32_t seg_bin(uint32_t len, uint32_t seg) { return (len * 64u + seg - 1) / seg; }

static __global__ __launch_bounds__(256) void msm_expand_kernel(const uint2* __restrict__ offsets, size_t nkeys,
                                                                uint32_t* __restrict__ cursor,
                                                                SegDesc* __restrict__ segs, uint32_t seg,
                                                                uint32_t* __restrict__ lenhist, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(cursor);
  ZK_YSHIFT(segs);
  ZK_YSHIFT(lenhist);
  __shared__ uint32_t lh[SEG_BINS];
  if (threadIdx.x < SEG_BINS) lh[threadIdx.x] = 0;
  __syncthreads();
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nkeys) {
    uint2 o = offsets[k], o1 = offsets[k + 1];
    cursor[k] = o.x;
    uint32_t s = o.y;
    uint32_t full = (o1.x - o.x) / seg, rem = (o1.x - o.x) % seg;
    for (uint32_t p = o.x; p < o1.x; p += seg, s++) {
      uint32_t e = p + seg < o1.x ? p + seg : o1.x;
      segs[s] = {(uint32_t)k, p, e};
    }
    if (full) atomicAdd(&lh[64], full);
    if (rem) atomicAdd(&lh[seg_bin(rem, seg)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < SEG_BINS && lh[threadIdx.x]) atomicAdd(&lenhist[threadIdx.x], lh[threadIdx.x]);
}

// order[] = the segment indices sorted by descending length (counting sort over SEG_BINS length classes), so that
// the 64 lanes of an accumulate wave walk chains of equal length: bucket sizes are Poisson-like and an unsorted
// launch idles ~25% of its lanes (every wave runs for its longest lane).
static __global__ __launch_bounds__(256) void msm_order_kernel(const SegDesc* __restrict__ segs,
                                                               const uint2* __restrict__ offsets, size_t nkeys,
                                                               uint32_t seg, const uint32_t* __restrict__ lenhist,
                                                               uint32_t* __restrict__ bincur,
                                                               uint32_t* __restrict__ order, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(segs);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(lenhist);
  ZK_YSHIFT(bincur);
  ZK_YSHIFT(order);
  __shared__ uint32_t lh[SEG_BINS], base[SEG_BINS];
  if (threadIdx.x < SEG_BINS) lh[threadIdx.x] = 0;
  __syncthreads();
  uint32_t nseg = offsets[nkeys].y;
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bin = 0, r = 0;
  if (s < nseg) {
    SegDesc d = segs[s];
    bin = seg_bin(d.end - d.start, seg);
    r = atomicAdd(&lh[bin], 1u);
  }
  __syncthreads();
  if (threadIdx.x < SEG_BINS) {
    uint32_t b = threadIdx.x, before = 0;
    for (uint32_t j = b + 1; j < SEG_BINS; j++) before += lenhist[j];
    base[b] = before + (lh[b] ? atomicAdd(&bincur[b], lh[b]) : 0u);
  }
  __syncthreads();
  if (s < nseg) order[base[bin] + r] = s;
}

// -------------------------------------------------------------------------------------------------- accumulate
// Waves per SIMD the accumulate kernel is compiled for.  The inlined Fq2 mixed addition takes 256 VGPRs + ~200 AGPR
// spill slots = one wave per SIMD.  Forcing two waves (256 registers, ~600 B of scratch per lane, scheduling barriers
// between the field multiplications to shorten live ranges) was measured and rejected: 0.96 ms against 0.73 ms per
// 52k-point launch -- with so few waves the multiplier needs the instruction-level parallelism across independent
// field multiplications more than it needs a second wave.
#ifndef ZK_ACC_WAVES_12
#define ZK_ACC_WAVES_12 3
#endif
#ifndef ZK_ACC_WAVES_8
#define ZK_ACC_WAVES_8 3
#endif
#ifndef ZK_PAIR_WAVES_12
#define ZK_PAIR_WAVES_12 2
#endif
template <class Fld>
constexpr int ACC_WAVES = sizeof(Fld) > 48 ? 1 : (sizeof(Fld) == 48 ? ZK_ACC_WAVES_12 : ZK_ACC_WAVES_8);
template <class Fld>
constexpr int PAIR_WAVES = sizeof(Fld) > 64 ? ZK_PAIR_WAVES_12 : 2;
template <class Fld>
__global__ __launch_bounds__(128, ACC_WAVES<Fld>) void msm_accumulate_kernel(const Affine<Fld>* __restrict__ bases0,
                                                            const Affine<Fld>* __restrict__ bases1, size_t pstride,
                                                            const uint32_t* __restrict__ sorted,
                                                            const SegDesc* __restrict__ segs,
                                                            const uint2* __restrict__ offsets, size_t nkeys,
                                                            const uint32_t* __restrict__ order,
                                                            XYZZ<Fld>* __restrict__ partial0, size_t ys) {
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(segs);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(order);
  // blockIdx.y: which of the (up to two) base vectors that share this scalar vector -- and therefore the sort
  const Affine<Fld>* __restrict__ bases = blockIdx.y ? bases1 : bases0;
  XYZZ<Fld>* __restrict__ partial = partial0 + blockIdx.y * pstride;
  // bounded grid + grid-stride loop: every workgroup of the launch is resident at once, so the launch does not sit
  // in its hardware queue waiting for wave slots (which stalls every other stream mapped to the same pipe)
  const uint32_t nseg = offsets[nkeys].y;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nseg; t += (size_t)gridDim.x * blockDim.x) {
    const uint32_t s = order[t];
    SegDesc d = segs[s];
    XYZZ<Fld> acc = XYZZ<Fld>::identity();
    // software pipeline: the next point's index and coordinates are in flight while the current one is added
    uint32_t e = sorted[d.start];
    Affine<Fld> pt = load_elem(bases + (e & 0x7fffffffu));
    for (uint32_t p = d.start; p < d.end; p++) {
      uint32_t e_next = e;
      Affine<Fld> pt_next = pt;
      if (p + 1 < d.end) {
        e_next = sorted[p + 1];
        pt_next = load_elem(bases + (e_next & 0x7fffffffu));
      }
      if (!pt.is_identity()) {
        Fld y = (e >> 31) ? pt.y.neg() : pt.y;
        acc = xyzz_madd(acc, pt.x, y);
      }
      e = e_next;
      pt = pt_next;
    }
    store_elem(partial + s, acc);
  }
}

// Extension-field variant: one PAIR of lanes per segment (quad.hpp pair_madd): 128 threads = 64 segments per workgroup.
template <class Fld>
__global__ __launch_bounds__(128, PAIR_WAVES<Fld>) void msm_accumulate_pair_kernel(const Affine<Fld>* __restrict__ bases0,
                                                                    const Affine<Fld>* __restrict__ bases1, size_t pstride,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    const SegDesc* __restrict__ segs,
                                                                    const uint2* __restrict__ offsets, size_t nkeys,
                                                                    const uint32_t* __restrict__ order,
                                                                    XYZZ<Fld>* __restrict__ partial0, size_t ys) {
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(segs);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(order);
  const Affine<Fld>* __restrict__ bases = blockIdx.y ? bases1 : bases0;
  XYZZ<Fld>* __restrict__ partial = partial0 + blockIdx.y * pstride;
  const uint32_t nseg = offsets[nkeys].y;
  const bool lb = (threadIdx.x & 1) != 0;
  const size_t npairs = (size_t)gridDim.x * (blockDim.x / 2);
  for (size_t t = (size_t)blockIdx.x * (blockDim.x / 2) + (threadIdx.x >> 1); t < nseg; t += npairs) {
    const uint32_t s = order[t];
    SegDesc d = segs[s];
    PairAcc<Fld> acc{Fld::one(), Fld::zero()};               // the identity: X = Y = 1, ZZ = ZZZ = 0
    uint32_t e = sorted[d.start];
    // my coordinate of the affine point: x for the even lane, y for the odd one (a pair reads one whole point)
    Fld pt = load_elem(reinterpret_cast<const Fld*>(bases + (e & 0x7fffffffu)) + (lb ? 1 : 0));
    for (uint32_t p = d.start; p < d.end; p++) {
      uint32_t e_next = e;
      Fld pt_next = pt;
      if (p + 1 < d.end) {
        e_next = sorted[p + 1];
        pt_next = load_elem(reinterpret_cast<const Fld*>(bases + (e_next & 0x7fffffffu)) + (lb ? 1 : 0));
      }
      // the identity sentinel is (0, 0): both coordinates zero (pair-uniform after the exchange)
      const bool ident = pt.is_zero() && pswap(pt).is_zero();
      if (!ident) {
        const Fld c = qsel(lb && (e >> 31) != 0, pt.neg(), pt);
        acc = pair_madd(acc, c, lb);
      }
      e = e_next;
      pt = pt_next;
    }
    Fld* o = reinterpret_cast<Fld*>(partial + s);            // X, Y, ZZ, ZZZ
    store_elem(o + (lb ? 1 : 0), acc.c0);
    store_elem(o + (lb ? 3 : 2), acc.c1);
  }
}

// ------------------------------------------------------------------------------------------------ finalize / reduce
// Everything after the accumulate kernel is a short chain of dependent group additions over few points, so it runs on
// the lane-cooperative addition of quad.hpp (one point per quad of lanes, 4 multiplication rounds per addition).
//
//   finalize        bucket = sum of its segment partials: one quad per bucket; buckets with more than FIN_SEQ segments
//                   (skewed digit distributions: degenerate scalars such as all ones) are summed by their whole
//                   workgroup afterwards, so no quad ever walks a long chain
//   reduce stage A  the buckets of one window, indexed by k = digit magnitude in [1, B], form a (HI+1) x LO matrix
//                   k = hi*LO + lo.  Row sums R_hi and column sums C_lo are PLAIN sums (trees of depth log2 LO / HI):
//                        sum_k k*bucket_k = LO * sum_hi hi*R_hi + sum_lo lo*C_lo
//   reduce stage B  bit slices of those two short weighted sums: TR_j = sum of R_hi over hi with bit j set, TC_j
//                   likewise -- again plain sums
//   host            X_w = sum_j 2^(j + log2 LO) TR_j + sum_j 2^j TC_j is a Horner walk over c bit positions that joins
//                   the walk over the windows (msm_fold): 2 c group operations per window on single points
// The dependent depth is (LO/64 - 1 + 6) + ~8 quad additions instead of the 27 full additions of a running-sum
// reduction, and the work stays at ~2 additions per bucket.
constexpr uint32_t FIN_SEQ = 16;
constexpr int QUAD_THREADS = 256;                 // 64 points per workgroup: one wave per SIMD of a CU
constexpr int QUAD_VL = QUAD_THREADS / 4;

template <class Fld>
__global__ __launch_bounds__(QUAD_THREADS, 2) void msm_finalize_kernel(const XYZZ<Fld>* __restrict__ partial0, size_t pstride,
                                                                   const uint2* __restrict__ offsets, size_t nkeys,
                                                                   XYZZ<Fld>* __restrict__ buckets0, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(offsets);
  const XYZZ<Fld>* __restrict__ partial = partial0 + blockIdx.y * pstride;
  XYZZ<Fld>* __restrict__ buckets = buckets0 + blockIdx.y * nkeys;
  extern __shared__ uint4 smem_fin[];
  XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_fin);
  __shared__ uint32_t nheavy, heavy[QUAD_VL];
  if (threadIdx.x == 0) nheavy = 0;
  __syncthreads();
  const int q = threadIdx.x & 3, vl = threadIdx.x >> 2;
  const size_t k = (size_t)blockIdx.x * QUAD_VL + vl;
  if (k < nkeys) {
    const uint32_t s0 = offsets[k].y, s1 = offsets[k + 1].y;
    if (s1 - s0 > FIN_SEQ) {
      if (q == 0) heavy[atomicAdd(&nheavy, 1u)] = (uint32_t)vl;
    } else {
      Fld acc = qidentity<Fld>(q);
      if (s1 > s0) acc = qload(partial + s0, q);
      for (uint32_t s = s0 + 1; s < s1; s++) acc = qadd(acc, qload(partial + s, q), q);
      qstore(buckets + k, q, acc);
    }
  }
  __syncthreads();
  // buckets with many segments (degenerate scalars): the whole workgroup sums one at a time -- strided accumulation
  // over its 64 quads, then a tree -- so that no quad walks a long chain.  Empty for well-spread scalars.
  const uint32_t nh = nheavy;
  for (uint32_t h = 0; h < nh; h++) {
    const size_t kh = (size_t)blockIdx.x * QUAD_VL + heavy[h];
    const uint32_t s0 = offsets[kh].y, s1 = offsets[kh + 1].y;
    Fld acc = qidentity<Fld>(q);
    for (uint32_t s = s0 + vl; s < s1; s += QUAD_VL) acc = qadd(acc, qload(partial + s, q), q);
    acc = wg_quad_sum(acc, sh, vl, q, QUAD_VL);
    if (vl == 0) qstore(buckets + kh, q, acc);
    __syncthreads();
  }
}

// stage A: group = row 0..HI or column 0..LO-1 of one bucket set; a workgroup of 64 quads handles 64 / nvl groups,
// nvl quads each (nvl = 64: one group per workgroup, shortest chain -- few groups, latency matters; nvl = 4: many
// groups, the serial part dominates and the waves stay full).  blockIdx.y = bucket set (base vector y, window w).
template <class Fld>
__global__ __launch_bounds__(QUAD_THREADS, 2) void msm_reduce_a_kernel(const XYZZ<Fld>* __restrict__ buckets0, uint32_t B,
                                                                   int lo_bits, int nvl, XYZZ<Fld>* __restrict__ out0) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ uint4 smem_red[];
  XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_red);
  const uint32_t LO = 1u << lo_bits, HI = B >> lo_bits;          // k = hi * LO + lo, 1 <= k <= B
  const uint32_t ngroups = HI + 1 + LO;
  const XYZZ<Fld>* __restrict__ wb = buckets0 + (size_t)blockIdx.y * B;     // bucket of digit magnitude k is wb[k - 1]
  const int q = threadIdx.x & 3, vl = threadIdx.x >> 2;
  const uint32_t g = blockIdx.x * (uint32_t)(QUAD_VL / nvl) + (uint32_t)(vl / nvl);
  const uint32_t sub = (uint32_t)(vl & (nvl - 1));
  Fld acc = qidentity<Fld>(q);
  if (g <= HI) {                                    // row hi = g: lo runs over the row
    for (uint32_t lo = sub; lo < LO; lo += (uint32_t)nvl) {
      uint32_t k = g * LO + lo;
      if (k >= 1 && k <= B) acc = qadd(acc, qload(wb + (k - 1), q), q);
    }
  } else if (g < ngroups) {                         // column lo = g - HI - 1: hi runs over the column
    const uint32_t lo = g - HI - 1;
    for (uint32_t hi = sub; hi <= HI; hi += (uint32_t)nvl) {
      uint32_t k = hi * LO + lo;
      if (k >= 1 && k <= B) acc = qadd(acc, qload(wb + (k - 1), q), q);
    }
  }
  acc = wg_quad_sum(acc, sh, vl, q, nvl);
  if (sub == 0 && g < ngroups) qstore(out0 + (size_t)blockIdx.y * ngroups + g, q, acc);
}

// stage B: slice j <= hb: TR_j over the rows, otherwise TC_(j - hb - 1) over the columns; same workgroup layout
template <class Fld>
__global__ __launch_bounds__(QUAD_THREADS, 2) void msm_reduce_b_kernel(const XYZZ<Fld>* __restrict__ rc0, uint32_t B, int lo_bits,
                                                                   int nvl, XYZZ<Fld>* __restrict__ out0) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ uint4 smem_red[];
  XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_red);
  const uint32_t LO = 1u << lo_bits, HI = B >> lo_bits;
  const uint32_t ngroups = HI + 1 + LO;
  int hb = 0;
  while ((1u << hb) < HI) hb++;                      // HI = 2^hb: row weights 0..HI have bits 0..hb
  const int nslices = hb + 1 + lo_bits;
  const XYZZ<Fld>* __restrict__ rc = rc0 + (size_t)blockIdx.y * ngroups;
  const int q = threadIdx.x & 3, vl = threadIdx.x >> 2;
  const int j = (int)blockIdx.x * (QUAD_VL / nvl) + vl / nvl;
  const uint32_t sub = (uint32_t)(vl & (nvl - 1));
  Fld acc = qidentity<Fld>(q);
  if (j <= hb) {
    for (uint32_t hi = sub; hi <= HI; hi += (uint32_t)nvl)
      if ((hi >> j) & 1u) acc = qadd(acc, qload(rc + hi, q), q);
  } else if (j < nslices) {
    const int jj = j - hb - 1;
    for (uint32_t lo = sub; lo < LO; lo += (uint32_t)nvl)
      if ((lo >> jj) & 1u) acc = qadd(acc, qload(rc + HI + 1 + lo, q), q);
  }
  acc = wg_quad_sum(acc, sh, vl, q, nvl);
  if (sub == 0 && j < nslices) qstore(out0 + (size_t)blockIdx.y * nslices + j, q, acc);
}

// Fixed-base table (zk_msm_precompute): row w holds 2^(start of window w) * P_i in affine form, so that every window's
// digit of a scalar selects a point of its own row and ALL windows share one bucket set: no per-window bucket
// reduction, no doublings in the final fold, and the window can be wider (fewer mixed additions per point).
template <class Fld>
__global__ __launch_bounds__(128) void msm_table_kernel(const Affine<Fld>* __restrict__ bases, size_t len, int c,
                                                       int nwin, int wide, Affine<Fld>* __restrict__ table) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  Affine<Fld> p = load_elem(bases + i);
  if (p.is_identity()) {
    for (int w = 0; w < nwin; w++) store_elem(table + (size_t)w * len + i, p);
    return;
  }
  XYZZ<Fld> acc = XYZZ<Fld>::from_affine(p);
  int left = 0;
  for (int w = 0; w < nwin;) {                       // one inlined doubling / one inlined inversion call site
    if (left == 0) {
      store_elem(table + (size_t)w * len + i, xyzz_to_affine(acc));
      left = w < wide ? c : c - 1;
      w++;
      if (w == nwin) break;
    }
    acc = xyzz_dbl(acc);
    left--;
  }
}

#endif  // __HIPCC__

// Field type the device kernels are instantiated with: the inline tower for G2 over 8-limb base fields.
template <class Fld>
struct KernelField {
  using type = Fld;
};
template <class P>
struct KernelField<Fp2T<P, false>> {
  using type = Fp2T<P, (P::N <= ZK_MUL_INLINE_LIMBS)>;
};

// ---------------------------------------------------------------------------------------------------- host
// One workspace slot = device scratch + a pinned host buffer for the (S, A) pairs + the event that marks the end of
// the slot's last launch.  Independent MSMs run on different slots / streams.
struct MsmSlot {
  DevBuf ws;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t ev = nullptr;
  hipEvent_t ev_sort = nullptr;     // the sort of the slot's last launch is complete (consumed by MSMs that share it)
  ~MsmSlot() {
    if (pinned) (void)hipHostFree(pinned);
    if (ev) (void)hipEventDestroy(ev);
    if (ev_sort) (void)hipEventDestroy(ev_sort);
  }
  hipError_t ensure_pinned(size_t b) {
    if (b <= pinned_bytes) return hipSuccess;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_bytes = 0;
    hipError_t e = hipHostMalloc(&pinned, b, hipHostMallocDefault);
    if (e == hipSuccess) pinned_bytes = b;
    return e;
  }
};

// A launched MSM (or two sharing one sort) whose (S, A) pairs are on their way to the slot's pinned buffer.
// msm_fold() waits for the event and folds on the host.
// the scalar vectors of a batched launch (host side of MsmScalars)
struct MsmBatchArg {
  int nb = 1;
  const void* p[16] = {nullptr};
};

struct MsmPending {
  bool active = false;
  int kwin = 0, c = 0, wide = 0, nb = 1, lo_bits = 0;
  int batch = 1;                                 // scalar vectors of the launch (results: [base vector][batch])
  MsmSlot* slot = nullptr;
  std::shared_ptr<const MsmTable> tab, tab2;     // keep the tables alive while the kernels run
  // the sort this launch produced (device pointers into the slot's workspace): another MSM over the SAME scalars with
  // the same window layout -- Groth16's a_query / b_g1_query / b_g2_query over the witness shares -- may run its
  // accumulate on it instead of sorting again (MsmTuning::share)
  struct Sort {
    const void* scalars = nullptr;
    const void* coef = nullptr;
    size_t npts = 0, part_len = 0, nkeys = 0, max_segs = 0;
    int c = 0, nwin = 0, wide = 0;
    uint32_t seg = 0, pre_stride = 0, pre_off = 0;
    const uint32_t *sorted = nullptr, *order = nullptr;
    const SegDesc* segs = nullptr;
    const uint2* offsets = nullptr;
  } sort;
};

// Ordering between the accumulate kernels of concurrent launches (the Groth16 prover runs the G2 accumulate ahead of the
// G1 ones: see engine_impl.hpp prove_begin).  The signalling launch records `signal_ev` right after its accumulate
// kernel and then raises `signal_flag`; a waiting launch (its sort work still runs ahead) spins on `wait_flag` on the
// host -- so that the event it then waits for on its stream is this proof's record, not a stale one -- before its own
// accumulate kernel.
struct MsmGate {
  hipEvent_t wait_ev = nullptr;
  std::atomic<int>* wait_flag = nullptr;
  hipEvent_t signal_ev = nullptr;
  std::atomic<int>* signal_flag = nullptr;
};

struct MsmTuning {
  size_t bigsort_min;
  MsmGate gate;
  const MsmPending* share = nullptr;   // reuse this launch's sort when it matches (same scalars, layout, table geometry)
};

// Window width: minimise nwin * (npts + 4 * buckets) -- mixed additions plus the per-bucket reduction work --
// with nwin = ceil((BITS+1)/c) windows of evenly spread width (see msm_launch); ties go to the wider window
// (more buckets = more lanes with shorter chains).
template <class FrP>
inline int msm_pick_c(size_t npts, bool g2 = false) {
  if (g2)
    if (const char* e = getenv("ZK_MSM_C_G2")) {
      int c = atoi(e);
      if (c >= 2 && c <= 20) return c;
    }
  if (const char* e = getenv("ZK_MSM_C")) {
    int c = atoi(e);
    if (c >= 2 && c <= 20) return c;
  }
  int best = 4;
  double best_cost = 1e300;
  for (int c = 4; c <= 20; c++) {                     // > 17 only pays from ~2^25 points on (cost model below)
    int nwin = (FrP::BITS + c) / c;
    int ceff = (FrP::BITS + nwin) / nwin;          // widest window after spreading BITS+1 bits over nwin windows
    // per-bucket work is priced at 4 additions up to 17 bits (tuned on 10^5..10^7 points) and at 10 above: measured
    // on BLS12-381, 20-bit windows lose 13% at 2^24 points and win 8% at 2^26
    double cost = (double)nwin * ((double)npts + (ceff > 17 ? 10.0 : 4.0) * (double)((size_t)1 << (ceff - 1)));
    if (cost <= best_cost) {
      best_cost = cost;
      best = c;
    }
  }
  return best;
}

// points per accumulate lane (a bucket longer than this is cut into segments)
inline uint32_t msm_pick_seg(size_t npts, bool g2 = false) {
  if (g2)
    if (const char* e = getenv("ZK_MSM_SEG_G2")) {
      int v = atoi(e);
      if (v >= 1 && v <= 1024) return (uint32_t)v;
    }
  if (const char* e = getenv("ZK_MSM_SEG")) {
    int v = atoi(e);
    if (v >= 1 && v <= 1024) return (uint32_t)v;
  }
  // Measured on MI355X (SHA-256 circuit, 111k-point MSMs): the accumulate kernel's duration is set by its
  // longest lane chain, so short segments win (8.9 ms/proof at 16 vs 10.9 at 64); for multi-million-point MSMs
  // the buckets are long anyway and 64 keeps the number of partial sums down.
  return npts >= ((size_t)1 << 21) ? 64u : (uint32_t)MSM_SEG_MAX;
}

template <class FrP>
inline void msm_plan_of(size_t npts, bool g2, int* out) {
  const int c_req = msm_pick_c<FrP>(npts ? npts : 1, g2);
  const int T = FrP::BITS + 1;
  const int nwin = (T + c_req - 1) / c_req;
  out[0] = (T + nwin - 1) / nwin;
  out[1] = nwin;
  out[2] = (int)msm_pick_seg(npts, g2);
  out[3] = g2 ? 28 : 10;
}

// Enqueue the whole device pipeline of one MSM on `st` (digits .. reduce), the asynchronous copy of the (S, A) pairs
// into the slot's pinned buffer and the slot's event; no host synchronisation.  bases2 (optional): a second base
// vector multiplied by the SAME scalars (Groth16's a_query and b_g1_query over the witness shares): one sort, and
// every later launch covers both through blockIdx.y.  Defined in msm_impl.hpp, instantiated once per (curve, group) in
// its own translation unit (msm_<curve>_g<k>.hip) so that the heavy kernels compile in parallel.
template <class FrP, class Fld>
int msm_launch(IEngine* eng, MsmSlot& slot, const MsmTuning& tune, const void* bases, const void* bases2,
               const void* scalars, size_t npts, const Fp<FrP>* coef_d, size_t part_len, hipStream_t st,
               MsmPending* out, const MsmBatchArg* batch = nullptr);
// zk_msm_precompute's table kernel (same translation units)
template <class FrP, class Fld>
int msm_table_launch(IEngine* eng, const void* bases, size_t len, int c, int nwin, int wide, void* table,
                     hipStream_t st);

// Wait for a launched MSM and fold on the host.  The device left, per window, the c bit slices of the weighted bucket
// sum (msm_reduce_b_kernel): X_w = sum_j 2^(j + lo_bits) TR_j + sum_j 2^j TC_j, and the result is sum_w 2^(start of w)
// X_w -- one Horner walk over all bit positions from the top: a doubling per bit and an addition per slice.
// results: [base vector v < nvec][scalar vector b < batch] at results[v * batch + b]
template <class Fld>
int msm_fold_batch(IEngine* eng, MsmPending& p, XYZZ<Fld>* results, int nvec) {
  const int batch = p.batch;
  for (int i = 0; i < nvec * batch; i++) results[i] = XYZZ<Fld>::identity();
  if (!p.active) return ZK_OK;
  p.active = false;
  hipError_t he = hipEventSynchronize(p.slot->ev);
  p.tab.reset();
  p.tab2.reset();
  if (he != hipSuccess) return eng->hip_fail(he, "msm event");
  const int kwin = p.kwin, c = p.c, wide = p.wide, lo_bits = p.lo_bits, nslices = p.c;
  const int hb = c - 1 - lo_bits;                  // row slices 0..hb come first, then lo_bits column slices
  const XYZZ<Fld>* hall = (const XYZZ<Fld>*)p.slot->pinned;
  if (nvec > p.nb) nvec = p.nb;
  // windows [w_lo, w_hi] of one vector, high to low, doublings only INSIDE the range:
  // sum_w 2^(start_w - start_w_lo) X_w with X_w = sum_j 2^(j + lo_bits) TR_j + sum_j 2^j TC_j
  auto fold_range = [&](const XYZZ<Fld>* h, int w_hi, int w_lo) -> XYZZ<Fld> {
    XYZZ<Fld> total = XYZZ<Fld>::identity();
    for (int w = w_hi; w >= w_lo; w--) {
      const XYZZ<Fld>* sl = h + (size_t)w * nslices;
      const int cw = (kwin == 1 || w < wide) ? c : c - 1;
      for (int t = cw - 1; t >= 0; t--) {
        total = xyzz_dbl_ni(total);
        const XYZZ<Fld>& s = t >= lo_bits ? sl[t - lo_bits] : sl[hb + 1 + t];
        if (!s.is_identity()) total = xyzz_add_ni(total, s);
      }
    }
    return total;
  };
  auto width_of = [&](int w_hi, int w_lo) {
    int bits = 0;
    for (int w = w_hi; w >= w_lo; w--) bits += (kwin == 1 || w < wide) ? c : c - 1;
    return bits;
  };
  // Table-free MSMs have one bucket set per window: ~250 doublings + ~2c additions per window on one host thread
  // (0.4 ms for G2, the gap between two table-free proofs).  With the context's worker pool the windows are folded in
  // FOLD_PARTS contiguous groups in parallel and the groups joined by one walk of doublings: the additions leave the
  // critical path.  (With a fixed-base table there is one window and nothing to split.)
  constexpr int FOLD_PARTS = 4;
  HostPool* pool = eng->host_pool();
  static const bool par_fold = !(getenv("ZK_PAR_FOLD") && atoi(getenv("ZK_PAR_FOLD")) == 0);
  // bucket sets of (base vector v, scalar vector b) start at hall + ((v * batch + b) * kwin) * nslices
  auto sets_of = [&](int v, int b) { return hall + ((size_t)v * batch + b) * kwin * nslices; };
  // only with free workers for every sub-task: this may itself be a pool task, and waiting for sub-tasks that nobody can
  // pick up would deadlock the pool
  if (batch > 1) {
    // a batch: one fold per (v, b); table-free folds (kwin > 1) are spread over free workers
    const int jobs = nvec * batch;
    if (kwin > 1 && pool && par_fold && pool->idle() >= jobs) {
      std::vector<std::future<void>> futs;
      for (int i = 1; i < jobs; i++) {
        XYZZ<Fld>* dst = results + i;
        const XYZZ<Fld>* h = sets_of(i / batch, i % batch);
        futs.push_back(pool->submit([=, &fold_range]() { *dst = fold_range(h, kwin - 1, 0); }));
      }
      results[0] = fold_range(sets_of(0, 0), kwin - 1, 0);
      for (auto& f : futs) f.get();
    } else {
      for (int i = 0; i < jobs; i++) results[i] = fold_range(sets_of(i / batch, i % batch), kwin - 1, 0);
    }
    return ZK_OK;
  }
  if (!pool || !par_fold || kwin < 2 * FOLD_PARTS || pool->idle() < 2 * FOLD_PARTS) {
    for (int v = 0; v < nvec; v++) results[v] = fold_range(sets_of(v, 0), kwin - 1, 0);
    return ZK_OK;
  }
  XYZZ<Fld> part[2][FOLD_PARTS];
  int lo_w[FOLD_PARTS], hi_w[FOLD_PARTS];
  for (int g = 0; g < FOLD_PARTS; g++) {          // group 0 = the highest windows
    hi_w[g] = kwin - 1 - (int)((long)kwin * g / FOLD_PARTS);
    lo_w[g] = kwin - (int)((long)kwin * (g + 1) / FOLD_PARTS);
  }
  std::vector<std::future<void>> futs;
  for (int v = 0; v < nvec; v++)
    for (int g = 0; g < FOLD_PARTS; g++) {
      if (v == 0 && g == 0) continue;              // this thread's share
      const XYZZ<Fld>* h = sets_of(v, 0);
      XYZZ<Fld>* dst = &part[v][g];
      const int a = hi_w[g], b = lo_w[g];
      futs.push_back(pool->submit([=, &fold_range]() { *dst = fold_range(h, a, b); }));
    }
  part[0][0] = fold_range(hall, hi_w[0], lo_w[0]);
  for (auto& f : futs) f.get();
  for (int v = 0; v < nvec; v++) {
    XYZZ<Fld> total = part[v][0];
    for (int g = 1; g < FOLD_PARTS; g++) {
      const int bits = width_of(hi_w[g], lo_w[g]);
      for (int i = 0; i < bits; i++) total = xyzz_dbl_ni(total);
      total = xyzz_add_ni(total, part[v][g]);
    }
    results[v] = total;
  }
  return ZK_OK;
}

// single scalar vector: result (and result2 for the second base vector of the launch)
template <class Fld>
int msm_fold(IEngine* eng, MsmPending& p, XYZZ<Fld>* result, XYZZ<Fld>* result2) {
  if (p.active && p.batch != 1) return eng->fail(ZK_ERR_GENERIC, "msm_fold on a batched launch");
  XYZZ<Fld> r[2];
  int rc = msm_fold_batch<Fld>(eng, p, r, result2 ? 2 : 1);
  *result = r[0];
  if (result2) *result2 = r[1];
  return rc;
}

// sum_p k_p * P_p for a handful of points by Straus' interleaving (4-bit windows, one shared doubling chain):
// 14 np + 252 + 64 np group operations instead of 381 np.  Host side: the in-mask term of d_msm, the king's
// unpack2 + sum over n points.
template <class FrP, class Fld>
inline XYZZ<Fld> host_straus(const XYZZ<Fld>* pts, const Fp<FrP>* k_mont, int np) {
  std::vector<XYZZ<Fld>> tab((size_t)np * 15);
  std::vector<Fp<FrP>> k((size_t)np);
  for (int p = 0; p < np; p++) {
    k[p] = k_mont[p].from_mont();
    tab[(size_t)p * 15] = pts[p];
    for (int d = 1; d < 15; d++)
      tab[(size_t)p * 15 + d] = (d & 1) ? xyzz_dbl_ni(tab[(size_t)p * 15 + d / 2])
                                        : xyzz_add_ni(tab[(size_t)p * 15 + d - 1], pts[p]);
  }
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  for (int nib = FrP::N * 8 - 1; nib >= 0; nib--) {
    for (int i = 0; i < 4; i++) acc = xyzz_dbl_ni(acc);
    for (int p = 0; p < np; p++) {
      uint32_t d = (k[p].v[nib / 8] >> (4 * (nib % 8))) & 15u;
      if (d) acc = xyzz_add_ni(acc, tab[(size_t)p * 15 + d - 1]);
    }
  }
  return acc;
}

template <class Cfg>
class MsmRunner {
 public:
  using FrP = typename Cfg::FrP;
  using Fr = Fp<FrP>;
  using Fq = Fp<typename Cfg::FqP>;
  using Fq2 = Fp2<typename Cfg::FqP>;

  static void plan(size_t npts, bool g2, int* out) { msm_plan_of<FrP>(npts, g2, out); }

  // launch on workspace slot `wslot`; the result is collected with finish_t
  template <class Fld>
  int launch_t(IEngine* eng, const void* bases, const void* scalars, size_t npts, const Fr* coef_d, size_t part_len,
               hipStream_t st, int wslot, MsmPending* pend, const void* bases2 = nullptr, MsmGate gate = MsmGate{},
               const MsmPending* share = nullptr, const MsmBatchArg* batch = nullptr) {
    if (wslot < 0 || wslot >= MSM_WS) return eng->fail(ZK_ERR_BAD_INPUT, "bad msm workspace slot");
    if (pend->active) return eng->fail(ZK_ERR_GENERIC, "msm workspace slot still in flight");
    MsmTuning tune{bigsort_min, gate, share};
    return msm_launch<FrP, Fld>(eng, slots_[wslot], tune, bases, bases2, scalars, npts, coef_d, part_len, st, pend,
                                batch);
  }
  template <class Fld>
  int finish_t(IEngine* eng, MsmPending* pend, XYZZ<Fld>* result, XYZZ<Fld>* result2 = nullptr) {
    return msm_fold<Fld>(eng, *pend, result, result2);
  }
  // blocking form
  template <class Fld>
  int run_t(IEngine* eng, const void* bases, const void* scalars, size_t npts, const Fr* coef_d, size_t part_len,
            XYZZ<Fld>* result, hipStream_t st, int wslot = 0, const void* bases2 = nullptr,
            XYZZ<Fld>* result2 = nullptr) {
    MsmPending pend;
    int rc = launch_t<Fld>(eng, bases, scalars, npts, coef_d, part_len, st, wslot, &pend, bases2);
    if (rc) return rc;
    return finish_t<Fld>(eng, &pend, result, result2);
  }

  template <class Fld>
  static void write_jacobian(void* out, const XYZZ<Fld>& p) {
    Jacobian<Fld> j = xyzz_to_jacobian(p);
    memcpy(out, &j, sizeof(j));
  }

  // G::msm: one Jacobian point to host memory.
  int run(IEngine* eng, int group, const void* bases, const void* scalars, size_t npts, const Fr* coef_d,
          size_t part_len, void* out, hipStream_t st) {
    if (npts && (!bases || !scalars)) return eng->fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (!out) return eng->fail(ZK_ERR_BAD_INPUT, "null output");
    if (group == ZK_G1) {
      XYZZ<Fq> r;
      int rc = run_t<Fq>(eng, bases, scalars, npts, coef_d, part_len, &r, st);
      if (rc) return rc;
      write_jacobian(out, r);
      return ZK_OK;
    }
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) {
        XYZZ<Fq2> r;
        int rc = run_t<Fq2>(eng, bases, scalars, npts, coef_d, part_len, &r, st);
        if (rc) return rc;
        write_jacobian(out, r);
        return ZK_OK;
      } else {
        return eng->fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
      }
    }
    return eng->fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
  }

  // sum_p coef[first + p] * mask_p over `count` Jacobian points (the in-mask term of d_msm's king step)
  template <class Fld>
  XYZZ<Fld> mask_term(const void* in_mask, int first, int count, const Fr* coef = nullptr) const {
    const Jacobian<Fld>* jm = (const Jacobian<Fld>*)in_mask;
    std::vector<XYZZ<Fld>> pts((size_t)count);
    for (int p = 0; p < count; p++) pts[p] = jacobian_to_xyzz(jm[p]);
    return host_straus<FrP, Fld>(pts.data(), coef ? coef : coef_h_.data() + first, count);
  }

  // d_msm for all n parties on this device (dmsm/mod.rs:59-102), fused (DESIGN.md "d_msm"):
  //   king output = sum_k unpack2(c_shares)[k] = sum_p coef_p * (msm_p + in_mask_p),  coef_p = sum_k U2[k][p]
  // and sum_p coef_p * msm_p is ONE msm over the n*len points with scalars pre-multiplied by coef_p; the in-mask
  // term is evaluated on the host while the device pipeline runs.
  // parties [first, first + count): bases/scalars [count][len]
  template <class Fld>
  int d_msm_range_t(IEngine* eng, const void* bases, const void* scalars, size_t len, int first, int count,
                    const void* in_mask, XYZZ<Fld>* result, hipStream_t st, int wslot = 0) {
    MsmPending pend;
    int rc = launch_t<Fld>(eng, bases, scalars, (size_t)count * len, coef_d_ + first, len, st, wslot, &pend);
    if (rc) return rc;
    XYZZ<Fld> mt = XYZZ<Fld>::identity();
    if (in_mask) mt = mask_term<Fld>(in_mask, first, count);
    XYZZ<Fld> r;
    rc = finish_t<Fld>(eng, &pend, &r);
    if (rc) return rc;
    *result = in_mask ? xyzz_add_ni(r, mt) : r;
    return ZK_OK;
  }
  template <class Fld>
  int d_msm_sum_t(IEngine* eng, const void* bases, const void* scalars, size_t len, const void* in_mask,
                  XYZZ<Fld>* result, hipStream_t st, int wslot = 0) {
    return d_msm_range_t<Fld>(eng, bases, scalars, len, 0, eng->n, in_mask, result, st, wslot);
  }

  template <class Fld>
  int d_msm_t(IEngine* eng, const void* bases, const void* scalars, size_t len, const void* in_mask,
              const void* out_mask, void* out, hipStream_t st) {
    const int n = eng->n;
    XYZZ<Fld> r;
    int rc = d_msm_sum_t<Fld>(eng, bases, scalars, len, in_mask, &r, st);
    if (rc) return rc;
    const Jacobian<Fld>* om = (const Jacobian<Fld>*)out_mask;
    Jacobian<Fld>* o = (Jacobian<Fld>*)out;
    for (int p = 0; p < n; p++) {
      XYZZ<Fld> v = r;
      if (om) v = xyzz_add_ni(v, jacobian_to_xyzz(om[p]));
      o[p] = xyzz_to_jacobian(v);
    }
    return ZK_OK;
  }

  int d_msm(IEngine* eng, int group, const void* bases, const void* scalars, size_t len, const void* in_mask,
            const void* out_mask, void* out, hipStream_t st) {
    if (!out) return eng->fail(ZK_ERR_BAD_INPUT, "null output");
    if (len && (!bases || !scalars)) return eng->fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (!coef_d_) return eng->fail(ZK_ERR_GENERIC, "d_msm coefficients not initialised");
    if (group == ZK_G1) return d_msm_t<Fq>(eng, bases, scalars, len, in_mask, out_mask, out, st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) return d_msm_t<Fq2>(eng, bases, scalars, len, in_mask, out_mask, out, st);
      else return eng->fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    }
    return eng->fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
  }

  // d_msm with explicit per-contributor coefficients (party subsets): bases/scalars [coef.size()][len]; every one
  // of the n parties receives the king's value plus its own out-mask.
  template <class Fld>
  int d_msm_coef_t(IEngine* eng, const void* bases, const void* scalars, size_t len, const std::vector<Fr>& coef,
                   const void* in_mask, const void* out_mask, void* out, hipStream_t st) {
    if (!out) return eng->fail(ZK_ERR_BAD_INPUT, "null output");
    const int np = (int)coef.size();
    Fr* cd = nullptr;
    hipError_t he = hipMalloc((void**)&cd, np * sizeof(Fr));
    if (he != hipSuccess) return eng->hip_fail(he, "hipMalloc coef");
    he = hipMemcpy(cd, coef.data(), np * sizeof(Fr), hipMemcpyHostToDevice);
    XYZZ<Fld> r;
    int rc = he == hipSuccess ? run_t<Fld>(eng, bases, scalars, (size_t)np * len, cd, len, &r, st) : eng->hip_fail(he, "memcpy");
    (void)hipFree(cd);
    if (rc) return rc;
    if (in_mask) r = xyzz_add_ni(r, mask_term<Fld>(in_mask, 0, np, coef.data()));
    const Jacobian<Fld>* om = (const Jacobian<Fld>*)out_mask;
    Jacobian<Fld>* o = (Jacobian<Fld>*)out;
    for (int p = 0; p < eng->n; p++) {
      XYZZ<Fld> v = r;
      if (om) v = xyzz_add_ni(v, jacobian_to_xyzz(om[p]));
      o[p] = xyzz_to_jacobian(v);
    }
    return ZK_OK;
  }

  // coef_p = sum_k U2[k][p]  (set by the engine once the PSS matrices exist)
  int set_coefs(IEngine* eng, const std::vector<Fr>& coef) {
    coef_h_ = coef;
    if (coef_d_) (void)hipFree(coef_d_);
    hipError_t e = hipMalloc((void**)&coef_d_, coef.size() * sizeof(Fr));
    if (e != hipSuccess) return eng->hip_fail(e, "hipMalloc coef");
    e = hipMemcpy(coef_d_, coef.data(), coef.size() * sizeof(Fr), hipMemcpyHostToDevice);
    if (e != hipSuccess) return eng->hip_fail(e, "hipMemcpy coef");
    return ZK_OK;
  }
  ~MsmRunner() {
    if (coef_d_) (void)hipFree(coef_d_);
  }

  // ---- fixed-base tables (zk_msm_precompute): process-wide registry keyed by address (engine.hpp TableRegistry)
  int table_c = 16;               // window bits of new tables (zk_ctx_set_option "msm_table_c")
  // ... of G2 tables ("msm_table_c_g2").  15 bits = 17 windows, 16 384 buckets: 6 % more mixed additions than 16 bits but
  // half the buckets in the G2 reduction, the latency chain a proof ends with (458-477 vs 423-453 proofs/s, same box).
  int table_c_g2 = getenv("ZK_TABLE_C_G2") ? atoi(getenv("ZK_TABLE_C_G2")) : 15;
  template <class Fld>
  int precompute_t(IEngine* eng, const void* bases, size_t len, hipStream_t st) {
    if (!bases || !len) return eng->fail(ZK_ERR_BAD_INPUT, "null base vector");
    const int T = FrP::BITS + 1;
    const int tc = IsExtField<Fld>::value ? table_c_g2 : table_c;
    const int nwin = (T + tc - 1) / tc;
    const int c = (T + nwin - 1) / nwin;
    if ((size_t)nwin * len >= ((size_t)1 << 31)) return eng->fail(ZK_ERR_BAD_INPUT, "base vector too long for a table");
    auto t = std::make_shared<MsmTable>();
    t->base = (const char*)bases;
    t->len = len;
    t->elem = sizeof(Affine<Fld>);
    t->c = c;
    t->nwin = nwin;
    t->wide = T - nwin * (c - 1);
    t->bits = FrP::BITS;
    t->device = eng->device;
    t->owner = eng;
    hipError_t he = hipMalloc(&t->data, (size_t)nwin * len * sizeof(Affine<Fld>));
    if (he != hipSuccess) return eng->hip_fail(he, "msm table");
    int rc = msm_table_launch<FrP, Fld>(eng, bases, len, c, nwin, t->wide, t->data, st);
    if (rc) return rc;
    he = hipStreamSynchronize(st);
    if (he != hipSuccess) return eng->hip_fail(he, "msm_table_kernel");
    {
      // does the vector hold any identity at all?  (once, here: an MSM over a vector without one skips the mask kernel)
      using KF = typename KernelField<Fld>::type;
      const size_t words = ((len + 63) / 64) * 2;
      uint32_t* mask_d = nullptr;
      he = hipMalloc(&mask_d, words * 4);
      if (he != hipSuccess) return eng->hip_fail(he, "identity mask");
      msm_skip_mask_kernel<KF><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st>>>((const Affine<KF>*)bases, nullptr, len,
                                                                                       mask_d, 0);
      std::vector<uint32_t> mask_h(words);
      he = hipMemcpyAsync(mask_h.data(), mask_d, words * 4, hipMemcpyDeviceToHost, st);
      if (he == hipSuccess) he = hipStreamSynchronize(st);
      (void)hipFree(mask_d);
      if (he != hipSuccess) return eng->hip_fail(he, "identity mask");
      bool any = false;
      for (size_t i = 0; i < (len + 31) / 32 && !any; i++) any = mask_h[i] != 0;
      t->any_identity = any;
    }
    TableRegistry::inst().add(std::move(t));
    return ZK_OK;
  }
  int forget_table(const void* bases) { return TableRegistry::inst().forget(bases); }
  // [window bits, digit windows] of the table registered for `bases`, or zeros
  void table_info(const void* bases, size_t elem, int* out) {
    size_t off;
    auto t = TableRegistry::inst().find(bases, 1, elem, FrP::BITS, &off);
    out[0] = t ? t->c : 0;
    out[1] = t ? t->nwin : 0;
  }

  // two-level sort from this many points on (zk_ctx_set_option "msm_bigsort_min"; env ZK_MSM_BIGSORT_MIN at start)
  size_t bigsort_min = getenv("ZK_MSM_BIGSORT_MIN") ? (size_t)atoll(getenv("ZK_MSM_BIGSORT_MIN")) : ((size_t)1 << 14);
  MsmSlot slots_[MSM_WS];
  Fr* coef_d_ = nullptr;
  std::vector<Fr> coef_h_;
};

}  // namespace zk
