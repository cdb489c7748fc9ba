// Client-side NTT stages of d_fft/d_ifft: `fft1_in_place`, dist-primitives/src/dfft/mod.rs:178-208.
//
// fft1 on a share vector of length n = m/l is a decimation-in-time radix-2 NTT over w_n (bit-reversed
// input, natural output) whose stage-s twiddle for in-block position k is w_{2^s}^(k+1) instead of
// w_{2^s}^k (factor starts at `factor_stride`, dfft/mod.rs:197); the king's rotate_right(1)
// (dfft/mod.rs:236) undoes the shift.  We reproduce exactly those values.
//
// Structure (DESIGN.md "fft1"): the log2(n) stages are cut into passes of <= 11 (first) / <= 9 (later)
// stages.  One workgroup = one tile of 2048 elements (64 KiB of LDS as two 16-byte-half planes, so that
// ds_read/write_b128 are conflict free) + the LDS-staged twiddles of the pass.  Pass 0 works on
// contiguous tiles with the shifted twiddles directly.  A later pass covering stages s0+1..s1 sees the
// vector as [h][r][c] (c < 2^s0 fastest): it multiplies element (r,c) by w_{2^s1}^((c+1)*bitrev(r)) from
// the full w_m^e table in HBM and then runs a *standard* DIT over r (derivation in DESIGN.md).
// Each thread keeps 4 elements in registers and performs two stages per LDS round trip.
#pragma once
#include "field.hpp"

namespace zk {

// Two tile sizes.  2^11-element tiles (512 threads, 64 KB of LDS, 9..11 stages per pass) for n > 2^14; 2^8-element tiles
// worked by ONE wave for 2^8 <= n <= 2^14 (the Groth16 circuits of the reference): as many passes, 8x the workgroups,
// and a workgroup that fits any single wave slot (pss.hpp small_groups for the measurement; ZK_SMALL_GROUPS=0 keeps the
// large tile from n = 2^11 up).
constexpr int NTT_TILE_BITS = 11;
constexpr int NTT_TILE_BITS_SMALL = 8;
constexpr int NTT_SMALL_MAX_LOG_N = 14;

struct NttPass {
  int s0, s1;   // stages s0+1 .. s1
  int cbits;    // log2 columns per tile
};

struct NttPlan {
  int log_n;
  int npass;
  NttPass pass[4];
};

inline int ntt_tile_bits(int log_n) { return log_n <= NTT_SMALL_MAX_LOG_N ? NTT_TILE_BITS_SMALL : NTT_TILE_BITS; }

// tb = tile bits; passes after the first take at most tb - 2 stages so that a tile keeps >= 4 adjacent columns
inline NttPlan make_ntt_plan(int k, int tb = NTT_TILE_BITS) {
  NttPlan p{};
  p.log_n = k;
  if (k <= tb) {
    p.npass = 1;
    p.pass[0] = {0, k, 0};
    return p;
  }
  const int later = tb - 2;
  int npass = 1 + (k - tb + later - 1) / later;
  int parts[4];
  int base = k / npass, rem = k % npass;
  for (int i = 0; i < npass; i++) parts[i] = base + (i < rem ? 1 : 0);
  for (int i = 1; i < npass; i++)
    if (parts[i] > later) {
      parts[0] += parts[i] - later;
      parts[i] = later;
    }
  p.npass = npass;
  int s = 0;
  for (int i = 0; i < npass; i++) {
    int rb = parts[i];
    int cb = tb - rb;
    if (cb > s) cb = s;
    p.pass[i] = {s, s + rb, cb};
    s += rb;
  }
  return p;
}

#if defined(__HIPCC__)

ZK_D uint32_t bitrev32(uint32_t x, int bits) { return bits == 0 ? 0u : (__brev(x) >> (32 - bits)); }

template <class F>
ZK_D F load_elem(const F* p) {
  static_assert(sizeof(F) % 16 == 0, "element must be a multiple of 16 bytes");
  F r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* o = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(F) / 16); i++) o[i] = q[i];
  return r;
}
template <class F>
ZK_D void store_elem(F* p, const F& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  const uint4* o = reinterpret_cast<const uint4*>(&v);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(F) / 16); i++) q[i] = o[i];
}

// LDS planes: element i lives at plane[k][i], k = 0..sizeof(F)/16-1.
template <class F>
struct LdsVec {
  static constexpr int H = sizeof(F) / 16;
  uint4* base;
  int stride;  // elements per plane
  ZK_D F get(int i) const {
    F r;
    uint4* o = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (int k = 0; k < H; k++) o[k] = base[k * stride + i];
    return r;
  }
  ZK_D void put(int i, const F& v) const {
    const uint4* o = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (int k = 0; k < H; k++) base[k * stride + i] = o[k];
  }
};

// LDS swizzle of the NTT tile (round 6).  The tile is two planes of 16-byte halves; `ds_read_b128` serves a wave in four
// 16-lane groups over 16 slots of 16 bytes (256 B), `ds_write_b128` in eight 8-lane groups over 8 slots (MI355X_MICROARCH.md,
// LDS).  The load / store phases walk consecutive elements (conflict free), but a butterfly round trip at stage offset
// sigma0 reads elements whose index has its lane bits at positions {0..sigma0-1, sigma0+2..}: the first two round trips of a
// pass without columns (pass 0) put a group's lanes on 2-4 slots -- 6.9 instead of 4 LDS cycles per read and 12.6 instead of
// 8 per write on average over the pass (r05_c2_sq_counters.json: 6.35 conflict cycles per LDS instruction), 8.0 / 14.4 for
// the one-wave tile.  Element i is kept at slot i ^ L(i >> 3), L a GF(2)-linear map onto the low four index bits found by
// search over the access patterns of every pass shape in use (tools/lds_swizzle_search.py): all reads and writes of the
// 2^11-element tile become conflict free (4.0 / 8.0), the one-wave tile reaches 4.0-4.8 / 8.0.  Linear: the slot of
// (a ^ b) is slot(a) ^ slot(b), so the four indices of a round trip cost one evaluation and three XORs with uniform values.
// MEASURED AND LEFT OFF (profiles/r06_ntt_ab.txt, same box, three rounds): with the swizzle d_fft 2^20 reads 0.643-0.647 ms
// against 0.642-0.643 without, the SHA-256 proof 613-619 against 603-620 proofs/s -- the pass is bound by VALU issue (95 % of
// its cycles), the conflict cycles hide under it and the swizzle's index arithmetic does not.  -DZK_NTT_SWIZZLE=1 builds it.
#ifndef ZK_NTT_SWIZZLE
#define ZK_NTT_SWIZZLE 0
#endif
template <int TB>
ZK_D uint32_t ntt_swz(uint32_t i) {
#if !ZK_NTT_SWIZZLE
  return i;
#else
  constexpr uint32_t M0 = TB >= 10 ? 0x51u : 0x0Du, M1 = TB >= 10 ? 0xFAu : 0x01u, M2 = TB >= 10 ? 0xE9u : 0x06u,
                     M3 = TB >= 10 ? 0xBAu : 0x02u;
  const uint32_t hi = i >> 3;
  return i ^ ((uint32_t)(__popc(hi & M0) & 1) | ((uint32_t)(__popc(hi & M1) & 1) << 1) | ((uint32_t)(__popc(hi & M2) & 1) << 2) |
              ((uint32_t)(__popc(hi & M3) & 1) << 3));
#endif
}

// One pass of fft1 over a batch of vectors.  grid = (n / TILE, batch), block = TILE / 4 threads, TILE = 2^TB.
//   data     : [batch][n]
//   tw_full  : w_m^e (direction of the transform), e in [0, m], m = n << log_l.  Stage twiddles
//              w_{2^rbits}^j = tw_full[j << (log_m - rbits)] are staged into LDS; the pre-twiddle of a
//              later pass is read from HBM/L2 once per element.
//   pass 0 (s0 == 0) uses the shifted exponent k+1 and no pre-twiddle.
//   add      : optional [batch][n] added at the final store (in_mask)
// optional read-only sources of the FIRST pass: vector y of the batch is read from p[y / per] + (y % per) * n instead of
// from `data` (circom_h transforms the caller's three QAP vectors into its work buffer without copying them first)
// (a batch of proofs passes three per proof: NTT_SRC_MAX = 3 x 16)
constexpr int NTT_SRC_MAX = 48;
template <class F>
struct NttSrc {
  const F* p[NTT_SRC_MAX];
  uint32_t per;
};
template <class F, int TB>
__global__ __launch_bounds__((1 << TB) / 4, TB >= 10 ? 4 : 16) void ntt_pass_kernel(F* __restrict__ data, int log_n, int s0,
                                                                               int s1, int cbits,
                                                                               const F* __restrict__ tw_full, int log_l,
                                                                               const F* __restrict__ add, NttSrc<F> src,
                                                                               int tws) {
  // tws: log2 stride of the stage-twiddle table kept in LDS.  0 = all R/2 + 1 twiddles of the pass (small tiles).  2 = every
  // fourth one: what every stage but the last two of a pass needs; the last round trip reads its three twiddles per thread
  // from the full table in HBM / L2 instead.  For the 2^11-element tile that makes the workgroup 72 KB of LDS instead of
  // 96 KB, i.e. TWO workgroups per CU instead of one -- the load / pre-twiddle / store phases of one overlap the
  // butterflies of the other (d_fft 2^20: see DESIGN.md "d_fft on one GPU").
  constexpr int NTT_TILE_BITS = TB;                  // shadow the namespace-level (large tile) constants
  constexpr int NTT_TILE = 1 << TB;
  constexpr int NTT_THREADS = NTT_TILE / 4;
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  extern __shared__ uint4 smem[];
  constexpr int H = sizeof(F) / 16;
  const int rbits = s1 - s0;
  const int R = 1 << rbits;
  const int C = 1 << cbits;
  const int hbbits = NTT_TILE_BITS - rbits - cbits;
  LdsVec<F> tile{smem, NTT_TILE};
  LdsVec<F> twl{smem + H * NTT_TILE, ((R / 2) >> tws) + 1};
  const int tw_sh = log_n + log_l - rbits;           // w_{2^rbits}^j = tw_full[j << tw_sh]
  // stage twiddle with exponent idx = j << sft (sft = the stage's shift): from LDS when the table holds it
  auto stage_tw = [&](uint32_t j, int sft) -> F {
    if (sft >= tws) return twl.get((j << sft) >> tws);
    return load_elem(tw_full + ((size_t)(j << sft) << tw_sh));
  };
  const int tid = threadIdx.x;
  const size_t n = (size_t)1 << log_n;
  F* vec = data + (size_t)blockIdx.y * n;
  const F* vin = src.p[0] ? src.p[blockIdx.y / src.per] + (size_t)(blockIdx.y % src.per) * n : vec;
  const F* addv = add ? add + (size_t)blockIdx.y * n : nullptr;

  // tile origin
  size_t h0;
  uint32_t c0;
  if (hbbits > 0 || cbits == s0) {  // tile spans whole rows of 2^s0 columns (cbits == s0)
    h0 = (size_t)blockIdx.x << hbbits;
    c0 = 0;
  } else {
    uint32_t tiles_per_h = 1u << (s0 - cbits);
    h0 = blockIdx.x / tiles_per_h;
    c0 = (blockIdx.x % tiles_per_h) << cbits;
  }
  const bool shifted = (s0 == 0);

  // stage twiddles into LDS
  for (int j = tid; j <= ((R / 2) >> tws); j += NTT_THREADS)
    twl.put(j, load_elem(tw_full + ((size_t)(j << tws) << tw_sh)));

  const uint32_t stid = ntt_swz<TB>((uint32_t)tid);
  // load 4 elements per thread (coalesced along c), pre-twiddle for later passes
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t x = tid + q * NTT_THREADS;
    uint32_t c = x & (C - 1);
    uint32_t r = (x >> cbits) & (R - 1);
    uint32_t hb = x >> (cbits + rbits);
    size_t gi = ((h0 + hb) << s1) + ((size_t)r << s0) + c0 + c;
    F v = load_elem(vin + gi);
    if (!shifted) {
      // w_{2^s1}^((c0+c+1)*rev(r)) = w_m^(e << (log_m - s1))
      uint64_t e = (uint64_t)(c0 + c + 1) * bitrev32(r, rbits);
      e &= ((uint64_t)1 << s1) - 1;
      v = v * load_elem(tw_full + (e << (log_n + log_l - s1)));
    }
    tile.put(stid ^ ntt_swz<TB>((uint32_t)(q * NTT_THREADS)), v);          // = ntt_swz(x): tid and q * NTT_THREADS share no bit
  }
  __syncthreads();

  const int sh = shifted ? 1 : 0;
  int sigma0 = 0;  // stages done so far within this pass
  if (rbits & 1) {
    // one radix-2 stage (two butterflies per thread)
#pragma unroll
    for (int q = 0; q < 2; q++) {
      uint32_t g = tid + q * NTT_THREADS;
      uint32_t c = g & (C - 1);
      uint32_t rest = g >> cbits;          // (hb, r_high) with bit 0 of r removed
      uint32_t rh = rest & ((R >> 1) - 1);
      uint32_t hb = rest >> (rbits - 1);
      uint32_t i0 = ntt_swz<TB>(((hb << rbits) + (rh << 1)) * C + c);
      uint32_t i1 = i0 ^ ntt_swz<TB>((uint32_t)C);                        // index + C: that bit is clear in i0
      F a = tile.get(i0), b = tile.get(i1);
      if (shifted) {
        // twiddle w_2^(0+1) = -1
        tile.put(i0, a - b);
        tile.put(i1, a + b);
      } else {
        tile.put(i0, a + b);
        tile.put(i1, a - b);
      }
    }
    sigma0 = 1;
    __syncthreads();
  }
  for (; sigma0 < rbits; sigma0 += 2) {
    uint32_t g = tid;
    uint32_t c = g & (C - 1);
    g >>= cbits;
    uint32_t rl = g & ((1u << sigma0) - 1);
    g >>= sigma0;
    uint32_t rh = g & ((1u << (rbits - sigma0 - 2)) - 1);
    uint32_t hb = g >> (rbits - sigma0 - 2);
    uint32_t rbase = (rh << (sigma0 + 2)) | rl;
    uint32_t step = (uint32_t)C << sigma0;
    // the four elements of the round trip: index bits (cbits + sigma0, + 1) are clear in the base, so + k * step = ^ k * step
    const uint32_t ib = ntt_swz<TB>(((hb << rbits) + rbase) * C + c);
    const uint32_t i1 = ib ^ ntt_swz<TB>(step), i2 = ib ^ ntt_swz<TB>(2 * step), i3 = ib ^ ntt_swz<TB>(3 * step);
    F v0 = tile.get(ib), v1 = tile.get(i1), v2 = tile.get(i2), v3 = tile.get(i3);
    // (round 6: in the first round trip of a later pass three of the four twiddles are w^0 = 1; a branch that skips those
    // products was built and measured on the same box -- d_fft 2^20 0.652-0.654 against 0.642-0.643 ms, the SHA-256 proof
    // 567-603 against 603-620 proofs/s (profiles/r06_ntt_ab.txt): the fifth inlined product site costs more than the three
    // products it saves; removed)
    // stage sigma0+1: pairs (v0,v1), (v2,v3); twiddle exponent (rl + sh) in units of w_{2^(sigma0+1)}
    {
      F w = stage_tw(rl + sh, rbits - sigma0 - 1);
      F t1 = v1 * w, t3 = v3 * w;
      v1 = v0 - t1;
      v0 = v0 + t1;
      v3 = v2 - t3;
      v2 = v2 + t3;
    }
    // stage sigma0+2: pairs (v0,v2) with q = rl, (v1,v3) with q = rl + 2^sigma0
    {
      F wa = stage_tw(rl + sh, rbits - sigma0 - 2);
      F wb = stage_tw(rl + (1u << sigma0) + sh, rbits - sigma0 - 2);
      F t2 = v2 * wa, t3 = v3 * wb;
      v2 = v0 - t2;
      v0 = v0 + t2;
      v3 = v1 - t3;
      v1 = v1 + t3;
    }
    tile.put(ib, v0);
    tile.put(i1, v1);
    tile.put(i2, v2);
    tile.put(i3, v3);
    __syncthreads();
  }

  // store
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t x = tid + q * NTT_THREADS;
    uint32_t c = x & (C - 1);
    uint32_t r = (x >> cbits) & (R - 1);
    uint32_t hb = x >> (cbits + rbits);
    size_t gi = ((h0 + hb) << s1) + ((size_t)r << s0) + c0 + c;
    F v = tile.get(stid ^ ntt_swz<TB>((uint32_t)(q * NTT_THREADS)));
    if (addv) v = v + load_elem(addv + gi);
    store_elem(vec + gi, v);
  }
}

// Small / generic path: one launch per stage, one thread per butterfly, straight from HBM.
// Used for n < 2048 and as an independent on-device cross-check of the tiled path.
template <class F>
__global__ void ntt_stage_simple_kernel(F* __restrict__ data, int log_n, int s, const F* __restrict__ tw_full,
                                        int log_l, size_t batch) {
  size_t n = (size_t)1 << log_n;
  size_t half = n >> 1;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= half * batch) return;
  size_t b = t / half;
  size_t i = t % half;
  size_t span = (size_t)1 << (s - 1);
  size_t k = i & (span - 1);
  size_t lo = ((i >> (s - 1)) << s) + k;
  F* vec = data + b * n;
  // w_{2^s}^(k+1) = w_m^((k+1) << (log_m - s))
  size_t e = ((k + 1) << (log_n + log_l - s)) & (((size_t)1 << (log_n + log_l)) - 1);
  F x = load_elem(vec + lo);
  F y = load_elem(vec + lo + span) * load_elem(tw_full + e);
  store_elem(vec + lo, x + y);
  store_elem(vec + lo + span, x - y);
}

// tw[e] = base^e for e in [0, count): two-level (2^10 split) build, one-off per context and size.
template <class F>
__global__ void powers_kernel(F* __restrict__ out, F base, size_t count, F scale) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  store_elem(out + e, base.pow_u64(e) * scale);
}

template <class F>
__global__ void vec_sum_kernel(F* __restrict__ out, const F* __restrict__ a, const F* __restrict__ b, size_t len) {
  __builtin_amdgcn_s_setprio(3);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) store_elem(out + i, load_elem(a + i) + load_elem(b + i));
}

template <class F>
__global__ void vec_add_kernel(F* __restrict__ x, const F* __restrict__ y, size_t len) {
  __builtin_amdgcn_s_setprio(3);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) store_elem(x + i, load_elem(x + i) + load_elem(y + i));
}

// x[r * xpitch + c] += y[r * ypitch + c], c < width, r < rows (rows of two differently pitched matrices: a batch's party
// rows [items][m/l] against one proof's mask rows [m/l])
template <class F>
__global__ void vec_add2d_kernel(F* __restrict__ x, size_t xpitch, const F* __restrict__ y, size_t ypitch, size_t width,
                                 size_t rows) {
  __builtin_amdgcn_s_setprio(3);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= width * rows) return;
  const size_t r = i / width, c = i % width;
  store_elem(x + r * xpitch + c, load_elem(x + r * xpitch + c) + load_elem(y + r * ypitch + c));
}

template <class F>
__global__ void vec_scale_kernel(F* __restrict__ x, F k, size_t len) {
  __builtin_amdgcn_s_setprio(3);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) store_elem(x + i, load_elem(x + i) * k);
}

template <class F>
__global__ void vec_mul_sub_kernel(F* __restrict__ out, const F* __restrict__ a, const F* __restrict__ b,
                                   const F* __restrict__ c, size_t len) {
  __builtin_amdgcn_s_setprio(3);   // on the circom_h chain: win issue arbitration against the bulk accumulate waves
  // grid-stride: the launch may be capped at a few hundred workgroups (each one has to find a free wave slot among the
  // accumulate waves of a proof; 2048 one-wave groups took 0.36 ms to get through, the arithmetic 0.02 ms)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
    store_elem(out + i, load_elem(a + i) * load_elem(b + i) - load_elem(c + i));
}

// dist-primitives/src/dfft/mod.rs:322-335: in-place bit-reversal permutation (swap when rev(i) > i).
template <class F>
__global__ void bitrev_kernel(F* __restrict__ x, int log_n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((size_t)1 << log_n)) return;
  size_t j = log_n == 0 ? 0 : (__brevll((unsigned long long)i) >> (64 - log_n));
  if (j > i) {
    F a = load_elem(x + i), b = load_elem(x + j);
    store_elem(x + i, b);
    store_elem(x + j, a);
  }
}

#endif  // __HIPCC__
}  // namespace zk
