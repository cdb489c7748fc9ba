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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ec.hpp"
#include "engine.hpp"
#include "ntt.hpp"

namespace zk {
#if defined(__HIPCC__)

constexpr int MSM_SEG = 64;           // max points per accumulate lane
constexpr int RED_THREADS = 256;      // bucket-reduce workgroup
constexpr int RED_G = 8;              // buckets per lane in bucket-reduce

struct SegDesc {
  uint32_t bucket, start, end;
};

// -------------------------------------------------------------------------------------------------- digits
// pass 0: histogram; pass 1: scatter.  coef (optional): per-part multiplier, part = i / part_len.
template <class FrP, int PASS>
__global__ void msm_digits_kernel(const Fp<FrP>* __restrict__ scalars, size_t npts, const Fp<FrP>* __restrict__ coef,
                                  size_t part_len, int c, int nwin, uint32_t* __restrict__ counts /* [nwin*B] */,
                                  uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  Fp<FrP> s = load_elem(scalars + i);
  if (coef) s = s * coef[i / part_len];
  s = s.from_mont();
  const uint32_t B = 1u << (c - 1);
  uint32_t carry = 0;
  constexpr int N = FrP::N;
  for (int w = 0; w < nwin; w++) {
    // low c bits, then shift the whole scalar right by c (static limb indices: stays in registers)
    uint32_t val = s.v[0] & ((1u << c) - 1);
#pragma unroll
    for (int q = 0; q < N - 1; q++) s.v[q] = (s.v[q] >> c) | (s.v[q + 1] << (32 - c));
    s.v[N - 1] >>= c;
    int32_t d = (int32_t)(val + carry);
    if ((uint32_t)d > B) {
      d -= (int32_t)(1u << c);
      carry = 1;
    } else {
      carry = 0;
    }
    if (d == 0) continue;
    uint32_t neg = d < 0 ? 1u : 0u;
    uint32_t b = (uint32_t)(neg ? -d : d) - 1;
    uint32_t key = (uint32_t)w * B + b;
    if (PASS == 0) {
      atomicAdd(counts + key, 1u);
    } else {
      uint32_t pos = atomicAdd(cursor + key, 1u);
      sorted[pos] = (uint32_t)i | (neg << 31);
    }
  }
}

// -------------------------------------------------------------------------------------------------- scan
// Exclusive scan of pairs {count, nseg(count)} over `len` keys in three launches.
constexpr int ISCAN_THREADS = 256;
constexpr int ISCAN_PER = 8;
constexpr int ISCAN_BLOCK = ISCAN_THREADS * ISCAN_PER;

ZK_D uint32_t nseg_of(uint32_t cnt) { return (cnt + MSM_SEG - 1) / MSM_SEG; }

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
                                                                   uint2* __restrict__ offsets, int mode) {
  __shared__ uint2 sh[ISCAN_THREADS];
  size_t base = (size_t)blockIdx.x * ISCAN_BLOCK + (size_t)threadIdx.x * ISCAN_PER;
  uint2 loc[ISCAN_PER];
  uint2 acc = make_uint2(0, 0);
#pragma unroll
  for (int i = 0; i < ISCAN_PER; i++) {
    uint32_t cnt = base + i < len ? counts[base + i] : 0u;
    loc[i] = acc;
    acc.x += cnt;
    acc.y += nseg_of(cnt);
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

static __global__ __launch_bounds__(ISCAN_THREADS) void iscan_carry_kernel(uint2* __restrict__ bt, size_t nblocks) {
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

// cursor[key] = offsets[key].x ; segment descriptors for every bucket
static __global__ void msm_expand_kernel(const uint2* __restrict__ offsets, size_t nkeys, uint32_t* __restrict__ cursor,
                                  SegDesc* __restrict__ segs) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nkeys) return;
  uint2 o = offsets[k], o1 = offsets[k + 1];
  cursor[k] = o.x;
  uint32_t s = o.y;
  for (uint32_t p = o.x; p < o1.x; p += MSM_SEG, s++) {
    uint32_t e = p + MSM_SEG < o1.x ? p + MSM_SEG : o1.x;
    segs[s] = {(uint32_t)k, p, e};
  }
}

// -------------------------------------------------------------------------------------------------- accumulate
template <class Fld>
__global__ __launch_bounds__(128) void msm_accumulate_kernel(const Affine<Fld>* __restrict__ bases,
                                                            const uint32_t* __restrict__ sorted,
                                                            const SegDesc* __restrict__ segs,
                                                            const uint2* __restrict__ offsets, size_t nkeys,
                                                            XYZZ<Fld>* __restrict__ partial) {
  size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nseg = offsets[nkeys].y;
  if (s >= nseg) return;
  SegDesc d = segs[s];
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  for (uint32_t p = d.start; p < d.end; p++) {
    uint32_t e = sorted[p];
    uint32_t idx = e & 0x7fffffffu;
    Affine<Fld> pt = load_elem(bases + idx);
    if (pt.is_identity()) continue;
    Fld y = (e >> 31) ? pt.y.neg() : pt.y;
    acc = xyzz_madd(acc, pt.x, y);
  }
  store_elem(partial + s, acc);
}

template <class Fld>
__global__ __launch_bounds__(128) void msm_finalize_kernel(const XYZZ<Fld>* __restrict__ partial,
                                                          const uint2* __restrict__ offsets, size_t nkeys,
                                                          XYZZ<Fld>* __restrict__ buckets) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nkeys) return;
  uint32_t s0 = offsets[k].y, s1 = offsets[k + 1].y;
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  if (s1 > s0) acc = load_elem(partial + s0);
  for (uint32_t s = s0 + 1; s < s1; s++) acc = xyzz_add_ni(acc, load_elem(partial + s));
  store_elem(buckets + k, acc);
}

// -------------------------------------------------------------------------------------------------- reduce
// Workgroup (w, blk) covers buckets [blk*RED_THREADS*RED_G, ...) of window w and emits
//   S = sum bucket_b,   A = sum (b - base + 1) * bucket_b      (base = first bucket of the workgroup)
template <class Fld>
ZK_D XYZZ<Fld> block_reduce_sum(XYZZ<Fld> v, XYZZ<Fld>* sh) {
  // Control flow is kept wave-uniform around the out-of-line group additions: every lane adds (idle lanes
  // add the identity) and only the store is predicated.
  int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int off = RED_THREADS / 2; off > 0; off >>= 1) {
    XYZZ<Fld> a = sh[tid];
    XYZZ<Fld> b = tid < off ? sh[tid + off] : XYZZ<Fld>::identity();
    XYZZ<Fld> r = xyzz_add_ni(a, b);
    __syncthreads();
    if (tid < off) sh[tid] = r;
    __syncthreads();
  }
  XYZZ<Fld> r = sh[0];
  __syncthreads();
  return r;
}

template <class Fld>
__global__ __launch_bounds__(RED_THREADS) void msm_reduce_kernel(const XYZZ<Fld>* __restrict__ buckets, uint32_t B,
                                                                uint32_t blocks_per_window,
                                                                XYZZ<Fld>* __restrict__ out /* [nwin][bpw][2] */) {
  extern __shared__ uint4 smem_red[];
  XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_red);
  uint32_t w = blockIdx.x / blocks_per_window, blk = blockIdx.x % blocks_per_window;
  uint32_t base = blk * RED_THREADS * RED_G;
  int tid = threadIdx.x;
  const XYZZ<Fld>* wb = buckets + (size_t)w * B;
  // per-lane suffix sums over its RED_G buckets
  XYZZ<Fld> run = XYZZ<Fld>::identity(), acc = XYZZ<Fld>::identity();
  for (int g = RED_G - 1; g >= 0; g--) {
    uint32_t b = base + tid * RED_G + g;
    XYZZ<Fld> bk = XYZZ<Fld>::identity();
    if (b < B) bk = load_elem(wb + b);
    run = xyzz_add_ni(run, bk);
    acc = xyzz_add_ni(acc, run);
  }
  // suffix scan of lane totals across the workgroup: suf[t] = sum_{t' >= t} run[t']
  sh[tid] = run;
  __syncthreads();
  for (int off = 1; off < RED_THREADS; off <<= 1) {
    XYZZ<Fld> a = sh[tid];
    XYZZ<Fld> b = XYZZ<Fld>::identity();
    if (tid + off < RED_THREADS) b = sh[tid + off];
    XYZZ<Fld> tv = xyzz_add_ni(a, b);
    __syncthreads();
    sh[tid] = tv;
    __syncthreads();
  }
  XYZZ<Fld> S = sh[0];
  // sum_t t * run[t] = sum_{j >= 1} suf[j]; the lane weight is t*RED_G, so multiply by RED_G afterwards
  XYZZ<Fld> mine = XYZZ<Fld>::identity();
  if (tid >= 1) mine = sh[tid];
  __syncthreads();
  XYZZ<Fld> T = block_reduce_sum(mine, sh);
  XYZZ<Fld> Asum = block_reduce_sum(acc, sh);
  XYZZ<Fld> TG = T;
  for (int g = 1; g < RED_G; g <<= 1) TG = xyzz_dbl_ni(TG);
  XYZZ<Fld> A = xyzz_add_ni(Asum, TG);
  if (tid == 0) {
    store_elem(out + ((size_t)blockIdx.x) * 2, S);
    store_elem(out + ((size_t)blockIdx.x) * 2 + 1, A);
  }
}

#endif  // __HIPCC__

// ---------------------------------------------------------------------------------------------------- host
template <class Cfg>
class MsmRunner {
 public:
  using FrP = typename Cfg::FrP;
  using Fr = Fp<FrP>;
  using Fq = Fp<typename Cfg::FqP>;
  using Fq2 = Fp2<typename Cfg::FqP>;

  static int pick_c(size_t npts) {
    if (const char* e = getenv("ZK_MSM_C")) {
      int c = atoi(e);
      if (c >= 2 && c <= 20) return c;
    }
    int lg = ilog2(npts ? npts : 1);
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > 17) c = 17;
    return c;
  }

  template <class Fld>
  int run_t(IEngine* eng, const void* bases, const void* scalars, size_t npts, const Fr* coef_d, size_t part_len,
            XYZZ<Fld>* result, hipStream_t st) {
#if defined(__HIPCC__)
    *result = XYZZ<Fld>::identity();
    if (npts == 0) return ZK_OK;
    if (npts >= ((size_t)1 << 31)) return eng->fail(ZK_ERR_BAD_INPUT, "msm too large");
    const int c = pick_c(npts);
    const int nwin = (FrP::BITS + c) / c;          // ceil((BITS+1)/c): room for the signed-digit carry
    const uint32_t B = 1u << (c - 1);
    const size_t nkeys = (size_t)nwin * B;
    const size_t max_sorted = npts * nwin;
    const size_t max_segs = nkeys + max_sorted / MSM_SEG + 1;
    const uint32_t bpw = (B + RED_THREADS * RED_G - 1) / (RED_THREADS * RED_G);
    const size_t iscan_blocks = (nkeys + ISCAN_BLOCK - 1) / ISCAN_BLOCK;

    // workspace layout
    size_t off = 0;
    auto take = [&](size_t bytes) {
      size_t o = off;
      off += (bytes + 255) & ~(size_t)255;
      return o;
    };
    size_t o_counts = take(nkeys * 4), o_cursor = take(nkeys * 4), o_offsets = take((nkeys + 1) * 8),
           o_bt = take(iscan_blocks * 8), o_sorted = take(max_sorted * 4), o_segs = take(max_segs * sizeof(SegDesc)),
           o_partial = take(max_segs * sizeof(XYZZ<Fld>)), o_buckets = take(nkeys * sizeof(XYZZ<Fld>)),
           o_out = take((size_t)nwin * bpw * 2 * sizeof(XYZZ<Fld>));
    hipError_t he = ws_.ensure(off);
    if (he != hipSuccess) return eng->hip_fail(he, "msm workspace");
    char* ws = (char*)ws_.p;
    uint32_t* counts = (uint32_t*)(ws + o_counts);
    uint32_t* cursor = (uint32_t*)(ws + o_cursor);
    uint2* offsets = (uint2*)(ws + o_offsets);
    uint2* bt = (uint2*)(ws + o_bt);
    uint32_t* sorted = (uint32_t*)(ws + o_sorted);
    SegDesc* segs = (SegDesc*)(ws + o_segs);
    XYZZ<Fld>* partial = (XYZZ<Fld>*)(ws + o_partial);
    XYZZ<Fld>* buckets = (XYZZ<Fld>*)(ws + o_buckets);
    XYZZ<Fld>* out = (XYZZ<Fld>*)(ws + o_out);

#define MSM_HIP(x)                                           \
  do {                                                       \
    hipError_t _e = (x);                                     \
    if (_e != hipSuccess) return eng->hip_fail(_e, #x);      \
  } while (0)
    const bool dbg = getenv("ZK_DEBUG_SYNC") != nullptr;
#define MSM_STAGE(name)                                                          \
  do {                                                                           \
    if (dbg) {                                                                   \
      hipError_t _e = hipStreamSynchronize(st);                                  \
      fprintf(stderr, "[zk msm] %s done (%s) npts=%zu c=%d nwin=%d\n", name,     \
              hipGetErrorString(_e), npts, c, nwin);                             \
      if (_e != hipSuccess) return eng->hip_fail(_e, name);                      \
    }                                                                            \
  } while (0)
    MSM_HIP(hipMemsetAsync(counts, 0, nkeys * 4, st));
    dim3 pg((unsigned)((npts + 255) / 256)), pb(256);
    constexpr bool IS_G2 = sizeof(Fld) != sizeof(Fq);
    {
    ProfScope ps_(eng->prof, PROF_MSM_SORT, st, (double)npts);
    msm_digits_kernel<FrP, 0><<<pg, pb, 0, st>>>((const Fr*)scalars, npts, coef_d, part_len ? part_len : npts, c, nwin,
                                                counts, nullptr, nullptr);
    MSM_STAGE("digits/count");
    iscan_block_kernel<<<dim3((unsigned)iscan_blocks), dim3(ISCAN_THREADS), 0, st>>>(counts, nkeys, bt, nullptr,
                                                                                     nullptr, 0);
    iscan_carry_kernel<<<dim3(1), dim3(ISCAN_THREADS), 0, st>>>(bt, iscan_blocks);
    iscan_block_kernel<<<dim3((unsigned)iscan_blocks), dim3(ISCAN_THREADS), 0, st>>>(counts, nkeys, nullptr, bt,
                                                                                     offsets, 1);
    MSM_STAGE("scan");
    msm_expand_kernel<<<dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st>>>(offsets, nkeys, cursor, segs);
    MSM_STAGE("expand");
    msm_digits_kernel<FrP, 1><<<pg, pb, 0, st>>>((const Fr*)scalars, npts, coef_d, part_len ? part_len : npts, c, nwin,
                                                nullptr, cursor, sorted);
    }
    MSM_STAGE("scatter");
    {
    ProfScope ps_(eng->prof, IS_G2 ? PROF_MSM_ACC_G2 : PROF_MSM_ACC_G1, st, (double)npts);
    msm_accumulate_kernel<Fld><<<dim3((unsigned)((max_segs + 127) / 128)), dim3(128), 0, st>>>(
        (const Affine<Fld>*)bases, sorted, segs, offsets, nkeys, partial);
    }
    MSM_STAGE("accumulate");
    {
    ProfScope ps_(eng->prof, PROF_MSM_REDUCE, st, (double)npts);
    msm_finalize_kernel<Fld><<<dim3((unsigned)((nkeys + 127) / 128)), dim3(128), 0, st>>>(partial, offsets, nkeys,
                                                                                          buckets);
    MSM_STAGE("finalize");
    size_t red_lds = RED_THREADS * sizeof(XYZZ<Fld>);
    static bool attr_set = false;
    if (!attr_set && red_lds > 48 * 1024) {
      MSM_HIP(hipFuncSetAttribute((const void*)msm_reduce_kernel<Fld>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)red_lds));
      attr_set = true;
    }
    msm_reduce_kernel<Fld><<<dim3((unsigned)(nwin * bpw)), dim3(RED_THREADS), red_lds, st>>>(buckets, B, bpw, out);
    }
    MSM_HIP(hipGetLastError());
    MSM_STAGE("reduce");
    std::vector<XYZZ<Fld>> h((size_t)nwin * bpw * 2);
    MSM_HIP(hipMemcpyAsync(h.data(), out, h.size() * sizeof(XYZZ<Fld>), hipMemcpyDeviceToHost, st));
    MSM_HIP(hipStreamSynchronize(st));
#undef MSM_HIP
#undef MSM_STAGE
    // host: window value = sum_blk (A_blk + blk*RED_THREADS*RED_G * S_blk); fold windows high -> low
    XYZZ<Fld> total = XYZZ<Fld>::identity();
    for (int w = nwin - 1; w >= 0; w--) {
      for (int i = 0; i < c; i++) total = xyzz_dbl_ni(total);
      XYZZ<Fld> wsum = XYZZ<Fld>::identity();
      XYZZ<Fld> run = XYZZ<Fld>::identity(), wt = XYZZ<Fld>::identity();
      for (int blk = (int)bpw - 1; blk >= 0; blk--) {
        const XYZZ<Fld>& S = h[((size_t)w * bpw + blk) * 2];
        const XYZZ<Fld>& A = h[((size_t)w * bpw + blk) * 2 + 1];
        wsum = xyzz_add_ni(wsum, A);
        if (blk >= 1) {
          run = xyzz_add_ni(run, S);      // sum_{blk' >= blk} S
          wt = xyzz_add_ni(wt, run);      // accumulates sum blk * S_blk
        }
      }
      if (bpw > 1) wsum = xyzz_add_ni(wsum, xyzz_mul_small(wt, (uint64_t)RED_THREADS * RED_G));
      total = xyzz_add_ni(total, wsum);
    }
    *result = total;
    return ZK_OK;
#else
    (void)eng; (void)bases; (void)scalars; (void)npts; (void)coef_d; (void)part_len; (void)result; (void)st;
    return ZK_ERR_GENERIC;
#endif
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
      if (!Cfg::HAS_G2) return eng->fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
      XYZZ<Fq2> r;
      int rc = run_t<Fq2>(eng, bases, scalars, npts, coef_d, part_len, &r, st);
      if (rc) return rc;
      write_jacobian(out, r);
      return ZK_OK;
    }
    return eng->fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
  }

  // d_msm for all n parties on this device (dmsm/mod.rs:59-102), fused (DESIGN.md "d_msm"):
  //   king output = sum_k unpack2(c_shares)[k] = sum_p coef_p * (msm_p + in_mask_p),  coef_p = sum_k U2[k][p]
  // and sum_p coef_p * msm_p is ONE msm over the n*len points with scalars pre-multiplied by coef_p; the
  // n in-mask points ride along as extra bases with scalars coef_p.
  // sum_p coef_p * (msm_p + in_mask_p): what the king reconstructs and sums (dmsm/mod.rs:85-86)
  template <class Fld>
  int d_msm_sum_t(IEngine* eng, const void* bases, const void* scalars, size_t len, const void* in_mask,
                  XYZZ<Fld>* result, hipStream_t st) {
    const int n = eng->n;
    XYZZ<Fld> r;
    int rc = run_t<Fld>(eng, bases, scalars, (size_t)n * len, coef_d_, len, &r, st);
    if (rc) return rc;
    if (in_mask) {
      // the n in-mask points contribute sum_p coef_p * mask_p (n host scalar multiplications)
      const Jacobian<Fld>* jm = (const Jacobian<Fld>*)in_mask;
      std::vector<Affine<Fld>> aff(n);
      for (int p = 0; p < n; p++) aff[p] = xyzz_to_affine(jacobian_to_xyzz(jm[p]));
      std::vector<char> mt = host_lincomb<Fld>(aff);
      r = xyzz_add_ni(r, *reinterpret_cast<XYZZ<Fld>*>(mt.data()));
    }
    *result = r;
    return ZK_OK;
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

  template <class Fld>
  std::vector<char> host_lincomb(const std::vector<Affine<Fld>>& pts) {
    XYZZ<Fld> acc = XYZZ<Fld>::identity();
    for (size_t p = 0; p < pts.size(); p++) {
      Fr k = coef_h_[p].from_mont();
      XYZZ<Fld> base = XYZZ<Fld>::from_affine(pts[p]);
      XYZZ<Fld> r = XYZZ<Fld>::identity();
      for (int i = FrP::N - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
          r = xyzz_dbl_ni(r);
          if ((k.v[i] >> b) & 1) r = xyzz_add_ni(r, base);
        }
      acc = xyzz_add_ni(acc, r);
    }
    std::vector<char> out(sizeof(XYZZ<Fld>));
    memcpy(out.data(), &acc, sizeof(acc));
    return out;
  }

  int d_msm(IEngine* eng, int group, const void* bases, const void* scalars, size_t len, const void* in_mask,
            const void* out_mask, void* out, hipStream_t st) {
    if (!out) return eng->fail(ZK_ERR_BAD_INPUT, "null output");
    if (len && (!bases || !scalars)) return eng->fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (!coef_d_) return eng->fail(ZK_ERR_GENERIC, "d_msm coefficients not initialised");
    if (group == ZK_G1) return d_msm_t<Fq>(eng, bases, scalars, len, in_mask, out_mask, out, st);
    if (group == ZK_G2) {
      if (!Cfg::HAS_G2) return eng->fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
      return d_msm_t<Fq2>(eng, bases, scalars, len, in_mask, out_mask, out, st);
    }
    return eng->fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
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

  DevBuf ws_;
  Fr* coef_d_ = nullptr;
  std::vector<Fr> coef_h_;
};

}  // namespace zk
