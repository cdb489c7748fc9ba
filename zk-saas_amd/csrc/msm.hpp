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
//   4. (gone)       : round 2 cut buckets into segments and ordered them by length; see "balanced partition"
//   5. accumulate   : equal contiguous ranges of the sorted entries, one per lane, mixed XYZZ additions   <- dominant
//   6. finalize     : buckets that straddle a lane boundary are summed from the lanes' edge partials
//   7. reduce       : sum_b (b+1) * bucket_b per window with per-lane suffix sums + an LDS tree across the
//                     workgroup ("wavefront-level bucket reduction"); one (S, A) pair per workgroup
//   8. host         : combines the few (S, A) pairs per window and folds windows high -> low
//                     (254 sequential doublings: latency-bound on any one lane, cheap on a CPU core).
#pragma once
#include <atomic>
#include <future>
#include <map>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "ec.hpp"
#include "engine.hpp"
#include "ntt.hpp"
#include "quad.hpp"

namespace zk {
#if defined(__HIPCC__)

constexpr size_t MSM_RANGE_MIN = 20, MSM_RANGE = 64;        // entries per accumulate lane: fewest (small MSMs), most
                                                            // (SHA-256 proof, same box, round 5: 14 -> 622-626, 26 -> 629-630
                                                            // against 646-649 proofs/s at 20)
constexpr size_t MSM_RANGE_MIN_G2 = 16, MSM_RANGE_G2 = 64;  // ... per lane quad of the extension-field kernel (32 until
                                                            // round 4; C5: 1.303 s at 32, 1.286 at 64, 1.280 at 128)
constexpr uint32_t FIN_SEQ = 16;      // a bucket spread over more accumulate lanes than this is summed by a workgroup
constexpr int MSM_WS = 30;            // independent workspaces: 6 per proof in flight (concurrent MSMs on separate
                                      // streams) + 6 per batch of proofs in flight (zk_groth16_prove_batch)

// Sort-stage arrays of a launch over TWO base vectors with per-vector sorts (their identity bases differ) live in two
// copies of one workspace region; blockIdx.y picks the copy: every sort-stage kernel shifts its array pointers by
// blockIdx.y * ys bytes (ys = 0: one sort shared by both vectors).
// (pointer arithmetic on the pointer itself, never through an integer: a pointer rebuilt from an integer is a GENERIC
// pointer to the compiler, its loads become flat_load and every wait on one of them waits for all memory traffic of the
// wave -- found in the accumulate kernel's ISA, where it serialised the index load, the gather and the bucket flush)
template <class T>
__device__ __forceinline__ T* zk_yshift(T* p, size_t bytes) {
  using C = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return p ? reinterpret_cast<T*>(reinterpret_cast<C*>(p) + bytes) : p;
}
#define ZK_YSHIFT(p) p = zk_yshift(p, (size_t)blockIdx.y * ys)

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

// Identity bases contribute nothing and are left out of the sort: the FIRST sort-stage kernel (the histogram pass, which
// also writes the canonical scalars the later passes read) looks at the base itself -- the identity sentinel (0, 0) is an
// affine point whose every byte is zero -- and gives the point a zero scalar.  A Groth16 CRS is full of identities: b_query
// holds one for every wire that no B-row mentions (59 % of the SHA-256 circuit's wires; half of the PACKED shares), and a
// lane that loads an identity idles while its wave adds (measured: 49 % of the lanes of the G2 accumulate active).
// Until round 6 a kernel of its own wrote a bit mask first (msm_skip_mask_kernel): 4 us alone, but on the chain of the
// table-free U-MSM inside a proof its span read 166-195 us (its 512 four-wave workgroups waiting for wave slots among the
// bulk accumulate waves: profiles/r06_c4tf_timeline.txt).  One launch, one dependent boundary and one buffer less now --
// and NO change in proofs/s (same-box A/B against the mask kernel, option "msm_skip_kernel": table-free 450-452 vs 434-456,
// with tables 645-652 vs 645-651, profiles/r06_skipfold_ab.txt): the chain waits for the chip elsewhere instead.
struct MsmBaseId {
  const void* b0 = nullptr;   // nullptr: no base of the launch is the identity (registered vectors know: zk_msm_precompute)
  const void* b1 = nullptr;   // second base vector of the launch (own sort when ys != 0: blockIdx.y picks the vector)
  uint32_t elem16 = 0;        // bytes per affine point / 16
  const uint32_t* skip = nullptr;   // A/B only (zk_ctx_set_option "msm_skip_kernel"): round 5's bit mask, written by a kernel of its own
  size_t skip_ys = 0;               // ... its stride between the two sorts, in words
};
ZK_D bool msm_base_zero(const void* bases, uint32_t elem16, uint32_t i) {
  const uint4* p = reinterpret_cast<const uint4*>(bases) + (size_t)i * elem16;
  uint4 a = p[0];
  for (uint32_t k = 1; k < elem16; k++) {
    const uint4 b = p[k];
    a.x |= b.x, a.y |= b.y, a.z |= b.z, a.w |= b.w;
  }
  return (a.x | a.y | a.z | a.w) == 0u;
}
// ys != 0 (grid.y = 2): one sort per base vector, each skips its own identities; ys == 0: one sort, a point is skipped when
// EVERY vector holds the identity there
ZK_D bool msm_base_is_identity(const MsmBaseId& id, uint32_t i, size_t ys) {
  if (id.skip) return ((id.skip[(size_t)blockIdx.y * id.skip_ys + (i >> 5)] >> (i & 31)) & 1u) != 0;
  if (!id.b0) return false;
  if (ys) return msm_base_zero(blockIdx.y ? id.b1 : id.b0, id.elem16, i);
  return msm_base_zero(id.b0, id.elem16, i) && (!id.b1 || msm_base_zero(id.b1, id.elem16, i));
}

// -------------------------------------------------------------------------------------------------- digits
// pass 0: histogram; pass 1: scatter.  coef (optional): per-part multiplier, part = i / part_len.
template <class FrP, int PASS>
__global__ void msm_digits_kernel(MsmScalars<Fp<FrP>> sc, const Fp<FrP>* __restrict__ coef,
                                  size_t part_len, int c, int nwin, int wide /* windows [0, wide) have c bits, the rest
                                  c-1 */, uint32_t pre_stride /* fixed-base table: rows of this many points, one per
                                  window; all windows share ONE bucket set; 0 = no table */, uint32_t pre_off,
                                  uint32_t* __restrict__ counts /* [nsets*B] */,
                                  uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted,
                                  MsmBaseId bid /* identity bases get a zero scalar in pass 0 */,
                                  Fp<FrP>* __restrict__ canon /* canonical scalars: written by pass 0, read by pass 1 */,
                                  size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(counts);
  ZK_YSHIFT(cursor);
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(canon);
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)sc.npts * sc.nb) return;
  const uint32_t vb = (uint32_t)(g / sc.npts), i = (uint32_t)(g % sc.npts);
  Fp<FrP> s;
  if (PASS == 0) {
    if (msm_base_is_identity(bid, i, ys)) {
      store_elem(canon + g, Fp<FrP>::zero());
      return;
    }
    s = load_elem(sc.p[vb] + i);
    if (coef) s = s * coef[i / part_len];
    s = s.from_mont();
    store_elem(canon + g, s);
  } else {
    s = load_elem(canon + g);
    if (s.is_zero()) return;
  }
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

// Kernels whose dynamic LDS exceeds the default limit: raise the limit once per (kernel, device)
inline hipError_t msm_lds_attr(const void* fn, size_t bytes, int device) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  std::lock_guard<std::mutex> g(mu);
  size_t& v = done[std::make_pair(fn, device)];
  if (v >= bytes) return hipSuccess;
  const size_t want = std::max(bytes, (size_t)160 * 1024);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  if (e == hipSuccess) v = want;
  return e;
}

// Bit mask of the identity bases of a vector (bit i of `skip`).  Since round 6 only zk_msm_precompute runs it, once per
// registered vector, to learn whether the vector holds an identity AT ALL (MsmTable::any_identity); the sorts look at the
// bases themselves (MsmBaseId above).
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

// `cap` = lanes the chip holds at once for this kernel: when the entries make one round of waves or more, the range
// length is stretched so that the launch is a WHOLE number of rounds -- a last round that is 5 % full costs as much as a
// full one (measured: 3.05 rounds of a G2 accumulate ran 25 % slower than 2.9)
ZK_D uint32_t msm_range_len(uint32_t entries, uint32_t nlanes, uint32_t tmin, uint32_t cap) {
  uint32_t T = (entries + nlanes - 1) / nlanes;
  if (T < tmin) T = tmin;
  const uint64_t round_entries = (uint64_t)cap * T;
  uint32_t rounds = (uint32_t)(entries / round_entries);
  if (rounds == 1) {
    // exactly one round is the slow case (79 against 100+ G multiplications/s): two rounds of shorter ranges instead,
    // unless that makes them shorter than 12 entries
    // (only when the launch has the lanes for it: ranges shorter than entries / nlanes would leave entries uncovered --
    // msm_pick_lanes sizes such launches at 2 cap lanes)
    const uint32_t T2 = (uint32_t)((entries + 2ull * cap - 1) / (2ull * cap));
    if (T2 >= 12 && (uint64_t)T2 * nlanes >= entries) return T2;
  }
  if (rounds >= 1) {
    const uint64_t lanes = (uint64_t)cap * rounds;
    const uint32_t Tr = (uint32_t)((entries + lanes - 1) / lanes);
    if (Tr > T) T = Tr;
  }
  return T;
}

// -------------------------------------------------------------------------------------------------- big sort
// For multi-million-point MSMs the two atomics-per-(point, window) passes above dominate (measured at 8 x 2^20
// points: 4.7 ms histogram + 12.2 ms scatter against 15.6 ms of accumulate).  The big-sort path is a two-level
// counting sort whose global atomics are per (workgroup tile, bin) instead of per entry AND WHOSE GLOBAL STORES ARE
// COALESCED RUNS (round 4): both levels sort their tile in LDS first and then stream it out, consecutive lanes to
// consecutive addresses.  Round 3 stored every entry from the lane that produced it -- 64 scattered 4- or 8-byte
// stores per wave instruction, which the L2 does not merge: the counters showed 3.4x (level 1) and 5.7x (level 2) the
// algorithmic bytes going to memory (profiles/r03_c3_pmc_hbm.json).
//   bin = (bucket set, top hi_bits of the bucket index)         nbins = nsets * 2^hi_bits  (at most 8192)
//   hist    : per tile of THR*PPT points of ONE scalar vector, LDS histogram over the bins -> one global add per
//             non-empty bin; the canonical scalar (times the party coefficient) is computed here once and cached; the
//             LAST workgroup to finish (ticket) scans the bin totals -> bin_base[], bin_cursor[]
//   scatter : same tile, scalars in registers.  Counts its entries per bin, reserves its range in every bin with one
//             global add, then goes over the windows in ROUNDS of `wgroup` windows (bins are set-major, so without a
//             fixed-base table the bins of a round are contiguous and so are the tile-sorted positions of its
//             entries; with a table all windows share the bins and the whole tile is one round): entries are ranked
//             into an LDS stage in bin order and the stage is streamed to tmp[] -- runs of
//             (tile points / 2^hi_bits) entries.  An entry is ONE 32-bit word {point index, sign, low bucket bits}
//             when that fits (2^20 points per party x 8 parties with 17-bit windows: 23 + 1 + 8), otherwise a word
//             plus a 16-bit low part in a second array
//   binsort : one workgroup per bin: counts of the 2^lo_bits low-bit buckets, LDS scan -> offsets[] of the bin's
//             keys (= bin_base + local prefix: the separate three-launch scan over all keys is gone) and the first
//             bucket of every accumulate lane that starts inside the bin (msm_lane_start_kernel's job on this path);
//             then chunks of THR*EPT entries are ranked into an LDS stage in bucket order and streamed to sorted[]
// The order inside a bucket is arbitrary, as before; bucket sums do not depend on it.
constexpr int BIG_HI = 8;                 // top bucket bits of a bin when nothing else decides (msm_big_hi)
constexpr int BIG_MAX_BINS = 8192;
constexpr int BIG_THREADS = 256;          // histogram tiles; scatter / binsort: 256 (small launches) or 1024
constexpr int BIG_PTS_PER_THREAD = 8;     // most points per thread (scalars are held in registers by the scatter)
constexpr int BIG_EPT = 16;               // entries per thread and chunk of the bin sort
constexpr size_t BIG_LDS_MAX = 156 * 1024;
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
// one step of the same walk with its state (shifted scalar, carry) kept by the caller: the scatter kernel goes over
// the windows in rounds and continues where the previous round stopped
template <class FrP>
__device__ __forceinline__ bool msm_next_digit(Fp<FrP>& s, uint32_t& carry, int cw, uint32_t* b, uint32_t* neg) {
  constexpr int N = FrP::N;
  const uint32_t val = s.v[0] & ((1u << cw) - 1);
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
  if (d == 0) return false;
  *neg = d < 0 ? 1u : 0u;
  *b = (uint32_t)(d < 0 ? -d : d) - 1;
  return true;
}

// exclusive scan of one value per thread over the workgroup: wave shuffles + one LDS round over the wave totals
template <int THR>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* sh /* THR / 64 words */, uint32_t* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THR / 64; i++) {
    const uint32_t t = sh[i];
    pre += i < wv ? t : 0u;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return pre + inc - v;
}

// Layout of the bins block (uint32): counts[nbins], ticket, base[nbins + 1], cursor[nbins]
ZK_HD size_t msm_bins_words(size_t nbins) { return 3 * nbins + 2; }

// scalar of (vector vb of the batch, point i); identity bases give zero (no digit, no entry)
template <class FrP>
__device__ __forceinline__ Fp<FrP> msm_canon_scalar(const MsmScalars<Fp<FrP>>& sc, const Fp<FrP>* coef, size_t part_len,
                                                    uint32_t vb, uint32_t i, const MsmBaseId& bid, size_t ys) {
  Fp<FrP> s = Fp<FrP>::zero();
  if (!msm_base_is_identity(bid, i, ys)) {
    s = load_elem(sc.p[vb] + i);
    if (coef) s = s * coef[i / part_len];
    s = s.from_mont();
  }
  return s;
}

// hist: grid.x = tiles_per_vec * nb tiles of BIG_THREADS * ppt points, each inside ONE scalar vector
template <class FrP>
__global__ __launch_bounds__(BIG_THREADS) void msm_hist_kernel(MsmScalars<Fp<FrP>> sc, const Fp<FrP>* __restrict__ coef,
                                                               size_t part_len, int c, int nwin, int wide, int hi_bits,
                                                               int lo_bits, int ppt, uint32_t tiles_per_vec,
                                                               uint32_t wmask /* 0: fixed-base table, all windows
                                                               share one bucket set; ~0: one set per window */,
                                                               int w_begin /* the launch sorts windows [w_begin, nwin)
                                                               only (a window group of a split MSM): set = w - w_begin */,
                                                               uint32_t* __restrict__ bins,
                                                               MsmBaseId bid,
                                                               Fp<FrP>* __restrict__ canon,
                                                               uint16_t* __restrict__ tile_counts /* optional
                                                               [tiles][bins of one vector]: the staged scatter, whose
                                                               tiles these then are, does not count again */,
                                                               size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bins);
  ZK_YSHIFT(canon);
  ZK_YSHIFT(tile_counts);
  extern __shared__ uint32_t big_lds[];
  const uint32_t nbl = sc.sets_per << hi_bits;                 // bins of one scalar vector
  const uint32_t nbins = nbl * sc.nb;
  const uint32_t vb = blockIdx.x / tiles_per_vec;
  const uint32_t pt0 = (blockIdx.x % tiles_per_vec) * (uint32_t)(BIG_THREADS * ppt);
  for (uint32_t b = threadIdx.x; b < nbl; b += BIG_THREADS) big_lds[b] = 0;
  __syncthreads();
  // the identity tests of the thread's points first: independent loads, in flight together (inside the loop below each
  // would sit in front of its scalar's load: two dependent memory latencies per point)
  uint32_t idmask = 0;
  if (bid.b0 || bid.skip)
    for (int k = 0; k < ppt; k++) {
      const uint32_t i = pt0 + (uint32_t)k * BIG_THREADS + threadIdx.x;
      if (i < sc.npts && msm_base_is_identity(bid, i, ys)) idmask |= 1u << k;
    }
  const MsmBaseId no_id{};
  for (int k = 0; k < ppt; k++) {
    const uint32_t i = pt0 + (uint32_t)k * BIG_THREADS + threadIdx.x;
    if (i >= sc.npts) break;
    const Fp<FrP> s = ((idmask >> k) & 1u) ? Fp<FrP>::zero() : msm_canon_scalar<FrP>(sc, coef, part_len, vb, i, no_id, ys);
    store_elem(canon + (size_t)vb * sc.npts + i, s);
    msm_for_each_digit<FrP>(s, c, nwin, wide, [&](int w, uint32_t b, uint32_t) {
      if (w >= w_begin) atomicAdd(&big_lds[(((uint32_t)(w - w_begin) & wmask) << hi_bits) | (b >> lo_bits)], 1u);
    });
  }
  __syncthreads();
  uint32_t* bin_counts = bins + (size_t)vb * nbl;
  if (tile_counts)
    for (uint32_t b = threadIdx.x; b < nbl; b += BIG_THREADS) tile_counts[(size_t)blockIdx.x * nbl + b] = (uint16_t)big_lds[b];
  uint32_t seen = 0;                 // RETURNING adds: a value that has come back is an add that has been performed
  for (uint32_t b0 = threadIdx.x; b0 < nbl; b0 += 8 * BIG_THREADS) {       // eight in flight, then their values
    uint32_t r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t b = b0 + (uint32_t)u * BIG_THREADS;
      const uint32_t n = b < nbl ? big_lds[b] : 0u;
      r[u] = n ? atomicAdd(&bin_counts[b], n) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) seen += r[u];
  }
  asm volatile("" ::"v"(seen));      // ... and this makes the wave wait for them before the barrier below
  // the last workgroup of this sort to get here scans the bin totals (saves the one-workgroup scan launch that sat on
  // every MSM's dependent chain).  The counts were added by device-scope atomics; they are read back the same way (the
  // per-XCD L2s are not coherent with each other)
  // (no __threadfence here: at device scope it writes back and invalidates the whole L2 -- measured 188 us for this
  // kernel and every concurrent kernel slowed down.  Every wave has its adds' return values before the barrier, the
  // ticket is taken after it, and the scanning workgroup reads with device-scope loads)
  __shared__ uint32_t last_sh;
  __syncthreads();
  // The ticket stays a RELAXED add.  ADVICE r4 asked for acq_rel at agent scope (the formal release / acquire edge between
  // the workgroups' count adds and the last workgroup's reads); built and measured in round 5: every workgroup's release
  // writes back its XCD's L2, and with the three witness sorts of a proof running side by side msm_hist went from 22-57 to
  // 101-112 us per launch (profiles/r05_c4_timeline.txt of that build) -- on every MSM's dependent chain.  What makes the
  // relaxed form correct on gfx950: the counts are device-scope atomic RMWs performed at the L2 / memory side (never in a
  // CU's L1), a wave has the RETURN VALUE of each of its adds before it reaches the barrier (so the adds have been
  // performed), the ticket is taken after the barrier, and the scanning workgroup reads the counts with device-scope
  // atomic loads.  -DZK_HIST_TICKET_ACQ_REL=1 builds the fenced form; tests/test_gpu_msm.py::
  // test_concurrent_sorts_stress compares hundreds of concurrent sorts' results with serial ones.
#if defined(ZK_HIST_TICKET_ACQ_REL) && ZK_HIST_TICKET_ACQ_REL
  if (threadIdx.x == 0)
    last_sh = __hip_atomic_fetch_add(bins + nbins, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
#else
  if (threadIdx.x == 0) last_sh = atomicAdd(bins + nbins, 1u) == gridDim.x - 1 ? 1u : 0u;
#endif
  __syncthreads();
  if (!last_sh) return;
  // the acquire half of the edge is free (one workgroup per sort invalidates its caches); the release half is what cost
  // 3.8 % of a proof and stays the hardware argument above (stated in include/zksaas.h, "Memory-model note")
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  uint32_t* bin_base = bins + nbins + 1;
  uint32_t* bin_cursor = bin_base + nbins + 1;
  // all counts into LDS first (independent loads, all in flight), then the scan reads LDS
  uint32_t* cnt = big_lds + BIG_THREADS / 64;
  for (uint32_t b = threadIdx.x; b < nbins; b += BIG_THREADS)
    cnt[b] = __hip_atomic_load(bins + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t per = (nbins + BIG_THREADS - 1) / BIG_THREADS;
  const uint32_t b0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; j++)
    if (b0 + j < nbins) sum += cnt[b0 + j];
  uint32_t tot;
  uint32_t run = block_scan_excl<BIG_THREADS>(sum, big_lds, &tot);
  for (uint32_t j = 0; j < per; j++)
    if (b0 + j < nbins) {
      bin_base[b0 + j] = run;
      bin_cursor[b0 + j] = run;
      run += cnt[b0 + j];
    }
  if (threadIdx.x == 0) bin_base[nbins] = tot;
}

// Small launches (everything stays in L2 / Infinity Cache; what counts is the latency of a short kernel that has to find
// room among the accumulate waves of the other MSMs in flight: few registers, little LDS): entries go straight from the
// lane that produced them to tmp[]
template <class FrP, bool WIDE>
__global__ __launch_bounds__(BIG_THREADS) void msm_scatter_direct_kernel(MsmScalars<Fp<FrP>> sc, int c, int nwin, int wide,
                                                                         int hi_bits, int lo_bits, int ppt,
                                                                         uint32_t tiles_per_vec, uint32_t wmask, int w_begin,
                                                                         uint32_t pre_stride, uint32_t pre_off,
                                                                         int idx_bits, uint32_t* __restrict__ bins,
                                                                         uint32_t* __restrict__ tmp,
                                                                         uint16_t* __restrict__ tmp_lo,
                                                                         const Fp<FrP>* __restrict__ canon, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bins);
  ZK_YSHIFT(tmp);
  ZK_YSHIFT(tmp_lo);
  ZK_YSHIFT(canon);
  extern __shared__ uint32_t big_lds[];
  const uint32_t nbl = sc.sets_per << hi_bits;
  const uint32_t nbins = nbl * sc.nb;
  const uint32_t vb = blockIdx.x / tiles_per_vec;
  const uint32_t pt0 = (blockIdx.x % tiles_per_vec) * (uint32_t)(BIG_THREADS * ppt);
  uint32_t* bin_cursor = bins + 2 * (size_t)nbins + 2 + (size_t)vb * nbl;
  uint32_t* cnt = big_lds;            // per-bin count of this tile, then the running local rank
  uint32_t* gbase = big_lds + nbl;    // start of this tile's range inside the bin
  const Fp<FrP>* __restrict__ my = canon + (size_t)vb * sc.npts;
  for (uint32_t b = threadIdx.x; b < nbl; b += BIG_THREADS) cnt[b] = 0;
  __syncthreads();
  for (int k = 0; k < ppt; k++) {
    const uint32_t i = pt0 + (uint32_t)k * BIG_THREADS + threadIdx.x;
    if (i >= sc.npts) break;
    msm_for_each_digit<FrP>(load_elem(my + i), c, nwin, wide, [&](int w, uint32_t b, uint32_t) {
      if (w >= w_begin) atomicAdd(&cnt[(((uint32_t)(w - w_begin) & wmask) << hi_bits) | (b >> lo_bits)], 1u);
    });
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nbl; b += BIG_THREADS) {
    const uint32_t n = cnt[b];
    gbase[b] = n ? atomicAdd(&bin_cursor[b], n) : 0u;
    cnt[b] = 0;
  }
  __syncthreads();
  const uint32_t lo_mask = (1u << lo_bits) - 1;
  for (int k = 0; k < ppt; k++) {
    const uint32_t i = pt0 + (uint32_t)k * BIG_THREADS + threadIdx.x;
    if (i >= sc.npts) break;
    msm_for_each_digit<FrP>(load_elem(my + i), c, nwin, wide, [&](int w, uint32_t b, uint32_t neg) {
      if (w < w_begin) return;
      const uint32_t lbin = (((uint32_t)(w - w_begin) & wmask) << hi_bits) | (b >> lo_bits);
      const uint32_t dst = gbase[lbin] + atomicAdd(&cnt[lbin], 1u);
      const uint32_t idx = pre_stride ? (uint32_t)w * pre_stride + pre_off + i : i;
      if (WIDE) {
        tmp[dst] = idx | (neg << 31);
        tmp_lo[dst] = (uint16_t)(b & lo_mask);
      } else {
        tmp[dst] = idx | (neg << idx_bits) | ((b & lo_mask) << (idx_bits + 1));
      }
    });
  }
}

// Entry formats of tmp[]: packed = {index : idx_bits, sign : 1, low bucket bits}; wide = {index | sign << 31} + 16-bit low part
template <class FrP, int THR, int PPT, bool WIDE>
__global__ __launch_bounds__(THR) void msm_scatter_kernel(MsmScalars<Fp<FrP>> sc, int c, int nwin, int wide, int hi_bits,
                                                          int lo_bits, uint32_t tiles_per_vec, uint32_t wmask, int w_begin,
                                                          int wgroup /* windows per round */, uint32_t pre_stride,
                                                          uint32_t pre_off, int idx_bits, uint32_t stage_cap,
                                                          uint32_t* __restrict__ bins, uint32_t* __restrict__ tmp,
                                                          uint16_t* __restrict__ tmp_lo,
                                                          const Fp<FrP>* __restrict__ canon,
                                                          const uint16_t* __restrict__ tile_counts, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bins);
  ZK_YSHIFT(tmp);
  ZK_YSHIFT(tmp_lo);
  ZK_YSHIFT(canon);
  ZK_YSHIFT(tile_counts);
  extern __shared__ uint32_t big_lds[];
  const uint32_t nbl = sc.sets_per << hi_bits;
  const uint32_t nbins = nbl * sc.nb;
  const uint32_t vb = blockIdx.x / tiles_per_vec;
  const uint32_t pt0 = (blockIdx.x % tiles_per_vec) * (uint32_t)(THR * PPT);
  uint32_t* bin_cursor = bins + 2 * (size_t)nbins + 2 + (size_t)vb * nbl;
  uint32_t* cur = big_lds;                    // [nbl] running tile-local position of the next entry of the bin
  uint32_t* dlt = cur + nbl;                  // [nbl] start of the tile's run inside the bin (global) - tile-local start
  uint32_t* scr = dlt + nbl;                  // [THR / 64] scan scratch
  uint32_t* stage = scr + THR / 64;           // [stage_cap]
  uint16_t* sbin = reinterpret_cast<uint16_t*>(stage + stage_cap);     // [stage_cap] bin of a staged entry - first of round
  uint16_t* slo = sbin + stage_cap;           // [stage_cap] (wide format)
  const Fp<FrP>* __restrict__ my = canon + (size_t)vb * sc.npts;
  if (tile_counts) {
    // ---- the histogram pass counted this very tile (same tiling): take its counts
    for (uint32_t b = threadIdx.x; b < nbl; b += THR) cur[b] = tile_counts[(size_t)blockIdx.x * nbl + b];
  } else {
    for (uint32_t b = threadIdx.x; b < nbl; b += THR) cur[b] = 0;
    __syncthreads();
    // ---- count
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const uint32_t i = pt0 + (uint32_t)k * THR + threadIdx.x;
      if (i < sc.npts)
        msm_for_each_digit<FrP>(load_elem(my + i), c, nwin, wide, [&](int w, uint32_t b, uint32_t) {
          if (w >= w_begin) atomicAdd(&cur[(((uint32_t)(w - w_begin) & wmask) << hi_bits) | (b >> lo_bits)], 1u);
        });
    }
  }
  __syncthreads();
  // ---- reserve the tile's range in every bin; cur <- exclusive scan (tile-sorted position of the bin's first entry)
  uint32_t tile_total;
  {
    const uint32_t per = (nbl + THR - 1) / THR;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++)
      if (b0 + j < nbl) sum += cur[b0 + j];
    uint32_t run = block_scan_excl<THR>(sum, scr, &tile_total);
    for (uint32_t j = 0; j < per; j++)
      if (b0 + j < nbl) {
        const uint32_t n = cur[b0 + j];
        const uint32_t g = n ? atomicAdd(&bin_cursor[b0 + j], n) : 0u;
        dlt[b0 + j] = g - run;
        cur[b0 + j] = run;
        run += n;
      }
  }
  __syncthreads();
  // ---- rounds of windows: rank into the stage, stream the stage out
  Fp<FrP> s[PPT];
  uint32_t carry[PPT];
#pragma unroll
  for (int k = 0; k < PPT; k++) {
    const uint32_t i = pt0 + (uint32_t)k * THR + threadIdx.x;
    s[k] = i < sc.npts ? load_elem(my + i) : Fp<FrP>::zero();
    carry[k] = 0;
    for (int w = 0; w < w_begin; w++) {          // the windows of another group: only their carry matters here
      uint32_t b, neg;
      (void)msm_next_digit<FrP>(s[k], carry[k], w < wide ? c : c - 1, &b, &neg);
    }
  }
  const uint32_t lo_mask = (1u << lo_bits) - 1;
  for (int w0 = w_begin; w0 < nwin; w0 += wgroup) {
    const int w1 = w0 + wgroup < nwin ? w0 + wgroup : nwin;
    const uint32_t lb_lo = ((uint32_t)(w0 - w_begin) & wmask) << hi_bits,
                   lb_hi = ((((uint32_t)(w1 - w_begin) - 1) & wmask) + 1) << hi_bits;
    const uint32_t r0 = cur[lb_lo], r1 = lb_hi < nbl ? cur[lb_hi] : tile_total;     // untouched so far: bins of this round
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const uint32_t i = pt0 + (uint32_t)k * THR + threadIdx.x;
      for (int w = w0; w < w1; w++) {
        uint32_t b, neg;
        if (!msm_next_digit<FrP>(s[k], carry[k], w < wide ? c : c - 1, &b, &neg)) continue;
        const uint32_t lbin = (((uint32_t)(w - w_begin) & wmask) << hi_bits) | (b >> lo_bits);
        const uint32_t j = atomicAdd(&cur[lbin], 1u) - r0;
        const uint32_t idx = pre_stride ? (uint32_t)w * pre_stride + pre_off + i : i;
        if (WIDE) {
          stage[j] = idx | (neg << 31);
          slo[j] = (uint16_t)(b & lo_mask);
        } else {
          stage[j] = idx | (neg << idx_bits) | ((b & lo_mask) << (idx_bits + 1));
        }
        sbin[j] = (uint16_t)(lbin - lb_lo);
      }
    }
    __syncthreads();
    const uint32_t nr = r1 - r0;
    for (uint32_t j = threadIdx.x; j < nr; j += THR) {
      const uint32_t dst = dlt[lb_lo + sbin[j]] + r0 + j;
      tmp[dst] = stage[j];
      if (WIDE) tmp_lo[dst] = slo[j];
    }
    __syncthreads();
  }
}

// one workgroup per bin; the bin's keys are [key0, key0 + 2^lo_bits) in the (set-major) key order
template <int THR, bool WIDE, bool STAGED>
__global__ __launch_bounds__(THR) void msm_binsort_kernel(const uint32_t* __restrict__ tmp,
                                                          const uint16_t* __restrict__ tmp_lo,
                                                          const uint32_t* __restrict__ bins, uint32_t nbins, int hi_bits,
                                                          int lo_bits, uint32_t keys_per_set_log2, int idx_bits,
                                                          uint32_t nkeys, uint32_t nlanes, uint32_t tmin, uint32_t cap,
                                                          uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted,
                                                          uint32_t* __restrict__ k0, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // short sort-stage kernel: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(tmp);
  ZK_YSHIFT(tmp_lo);
  ZK_YSHIFT(bins);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(k0);
  constexpr uint32_t CH = THR * BIG_EPT;
  extern __shared__ uint32_t big_lds[];
  const uint32_t nlo = 1u << lo_bits;
  uint32_t* cur = big_lds;                  // [nlo] next free slot of the bucket in sorted[]
  uint32_t* coff = cur + nlo;               // [nlo + 1] chunk-local exclusive offsets
  uint32_t* scr = coff + nlo + 1;           // [THR / 64]
  uint32_t* stage = scr + THR / 64;         // [CH]
  uint16_t* slo = reinterpret_cast<uint16_t*>(stage + CH);       // [CH]
  const uint32_t* __restrict__ bin_base = bins + nbins + 1;
  const uint32_t bin = blockIdx.x;
  const uint32_t set = bin >> hi_bits, hi = bin & ((1u << hi_bits) - 1);
  const size_t key0 = ((size_t)set << keys_per_set_log2) + ((size_t)hi << lo_bits);
  const uint32_t e0 = bin_base[bin], e1 = bin_base[bin + 1], total = bin_base[nbins];
  const uint32_t idx_mask = WIDE ? 0u : (1u << idx_bits) - 1;
  auto lo_of = [&](uint32_t e, uint32_t word) -> uint32_t { return WIDE ? (uint32_t)tmp_lo[e] : word >> (idx_bits + 1); };
  for (uint32_t j = threadIdx.x; j < nlo; j += THR) cur[j] = 0;
  __syncthreads();
  // The first two chunks of the bin stay in REGISTERS from this counting pass to the placement pass (packed format): a bin of
  // the 8 x 2^20-point d_msm is two chunks, so tmp[] is read once instead of twice
  constexpr bool KEEP = STAGED && !WIDE;
  uint32_t kept0[KEEP ? BIG_EPT : 1], kept1[KEEP ? BIG_EPT : 1];
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < BIG_EPT; k++) {
      const uint32_t ea = e0 + (uint32_t)k * THR + threadIdx.x, eb = ea + CH;
      kept0[k] = ea < e1 ? tmp[ea] : 0u;
      kept1[k] = eb < e1 ? tmp[eb] : 0u;
    }
#pragma unroll
    for (int k = 0; k < BIG_EPT; k++) {
      const uint32_t ea = e0 + (uint32_t)k * THR + threadIdx.x, eb = ea + CH;
      if (ea < e1) atomicAdd(&cur[kept0[k] >> (idx_bits + 1)], 1u);
      if (eb < e1) atomicAdd(&cur[kept1[k] >> (idx_bits + 1)], 1u);
    }
    for (uint32_t e = e0 + 2 * CH + threadIdx.x; e < e1; e += THR) atomicAdd(&cur[tmp[e] >> (idx_bits + 1)], 1u);
  } else {
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += THR) atomicAdd(&cur[lo_of(e, WIDE ? 0u : tmp[e])], 1u);
  }
  __syncthreads();
  {
    // offsets of the bin's keys; cur <- first slot of every bucket
    const uint32_t per = (nlo + THR - 1) / THR;
    const uint32_t j0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++)
      if (j0 + j < nlo) sum += cur[j0 + j];
    uint32_t tot;
    uint32_t run = e0 + block_scan_excl<THR>(sum, scr, &tot);
    for (uint32_t j = 0; j < per; j++)
      if (j0 + j < nlo) {
        const uint32_t n = cur[j0 + j];
        cur[j0 + j] = run;
        offsets[key0 + j0 + j] = run;
        run += n;
      }
    if (bin == nbins - 1 && threadIdx.x == 0) offsets[nkeys] = total;
  }
  __syncthreads();
  {
    // first bucket of every accumulate lane whose range starts inside this bin (msm_bucket_of on the local offsets)
    const uint32_t T = msm_range_len(total, nlanes, tmin, cap);
    const uint32_t l0 = (e0 + T - 1) / T, l1 = (e1 + T - 1) / T;       // lanes with e0 <= lane * T < e1
    for (uint32_t lane = l0 + threadIdx.x; lane < l1 && lane < nlanes; lane += THR) {
      const uint32_t a = lane * T;
      uint32_t lo = 0, hi_ = nlo - 1;
      while (lo < hi_) {
        const uint32_t mid = (lo + hi_) >> 1;
        if (cur[mid + 1] <= a) lo = mid + 1;
        else hi_ = mid;
      }
      k0[lane] = (uint32_t)key0 + lo;
    }
  }
  __syncthreads();
  // a bin far longer than the even share: scalars that repeat a value put most of their entries into ONE bucket, so
  // consecutive lanes draw consecutive slots and the direct stores coalesce by themselves -- while the chunked path below
  // would walk the bin 16 k entries at a time through one workgroup, ten barriers per chunk (measured on a 2^22-point MSM
  // with 60 % ones: this kernel 5.5 ms of 8.4; one atomic per wave when all its lanes hold the same key was measured on top of
  // this path and is slower: 32.7 against 26.8 ms at 2^24 points)
  const bool long_bin = STAGED && e1 - e0 > 8u * CH;
  if (!STAGED || long_bin) {
    // small launches: straight from the lane that read an entry to its bucket's next slot
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += THR) {
      const uint32_t w_ = tmp[e];
      const uint32_t out = WIDE ? w_ : (w_ & idx_mask) | (((w_ >> idx_bits) & 1u) << 31);
      sorted[atomicAdd(&cur[lo_of(e, w_)], 1u)] = out;
    }
    return;
  }
  // one chunk: rank its entries into the LDS stage in bucket order, stream the stage out (`load(k, j)` = entry j of the chunk)
  auto chunk = [&](uint32_t cb, uint32_t n, auto load) {
    for (uint32_t j = threadIdx.x; j <= nlo; j += THR) coff[j] = 0;
    __syncthreads();
    uint32_t word[BIG_EPT], rk[BIG_EPT];
#pragma unroll
    for (int k = 0; k < BIG_EPT; k++) {
      const uint32_t j = (uint32_t)k * THR + threadIdx.x;
      word[k] = j < n ? load(k, cb + j) : 0u;
    }
#pragma unroll
    for (int k = 0; k < BIG_EPT; k++) {
      const uint32_t j = (uint32_t)k * THR + threadIdx.x;
      if (j < n) rk[k] = atomicAdd(&coff[lo_of(cb + j, word[k])], 1u);
    }
    __syncthreads();
    {
      const uint32_t per = (nlo + THR - 1) / THR;
      const uint32_t j0 = threadIdx.x * per;
      uint32_t sum = 0;
      for (uint32_t j = 0; j < per; j++)
        if (j0 + j < nlo) sum += coff[j0 + j];
      uint32_t tot;
      uint32_t run = block_scan_excl<THR>(sum, scr, &tot);
      for (uint32_t j = 0; j < per; j++)
        if (j0 + j < nlo) {
          const uint32_t c_ = coff[j0 + j];
          coff[j0 + j] = run;
          run += c_;
        }
      if (threadIdx.x == 0) coff[nlo] = n;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIG_EPT; k++) {
      const uint32_t j = (uint32_t)k * THR + threadIdx.x;
      if (j < n) {
        const uint32_t l = lo_of(cb + j, word[k]);
        const uint32_t p = coff[l] + rk[k];
        if (WIDE) {
          stage[p] = word[k];
        } else {
          const uint32_t w_ = word[k];
          stage[p] = (w_ & idx_mask) | (((w_ >> idx_bits) & 1u) << 31);
        }
        slo[p] = (uint16_t)l;
      }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < n; j += THR) {
      const uint32_t l = slo[j];
      sorted[cur[l] + j - coff[l]] = stage[j];
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nlo; j += THR) cur[j] += coff[j + 1] - coff[j];
    __syncthreads();
  };
  uint32_t cb = e0;
  if constexpr (KEEP) {
    if (cb < e1) {
      chunk(cb, e1 - cb < CH ? e1 - cb : CH, [&](int k, uint32_t) { return kept0[k]; });
      cb += CH;
    }
    if (cb < e1) {
      chunk(cb, e1 - cb < CH ? e1 - cb : CH, [&](int k, uint32_t) { return kept1[k]; });
      cb += CH;
    }
  }
  for (; cb < e1; cb += CH) chunk(cb, e1 - cb < CH ? e1 - cb : CH, [&](int, uint32_t e) { return tmp[e]; });
}

// -------------------------------------------------------------------------------------------------- scan
// Exclusive scan of the per-key counts over `len` keys in three launches: offsets[k] = first sorted entry of key k,
// offsets[len] = number of entries.
constexpr int ISCAN_THREADS = 256;
constexpr int ISCAN_PER = 8;
constexpr int ISCAN_BLOCK = ISCAN_THREADS * ISCAN_PER;

ZK_D uint32_t block_scan_u32(uint32_t v, uint32_t* sh, uint32_t* total) {
  int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int off = 1; off < ISCAN_THREADS; off <<= 1) {
    uint32_t t = sh[tid];
    if (tid >= off) t += sh[tid - off];
    __syncthreads();
    sh[tid] = t;
    __syncthreads();
  }
  uint32_t ex = tid ? sh[tid - 1] : 0u;
  *total = sh[ISCAN_THREADS - 1];
  return ex;
}

// mode 0: write block totals; mode 1: write the exclusive scan (and, for the global-atomics sort, a copy as the scatter
// cursors)
static __global__ __launch_bounds__(ISCAN_THREADS) void iscan_block_kernel(const uint32_t* __restrict__ counts, size_t len,
                                                                   uint32_t* __restrict__ block_tot,
                                                                   const uint32_t* __restrict__ carry,
                                                                   uint32_t* __restrict__ offsets,
                                                                   uint32_t* __restrict__ cursor, int mode, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(counts);
  ZK_YSHIFT(block_tot);
  ZK_YSHIFT(carry);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(cursor);
  __shared__ uint32_t sh[ISCAN_THREADS];
  size_t base = (size_t)blockIdx.x * ISCAN_BLOCK + (size_t)threadIdx.x * ISCAN_PER;
  uint32_t loc[ISCAN_PER];
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < ISCAN_PER; i++) {
    uint32_t cnt = base + i < len ? counts[base + i] : 0u;
    loc[i] = acc;
    acc += cnt;
  }
  uint32_t tot;
  uint32_t ex = block_scan_u32(acc, sh, &tot);
  if (mode == 0) {
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
    return;
  }
  uint32_t cr = carry[blockIdx.x];
#pragma unroll
  for (int i = 0; i < ISCAN_PER; i++)
    if (base + i < len) {
      offsets[base + i] = cr + ex + loc[i];
      if (cursor) cursor[base + i] = cr + ex + loc[i];
    }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == ISCAN_THREADS - 1) offsets[len] = cr + tot;
}

static __global__ __launch_bounds__(ISCAN_THREADS) void iscan_carry_kernel(uint32_t* __restrict__ bt, size_t nblocks,
                                                                           size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(bt);
  __shared__ uint32_t sh[ISCAN_THREADS];
  uint32_t running = 0;
  for (size_t b0 = 0; b0 < nblocks; b0 += ISCAN_BLOCK) {
    size_t base = b0 + (size_t)threadIdx.x * ISCAN_PER;
    uint32_t loc[ISCAN_PER];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ISCAN_PER; i++) {
      uint32_t v = base + i < nblocks ? bt[base + i] : 0u;
      loc[i] = acc;
      acc += v;
    }
    uint32_t tot;
    uint32_t ex = block_scan_u32(acc, sh, &tot);
#pragma unroll
    for (int i = 0; i < ISCAN_PER; i++)
      if (base + i < nblocks) bt[base + i] = running + ex + loc[i];
    running += tot;
    __syncthreads();
  }
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
#define ZK_ACC_WAVES_8 4          // round 4: 128 VGPRs with the lazy mixed addition (4 dwords of scratch, read twice per step)
#endif
template <class Fld>
constexpr int ACC_WAVES = sizeof(Fld) == 48 ? ZK_ACC_WAVES_12 : ZK_ACC_WAVES_8;       // base-field (G1) kernel only
// BALANCED PARTITION.  The sorted entry array (offsets[nkeys] entries, grouped by bucket) is cut into `nlanes`
// contiguous ranges of T = ceil(entries / nlanes) entries, one per lane (per lane pair in G2), whatever the bucket
// boundaries are: every lane of the launch performs the same number of mixed additions, so a wave has no idle lanes
// waiting for its longest chain, there is no tail of short waves, and the sort needs no per-bucket segment
// descriptors nor a by-length ordering pass (round 2 cut buckets into segments of <= 16..64 points and counting-sorted
// the segments by length: two more launches on every MSM's dependent chain, 1.33 rounds of waves on a batch of eight).
// A lane walks its range and flushes its running sum whenever the bucket changes:
//   * a bucket that lies wholly inside the range        -> buckets[k] (final: the finalize kernel does not touch it)
//   * the first run when its bucket began in an earlier lane (whether or not it ends here) -> head[lane]
//   * the last run when its bucket continues in the next lane (and began here)             -> tail[lane]
// so bucket k, spanning lanes l0 = offsets[k] / T .. l1 = (offsets[k+1] - 1) / T with l1 > l0, is tail[l0] +
// head[l0+1] + .. + head[l1] (msm_finalize_kernel, which also writes the identity into empty buckets): one extra
// addition per lane boundary, ~1/T of the accumulate's.  Buckets spread over more than FIN_SEQ lanes (degenerate
// scalars: all ones) are summed by whole workgroups (msm_heavy_kernel) from a list the accumulate lanes leave.
// entries per lane: the launch has `nlanes` lanes for the most entries the sort could have produced; when it produced
// fewer (identity bases, zero digits) the ranges stay at least `tmin` long and the surplus lanes have nothing to do --
// every lane boundary costs a full addition in the finalize kernel
// Stores of the accumulate kernels' partial sums, written as inline assembly ON PURPOSE.  gfx950 counts vector loads and
// stores in one counter (vmcnt) and the compiler, seeing both kinds pending, can no longer wait for "all but the N
// newest" operations: every wait inside the loop -- for the index fetched two points ahead, for a bucket boundary --
// became s_waitcnt vmcnt(0), i.e. a wait for the point gather issued a moment earlier (read in the ISA; in a batch
// the accumulate kernels ran 20 % slower than round 2's, which stored once, after its loop).  A store the compiler does
// not see leaves its bookkeeping to loads, which return in order; the hardware still counts the store, so a wait can
// only be longer than computed, never shorter, and the data registers are read when the store issues.
typedef uint32_t zk_u32x4 __attribute__((ext_vector_type(4)));
template <int I, int N>
__device__ __forceinline__ void zk_store_chunks(const void* p, const zk_u32x4* v) {
  if constexpr (I < N) {
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2" : : "v"(p), "v"(v[I]), "n"(16 * I) : "memory");
    zk_store_chunks<I + 1, N>(p, v);
  }
}
template <class F>
__device__ __forceinline__ void store_elem_untracked(F* p, const F& v) {
  static_assert(sizeof(F) % 16 == 0 && sizeof(F) <= 4096, "element must be a multiple of 16 bytes");
  zk_u32x4 c[sizeof(F) / 16];
  __builtin_memcpy(c, &v, sizeof(F));
  zk_store_chunks<0, (int)(sizeof(F) / 16)>((const void*)p, c);
  // the hazard recogniser does not see these stores either: gfx9-family parts want one wait state between a VMEM store of
  // more than 64 bits and a VALU write of its data registers (the caller resets the stored sum right afterwards)
  asm volatile("s_nop 0" ::: "memory");
}

// bucket that contains entry a (offsets[k] <= a < offsets[k+1]; empty buckets are skipped by construction)
ZK_D uint32_t msm_bucket_of(const uint32_t* __restrict__ offsets, uint32_t nkeys, uint32_t a) {
  uint32_t lo = 0, hi = nkeys;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid + 1] <= a) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// first bucket of every lane's range, by one thread per lane ahead of the accumulate kernel: inside it the 18 dependent
// loads of the search would open every wave (measured in a batch of eight with four MSMs in flight: accumulate slots
// 17.6 + 10.2 ms with the search inside against 14.8 + 8.2 ms for round 2's per-bucket segments)
static __global__ __launch_bounds__(256) void msm_lane_start_kernel(const uint32_t* __restrict__ offsets, uint32_t nkeys,
                                                                    uint32_t nlanes, uint32_t tmin, uint32_t cap,
                                                                    uint32_t* __restrict__ k0, size_t ys) {
  __builtin_amdgcn_s_setprio(3);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(k0);
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= nlanes) return;
  const uint32_t total = offsets[nkeys];
  const uint32_t T = msm_range_len(total, nlanes, tmin, cap);
  if ((uint64_t)lane * T >= total) return;
  k0[lane] = msm_bucket_of(offsets, nkeys, lane * T);
}

// HEAVY LIST (buckets spread over more than FIN_SEQ accumulate lanes).  heavy[0..1] is ONE 64-bit counter {buckets listed :
// high word | virtual workgroups reserved : low word}; entry i (in the order of the counter's high word) is {bucket id,
// first virtual workgroup} at heavy[2 + 2 i].  A bucket of L lanes reserves S = 1 + L / 256 virtual workgroups, at most 1024
// (msm_heavy_kernel: each sums a chunk of L / S lanes; the chunk sums of a bucket with S > 1 go to hpart[] and the finalize
// kernel's extra workgroups sum them) -- round 4: an MSM over a PUBLIC witness has buckets of a million lanes (values that
// repeat), and ONE 16-quad workgroup walking such a bucket made a 2^24-point BLS12-381 MSM 121 ms instead of 39.
ZK_D uint32_t msm_heavy_splits(uint32_t L) {          // (integer only: this sits in the accumulate kernel's flush block)
  const uint32_t s = 1 + (L >> 8);
  return s < 1024 ? s : 1024;
}
ZK_D void msm_report_heavy(uint32_t* __restrict__ heavy, uint32_t k, uint32_t L) {
  const unsigned long long old =
      atomicAdd(reinterpret_cast<unsigned long long*>(heavy), (1ull << 32) | (unsigned long long)msm_heavy_splits(L));
  const uint32_t slot = (uint32_t)(old >> 32);
  heavy[2 + 2 * slot] = k;
  heavy[3 + 2 * slot] = (uint32_t)old;
}
// capacity of the list (entries) and of hpart[] (virtual workgroups) for a launch of nlanes accumulate lanes
inline size_t msm_heavy_cap(size_t nlanes) { return nlanes / FIN_SEQ + 8; }
inline size_t msm_heavy_vcap(size_t nlanes) { return nlanes / FIN_SEQ + nlanes / 256 + 8; }

// Base fields whose running sums the accumulate kernel keeps as lazy residues (field.hpp, LAZY_OK); ZK_ACC_LAZY=0 at build
// time restores the canonical form.  Same-box A/B on BN254 (profiles/r04_acc_lazy_ab.txt): d_msm 8 x 2^20 12.40 against
// 12.78 ms (-3 %), C4 599 against 592 proofs/s when compiled for four waves (at three waves the kernel takes 140
// VGPRs and the table-free proof, whose MSMs overlap, loses 4 %).  The 12-limb base fields are left canonical: there
// the lazy form spills (20 / 39 dwords at three waves) and C5 measures 1.31 against 1.285 s.
#ifndef ZK_ACC_LAZY
#define ZK_ACC_LAZY 1
#endif
template <class F>
constexpr bool acc_lazy_v = false;
#if defined(__HIP_DEVICE_COMPILE__)
template <class P>
constexpr bool acc_lazy_v<Fp<P>> = ZK_ACC_LAZY && Fp<P>::LAZY_OK && P::N <= 8;      // (the lazy operations exist in the device pass only)
#endif

template <class Fld>
__global__ __launch_bounds__(128, ACC_WAVES<Fld>) void msm_accumulate_kernel(const Affine<Fld>* __restrict__ bases0,
                                                            const Affine<Fld>* __restrict__ bases1,
                                                            const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ offsets, uint32_t nkeys,
                                                            uint32_t nlanes, uint32_t tmin, uint32_t cap,
                                                            XYZZ<Fld>* __restrict__ buckets0,
                                                            XYZZ<Fld>* __restrict__ edge0 /* [NB][2][nlanes]: head, tail */,
                                                            uint32_t* __restrict__ heavy /* the heavy list (msm_report_heavy) */,
                                                            const uint32_t* __restrict__ k0, size_t ys, int prio) {
  // prio != 0: the MSM at the END of a proof's critical chain (U, behind circom_h): its waves win issue arbitration against
  // the accumulate waves of the witness MSMs that still share the SIMDs when it starts (they have slack, it has none)
  if (prio) __builtin_amdgcn_s_setprio(2);
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(heavy);
  ZK_YSHIFT(k0);
  // blockIdx.y: which of the (up to two) base vectors that share this scalar vector -- and therefore the sort
  const Affine<Fld>* __restrict__ bases = blockIdx.y ? bases1 : bases0;
  XYZZ<Fld>* __restrict__ buckets = buckets0 + (size_t)blockIdx.y * nkeys;
  XYZZ<Fld>* __restrict__ head = edge0 + (size_t)blockIdx.y * 2 * nlanes;
  XYZZ<Fld>* __restrict__ tail = head + nlanes;
  const uint32_t total = offsets[nkeys];
  const uint32_t T = msm_range_len(total, nlanes, tmin, cap);
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= nlanes || (uint64_t)lane * T >= total) return;
  const uint32_t a = lane * T, b = a + T < total ? a + T : total;
  uint32_t k = k0[lane];
  bool cont = offsets[k] < a;
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  // Software pipeline.  The affine point is needed by the first two of the ten multiplications of a mixed addition
  // only (U2 = x ZZ, S2 = y ZZZ): the NEXT point's coordinates are requested right after them, INTO THE SAME REGISTERS,
  // and arrive under the other eight; the index is fetched two points ahead.  (Loading the next point into a second
  // set of registers and moving it over at the end of the step -- round 2's form -- makes the compiler wait for the
  // gather before the moves: s_waitcnt vmcnt(0) at the bottom of every iteration, no overlap at all.)
  uint32_t e = sorted[a];
  uint32_t e1 = sorted[a + 1 < b ? a + 1 : a];
  Affine<Fld> pt = load_elem(bases + (e & 0x7fffffffu));
  for (uint32_t p = a; p < b; p++) {
    const uint32_t e_cur = e;
    // end of the current bucket: requested here, looked at after the addition (the same word for a whole run, so it
    // comes from L1); nothing loaded is carried from one step to the next -- a value loaded inside the flush block and
    // carried over costs a move there, and the move a wait for ALL loads of the wave, the gather included, on the two
    // steps out of three on which some lane of a wave meets a bucket boundary
    uint32_t end_k = offsets[k + 1];
    const bool skip_pt = pt.is_identity();           // only when the sort kept identity bases (ZK_MSM_SKIP_IDENTITY=0)
    const bool was_id = acc.is_identity();
    Fld U2 = Fld::zero(), S2 = Fld::zero();
    {
      const Fld y2 = (e_cur >> 31) ? pt.y.neg() : pt.y;
      if (!skip_pt) {
        if (was_id) {
          acc = XYZZ<Fld>{pt.x, y2, Fld::one(), Fld::one()};
        } else if constexpr (acc_lazy_v<Fld>) {
          U2 = Fld::mul_lazy(pt.x, acc.ZZ);
          S2 = Fld::mul_lazy(y2, acc.ZZZ);
        } else {
          U2 = pt.x * acc.ZZ;
          S2 = y2 * acc.ZZZ;
        }
      }
    }
    // pt is dead from here on: the next point's coordinates land in its registers.  Both loads are UNCONDITIONAL (the
    // last steps of a range fetch its last entry again): behind a branch the compiler does not know how many loads
    // are in flight and turns the next wait into a wait for all of them
    e = e1;
    pt = load_elem(bases + (e & 0x7fffffffu));
    e1 = sorted[p + 2 < b ? p + 2 : b - 1];
    if constexpr (acc_lazy_v<Fld>) {
      // the running sum as LAZY residues in [0, 2p) (field.hpp): no conditional subtraction behind eight of the ten products
      if (!skip_pt && !was_id) {
        const Fld P = Fld::sub_lazy(U2, acc.X);
        const Fld R = Fld::sub_lazy(S2, acc.Y);
        if (P.is_zero_lazy()) {
          if (R.is_zero_lazy()) {
            const Affine<Fld> q = load_elem(bases + (e_cur & 0x7fffffffu));
            acc = xyzz_dbl_affine(q.x, (e_cur >> 31) ? q.y.neg() : q.y);
          } else {
            acc = XYZZ<Fld>::identity();
          }
        } else {
          const Fld PP = Fld::mul_lazy(P, P);
          const Fld PPP = Fld::mul_lazy(P, PP);
          const Fld Q = Fld::mul_lazy(acc.X, PP);
          const Fld X3 = Fld::sub_lazy(Fld::sub_lazy(Fld::mul_lazy(R, R), PPP), Fld::dbl_lazy(Q));
          const Fld Y3 = Fld::mul_sub_mul_lazy(R, Fld::sub_lazy(Q, X3), acc.Y, PPP);
          acc = XYZZ<Fld>{X3, Y3, Fld::mul_lazy(acc.ZZ, PP), Fld::mul_lazy(acc.ZZZ, PPP)};
        }
      }
    } else if (!skip_pt && !was_id) {
      const Fld P = U2 - acc.X;
      const Fld R = S2 - acc.Y;
      if (P.is_zero()) {
        if (R.is_zero()) {                           // the same point again (degenerate inputs): fetch it once more
          const Affine<Fld> q = load_elem(bases + (e_cur & 0x7fffffffu));
          acc = xyzz_dbl_affine(q.x, (e_cur >> 31) ? q.y.neg() : q.y);
        } else {
          acc = XYZZ<Fld>::identity();
        }
      } else {
        const Fld PP = P.sqr();
        const Fld PPP = P * PP;
        const Fld Q = acc.X * PP;
        const Fld X3 = R.sqr() - PPP - Q.dbl();
        const Fld Y3 = Fld::mul_sub_mul(R, Q - X3, acc.Y, PPP);      // one reduction for the two products (field.hpp)
        acc = XYZZ<Fld>{X3, Y3, acc.ZZ * PP, acc.ZZZ * PPP};
      }
    }
    while (end_k <= p) {                            // empty buckets behind the last flush (rare)
      k++;
      end_k = offsets[k + 1];
    }
    if (p + 1 == end_k || p + 1 == b) {             // the run ends: with its bucket, or with the range
      XYZZ<Fld>* dst = cont ? head + lane : (p + 1 == end_k ? buckets + k : tail + lane);
      if constexpr (acc_lazy_v<Fld>) acc = XYZZ<Fld>{acc.X.canon(), acc.Y.canon(), acc.ZZ.canon(), acc.ZZZ.canon()};
      store_elem_untracked(dst, acc);
      // the lane in which a bucket BEGINS reports it when it spreads over many lanes (once per bucket and sort)
      if (!cont && p + 1 != end_k && (blockIdx.y == 0 || ys != 0) && (end_k - 1) / T - lane > FIN_SEQ)
        msm_report_heavy(heavy, k, (end_k - 1) / T - lane);        // at most nlanes / FIN_SEQ such buckets: the list's capacity
      acc = XYZZ<Fld>::identity();
      cont = false;
      k += p + 1 == end_k ? 1u : 0u;                // (the last step of the range may leave k one past: unused)
    }
  }
}

// Extension-field variant with ONE BASE-FIELD VALUE PER LANE (quad.hpp split_madd): a QUAD of lanes per range, 128 threads =
// 32 ranges per workgroup.  P = parameters of the base field; points and partial sums are read and written in the
// layout of Affine<Fq2> / XYZZ<Fq2> (lane q of a quad owns base-field element q of a point, q and 4 + q of a sum).
// waves per SIMD the registers allow: 127 VGPRs for 8-limb base fields (four waves), 176 for 12-limb ones (two)
template <class P>
constexpr int SPLIT_WAVES = P::N > 8 ? 3 : 4;
template <class P>
__global__ __launch_bounds__(128, SPLIT_WAVES<P>) void msm_accumulate_split_kernel(const void* __restrict__ bases0,
                                                                 const void* __restrict__ bases1,
                                                                 const uint32_t* __restrict__ sorted,
                                                                 const uint32_t* __restrict__ offsets, uint32_t nkeys,
                                                                 uint32_t nlanes, uint32_t tmin, uint32_t cap,
                                                                 void* __restrict__ buckets0, void* __restrict__ edge0,
                                                                 uint32_t* __restrict__ heavy,
                                                                 const uint32_t* __restrict__ k0, size_t ys) {
  using F = Fp<P>;
  constexpr size_t AFF = 4, SUM = 8;                 // base-field elements per affine point / per XYZZ sum
  ZK_YSHIFT(sorted);
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(heavy);
  ZK_YSHIFT(k0);
  const F* __restrict__ bases = reinterpret_cast<const F*>(blockIdx.y ? bases1 : bases0);
  F* __restrict__ buckets = reinterpret_cast<F*>(buckets0) + (size_t)blockIdx.y * nkeys * SUM;
  F* __restrict__ head = reinterpret_cast<F*>(edge0) + (size_t)blockIdx.y * 2 * nlanes * SUM;
  F* __restrict__ tail = head + (size_t)nlanes * SUM;
  const uint32_t total = offsets[nkeys];
  const uint32_t T = msm_range_len(total, nlanes, tmin, cap);
  const uint32_t q = threadIdx.x & 3;
  const bool comp = (q & 1) != 0, half = (q & 2) != 0;
  const uint32_t lane = blockIdx.x * (blockDim.x / 4) + (threadIdx.x >> 2);        // quad index = range index
  if (lane >= nlanes || (uint64_t)lane * T >= total) return;
  const uint32_t a = lane * T, b = a + T < total ? a + T : total;
  uint32_t k = k0[lane];
  bool cont = offsets[k] < a;
  uint32_t end_k = offsets[k + 1];
  uint32_t end_n = offsets[k + 2 <= nkeys ? k + 2 : nkeys];
  SplitAcc<P> acc = split_identity<P>(comp);
  uint32_t e = sorted[a];
  uint32_t e1 = a + 1 < b ? sorted[a + 1] : e;
  F pt = load_elem(bases + (size_t)(e & 0x7fffffffu) * AFF + q);
  for (uint32_t p = a; p < b; p++) {
    const uint32_t e_next = e1;
    if (p + 2 < b) e1 = sorted[p + 2];
    F pt_next = pt;
    if (p + 1 < b) pt_next = load_elem(bases + (size_t)(e_next & 0x7fffffffu) * AFF + q);
    // the identity sentinel is (0, 0): all four base-field elements zero (quad-uniform after the exchanges)
    uint32_t z = pt.is_zero() ? 1u : 0u;
    z &= qperm_u32<1, 0, 3, 2>(z);
    z &= qperm_u32<2, 3, 0, 1>(z);
    if (!z) {
      const F c = qsel(half && (e >> 31) != 0, pt.neg(), pt);
      acc = split_madd<P>(acc, c, half, comp);
    }
    if (p + 1 == end_k || p + 1 == b) {
      F* dst = cont ? head + (size_t)lane * SUM : (p + 1 == end_k ? buckets + (size_t)k * SUM : tail + (size_t)lane * SUM);
      store_elem(dst + q, acc.c0);                         // X | Y
      store_elem(dst + 4 + q, acc.c1);                     // ZZ | ZZZ
      if (q == 0 && !cont && p + 1 != end_k && (blockIdx.y == 0 || ys != 0) && (end_k - 1) / T - lane > FIN_SEQ)
        msm_report_heavy(heavy, k, (end_k - 1) / T - lane);
      acc = split_identity<P>(comp);
      cont = false;
      if (p + 1 < b) {
        k++;
        end_k = end_n;
        while (end_k == p + 1) {                    // empty buckets in between (rare)
          k++;
          end_k = offsets[k + 1];
        }
        end_n = offsets[k + 2 <= nkeys ? k + 2 : nkeys];
      }
    }
    e = e_next;
    pt = pt_next;
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
constexpr int QUAD_THREADS = 256;                 // most: 64 points per workgroup, one wave per SIMD of a CU
constexpr int QUAD_VL = QUAD_THREADS / 4;
// Threads per workgroup of the heavy / reduce kernels.  256 (64 quads: shortest trees) when an MSM runs alone; 64 -- ONE
// wave -- by default: a four-wave workgroup needs four free wave slots with registers on ONE CU at the same moment, and
// among the accumulate waves of a BATCH's MSMs it waits for them (profiles/r03_b8_timeline_wide_tails.txt: 1.0-1.3 ms
// for a heavy-bucket launch that reads one word and exits, 0.6-1.8 ms per reduce stage).
inline int quad_threads(bool batched) {
  return batched ? 64 : 256;          // one proof at a time: 256 measured better (451 vs 409 proofs/s)
}

// Buckets spread over more than FIN_SEQ accumulate lanes (listed by the lanes they begin in, msm_report_heavy): one-wave
// workgroups, one VIRTUAL workgroup at a time (grid-stride): strided accumulation over its 16 quads, then a tree, so
// that no quad walks a long chain.  A bucket with one virtual workgroup is finished here; the chunk sums of a longer one go
// to hpart[] and are summed by the finalize kernel's extra workgroups.  The list is empty for well-spread scalars and the
// launch ends at once.  grid.y = base vector; with two sorts (ys != 0) each has its own list, with one shared sort both
// vectors use list 0.
template <class Fld>
__global__ __launch_bounds__(64) void msm_heavy_kernel(const XYZZ<Fld>* __restrict__ edge0, uint32_t nlanes, uint32_t tmin,
                                                       uint32_t cap, const uint32_t* __restrict__ offsets, uint32_t nkeys,
                                                       XYZZ<Fld>* __restrict__ buckets0,
                                                       const uint32_t* __restrict__ heavy,
                                                       XYZZ<Fld>* __restrict__ hpart0, uint32_t vcap, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(offsets);
  ZK_YSHIFT(heavy);
  const uint32_t nh = heavy[1], V = heavy[0];                  // (little-endian halves of the 64-bit counter)
  if (blockIdx.x >= V) return;
  const XYZZ<Fld>* __restrict__ head = edge0 + (size_t)blockIdx.y * 2 * nlanes;
  const XYZZ<Fld>* __restrict__ tail = head + nlanes;
  XYZZ<Fld>* __restrict__ buckets = buckets0 + (size_t)blockIdx.y * nkeys;
  XYZZ<Fld>* __restrict__ hpart = hpart0 + (size_t)blockIdx.y * vcap;
  extern __shared__ uint4 smem_fin[];
  XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_fin);
  const uint32_t T = msm_range_len(offsets[nkeys], nlanes, tmin, cap);
  const int q = threadIdx.x & 3, vl = threadIdx.x >> 2;
  const int nq = (int)blockDim.x / 4;                   // quads of this workgroup
  for (uint32_t v = blockIdx.x; v < V; v += gridDim.x) {
    // the listed bucket this virtual workgroup belongs to: the last entry whose first virtual workgroup is <= v
    uint32_t lo = 0, hi = nh;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (heavy[3 + 2 * mid] <= v) lo = mid;
      else hi = mid;
    }
    const uint32_t kh = heavy[2 + 2 * lo], vb = heavy[3 + 2 * lo];
    const uint32_t S = (lo + 1 < nh ? heavy[3 + 2 * (lo + 1)] : V) - vb, s = v - vb;
    const uint32_t o0 = offsets[kh], o1 = offsets[kh + 1];
    const uint32_t l0 = o0 / T, l1 = (o1 - 1) / T;
    const uint32_t chunk = (l1 - l0 + S - 1) / S;
    const uint32_t first = l0 + 1 + s * chunk, last = first + chunk - 1 < l1 ? first + chunk - 1 : l1;
    Fld acc = (s == 0 && vl == 0) ? qload(tail + l0, q) : qidentity<Fld>(q);
    for (uint32_t l = first + (uint32_t)vl; l <= last; l += (uint32_t)nq) acc = qadd(acc, qload(head + l, q), q);
    acc = wg_quad_sum(acc, sh, vl, q, nq);
    if (vl == 0) qstore(S == 1 ? buckets + kh : hpart + v, q, acc);
    __syncthreads();
  }
}

// finalize: one quad per bucket, ONE-WAVE workgroups (a workgroup that needs four wave slots on one CU at once waits for
// milliseconds among the accumulate waves of a batch: measured 5 ms for a 0.3 ms pass).  Empty bucket -> identity; bucket
// inside one lane's range, or already summed by msm_heavy_kernel -> untouched; otherwise tail + heads.  (Forming the sum
// inside the reduction's bucket load instead was measured and rejected: it puts independent work into the reduction's
// dependent chains -- 434 vs 490 proofs/s one at a time.)
constexpr int FIN_THREADS = 64;
template <class Fld>
__global__ __launch_bounds__(FIN_THREADS) void msm_finalize_kernel(const XYZZ<Fld>* __restrict__ edge0, uint32_t nlanes,
                                                               uint32_t tmin, uint32_t cap,
                                                               const uint32_t* __restrict__ offsets, uint32_t nkeys,
                                                               XYZZ<Fld>* __restrict__ buckets0,
                                                               const uint32_t* __restrict__ heavy,
                                                               const XYZZ<Fld>* __restrict__ hpart0, uint32_t vcap,
                                                               uint32_t fin_wgs, size_t ys) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound: win issue arbitration against the bulk accumulate waves
  ZK_YSHIFT(offsets);
  const XYZZ<Fld>* __restrict__ head = edge0 + (size_t)blockIdx.y * 2 * nlanes;
  const XYZZ<Fld>* __restrict__ tail = head + nlanes;
  XYZZ<Fld>* __restrict__ buckets = buckets0 + (size_t)blockIdx.y * nkeys;
  const int q = threadIdx.x & 3;
  if (blockIdx.x >= fin_wgs) {
    // extra workgroups: the chunk sums msm_heavy_kernel left for buckets with several virtual workgroups
    ZK_YSHIFT(heavy);
    const uint32_t nh = heavy[1], V = heavy[0];
    const XYZZ<Fld>* __restrict__ hpart = hpart0 + (size_t)blockIdx.y * vcap;
    extern __shared__ uint4 smem_fin[];
    XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_fin);
    const int vl = threadIdx.x >> 2;
    for (uint32_t h = blockIdx.x - fin_wgs; h < nh; h += gridDim.x - fin_wgs) {
      const uint32_t vb = heavy[3 + 2 * h], S = (h + 1 < nh ? heavy[3 + 2 * (h + 1)] : V) - vb;
      if (S == 1) continue;                             // (workgroup-uniform)
      Fld acc = qidentity<Fld>(q);
      for (uint32_t j = (uint32_t)vl; j < S; j += FIN_THREADS / 4) acc = qadd(acc, qload(hpart + vb + j, q), q);
      acc = wg_quad_sum(acc, sh, vl, q, FIN_THREADS / 4);
      if (vl == 0) qstore(buckets + heavy[2 + 2 * h], q, acc);
      __syncthreads();
    }
    return;
  }
  const uint32_t T = msm_range_len(offsets[nkeys], nlanes, tmin, cap);
  // grid-stride over the buckets: the launch is capped so that it does not queue thousands of workgroups behind the
  // accumulate waves
  for (size_t k = (size_t)blockIdx.x * (FIN_THREADS / 4) + (threadIdx.x >> 2); k < nkeys;
       k += (size_t)fin_wgs * (FIN_THREADS / 4)) {
    const uint32_t o0 = offsets[k], o1 = offsets[k + 1];
    if (o0 == o1) {
      qstore(buckets + k, q, qidentity<Fld>(q));
      continue;
    }
    const uint32_t l0 = o0 / T, l1 = (o1 - 1) / T;
    if (l1 == l0 || l1 - l0 > FIN_SEQ) continue;
    Fld acc = qload(tail + l0, q);
    for (uint32_t l = l0 + 1; l <= l1; l++) acc = qadd(acc, qload(head + l, q), q);
    qstore(buckets + k, q, acc);
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
  const uint32_t g = blockIdx.x * (uint32_t)((int)blockDim.x / 4 / nvl) + (uint32_t)(vl / nvl);
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
                                                                   int nvl, XYZZ<Fld>* __restrict__ out0,
                                                                   const uint32_t* __restrict__ cnt_src, size_t cnt_stride,
                                                                   uint32_t ncnt, uint32_t* __restrict__ cnt_dst) {
  __builtin_amdgcn_s_setprio(3);
  // out0 and cnt_dst are the slot's pinned host buffer: the last kernel of the chain leaves the slices (and the sorted-entry
  // counts of the launch's sorts, for zk_msm_stats) where the host folds them -- no copy kernels behind it
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < ncnt)
    cnt_dst[2 * threadIdx.x] = *(const uint32_t*)((const char*)cnt_src + threadIdx.x * cnt_stride);
  extern __shared__ uint4 smem_red[];
  XYZZ<Fld>* sh = reinterpret_cast<XYZZ<Fld>*>(smem_red);
  const uint32_t LO = 1u << lo_bits, HI = B >> lo_bits;
  const uint32_t ngroups = HI + 1 + LO;
  int hb = 0;
  while ((1u << hb) < HI) hb++;                      // HI = 2^hb: row weights 0..HI have bits 0..hb
  const int nslices = hb + 1 + lo_bits;
  const XYZZ<Fld>* __restrict__ rc = rc0 + (size_t)blockIdx.y * ngroups;
  const int q = threadIdx.x & 3, vl = threadIdx.x >> 2;
  const int j = (int)blockIdx.x * ((int)blockDim.x / 4 / nvl) + vl / nvl;
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
  ~MsmSlot() {
    if (pinned) (void)hipHostFree(pinned);
    if (ev) (void)hipEventDestroy(ev);
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
  int w0 = 0;                                    // first window of the launch (a window group of a split MSM)
  bool tabbed = false;                           // fixed-base table: one bucket set, every window c bits wide
  int batch = 1;                                 // scalar vectors of the launch (results: [base vector][batch])
  size_t stats_off = 0, offered = 0;             // statistics: where the sorts' entry counts land in the pinned buffer;
  int nsorts = 1;                                // (point, window) pairs offered to the sort
  bool g2 = false;
  MsmSlot* slot = nullptr;
  std::shared_ptr<const MsmTable> tab, tab2;     // keep the tables alive while the kernels run
};

// Ordering between the accumulate kernels of concurrent launches (the Groth16 prover runs the G2 accumulate ahead of the
// G1 ones: see engine_impl.hpp prove_begin).  The signalling launch records `signal_ev` right after its accumulate
// kernel and then raises `signal_flag`; a waiting launch (its sort work still runs ahead) spins on `wait_flag` on the
// host -- so that the event it then waits for on its stream is this proof's record, not a stale one -- before its own
// accumulate kernel.
struct MsmGate {
  hipEvent_t sorted_ev = nullptr;     // recorded on the launch's stream after its sort, before its accumulate kernel,
  std::atomic<int>* sorted_cnt = nullptr;   // ... and counted here once recorded
  // "all sorts first": before its accumulate kernel the launch waits (host: until `sorted_cnt` reaches `sorted_need`, so
  // that every event below is THIS proof's record; then on its stream) for the sorts of the proof's other MSMs
  const hipEvent_t* wait_sorted = nullptr;
  int n_wait_sorted = 0, sorted_need = 0;
  hipEvent_t wait_ev = nullptr;
  std::atomic<int>* wait_flag = nullptr;
  hipEvent_t signal_ev = nullptr;
  std::atomic<int>* signal_flag = nullptr;
};

struct MsmTuning {
  size_t bigsort_min;
  MsmGate gate;
  int c_force = 0;       // window bits asked for by zk_ctx_set_option "msm_c" / "msm_c_g2" (0: the cost model)
  // dynamic LDS bytes asked for with every accumulate workgroup (zk_ctx_set_option "msm_acc_lds"; the kernels use none):
  // caps the accumulate workgroups a CU holds, i.e. leaves wave slots and registers free for the short kernels of a
  // proof's critical chain, which otherwise wait for an accumulate wave to retire before they can become resident
  unsigned acc_lds = 0;
  bool skip_kernel = false;   // A/B only: identity bases through round 5's separate mask kernel instead of MsmBaseId's test
  int sort_lo_tab = 0;        // A/B only (zk_ctx_set_option "msm_sort_lo_tab"): low bucket bits per bin of a small TABLE sort (0 = 7)
};

// Window width: minimise nwin * (npts + 4 * buckets) -- mixed additions plus the per-bucket reduction work --
// with nwin = ceil((BITS+1)/c) windows of evenly spread width (see msm_launch); ties go to the wider window
// (more buckets = more lanes with shorter chains).
template <class FrP>
inline int msm_pick_c(size_t npts, bool g2 = false, int c_force = 0) {
  if (c_force >= 2 && c_force <= 20) return c_force;      // zk_ctx_set_option "msm_c" / "msm_c_g2" (tests force widths)
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

// Accumulate lanes of a launch with at most `max_entries` sorted entries (msm.hpp "balanced partition"): entries per
// lane `t` and the lane count that covers max_entries at that length.  Measured on batches of 1 / 2 / 4 / 8 119k-point
// MSMs alone on the chip (1.9 M .. 15 M entries; the chip holds 196 608 lanes at three waves per SIMD, 262 144 at the four
// the BN254 kernel is compiled for since round 4;
// profiles/r03_range_sweep.txt): what matters is how many ROUNDS of waves a launch makes -- >= 1.6 rounds run at
// 98-116 G multiplications/s, exactly one round at 79 (a grid that just fills the chip leaves the dispatcher no slack and
// all its waves march in step), fewer than one underfills -- while every lane boundary costs one full addition (14
// multiplications; 42 in G2) in the finalize kernel.  So: ~2.4 rounds, but never fewer than `lo` entries per lane (small
// launches: the MSMs of ONE proof run four at a time and fill the chip together) nor more than `hi`.
struct MsmLanes {
  uint32_t nlanes, tmin, cap;
};
inline MsmLanes msm_pick_lanes(size_t max_entries, int waves, bool pair, int lanes_per_range = 0) {
  if (!lanes_per_range) lanes_per_range = pair ? 2 : 1;
  const size_t cap = (size_t)1024 * waves * (64 / lanes_per_range);
  const size_t lo = pair ? MSM_RANGE_MIN_G2 : MSM_RANGE_MIN;
  const size_t hi = pair ? MSM_RANGE_G2 : MSM_RANGE;
  size_t t = std::min(hi, std::max(lo, (size_t)((double)max_entries / (2.4 * (double)cap))));
  size_t nl = std::max<size_t>(1, (max_entries + t - 1) / t);
  // between one and two rounds at this length msm_range_len splits the entries over TWO rounds of shorter ranges: the
  // launch needs the lanes of two rounds then (found by the 2^18-point BLS12-381 prover test: 4.4 M entries at 20 per
  // lane made 222 823 lanes, the two-round length of 12 covered 2.7 M entries and the rest was never added)
  if (max_entries >= cap * t && max_entries < 2 * cap * t && (max_entries + 2 * cap - 1) / (2 * cap) >= 12)
    nl = std::max(nl, 2 * cap);
  return MsmLanes{(uint32_t)nl, (uint32_t)t, (uint32_t)cap};
}

template <class FrP>
inline void msm_plan_of(size_t npts, bool g2, int* out, int c_force = 0) {
  const int c_req = msm_pick_c<FrP>(npts ? npts : 1, g2, c_force);
  const int T = FrP::BITS + 1;
  const int nwin = (T + c_req - 1) / c_req;
  out[0] = (T + nwin - 1) / nwin;
  out[1] = nwin;
  {
    const size_t entries = npts * (size_t)nwin;
    out[2] = (int)msm_pick_lanes(entries, g2 ? 2 : 3, g2).tmin;     // sorted entries (= mixed additions) per accumulate lane
  }
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
// det_pack over extension-field points, quad-split (pack_split.hpp; same translation units: inline 12-limb products).
// dig: the parties' joint-sparse-form digit columns [n][jlen] (groth16.hpp jsf_digits); G1 instantiations return an error.
template <class FrP, class Fld>
int pack_points_split_launch(IEngine* eng, const void* points, size_t nchunks, int n, const uint8_t* dig, int jlen,
                             const void* beta /* one base-field element, host */, void* shares, hipStream_t st);
// fixed-base multiplication over extension-field points, quad-split (pack_split.hpp fixed_base_mul_split_kernel); wb = 8 or 16
template <class FrP, class Fld>
int base_mul_split_launch(IEngine* eng, const void* scalars, size_t len, const void* table, int nwin, int wb, void* out,
                          hipStream_t st);
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
  hipError_t he = eng->event_wait(p.slot->ev);
  if (he == hipErrorNotReady) {
    // the tables stay referenced (the kernels may still run); the caller aborts the job
    p.active = true;
    return eng->fail(ZK_ERR_GENERIC, std::string("msm fold: the completion event of a ") + (p.g2 ? "G2" : "G1") +
                                         " MSM chain did not signal within the deadline (zk_ctx_set_option wait_deadline_ms)");
  }
  p.tab.reset();
  p.tab2.reset();
  if (he != hipSuccess) return eng->hip_fail(he, "msm event");
  const int kwin = p.kwin, c = p.c, wide = p.wide, lo_bits = p.lo_bits, nslices = p.c;
  const int hb = c - 1 - lo_bits;                  // row slices 0..hb come first, then lo_bits column slices
  const XYZZ<Fld>* hall = (const XYZZ<Fld>*)p.slot->pinned;
  {
    // statistics (zk_msm_stats): mixed additions performed = sorted entries, per base vector
    const uint32_t* cnt = (const uint32_t*)((const char*)p.slot->pinned + p.stats_off);
    uint64_t adds = 0;
    for (int v = 0; v < p.nb; v++) adds += cnt[2 * (p.nsorts == 2 ? v : 0)];
    eng->msm_adds[p.g2 ? 1 : 0].fetch_add(adds, std::memory_order_relaxed);
    eng->msm_offered[p.g2 ? 1 : 0].fetch_add(p.offered, std::memory_order_relaxed);
  }
  if (nvec > p.nb) nvec = p.nb;
  // windows [w_lo, w_hi] of one vector, high to low, doublings only INSIDE the range:
  // sum_w 2^(start_w - start_w_lo) X_w with X_w = sum_j 2^(j + lo_bits) TR_j + sum_j 2^j TC_j
  auto fold_range = [&](const XYZZ<Fld>* h, int w_hi, int w_lo) -> XYZZ<Fld> {
    XYZZ<Fld> total = XYZZ<Fld>::identity();
    for (int w = w_hi; w >= w_lo; w--) {
      const XYZZ<Fld>* sl = h + (size_t)w * nslices;
      const int cw = (p.tabbed || w + p.w0 < wide) ? c : c - 1;
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
    for (int w = w_hi; w >= w_lo; w--) bits += (p.tabbed || w + p.w0 < wide) ? c : c - 1;
    return bits;
  };
  // Table-free MSMs have one bucket set per window: ~250 doublings + c additions per window on one host thread
  // (0.4 ms for G2, the gap between two table-free proofs).  With the context's worker pool the windows are folded in
  // contiguous groups in parallel (below).  (With a fixed-base table there is one window and nothing to split.)
  constexpr int FOLD_PARTS_MAX = 8;
  HostPool* pool = eng->host_pool();
  constexpr bool par_fold = true;
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
      for (auto& f : futs)
        while (!pool->wait_helping(f, eng->deadline_from_now())) {}      // the sub-tasks reference this frame: never leave before them
    } else {
      for (int i = 0; i < jobs; i++) results[i] = fold_range(sets_of(i / batch, i % batch), kwin - 1, 0);
    }
    return ZK_OK;
  }
  const int max_parts = !pool || !par_fold ? 1 : std::min({FOLD_PARTS_MAX, kwin / 2, (pool->idle() + 1) / std::max(nvec, 1)});
  if (max_parts < 2) {
    for (int v = 0; v < nvec; v++) results[v] = fold_range(sets_of(v, 0), kwin - 1, 0);
    return ZK_OK;
  }
  // Part g folds its windows AND carries its result to its absolute position (start of its lowest window) by doublings, so
  // that the parts only have to be added.  Every bit position below a part's top costs it one doubling (the walk is
  // sequential in those, 0.35 us each for BN254 G1 on the host, 1.06 us for G2), every slice one addition (0.19 / 0.68):
  // the part holding the top window can hold little else; lower parts take more windows.  Greedy partition from the top
  // for the smallest cost bound that needs no more than max_parts parts.
  std::vector<int> start(kwin + 1, 0);             // bit position of window w
  for (int w = 0; w < kwin; w++) start[w + 1] = start[w] + width_of(w, w);
  constexpr double ADD_COST = 0.6;                 // one addition in units of a doubling
  auto partition = [&](double bound, int* hi_w, int* lo_w) {
    int np = 0, w = kwin - 1;
    while (w >= 0) {
      if (np == FOLD_PARTS_MAX) return FOLD_PARTS_MAX + 1;
      hi_w[np] = w;
      double cost = start[w + 1] + ADD_COST * nslices;      // the part's first window: all doublings down to bit 0
      w--;
      while (w >= 0 && cost + ADD_COST * nslices <= bound) cost += ADD_COST * nslices, w--;
      lo_w[np++] = w + 1;
    }
    return np;
  };
  int hi_w[FOLD_PARTS_MAX], lo_w[FOLD_PARTS_MAX];
  double lo_b = start[kwin] + ADD_COST * nslices, hi_b = start[kwin] + ADD_COST * nslices * kwin;
  for (int it = 0; it < 24; it++) {
    const double mid = 0.5 * (lo_b + hi_b);
    if (partition(mid, hi_w, lo_w) <= max_parts) hi_b = mid;
    else lo_b = mid;
  }
  const int nparts = partition(hi_b, hi_w, lo_w);
  auto fold_part = [&fold_range, &start](const XYZZ<Fld>* h, int w_hi, int w_lo) {
    XYZZ<Fld> t = fold_range(h, w_hi, w_lo);
    if (!t.is_identity())
      for (int i = 0; i < start[w_lo]; i++) t = xyzz_dbl_ni(t);
    return t;
  };
  XYZZ<Fld> part[2][FOLD_PARTS_MAX];
  std::vector<std::future<void>> futs;
  for (int v = 0; v < nvec; v++)
    for (int g = 0; g < nparts; g++) {
      if (v == 0 && g == 0) continue;              // this thread's share
      const XYZZ<Fld>* h = sets_of(v, 0);
      XYZZ<Fld>* dst = &part[v][g];
      const int a = hi_w[g], b = lo_w[g];
      futs.push_back(pool->submit([=, &fold_part]() { *dst = fold_part(h, a, b); }));
    }
  part[0][0] = fold_part(hall, hi_w[0], lo_w[0]);
  for (auto& f : futs)
    while (!pool->wait_helping(f, eng->deadline_from_now())) {}          // the sub-tasks reference this frame: never leave before them
  for (int v = 0; v < nvec; v++) {
    XYZZ<Fld> total = part[v][0];
    for (int g = 1; g < nparts; g++) total = xyzz_add_ni(total, part[v][g]);
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

  void plan(size_t npts, bool g2, int* out) const { msm_plan_of<FrP>(npts, g2, out, g2 ? c_g2 : c_g1); }
  int c_g1 = 0, c_g2 = 0;        // zk_ctx_set_option "msm_c" / "msm_c_g2": forced window bits of table-free MSMs (0 = cost model)

  // launch on workspace slot `wslot`; the result is collected with finish_t
  template <class Fld>
  int launch_t(IEngine* eng, const void* bases, const void* scalars, size_t npts, const Fr* coef_d, size_t part_len,
               hipStream_t st, int wslot, MsmPending* pend, const void* bases2 = nullptr, MsmGate gate = MsmGate{},
               const MsmBatchArg* batch = nullptr) {
    if (wslot < 0 || wslot >= MSM_WS) return eng->fail(ZK_ERR_BAD_INPUT, "bad msm workspace slot");
    if (pend->active) return eng->fail(ZK_ERR_GENERIC, "msm workspace slot still in flight");
    MsmTuning tune{bigsort_min, gate, IsExtField<Fld>::value ? c_g2 : c_g1, acc_lds, skip_kernel, sort_lo_tab};
    return msm_launch<FrP, Fld>(eng, slots_[wslot], tune, bases, bases2, scalars, npts, coef_d, part_len, st, pend,
                                batch);
  }
  template <class Fld>
  int finish_t(IEngine* eng, MsmPending* pend, XYZZ<Fld>* result, XYZZ<Fld>* result2 = nullptr) {
    return msm_fold<Fld>(eng, *pend, result, result2);
  }
  // (Round 4 measured a large lone MSM as TWO window groups on two streams -- the sort of the upper windows under the
  // accumulate of the lower ones: 13.3-13.4 ms with the staged sort kernels, 15.2-15.3 with small-workgroup ones, against
  // 13.1-13.7 for the single launch: the accumulate kernel wants the chip to itself.  The path was removed in round 5.)
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
  // The table has 2 n entries: [0, n) by party id, [n, 2 n) the window of set_window.
  int set_coefs(IEngine* eng, const std::vector<Fr>& coef) {
    coef_h_ = coef;
    coef_h_.resize(2 * coef.size(), Fr::zero());
    window_.clear();
    if (coef_d_) (void)hipFree(coef_d_);
    hipError_t e = hipMalloc((void**)&coef_d_, coef_h_.size() * sizeof(Fr));
    if (e != hipSuccess) return eng->hip_fail(e, "hipMalloc coef");
    e = hipMemcpy(coef_d_, coef_h_.data(), coef_h_.size() * sizeof(Fr), hipMemcpyHostToDevice);
    if (e != hipSuccess) return eng->hip_fail(e, "hipMemcpy coef");
    return ZK_OK;
  }
  // window [n, n + parties.size()) = the coefficients of an arbitrary party list (a rank's parties under a general
  // party -> rank map): callers then address them as the range (first = n, count)
  std::vector<int> window_;
  int set_window(IEngine* eng, const std::vector<int>& parties) {
    if (parties == window_) return ZK_OK;
    const size_t nn = coef_h_.size() / 2;
    if (!coef_d_ || parties.size() > nn) return eng->fail(ZK_ERR_BAD_INPUT, "bad party window");
    for (size_t i = 0; i < parties.size(); i++) {
      if (parties[i] < 0 || (size_t)parties[i] >= nn) return eng->fail(ZK_ERR_BAD_INPUT, "bad party id");
      coef_h_[nn + i] = coef_h_[(size_t)parties[i]];
    }
    hipError_t e = hipMemcpy(coef_d_ + nn, coef_h_.data() + nn, parties.size() * sizeof(Fr), hipMemcpyHostToDevice);
    if (e != hipSuccess) return eng->hip_fail(e, "hipMemcpy coef window");
    window_ = parties;
    return ZK_OK;
  }
  ~MsmRunner() {
    if (coef_d_) (void)hipFree(coef_d_);
  }

  // ---- fixed-base tables (zk_msm_precompute): process-wide registry keyed by address (engine.hpp TableRegistry)
  // Window bits of new tables: 0 = by the vector's length (below), else what zk_ctx_set_option "msm_table_c" /
  // "msm_table_c_g2" asked for (validated there: 0 or 8..22).  A table folds all windows into ONE set of 2^(c-1) buckets, so a
  // bucket receives len * nwin / 2^(c-1) entries: once that is more than FIN_SEQ (16) accumulate ranges long, every
  // bucket goes through the heavy-bucket path meant for degenerate scalars and the MSM is 2-3x slower than table-free
  // (measured, tools/tab_c3.py, G1 d_msm over 2^19 / 2^20 / 2^23 points: c = 16 2.53 / 3.48 / 18.8 ms, c = 17 1.10 /
  // 1.95 / 20.7, c = 20 1.45 / 2.2 / 11.8; table-free 1.59 / 2.77 / 12.7; below 2^19 points c = 16 is best: 0.78 against
  // 1.15 ms table-free at 2^18).  G1: 15 bits (17 windows, 16 384 buckets) up to 2^17 points -- the SHA-256 proof's four G1
  // MSMs: 6 % more mixed additions than 16 bits but half the buckets in the reduction every chain ends with: 634 / 649 /
  // 653 / 651 vs 630 / 639 / 638 / 638 proofs/s, same box, round 5; 14 bits already sends every bucket down the heavy path:
  // 512-542 --, 16 bits below 2^19 points, 18 below 2^22 (re-rounded to evenly spread windows: 17 bits
  // = 15 windows on BN254's 254-bit Fr, 18 bits on BLS12-381's 255-bit Fr), 20 (13 windows) from there.
  // G2: 15 bits = 17 windows, 16 384 buckets below 2^20 points (6 % more mixed additions than 16 bits but half the
  // buckets in the G2 reduction, the latency chain a proof ends with: 458-477 vs 423-453 proofs/s, same box), 19 (14
  // windows) from there (2^21 points: 11.1 ms against 17.8 at 15 bits and 11.8 table-free).
  int table_c = 0;
  int table_c_g2 = 0;
  static int table_c_auto(size_t len, bool g2) {
    if (g2) return len < ((size_t)1 << 20) ? 15 : 19;
    return len <= ((size_t)1 << 17) ? 15 : len < ((size_t)1 << 19) ? 16 : len < ((size_t)1 << 22) ? 18 : 20;
  }
  template <class Fld>
  int precompute_t(IEngine* eng, const void* bases, size_t len, hipStream_t st) {
    if (!bases || !len) return eng->fail(ZK_ERR_BAD_INPUT, "null base vector");
    const int T = FrP::BITS + 1;
    const int tc_opt = IsExtField<Fld>::value ? table_c_g2 : table_c;
    const int tc = tc_opt > 0 ? tc_opt : table_c_auto(len, IsExtField<Fld>::value);
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

  // two-level sort from this many points on (zk_ctx_set_option "msm_bigsort_min")
  size_t bigsort_min = (size_t)1 << 14;
  unsigned acc_lds = 0;           // zk_ctx_set_option "msm_acc_lds" (MsmTuning::acc_lds)
  bool skip_kernel = false;       // zk_ctx_set_option "msm_skip_kernel" (MsmTuning::skip_kernel)
  int sort_lo_tab = 0;            // zk_ctx_set_option "msm_sort_lo_tab" (MsmTuning::sort_lo_tab)
  MsmSlot slots_[MSM_WS];
  Fr* coef_d_ = nullptr;
  std::vector<Fr> coef_h_;
};

}  // namespace zk
