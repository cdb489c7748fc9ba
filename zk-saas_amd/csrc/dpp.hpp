// d_pp: the king's part of the distributed partial products (dist-primitives/src/dpp/mod.rs:41-76), three launches.
//
// The reference unpacks num || den, divides element by element (one inverse() each, :54-57), runs a serial prefix product
// (:62-65) and packs.  The same field elements are produced here WITHOUT a division per element:
//
//   prefix_i = prod_{k<=i} num_k / den_k = N_i * S_{i+1} / D,   N_i = prod_{k<=i} num_k,  S_i = prod_{k>=i} den_k,
//                                                                D   = prod_k den_k  (zero iff some den_k is zero)
//
// i.e. a prefix scan of the numerators, a suffix scan of the denominators and ONE inversion for the whole vector (field
// arithmetic is exact, so the shares are bit-identical to the divide-then-scan form; 5 multiplications per element where
// Montgomery's trick needs 6 plus its inversions).
//
//   dpp_tile_kernel   : one workgroup per tile of 256 E elements (E = 8, or 4 for short vectors): unpack2 of num and den (coalesced share loads),
//                       transposed through LDS so that each thread owns E consecutive elements; thread-local prefix /
//                       suffix, the 256 thread totals scanned by one wave each; writes
//                       y_i = (N_i / N_before_tile) * (S_{i+1} / S_after_tile) and the tile's two totals.
//   dpp_carry_kernel  : one workgroup: exclusive prefix of the numerator totals, exclusive suffix of the denominator totals,
//                       the inversion of D  ->  c_tile = N_before_tile * S_after_tile / D;  raises *err when D = 0
//                       (the reference panics on inverse().unwrap(), dpp/mod.rs:55).
//   dpp_finish_kernel : prefix_i = y_i * c_tile -> pack (fresh randomness) -> store; when the caller runs all parties on
//                       this device it also carries the deg_red that follows (dpp/mod.rs:86, deg_red.rs:80-126): the
//                       king of that round would unpack2(pack(prefix, r1) + in_mask) = prefix + unpack2(in_mask) -- the
//                       first pack's randomness cancels exactly -- so the kernel adds unpack2 of the in-mask column to the
//                       prefix products, packs with deg_red's randomness stream and adds the out-mask: the 2 GiB of
//                       intermediate shares (C5) are never written or read back.
//
// Algorithmic bytes: read 2 n (m/l) B, write n (m/l) B, + 2 m B for y.
#pragma once
#include "pss.hpp"

namespace zk {
#if defined(__HIPCC__)

constexpr int DPP_THREADS = 256;
// consecutive elements per thread (template parameter E of the tile kernel): 8 -> 2048-element tiles, the thread-total scans
// amortised over 8 elements.  E = 4 was measured (same box): slower at 2^20 (more tiles for the carry kernel), a tie at 2^24.
constexpr int DPP_E_LONG = 8;
constexpr int DPP_CARRY_THREADS = 1024;
template <int E>
struct DppGeom {
  static constexpr int TILE = DPP_THREADS * E;            // elements per workgroup
  static constexpr int LDS_SLOTS = TILE + TILE / E;
};

// LDS position of tile element e: one spare slot after every E elements, so that the per-thread walks (thread t
// reads element t * DPP_E + i) fall on distinct banks; the spare slots hold the per-thread numerator totals.
template <int E>
ZK_D int dpp_pos(int e) { return e + e / E; }
template <int E>
ZK_D int dpp_spare(int t) { return t * (E + 1) + E; }

template <class F>
ZK_D F wave_up(const F& v, int off) {
  F r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = __shfl_up(v.v[i], (unsigned)off, 64);
  return r;
}
template <class F>
ZK_D F wave_bcast(const F& v, int lane) {
  F r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = __shfl(v.v[i], lane, 64);
  return r;
}
// inclusive product scan over the 64 lanes of a wave (six multiplications per lane)
template <class F>
ZK_D F wave_scan_mul(F v) {
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    F p = wave_up(v, off) * v;
    if (lane >= off) v = p;
  }
  return v;
}

// ONE wave turns the DPP_THREADS per-thread totals at positions pos(t) of `lds` into their exclusive prefix products (or
// exclusive suffix products when `reverse`), in place; returns the product of all of them in every lane.
template <bool reverse, class F, class PosFn>
ZK_D F dpp_scan_totals(const LdsVec<F>& lds, PosFn pos) {
  constexpr int Q = DPP_THREADS / 64;
  const int lane = threadIdx.x & 63;
  F p[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int idx = lane * Q + q;
    const F t = lds.get(pos(reverse ? DPP_THREADS - 1 - idx : idx));
    p[q] = q ? p[q - 1] * t : t;
  }
  const F inc = wave_scan_mul(p[Q - 1]);
  F ex = wave_up(inc, 1);
  if (lane == 0) ex = F::one();
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int idx = lane * Q + q;
    lds.put(pos(reverse ? DPP_THREADS - 1 - idx : idx), q ? ex * p[q - 1] : ex);
  }
  return wave_bcast(inc, 63);
}

// unpack2 (or the Lagrange form for a party subset) of E / L chunks per thread, lane-adjacent chunks adjacent in memory,
// into the tile's LDS slots in element order; chunks past the end contribute ones
template <class P, int L, int E>
ZK_D void dpp_unpack_to_lds(const Fp<P>* __restrict__ sh, int np, size_t nchunks, size_t pitch,
                            const Fp<P>* __restrict__ U, size_t chunk0, const LdsVec<Fp<P>>& buf) {
  using F = Fp<P>;
  constexpr int K = E / L;
  const int tid = threadIdx.x;
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    const int c = k * DPP_THREADS + tid;
    const size_t j = chunk0 + c;
    F s[L];
    if (j < nchunks) {
#pragma unroll
      for (int i = 0; i < L; i++) s[i] = F::zero();
      unpack_accumulate<F, L>(s, U, np, [&](int r) { return load_elem(sh + (size_t)r * pitch + j); });
    } else {
#pragma unroll
      for (int i = 0; i < L; i++) s[i] = F::one();
    }
#pragma unroll
    for (int i = 0; i < L; i++) buf.put(dpp_pos<E>(c * L + i), s[i]);
  }
}

// shares [np][pitch] of num and den -> y (natural element order j * l + i), tile totals
template <class P, int L, int E>
__global__ __launch_bounds__(DPP_THREADS) void dpp_tile_kernel(const Fp<P>* __restrict__ num,
                                                              const Fp<P>* __restrict__ den, int np, size_t nchunks,
                                                              size_t pitch, const Fp<P>* __restrict__ U /* [l][np] */,
                                                              Fp<P>* __restrict__ y, Fp<P>* __restrict__ tile_n,
                                                              Fp<P>* __restrict__ tile_d) {
  using F = Fp<P>;
  constexpr int K = E / L;                         // chunks per thread
  static_assert(K >= 1 && K * L == E, "packing factor must divide E");
  constexpr int TC = DPP_THREADS * K;              // chunks per tile
  extern __shared__ uint4 smem[];
  LdsVec<F> buf;                 // (assigned, not brace-initialised: a constant aggregate holding the LDS address
  buf.base = smem;               //  would be emitted as a static initialiser, which the backend rejects)
  buf.stride = DppGeom<E>::LDS_SLOTS;
  const int tid = threadIdx.x;
  const size_t chunk0 = (size_t)blockIdx.x * TC;

  // numerators: a[i] = product of this thread's elements 0..i
  dpp_unpack_to_lds<P, L, E>(num, np, nchunks, pitch, U, chunk0, buf);
  __syncthreads();
  F a[E];
  a[0] = buf.get(dpp_pos<E>(tid * E));
#pragma unroll
  for (int i = 1; i < E; i++) a[i] = a[i - 1] * buf.get(dpp_pos<E>(tid * E + i));
  const F tot_a = a[E - 1];
  __syncthreads();

  // denominators: a[i] *= product of this thread's elements i+1..E-1
  dpp_unpack_to_lds<P, L, E>(den, np, nchunks, pitch, U, chunk0, buf);
  __syncthreads();
  F b = buf.get(dpp_pos<E>(tid * E + E - 1));
#pragma unroll
  for (int i = E - 2; i >= 0; i--) {
    a[i] = a[i] * b;
    b = b * buf.get(dpp_pos<E>(tid * E + i));
  }
  __syncthreads();
  // thread totals: numerators in the spare slots, denominators over the (dead) element slots
  buf.put(dpp_spare<E>(tid), tot_a);
  buf.put(dpp_pos<E>(tid), b);
  __syncthreads();
  const int wave = tid >> 6;
  if (wave == 0) {
    const F tot = dpp_scan_totals<false>(buf, [](int t) { return dpp_spare<E>(t); });
    if (tid == 0) store_elem(tile_n + blockIdx.x, tot);
  } else if (wave == 1) {
    const F tot = dpp_scan_totals<true>(buf, [](int t) { return dpp_pos<E>(t); });
    if (tid == 64) store_elem(tile_d + blockIdx.x, tot);
  }
  __syncthreads();
  const F c = buf.get(dpp_spare<E>(tid)) * buf.get(dpp_pos<E>(tid));
  __syncthreads();
#pragma unroll
  for (int i = 0; i < E; i++) buf.put(dpp_pos<E>(tid * E + i), a[i] * c);
  __syncthreads();
  const size_t e0 = chunk0 * L, total = nchunks * L;
#pragma unroll
  for (int k = 0; k < E; k++) {
    const int e = k * DPP_THREADS + tid;
    if (e0 + e < total) store_elem(y + e0 + e, buf.get(dpp_pos<E>(e)));
  }
}

// exclusive product scan over the threads of a (multiple-of-64, <= 1024 thread) workgroup; `sh` holds 2 x 16 elements.
// Wave scans (six products per lane), then ONE wave scans the <= 16 wave totals while the others wait at the barrier (a
// loop over the wave totals in every wave made sixteen waves issue the same sixteen products on four SIMDs: ~40 us per scan).
template <class F>
ZK_D F block_scan_mul_exclusive(const F& v, F* sh, F* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const F inc = wave_scan_mul(v);
  __syncthreads();                       // `sh` may still be read from an earlier scan
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const F t = lane < nw ? sh[lane] : F::one();
    const F s = wave_scan_mul(t);        // inclusive prefix products of the wave totals
    if (lane < nw) sh[16 + lane] = s;
  }
  __syncthreads();
  const F before = wave ? sh[16 + wave - 1] : F::one();
  *total = sh[16 + nw - 1];
  F ex = wave_up(inc, 1);
  if (lane == 0) ex = F::one();
  return before * ex;
}

// tile totals -> c_tile = (product of the numerator totals before the tile) * (product of the denominator totals after it)
// / (product of all denominators).  tile_n is overwritten with the exclusive prefixes.
//
// TWO workgroups (round 6).  Round 5 ran one: two block scans (~25 us), then the one inversion on one lane (~45 us,
// batched divsteps) with every other wave parked, then the c_tile pass -- 94 us at 2^20, a quarter of the call.  The
// inversion only needs D = the product of all denominator totals, which a tree product gives long before the scans end:
// workgroup 1 forms D, inverts it and publishes 1 / D; workgroup 0 runs the two scans and the prefix store meanwhile and
// picks the inverse up when it needs it.  Hand-off without fences (a device-scope release would write back the L2 the tile
// kernel has just filled): the limbs of 1 / D go out as device-scope atomic exchanges whose RETURN VALUES the wave holds
// before it raises the flag (so they have been performed at the memory side), the reader polls the flag and reads the
// limbs with device-scope atomic loads -- the construction of msm_hist's ticket (include/zksaas.h, memory-model note).
//   sync[0] = error flag (zero denominator), sync[1] = ready flag, sync[8 .. 8 + N) = limbs of 1 / D; sync[0..1] zeroed per call.
template <class F>
__global__ __launch_bounds__(DPP_CARRY_THREADS) void dpp_carry_kernel(F* __restrict__ tile_n, const F* __restrict__ tile_d,
                                                                      size_t ntiles, F* __restrict__ ctile,
                                                                      int* __restrict__ sync) {
  __shared__ F sh[32];
  __shared__ F inv_sh;
  const size_t tid = threadIdx.x, nt = blockDim.x;
  const size_t q = (ntiles + nt - 1) / nt;
  // thread t owns the numerator range t and the denominator range nt-1-t: one forward scan serves prefix and suffix
  const size_t nlo = tid * q < ntiles ? tid * q : ntiles, nhi = nlo + q < ntiles ? nlo + q : ntiles;
  const size_t r = nt - 1 - tid;
  const size_t dlo = r * q < ntiles ? r * q : ntiles, dhi = dlo + q < ntiles ? dlo + q : ntiles;
  uint32_t* const inv_g = reinterpret_cast<uint32_t*>(sync) + 8;
  uint32_t* const ready = reinterpret_cast<uint32_t*>(sync) + 1;
  if (blockIdx.x == 1) {
    // ---- the inverter: D, 1 / D, publish
    F pd = F::one();
    for (size_t i = dlo; i < dhi; i++) pd = pd * load_elem(tile_d + i);
    F tot_d;
    (void)block_scan_mul_exclusive(pd, sh, &tot_d);
    if (tid < 64) {
      // by the batched divsteps on ONE wave: every lane holds the same total, and handing it to the loop through
      // readfirstlane lets the compiler keep the state in SGPRs instead of running a one-lane vector loop
      F t;
#pragma unroll
      for (int i = 0; i < F::N; i++) t.v[i] = __builtin_amdgcn_readfirstlane(tot_d.v[i]);
      const bool zero = t.is_zero();
      const F inv = zero ? F::zero() : t.inverse_safegcd();
      if (tid == 0 && zero) atomicExch(sync, 1);
      uint32_t mine = 0;
#pragma unroll
      for (int i = 0; i < F::N; i++) mine = (int)tid == i ? inv.v[i] : mine;
      uint32_t old = 0;
      if (tid < (size_t)F::N) old = atomicExch(inv_g + tid, mine);
      asm volatile("" ::"v"(old));                       // the wave has the exchanges' return values: they have been performed
      if (tid == 0) (void)atomicExch(ready, 1u);         // (a read-modify-write like the limbs: performed at the memory side)
    }
    return;
  }
  F pn = F::one(), pd = F::one();
  for (size_t i = nlo; i < nhi; i++) pn = pn * load_elem(tile_n + i);
  for (size_t i = dlo; i < dhi; i++) pd = pd * load_elem(tile_d + i);
  F tot_n, tot_d;
  F run = block_scan_mul_exclusive(pn, sh, &tot_n);
  F sd = block_scan_mul_exclusive(pd, sh, &tot_d);
  for (size_t i = nlo; i < nhi; i++) {
    const F t = load_elem(tile_n + i);
    store_elem(tile_n + i, run);
    run = run * t;
  }
  if (tid < 64) {
    // the inverse from workgroup 1 (normally there already: this workgroup's scans take about as long as the inversion)
    // BOUNDED poll (~50 ms): if workgroup 1 has not been given a slot by then -- a chip held by other work -- this wave
    // inverts the total itself; the kernel cannot wait for ever on a workgroup that is not resident
    uint32_t got = 0;
    if (tid == 0)
      for (int spin = 0; spin < 400000; spin++) {
        got = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got) break;
        __builtin_amdgcn_s_sleep(2);
      }
    got = __builtin_amdgcn_readfirstlane(got);
    F inv;
    if (got) {
      uint32_t limb = 0;
      // (lanes 0..N-1 read after lane 0 has seen the flag: one wave, in order)
      if (tid < (size_t)F::N) limb = __hip_atomic_load(inv_g + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < F::N; i++) inv.v[i] = __shfl(limb, i, 64);
    } else {
      F t;
#pragma unroll
      for (int i = 0; i < F::N; i++) t.v[i] = __builtin_amdgcn_readfirstlane(tot_d.v[i]);
      const bool zero = t.is_zero();
      inv = zero ? F::zero() : t.inverse_safegcd();
      if (tid == 0 && zero) atomicExch(sync, 1);
    }
    if (tid == 0) inv_sh = inv;
  }
  __threadfence_block();
  __syncthreads();
  sd = sd * inv_sh;
  for (size_t i = dhi; i-- > dlo;) {
    store_elem(ctile + i, load_elem(tile_n + i) * sd);
    sd = sd * load_elem(tile_d + i);
  }
}

// prefix products -> fresh shares [n][nchunks]; with in_mask / out_mask the deg_red round that follows d_pp's king round
// when all parties live on this device (see the header)
template <class P, int L>
__global__ __launch_bounds__(KING_THREADS) void dpp_finish_kernel(const Fp<P>* __restrict__ y,
                                                                 const Fp<P>* __restrict__ ctile, size_t nchunks,
                                                                 const Fp<P>* __restrict__ in_mask,
                                                                 const Fp<P>* __restrict__ out_mask,
                                                                 const Fp<P>* __restrict__ U /* [l][n], all parties */,
                                                                 const Fp<P>* __restrict__ Pm,
                                                                 const PackL2<Fp<P>>* __restrict__ k2, RngSeed seed,
                                                                 Fp<P>* __restrict__ out, uint32_t tile_elems) {
  using F = Fp<P>;
  constexpr int T = L, N = 4 * L;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nchunks) return;
  const F c = load_elem(ctile + (j * L) / tile_elems);
  F sec[L + T];
#pragma unroll
  for (int i = 0; i < L; i++) sec[i] = load_elem(y + j * L + i) * c;
  if (in_mask) {
    F mk[L];
#pragma unroll
    for (int i = 0; i < L; i++) mk[i] = F::zero();
    unpack_accumulate<F, L>(mk, U, N, [&](int s) { return load_elem(in_mask + (size_t)s * nchunks + j); });
#pragma unroll
    for (int i = 0; i < L; i++) sec[i] = sec[i] + mk[i];
  }
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
    const size_t o = (size_t)p * nchunks + j;
    F acc = sh[p];
    if (out_mask) acc = acc + load_elem(out_mask + o);
    store_elem(out + o, acc);
  }
}

#endif  // __HIPCC__
}  // namespace zk
