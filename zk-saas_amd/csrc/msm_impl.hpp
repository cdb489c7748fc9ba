// Definitions of msm_launch / msm_table_launch (declared in msm.hpp).  Included ONLY by the msm_<curve>_g<k>.hip
// translation units, each of which instantiates them for one (scalar field, coordinate field) pair: the Pippenger
// kernels are the most expensive part of the build and compile in parallel this way.
#pragma once
#include "msm.hpp"

namespace zk {

template <class FrP, class Fld>
int msm_launch(IEngine* eng, MsmSlot& slot, const MsmTuning& tune, const void* bases, const void* bases2,
               const void* scalars, size_t npts, const Fp<FrP>* coef_d, size_t part_len, hipStream_t st,
               MsmPending* pend) {
  using Fr = Fp<FrP>;
  pend->active = false;
  const unsigned NB = bases2 ? 2u : 1u;
  if (npts == 0) return ZK_OK;
  if (npts >= ((size_t)1 << 31)) return eng->fail(ZK_ERR_BAD_INPUT, "msm too large");
  constexpr bool G2FLD = IsExtField<Fld>::value;
  // fixed-base table registered for this base vector (and the same window layout / offset for the second one)?
  size_t toff = 0, toff2 = 0;
  std::shared_ptr<const MsmTable> tab = TableRegistry::inst().find(bases, npts, sizeof(Affine<Fld>), FrP::BITS, &toff), tab2;
  if (tab && NB == 2) {
    tab2 = TableRegistry::inst().find(bases2, npts, sizeof(Affine<Fld>), FrP::BITS, &toff2);
    if (!tab2 || tab2->len != tab->len || toff2 != toff || tab2->c != tab->c) tab = nullptr;
    else bases2 = tab2->data;
  }
  if (tab) bases = tab->data;
  const int c_req = tab ? tab->c : msm_pick_c<FrP>(npts, G2FLD);
  // BITS+1 bits (room for the signed-digit carry) are spread EVENLY over the windows: `wide` windows of c bits and
  // nwin-wide of c-1.  A plain c-bit split leaves a top window of a few bits (254 = 19*13 + 7) whose 64 buckets
  // each receive npts/64 points: hot atomics in the sort, long chains, and a heavy-bucket pass in every MSM.
  const int T = FrP::BITS + 1;
  const int nwin = (T + c_req - 1) / c_req;
  const int c = (T + nwin - 1) / nwin;            // widest window
  const int wide = T - nwin * (c - 1);            // 1 <= wide <= nwin
  const uint32_t B = 1u << (c - 1);
  const int kwin = tab ? 1 : nwin;                // bucket sets: with a table all windows share one
  const size_t nkeys = (size_t)kwin * B;
  const size_t max_sorted = npts * nwin;
  if (max_sorted >= ((size_t)1 << 32)) return eng->fail(ZK_ERR_BAD_INPUT, "msm too large (points x windows >= 2^32)");
  const uint32_t pre_stride = tab ? (uint32_t)tab->len : 0u, pre_off = tab ? (uint32_t)toff : 0u;
  uint32_t seg = msm_pick_seg(npts, G2FLD);
  {
    // keep the average bucket at no more than ~4 segments: with 2^26 points and 2^16 buckets per window a 64-point
    // segment would leave 16 partial sums per bucket, i.e. every bucket on the slow heavy-bucket path
    const size_t avg = (tab ? npts * nwin : npts) >> (c - 1);
    uint32_t want = 1;
    while ((size_t)want * 4 < avg && want < 1024) want <<= 1;
    const bool overridden = getenv("ZK_MSM_SEG") || (G2FLD && getenv("ZK_MSM_SEG_G2"));
    if (!overridden && want > seg) seg = want;
  }
  const size_t max_segs = nkeys + max_sorted / seg + 1;   // every bucket has < count/seg + 1 segments
  constexpr int RED_THREADS = red_threads<Fld>();
  constexpr int RED_G = red_g<Fld>();
  const uint32_t bpw = (B + RED_THREADS * RED_G - 1) / (RED_THREADS * RED_G);
  const size_t iscan_blocks = (nkeys + ISCAN_BLOCK - 1) / ISCAN_BLOCK;

  // workspace layout
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_counts = take(nkeys * 4), o_lenhist = take(2 * SEG_BINS * 4), o_order = take(max_segs * 4), o_cursor = take(nkeys * 4), o_offsets = take((nkeys + 1) * 8),
         o_bt = take(iscan_blocks * 8), o_sorted = take(max_sorted * 4), o_segs = take(max_segs * sizeof(SegDesc)),
         o_partial = take(NB * max_segs * sizeof(XYZZ<Fld>)), o_buckets = take(NB * nkeys * sizeof(XYZZ<Fld>)),
         o_out = take(NB * (size_t)kwin * bpw * 2 * sizeof(XYZZ<Fld>)), o_heavy = take((nkeys + 1) * 4);
  // big-sort path (see the kernels): bins = (window, top BIG_HI bits of the bucket), low bits sorted per bin
    const int lo_bits = c - 1 - BIG_HI;
  const bool big = !tab && npts >= tune.bigsort_min && lo_bits >= 1 && lo_bits <= 12;
  size_t o_bins = 0, o_tmp = 0;
  if (big) {
    o_bins = take((3 * ((size_t)nwin << BIG_HI) + 1) * 4);
    o_tmp = take(max_sorted * sizeof(uint2));
  }
  hipError_t he = slot.ws.ensure(off);
  if (he != hipSuccess) return eng->hip_fail(he, "msm workspace");
  const size_t out_bytes = NB * (size_t)kwin * bpw * 2 * sizeof(XYZZ<Fld>);
  he = slot.ensure_pinned(out_bytes);
  if (he != hipSuccess) return eng->hip_fail(he, "msm pinned buffer");
  if (!slot.ev) {
    he = hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming);
    if (he != hipSuccess) return eng->hip_fail(he, "msm event");
  }
  char* ws = (char*)slot.ws.p;
  uint32_t* counts = (uint32_t*)(ws + o_counts);
  uint32_t* lenhist = (uint32_t*)(ws + o_lenhist);   // [SEG_BINS] histogram, [SEG_BINS] cursors
  uint32_t* order = (uint32_t*)(ws + o_order);
  uint32_t* cursor = (uint32_t*)(ws + o_cursor);
  uint2* offsets = (uint2*)(ws + o_offsets);
  uint2* bt = (uint2*)(ws + o_bt);
  uint32_t* sorted = (uint32_t*)(ws + o_sorted);
  SegDesc* segs = (SegDesc*)(ws + o_segs);
  using KF = typename KernelField<Fld>::type;     // same layout as Fld
  static_assert(sizeof(KF) == sizeof(Fld), "kernel field layout");
  XYZZ<KF>* partial = (XYZZ<KF>*)(ws + o_partial);
  XYZZ<KF>* buckets = (XYZZ<KF>*)(ws + o_buckets);
  XYZZ<KF>* out = (XYZZ<KF>*)(ws + o_out);
  uint32_t* heavy = (uint32_t*)(ws + o_heavy);

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
  MSM_HIP(hipMemsetAsync(counts, 0, o_lenhist + 2 * SEG_BINS * 4 - o_counts, st));   // counts and lenhist
  dim3 pg((unsigned)((npts + 255) / 256)), pb(256);
  const size_t plen = part_len ? part_len : npts;
  {
  ProfScope ps_(eng->prof, PROF_MSM_SORT, st, (double)npts);
  if (big) {
    const uint32_t nbins = (uint32_t)nwin << BIG_HI;
    uint32_t* bin_counts = (uint32_t*)(ws + o_bins);
    uint32_t* bin_base = bin_counts + nbins;
    uint32_t* bin_cursor = bin_base + nbins + 1;
    uint2* tmp = (uint2*)(ws + o_tmp);
    MSM_HIP(hipMemsetAsync(bin_counts, 0, nbins * 4, st));
    const unsigned tiles = (unsigned)((npts + BIG_TILE - 1) / BIG_TILE);
    msm_part_hist_kernel<FrP><<<dim3(tiles), dim3(BIG_THREADS), nbins * 4, st>>>((const Fr*)scalars, npts, coef_d, plen,
                                                                                c, nwin, wide, lo_bits, bin_counts);
    msm_bin_scan_kernel<<<dim3(1), dim3(BIG_THREADS), 0, st>>>(bin_counts, nbins, bin_base, bin_cursor);
    msm_part_scatter_kernel<FrP><<<dim3(tiles), dim3(BIG_THREADS), 2 * nbins * 4, st>>>(
        (const Fr*)scalars, npts, coef_d, plen, c, nwin, wide, lo_bits, bin_cursor, tmp);
    msm_bin_sort_kernel<<<dim3(nbins), dim3(BIG_THREADS), 0, st>>>(tmp, bin_base, lo_bits, (uint32_t)(c - 1), counts,
                                                                   sorted);
    MSM_STAGE("big sort");
  } else {
    msm_digits_kernel<FrP, 0><<<pg, pb, 0, st>>>((const Fr*)scalars, npts, coef_d, plen, c, nwin, wide, pre_stride,
                                                pre_off, counts, nullptr, nullptr);
    MSM_STAGE("digits/count");
  }
  iscan_block_kernel<<<dim3((unsigned)iscan_blocks), dim3(ISCAN_THREADS), 0, st>>>(counts, nkeys, bt, nullptr,
                                                                                   nullptr, 0, seg);
  iscan_carry_kernel<<<dim3(1), dim3(ISCAN_THREADS), 0, st>>>(bt, iscan_blocks);
  iscan_block_kernel<<<dim3((unsigned)iscan_blocks), dim3(ISCAN_THREADS), 0, st>>>(counts, nkeys, nullptr, bt,
                                                                                   offsets, 1, seg);
  MSM_STAGE("scan");
  msm_expand_kernel<<<dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st>>>(offsets, nkeys, cursor, segs, seg,
                                                                                 lenhist);
  msm_order_kernel<<<dim3((unsigned)((max_segs + 255) / 256)), dim3(256), 0, st>>>(segs, offsets, nkeys, seg, lenhist,
                                                                                   lenhist + SEG_BINS, order);
  MSM_STAGE("expand");
  if (!big)
    msm_digits_kernel<FrP, 1><<<pg, pb, 0, st>>>((const Fr*)scalars, npts, coef_d, plen, c, nwin, wide, pre_stride,
                                                pre_off, nullptr, cursor, sorted);
  }
  MSM_STAGE("scatter");
  {
  ProfScope ps_(eng->prof, G2FLD ? PROF_MSM_ACC_G2 : PROF_MSM_ACC_G1, st, (double)npts * NB);
  size_t acc_wgs = (max_segs + 127) / 128;
  {
    static const int cap_g1 = getenv("ZK_ACC_WGS_G1") ? atoi(getenv("ZK_ACC_WGS_G1")) : 0;
    static const int cap_g2 = getenv("ZK_ACC_WGS_G2") ? atoi(getenv("ZK_ACC_WGS_G2")) : 0;
    const int cap = G2FLD ? cap_g2 : cap_g1;
    if (cap > 0 && acc_wgs > (size_t)cap) acc_wgs = (size_t)cap;
  }
  msm_accumulate_kernel<KF><<<dim3((unsigned)acc_wgs, NB), dim3(128), 0, st>>>(
      (const Affine<KF>*)bases, (const Affine<KF>*)bases2, max_segs, sorted, segs, offsets, nkeys, order, partial);
  }
  MSM_STAGE("accumulate");
  {
  ProfScope ps_(eng->prof, G2FLD ? PROF_MSM_REDUCE_G2 : PROF_MSM_REDUCE, st, (double)nkeys * NB);   // units: buckets
  MSM_HIP(hipMemsetAsync(heavy, 0, 4, st));
  msm_finalize_kernel<KF><<<dim3((unsigned)((nkeys + 127) / 128), NB), dim3(128), 0, st>>>(
      partial, max_segs, offsets, nkeys, buckets, heavy);
  {
    size_t fin_lds = FIN_HEAVY_THREADS * sizeof(XYZZ<Fld>);
    // small fixed grid (it strides over the heavy list, which is empty for well-spread scalars): a launch of many
    // workgroups of this register-hungry kernel would wait for whole SIMDs to drain just to find nothing to do
    static const unsigned heavy_wgs = getenv("ZK_FIN_HEAVY_WGS") ? (unsigned)atoi(getenv("ZK_FIN_HEAVY_WGS")) : 48u;
    msm_finalize_heavy_kernel<KF><<<dim3(heavy_wgs ? heavy_wgs : 48u, NB), dim3(FIN_HEAVY_THREADS), fin_lds, st>>>(partial, max_segs, offsets,
                                                                                            nkeys, heavy, buckets);
  }
  MSM_STAGE("finalize");
  size_t red_lds = 2 * RED_THREADS * sizeof(XYZZ<Fld>);
  if (red_lds > 48 * 1024) {
    static std::mutex attr_mu;
    static std::vector<int> attr_done;          // per device (the attribute belongs to the device's code object)
    std::lock_guard<std::mutex> lk(attr_mu);
    if (std::find(attr_done.begin(), attr_done.end(), eng->device) == attr_done.end()) {
      MSM_HIP(hipFuncSetAttribute((const void*)msm_reduce_kernel<KF, RED_THREADS, RED_G>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)red_lds));
      attr_done.push_back(eng->device);
    }
  }
  msm_reduce_kernel<KF, RED_THREADS, RED_G><<<dim3((unsigned)(kwin * bpw), NB), dim3(RED_THREADS), red_lds, st>>>(
      buckets, nkeys, B, bpw, out);
  }
  MSM_HIP(hipGetLastError());
  MSM_STAGE("reduce");
  MSM_HIP(hipMemcpyAsync(slot.pinned, out, out_bytes, hipMemcpyDeviceToHost, st));
  MSM_HIP(hipEventRecord(slot.ev, st));
#undef MSM_HIP
#undef MSM_STAGE
  pend->active = true;
  pend->kwin = kwin;
  pend->c = c;
  pend->wide = wide;
  pend->nb = (int)NB;
  pend->red_k = RED_THREADS * RED_G;
  pend->bpw = bpw;
  pend->slot = &slot;
  pend->tab = std::move(tab);
  pend->tab2 = std::move(tab2);
  return ZK_OK;
}

template <class FrP, class Fld>
int msm_table_launch(IEngine* eng, const void* bases, size_t len, int c, int nwin, int wide, void* table,
                     hipStream_t st) {
  using KF = typename KernelField<Fld>::type;
  msm_table_kernel<KF><<<dim3((unsigned)((len + 127) / 128)), dim3(128), 0, st>>>(
      (const Affine<KF>*)bases, len, c, nwin, wide, (Affine<KF>*)table);
  hipError_t he = hipGetLastError();
  if (he != hipSuccess) return eng->hip_fail(he, "msm_table_kernel");
  return ZK_OK;
}

#define ZK_INSTANTIATE_MSM(FRP, FLD)                                                                              \
  template int msm_launch<FRP, FLD>(IEngine*, MsmSlot&, const MsmTuning&, const void*, const void*, const void*, \
                                    size_t, const Fp<FRP>*, size_t, hipStream_t, MsmPending*);                   \
  template int msm_table_launch<FRP, FLD>(IEngine*, const void*, size_t, int, int, int, void*, hipStream_t);

}  // namespace zk
