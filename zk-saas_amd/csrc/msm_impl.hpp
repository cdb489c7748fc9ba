// Definitions of msm_launch / msm_table_launch (declared in msm.hpp).  Included ONLY by the msm_<curve>_g<k>.hip
// translation units, each of which instantiates them for one (scalar field, coordinate field) pair: the Pippenger
// kernels are the most expensive part of the build and compile in parallel this way.
#pragma once
#include "msm.hpp"
#include "pack_split.hpp"

namespace zk {

template <class FrP, class Fld>
int msm_launch(IEngine* eng, MsmSlot& slot, const MsmTuning& tune, const void* bases, const void* bases2,
               const void* scalars, size_t npts, const Fp<FrP>* coef_d, size_t part_len, hipStream_t st,
               MsmPending* pend, const MsmBatchArg* ba) {
  using Fr = Fp<FrP>;
  pend->active = false;
  const unsigned NB = bases2 ? 2u : 1u;
  const size_t batch = ba ? (size_t)ba->nb : 1;    // scalar vectors multiplied against the same base vector(s)
  if (batch < 1 || batch > (size_t)MSM_MAXB) return eng->fail(ZK_ERR_BAD_INPUT, "bad msm batch");
  pend->batch = (int)batch;
  if (npts == 0) return ZK_OK;
  if (npts * batch >= ((size_t)1 << 31)) return eng->fail(ZK_ERR_BAD_INPUT, "msm too large");
  constexpr bool G2FLD = IsExtField<Fld>::value;
  const void* const bases_in = bases;            // the caller's points (the identity test reads them, not the table rows)
  const void* const bases2_in = bases2;
  // fixed-base table registered for this base vector (and the same window layout / offset for the second one)?
  size_t toff = 0, toff2 = 0;
  std::shared_ptr<const MsmTable> tab = TableRegistry::inst().find(bases, npts, sizeof(Affine<Fld>), FrP::BITS, &toff), tab2;
  if (tab && NB == 2) {
    tab2 = TableRegistry::inst().find(bases2, npts, sizeof(Affine<Fld>), FrP::BITS, &toff2);
    if (!tab2 || tab2->len != tab->len || toff2 != toff || tab2->c != tab->c) tab = nullptr;
    else bases2 = tab2->data;
  }
  if (tab) bases = tab->data;
  const int c_req = tab ? tab->c : msm_pick_c<FrP>(npts, G2FLD, tune.c_force);
  // BITS+1 bits (room for the signed-digit carry) are spread EVENLY over the windows: `wide` windows of c bits and
  // nwin-wide of c-1.  A plain c-bit split leaves a top window of a few bits (254 = 19*13 + 7) whose 64 buckets
  // each receive npts/64 points: hot atomics in the sort, long chains, and a heavy-bucket pass in every MSM.
  const int T = FrP::BITS + 1;
  const int nwin = (T + c_req - 1) / c_req;
  const int c = (T + nwin - 1) / nwin;            // widest window
  const int wide = T - nwin * (c - 1);            // 1 <= wide <= nwin
  const uint32_t B = 1u << (c - 1);
  // the kernels sort and sum a window range [w_begin, w_end): every launch covers all of them (a split into window groups
  // on two streams was measured in round 4 and removed, see MsmRunner)
  const int w_begin = 0, w_end = nwin;
  const int nwin_r = w_end - w_begin;
  const int kwin = tab ? 1 : nwin_r;              // bucket sets per scalar vector: with a table all windows share one
  const size_t nsets = batch * (size_t)kwin;      // bucket sets of the launch
  const size_t nkeys = nsets * B;
  const size_t max_sorted = npts * batch * nwin_r;
  if (max_sorted >= ((size_t)1 << 32)) return eng->fail(ZK_ERR_BAD_INPUT, "msm too large (points x windows >= 2^32)");
  const uint32_t pre_stride = tab ? (uint32_t)tab->len : 0u, pre_off = tab ? (uint32_t)toff : 0u;
  // accumulate lanes (msm.hpp "balanced partition"): every lane adds the same number of sorted entries
  // Extension field: a QUAD of lanes per range, one base-field value per lane (quad.hpp split_madd).  Rounds 2-3 held whole
  // Fq2 values per lane -- a pair of lanes per range, or one lane on large launches of 8-limb curves: 256 registers with
  // 13-99 spilled dwords, BLS12-381 G2 at a third of the multiplier's peak; those kernels are gone (measured with the quad
  // form: a 2^24-constraint BLS12-381 proof 1.42 -> 1.28 s, the SHA-256 proof 561 -> 593 proofs/s, table-free 395 -> 428)
  const MsmLanes ml = G2FLD ? msm_pick_lanes(max_sorted, SPLIT_WAVES<typename BaseParams<Fld>::type>, true, 4)
                            : msm_pick_lanes(max_sorted, ACC_WAVES<Fld>, false, 0);
  const uint32_t nlanes = ml.nlanes, tmin = ml.tmin, cap = ml.cap;
  // reduction geometry (msm.hpp "reduce stage A / B"): digit magnitudes k = hi * LO + lo in [1, B]
  const int lo_bits = c / 2;                       // LO = 2^lo_bits columns, HI = B / LO rows (+ the row of k = B)
  const uint32_t red_groups = (B >> lo_bits) + 1 + (1u << lo_bits);
  const int nslices = c;                           // (log2 HI + 1) row slices + lo_bits column slices
  const size_t iscan_blocks = (nkeys + ISCAN_BLOCK - 1) / ISCAN_BLOCK;

  // workspace layout
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  // identity bases are left out of the sort (msm.hpp MsmBaseId: the first sort-stage kernel gives them a zero scalar).
  // Two base vectors over the same scalars then get their OWN sorts (their identities differ: a fused sort could only
  // skip a point that is the identity in both) -- same kernels, grid.y = 2, the sort-stage arrays in two copies of one
  // workspace region (ZK_YSHIFT in the kernels).
  constexpr bool skip_on = true;
  // registered vectors know whether they hold an identity at all (zk_msm_precompute): without one there is no mask
  const bool none = tab && !tab->any_identity && (NB == 1 || (tab2 && !tab2->any_identity));
  const unsigned NS = (NB == 2 && !none) ? 2u : 1u;       // sorts of this launch
  // ---- big-sort plan (msm.hpp "big sort"): workgroup shape, points per tile, bin split, entry format
  constexpr size_t large_min = (size_t)4 << 20;
  const bool large = npts * batch >= large_min;    // multi-million-point launches: 1024-thread workgroups, one per CU
  const int sthr = large ? 1024 : 256;
  // small launches: ~1024 tiles so that they still fill the chip, up to 16 points per thread
  int ppt = large ? BIG_PTS_PER_THREAD : 16;
  if (!large)
    while (ppt > 1 && ((npts + (size_t)sthr * ppt - 1) / ((size_t)sthr * ppt)) * batch < 1024) ppt >>= 1;
  const size_t stage_max = 16384;
  if (large && tab)
    while (ppt > 1 && (size_t)nwin * sthr * ppt > stage_max) ppt >>= 1;       // with a table the whole tile is one round
  // small launches: at least 7 low bits per bin (a table-free proof has 12-bit buckets in 20 sets: 8 top bits made 5120
  // bins of 16 buckets, one workgroup each -- measured 428 -> 442-447 proofs/s table-free with 5 top bits)
  int sort_hi = std::min(msm_big_hi(nsets), std::max(1, c - 1 - (tab && tune.sort_lo_tab ? tune.sort_lo_tab : 7))), sort_lo;
  if (large) {
    // runs of level 1 are (tile entries per bucket set) / 2^hi long, runs of level 2 (chunk) / 2^lo: balance them
    auto lg = [](size_t v) { int l = 0; while (((size_t)1 << (l + 1)) <= v) l++; return l; };
    const int tp_eff = lg((size_t)sthr * ppt * (tab ? nwin : 1)), ch = lg((size_t)sthr * BIG_EPT);
    sort_hi = (c - 1 + tp_eff - ch + 1) / 2;
    if (sort_hi > c - 2) sort_hi = c - 2;
    if (c - 1 - sort_hi > 12) sort_hi = c - 1 - 12;
    while (sort_hi > 0 && (nsets << sort_hi) > (size_t)BIG_MAX_BINS) sort_hi--;
  }
  sort_lo = c - 1 - sort_hi;
  int idx_bits = 1;
  {
    const size_t max_idx = tab ? (size_t)nwin * tab->len : npts;
    while (((size_t)1 << idx_bits) < max_idx) idx_bits++;
  }
  const bool wide_fmt = idx_bits + 1 + sort_lo > 32;
  const size_t nbl = (size_t)kwin << sort_hi;      // bins of one scalar vector
  size_t stage_cap = 0;
  if (large) {
    const size_t fixed = 8 * nbl + 4 * (size_t)(sthr / 64) + 64;
    if (BIG_LDS_MAX > fixed) stage_cap = std::min(stage_max, (BIG_LDS_MAX - fixed) / (wide_fmt ? 8 : 6)) & ~(size_t)63;
  }
  const size_t tile_pts = (size_t)sthr * ppt;
  const bool big = npts * batch >= tune.bigsort_min && sort_hi >= 1 && sort_lo >= 1 && sort_lo <= 12 &&
                   (nsets << sort_hi) <= (size_t)BIG_MAX_BINS &&
                   (large ? stage_cap >= (tab ? (size_t)nwin * tile_pts : tile_pts) : 8 * nbl <= BIG_LDS_MAX);
  const int wgroup = tab ? nwin : (int)std::min<size_t>((size_t)nwin, std::max<size_t>(1, stage_cap / tile_pts));
  const size_t nbins_tot = nsets << sort_hi;
  // ---- sort region (replicated NS times)
  size_t o_counts = take(nkeys * 4), o_heavy = take(msm_heavy_cap(nlanes) * 8 + 16),
         o_bins = take(big ? msm_bins_words(nbins_tot) * 4 : 0),     // right behind the heavy list: one zeroing launch
         o_cursor = take(nkeys * 4), o_offsets = take((nkeys + 1) * 4), o_bt = take(iscan_blocks * 4),
         o_sorted = take(max_sorted * 4);
  const size_t o_skip = take(((npts + 63) / 64) * 8);
  const size_t o_k0 = take((size_t)nlanes * 4);                // first bucket of every accumulate lane
  const size_t o_canon = take(npts * batch * sizeof(Fr));      // canonical scalars (written by the first sort pass)
  size_t o_tmp = 0, o_tmp_lo = 0, o_tcnt = 0;
  // staged scatter: the histogram pass runs on the scatter's own tiles and hands over its per-tile counts (2 B per tile and
  // bin), so that the scatter does not walk the digits a third time
  const size_t tiles_big = ((npts + tile_pts - 1) / tile_pts) * batch;
  const bool use_tcnt = big && large && tile_pts % BIG_THREADS == 0 && tile_pts < 65536;
  if (big) {
    o_tmp = take(max_sorted * 4);
    if (wide_fmt) o_tmp_lo = take(max_sorted * 2);
    if (use_tcnt) o_tcnt = take(tiles_big * nbl * 2);
  }
  const size_t sort_region = off;
  const size_t ys = NS == 2 ? sort_region : 0;     // byte distance between the two copies
  off = sort_region * NS;
  // ---- per base vector
  size_t o_edge = take(NB * 2 * (size_t)nlanes * sizeof(XYZZ<Fld>)), o_buckets = take(NB * nkeys * sizeof(XYZZ<Fld>)),
         o_hpart = take(NB * msm_heavy_vcap(nlanes) * sizeof(XYZZ<Fld>)),      // chunk sums of split heavy buckets
         o_rc = take(NB * nsets * red_groups * sizeof(XYZZ<Fld>));
  hipError_t he = slot.ws.ensure(off);
  if (he != hipSuccess) return eng->hip_fail(he, "msm workspace");
  const size_t out_bytes = NB * nsets * nslices * sizeof(XYZZ<Fld>);
  he = slot.ensure_pinned(out_bytes + 64);           // + the sorted-entry counts of the launch's sorts (msm statistics)
  if (he != hipSuccess) return eng->hip_fail(he, "msm pinned buffer");
  if (!slot.ev) {
    he = hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming);
    if (he != hipSuccess) return eng->hip_fail(he, "msm event");
  }
  char* ws = (char*)slot.ws.p;
  uint32_t* counts = (uint32_t*)(ws + o_counts);
  uint32_t* cursor = (uint32_t*)(ws + o_cursor);
  uint32_t* heavy = (uint32_t*)(ws + o_heavy);
  uint32_t* k0 = (uint32_t*)(ws + o_k0);
  uint32_t* offsets = (uint32_t*)(ws + o_offsets);
  uint32_t* bt = (uint32_t*)(ws + o_bt);
  uint32_t* sorted = (uint32_t*)(ws + o_sorted);
  using KF = typename KernelField<Fld>::type;     // same layout as Fld
  static_assert(sizeof(KF) == sizeof(Fld), "kernel field layout");
  XYZZ<KF>* edge = (XYZZ<KF>*)(ws + o_edge);         // per base vector: head[nlanes], tail[nlanes]
  XYZZ<KF>* buckets = (XYZZ<KF>*)(ws + o_buckets);
  XYZZ<KF>* rc = (XYZZ<KF>*)(ws + o_rc);
  XYZZ<KF>* hpart = (XYZZ<KF>*)(ws + o_hpart);
  const uint32_t vcap = (uint32_t)msm_heavy_vcap(nlanes);
  // the slices go straight to the slot's pinned host buffer (device-visible, coherent: hipHostMalloc's default)
  XYZZ<KF>* out = (XYZZ<KF>*)slot.pinned;

#define MSM_HIP(x)                                           \
do {                                                       \
  hipError_t _e = (x);                                     \
  if (_e != hipSuccess) return eng->hip_fail(_e, #x);      \
} while (0)
// stage marker: with -DZK_MSM_DEBUG_SYNC (debug builds only) the stream is synchronised and checked after every stage
#ifdef ZK_MSM_DEBUG_SYNC
#define MSM_STAGE(name)                                                          \
do {                                                                           \
  hipError_t _e = hipStreamSynchronize(st);                                    \
  fprintf(stderr, "[zk msm] %s done (%s) npts=%zu c=%d nwin=%d\n", name,       \
          hipGetErrorString(_e), npts, c, nwin);                               \
  if (_e != hipSuccess) return eng->hip_fail(_e, name);                        \
} while (0)
#else
#define MSM_STAGE(name) do { } while (0)
#endif
  {
  // counts (the two-level sort writes every offset itself; it needs its bin counters and the ticket zeroed) and the
  // heavy-bucket counter
  if (big) MSM_HIP(msm_zero(heavy, o_bins + (nbins_tot + 1) * 4 - o_heavy, st, NS, ys));
  else MSM_HIP(msm_zero(counts, o_heavy + 16 - o_counts, st, NS, ys));
  dim3 pg((unsigned)((npts + 255) / 256), NS), pb(256);
  dim3 pgb((unsigned)((npts * batch + 255) / 256), NS);     // one thread per (scalar vector, point)
  MsmScalars<Fr> sc{};
  for (size_t b = 0; b < batch; b++) sc.p[b] = (const Fr*)(ba ? ba->p[b] : scalars);
  sc.npts = (uint32_t)npts;
  sc.nb = (uint32_t)batch;
  sc.sets_per = (uint32_t)kwin;
  Fr* canon = (Fr*)(ws + o_canon);
  // identity bases: the first sort-stage kernel looks at the caller's points itself (msm.hpp MsmBaseId)
  MsmBaseId bid;
  if (skip_on && !none) {
    bid.b0 = bases_in;
    bid.b1 = bases2_in;
    bid.elem16 = (uint32_t)(sizeof(Affine<KF>) / 16);
    if (tune.skip_kernel) {
      uint32_t* skip = (uint32_t*)(ws + o_skip);
      msm_skip_mask_kernel<KF><<<pg, pb, 0, st>>>((const Affine<KF>*)bases_in, (const Affine<KF>*)bases2_in, npts, skip, ys);
      bid.skip = skip;
      bid.skip_ys = ys / 4;
    }
  }
  const size_t plen = part_len ? part_len : npts;
  {
  ProfScope ps_(eng->prof, PROF_MSM_SORT, st, (double)npts * batch);
  if (big) {
    uint32_t* bins = (uint32_t*)(ws + o_bins);
    uint32_t* tmp = (uint32_t*)(ws + o_tmp);
    uint16_t* tmp_lo = (uint16_t*)(ws + o_tmp_lo);
    const uint32_t wmask = tab ? 0u : ~0u;
    // hist: 256-thread tiles of its own (any tiling of the points gives the same bin totals)
    uint16_t* tcnt = use_tcnt ? (uint16_t*)(ws + o_tcnt) : nullptr;
    {
      const int hp = use_tcnt ? (int)(tile_pts / BIG_THREADS) : ppt;          // the scatter's tiles
      const unsigned tpv = (unsigned)((npts + (size_t)BIG_THREADS * hp - 1) / ((size_t)BIG_THREADS * hp));
      const size_t hl = (nbins_tot + BIG_THREADS / 64) * 4;      // tile histogram (one vector's bins); all bins for the last workgroup's scan
      if (hl > 48 * 1024) MSM_HIP(msm_lds_attr((const void*)msm_hist_kernel<FrP>, hl, eng->device));
      msm_hist_kernel<FrP><<<dim3(tpv * (unsigned)batch, NS), dim3(BIG_THREADS), hl, st>>>(
          sc, coef_d, plen, c, w_end, wide, sort_hi, sort_lo, hp, tpv, wmask, w_begin, bins, bid, canon, tcnt, ys);
    }
    if (large) {
      const unsigned tpv = (unsigned)((npts + tile_pts - 1) / tile_pts);
      const size_t l1 = (2 * nbl + (size_t)(sthr / 64)) * 4 + stage_cap * (wide_fmt ? 8 : 6);
      // the points per thread are a template parameter of the staged scatter (its scalars live in registers): with a
      // table the whole tile is one round of nwin windows and the tile shrinks to fit the stage (ppt 4 / 2 / 1)
#define ZK_SCATTER(P_, W_)                                                                                              \
  do {                                                                                                                 \
    if (l1 > 48 * 1024) MSM_HIP(msm_lds_attr((const void*)msm_scatter_kernel<FrP, 1024, P_, W_>, l1, eng->device));    \
    msm_scatter_kernel<FrP, 1024, P_, W_><<<dim3(tpv * (unsigned)batch, NS), dim3(1024), l1, st>>>(                    \
        sc, c, w_end, wide, sort_hi, sort_lo, tpv, wmask, w_begin, wgroup, pre_stride, pre_off, idx_bits, (uint32_t)stage_cap,   \
        bins, tmp, tmp_lo, canon, tcnt, ys);                                                                           \
  } while (0)
#define ZK_SCATTER_P(P_)                  \
  do {                                    \
    if (wide_fmt) ZK_SCATTER(P_, true);   \
    else ZK_SCATTER(P_, false);           \
  } while (0)
      switch (ppt) {
        case BIG_PTS_PER_THREAD: ZK_SCATTER_P(BIG_PTS_PER_THREAD); break;
        case 4: ZK_SCATTER_P(4); break;
        case 2: ZK_SCATTER_P(2); break;
        case 1: ZK_SCATTER_P(1); break;
        default: return eng->fail(ZK_ERR_GENERIC, "msm: unsupported scatter tile");
      }
#undef ZK_SCATTER_P
#undef ZK_SCATTER
    } else {
      const unsigned tpv = (unsigned)((npts + (size_t)BIG_THREADS * ppt - 1) / ((size_t)BIG_THREADS * ppt));
      const size_t l1 = 2 * nbl * 4;
      if (l1 > 48 * 1024) MSM_HIP(msm_lds_attr(wide_fmt ? (const void*)msm_scatter_direct_kernel<FrP, true> : (const void*)msm_scatter_direct_kernel<FrP, false>, l1, eng->device));
      if (wide_fmt)
        msm_scatter_direct_kernel<FrP, true><<<dim3(tpv * (unsigned)batch, NS), dim3(BIG_THREADS), l1, st>>>(
            sc, c, w_end, wide, sort_hi, sort_lo, ppt, tpv, wmask, w_begin, pre_stride, pre_off, idx_bits, bins, tmp, tmp_lo, canon, ys);
      else
        msm_scatter_direct_kernel<FrP, false><<<dim3(tpv * (unsigned)batch, NS), dim3(BIG_THREADS), l1, st>>>(
            sc, c, w_end, wide, sort_hi, sort_lo, ppt, tpv, wmask, w_begin, pre_stride, pre_off, idx_bits, bins, tmp, tmp_lo, canon, ys);
    }
    const size_t l2 = (2 * ((size_t)1 << sort_lo) + 1 + (size_t)(sthr / 64)) * 4 + (large ? (size_t)sthr * BIG_EPT * 6 : 0);
#define ZK_BINSORT(THR_, W_, S_)                                                                                      \
  do {                                                                                                                \
    if (l2 > 48 * 1024) MSM_HIP(msm_lds_attr((const void*)msm_binsort_kernel<THR_, W_, S_>, l2, eng->device));        \
    msm_binsort_kernel<THR_, W_, S_><<<dim3((unsigned)nbins_tot, NS), dim3(THR_), l2, st>>>(                          \
        tmp, tmp_lo, bins, (uint32_t)nbins_tot, sort_hi, sort_lo, (uint32_t)(c - 1), idx_bits, (uint32_t)nkeys, nlanes, \
        tmin, cap, offsets, sorted, k0, ys);                                                                          \
  } while (0)
    if (large) {
      if (wide_fmt) ZK_BINSORT(1024, true, true);
      else ZK_BINSORT(1024, false, true);
    } else {
      if (wide_fmt) ZK_BINSORT(256, true, false);
      else ZK_BINSORT(256, false, false);
    }
#undef ZK_BINSORT
    MSM_STAGE("big sort");
  } else {
    msm_digits_kernel<FrP, 0><<<pgb, pb, 0, st>>>(sc, coef_d, plen, c, nwin, wide, pre_stride, pre_off, counts, nullptr,
                                                 nullptr, bid, canon, ys);
    MSM_STAGE("digits/count");
    iscan_block_kernel<<<dim3((unsigned)iscan_blocks, NS), dim3(ISCAN_THREADS), 0, st>>>(counts, nkeys, bt, nullptr,
                                                                                         nullptr, nullptr, 0, ys);
    iscan_carry_kernel<<<dim3(1, NS), dim3(ISCAN_THREADS), 0, st>>>(bt, iscan_blocks, ys);
    iscan_block_kernel<<<dim3((unsigned)iscan_blocks, NS), dim3(ISCAN_THREADS), 0, st>>>(counts, nkeys, nullptr, bt,
                                                                                         offsets, cursor, 1, ys);
    MSM_STAGE("scan");
    msm_lane_start_kernel<<<dim3((nlanes + 255) / 256, NS), dim3(256), 0, st>>>(offsets, (uint32_t)nkeys, nlanes, tmin, cap, k0, ys);
    msm_digits_kernel<FrP, 1><<<pgb, pb, 0, st>>>(sc, coef_d, plen, c, nwin, wide, pre_stride, pre_off, nullptr, cursor,
                                                 sorted, bid, canon, ys);
  }
  }
  }
  MSM_STAGE("scatter");
  {
  ProfScope ps_(eng->prof, G2FLD ? PROF_MSM_ACC_G2 : PROF_MSM_ACC_G1, st, (double)npts * NB * batch);
  if (tune.gate.sorted_ev) {
    MSM_HIP(hipEventRecord(tune.gate.sorted_ev, st));
    if (tune.gate.sorted_cnt) tune.gate.sorted_cnt->fetch_add(1, std::memory_order_release);
  }
  // the two host-side gates are bounded by the context's deadline (round 6): a launch that never comes -- the task that
  // would raise the flag failed before it, or never ran -- is an error with a name, not a spin for ever
  if (tune.gate.n_wait_sorted) {
    if (!eng->spin_until([&] { return tune.gate.sorted_cnt->load(std::memory_order_acquire) >= tune.gate.sorted_need; }))
      return eng->fail(ZK_ERR_GENERIC, "msm launch: the sorts of the proof's other MSMs were not all enqueued within the deadline (" +
                                           std::to_string(tune.gate.sorted_cnt->load()) + " of " + std::to_string(tune.gate.sorted_need) + ")");
    for (int i = 0; i < tune.gate.n_wait_sorted; i++) MSM_HIP(hipStreamWaitEvent(st, tune.gate.wait_sorted[i], 0));
  }
  if (tune.gate.wait_ev) {
    if (tune.gate.wait_flag && !eng->spin_until([&] { return tune.gate.wait_flag->load(std::memory_order_acquire) != 0; }))
      return eng->fail(ZK_ERR_GENERIC, "msm launch: the accumulate kernel ahead of this one in the batch's chain was not enqueued within the deadline");
    MSM_HIP(hipStreamWaitEvent(st, tune.gate.wait_ev, 0));
  }
  if constexpr (G2FLD) {
    msm_accumulate_split_kernel<typename BaseParams<Fld>::type><<<dim3((nlanes + 31) / 32, NB), dim3(128), tune.acc_lds, st>>>(
        bases, bases2, sorted, offsets, (uint32_t)nkeys, nlanes, tmin, cap, buckets, edge, heavy, k0, ys);
  } else {
    msm_accumulate_kernel<KF><<<dim3((nlanes + 127) / 128, NB), dim3(128), tune.acc_lds, st>>>(
        (const Affine<KF>*)bases, (const Affine<KF>*)bases2, sorted, offsets, (uint32_t)nkeys, nlanes, tmin, cap, buckets, edge,
        heavy, k0, ys, 0);
  }
  if (tune.gate.signal_ev) {
    MSM_HIP(hipEventRecord(tune.gate.signal_ev, st));
    if (tune.gate.signal_flag) tune.gate.signal_flag->store(1, std::memory_order_release);
  }
  }
  MSM_STAGE("accumulate");
  {
  ProfScope ps_(eng->prof, G2FLD ? PROF_MSM_REDUCE_G2 : PROF_MSM_REDUCE, st, (double)nkeys * NB);   // units: buckets
  const int qt = quad_threads(batch > 1), qvl = qt / 4;
  const size_t quad_lds = (size_t)qvl * sizeof(XYZZ<Fld>);
  // buckets spread over many lanes (none for well-spread scalars: the workgroups read a zero count and leave)
  // (always one-wave workgroups: with 256 threads this launch, which normally reads one word and leaves, waited 90-150 us
  // for four free wave slots on one CU in a single proof's timeline)
  msm_heavy_kernel<KF><<<dim3(2048, NB), dim3(64), (size_t)16 * sizeof(XYZZ<Fld>), st>>>(edge, nlanes, tmin, cap, offsets, (uint32_t)nkeys,
                                                                    buckets, heavy, hpart, vcap, ys);
  MSM_STAGE("heavy buckets");
  {
    // capped grid (grid-stride inside): enough one-wave workgroups to cover the chip a few times over
    const size_t fin_wgs = std::min<size_t>((nkeys + FIN_THREADS / 4 - 1) / (FIN_THREADS / 4), 8192);
    // + workgroups that sum the chunk sums of split heavy buckets (they read the list's counter and leave, normally)
    const unsigned fin_extra = 256;
    msm_finalize_kernel<KF><<<dim3((unsigned)fin_wgs + fin_extra, NB), dim3(FIN_THREADS),
                              (size_t)(FIN_THREADS / 4) * sizeof(XYZZ<Fld>), st>>>(
        edge, nlanes, tmin, cap, offsets, (uint32_t)nkeys, buckets, heavy, hpart, vcap, (uint32_t)fin_wgs, ys);
  }
  MSM_STAGE("finalize");
  // quads per group: few groups (one bucket set) -> whole workgroups per group, shortest dependent chain; many groups
  // (one bucket set per window) -> 4 quads per group, waves stay full
  // As many quads per group as keep the whole launch resident at once (a second generation of workgroups doubles a
  // kernel that is one dependent chain): the chip holds 1024 SIMDs x (2 waves of the extension-field kernels, 3 of the
  // base-field ones) x 16 quads.
  const size_t tot_groups = (size_t)red_groups * NB * nsets;
  const size_t tot_slices = (size_t)nslices * NB * nsets;
  const size_t cap_quads = (size_t)1024 * (G2FLD ? 2 : 3) * 16;
  auto pick_nvl = [&](size_t groups) {
    int v = qvl;
    while (v > 4 && groups * (size_t)v > cap_quads) v >>= 1;
    return v;
  };
  int nvl_a = pick_nvl(tot_groups), nvl_b = pick_nvl(tot_slices);
  if (nvl_a > qvl) nvl_a = qvl;
  if (nvl_b > qvl) nvl_b = qvl;
  const unsigned gpw_a = (unsigned)(qvl / nvl_a), gpw_b = (unsigned)(qvl / nvl_b);
  msm_reduce_a_kernel<KF><<<dim3((red_groups + gpw_a - 1) / gpw_a, NB * (unsigned)nsets), dim3((unsigned)qt), quad_lds, st>>>(
      buckets, B, lo_bits, nvl_a, rc);
  msm_reduce_b_kernel<KF><<<dim3(((unsigned)nslices + gpw_b - 1) / gpw_b, NB * (unsigned)nsets), dim3((unsigned)qt), quad_lds,
                            st>>>(rc, B, lo_bits, nvl_b, out,
                                  // mixed additions actually performed = sorted entries (identity bases and zero digits
                                  // leave none): one count per sort
                                  offsets + nkeys, ys, NS, (uint32_t*)((char*)slot.pinned + out_bytes));
  }
  MSM_HIP(hipGetLastError());
  MSM_STAGE("reduce");
  MSM_HIP(hipEventRecord(slot.ev, st));
#undef MSM_HIP
#undef MSM_STAGE
  pend->active = true;
  pend->kwin = kwin;
  pend->c = c;
  pend->wide = wide;
  pend->nb = (int)NB;
  pend->lo_bits = lo_bits;
  pend->slot = &slot;
  pend->stats_off = out_bytes;
  pend->nsorts = (int)NS;
  pend->g2 = G2FLD;
  pend->offered = npts * batch * NB * (size_t)nwin_r;
  pend->w0 = w_begin;
  pend->tabbed = (bool)tab;
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

template <class FrP, class Fld>
int pack_points_split_launch(IEngine* eng, const void* points, size_t nchunks, int n, const uint8_t* dig, int jlen,
                             const void* beta, void* shares, hipStream_t st) {
  if constexpr (IsExtField<Fld>::value) {
    using P = typename BaseParams<Fld>::type;
    Fp<P> b;
    memcpy(&b, beta, sizeof(b));
    pss_pack_points_jsf_split_kernel<FrP, P><<<dim3((unsigned)((nchunks * 4 + 127) / 128), (unsigned)n), dim3(128), 0, st>>>(
        (const Affine<Fp2<P>>*)points, nchunks, n, dig, jlen, b, (Affine<Fp2<P>>*)shares);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return eng->hip_fail(he, "pss_pack_points_jsf_split_kernel");
    return ZK_OK;
  } else {
    return eng->fail(ZK_ERR_BAD_INPUT, "the quad-split pack kernel is for extension-field points");
  }
}

template <class FrP, class Fld>
int base_mul_split_launch(IEngine* eng, const void* scalars, size_t len, const void* table, int nwin, int wb, void* out,
                          hipStream_t st) {
  if constexpr (IsExtField<Fld>::value) {
    using P = typename BaseParams<Fld>::type;
    const dim3 grid((unsigned)((len * 4 + 127) / 128)), block(128);
    if (wb == 16)
      fixed_base_mul_split_kernel<FrP, P, 16><<<grid, block, 0, st>>>((const Fp<FrP>*)scalars, len, (const Affine<Fp2<P>>*)table,
                                                                      nwin, (Affine<Fp2<P>>*)out);
    else
      fixed_base_mul_split_kernel<FrP, P, 8><<<grid, block, 0, st>>>((const Fp<FrP>*)scalars, len, (const Affine<Fp2<P>>*)table,
                                                                     nwin, (Affine<Fp2<P>>*)out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return eng->hip_fail(he, "fixed_base_mul_split_kernel");
    return ZK_OK;
  } else {
    return eng->fail(ZK_ERR_BAD_INPUT, "the quad-split fixed-base kernel is for extension-field points");
  }
}

#define ZK_INSTANTIATE_MSM(FRP, FLD)                                                                              \
  template int msm_launch<FRP, FLD>(IEngine*, MsmSlot&, const MsmTuning&, const void*, const void*, const void*, \
                                    size_t, const Fp<FRP>*, size_t, hipStream_t, MsmPending*, const MsmBatchArg*);  \
  template int msm_table_launch<FRP, FLD>(IEngine*, const void*, size_t, int, int, int, void*, hipStream_t);     \
  template int pack_points_split_launch<FRP, FLD>(IEngine*, const void*, size_t, int, const uint8_t*, int, const void*, void*, hipStream_t); \
  template int base_mul_split_launch<FRP, FLD>(IEngine*, const void*, size_t, const void*, int, int, void*, hipStream_t);

}  // namespace zk
