// Part of class Engine<Cfg> (engine_impl.hpp includes this file INSIDE the class body): PSS entry points, vector helpers, fft1, the king of d_fft, deg_red, d_pp.
// Split out of engine_impl.hpp in round 5 (one 3 400-line class body had stopped being navigable); not a stand-alone header.

  // ---------------------------------------------------------------- PSS entry points
  template <int L>
  int pack_l(const Fr* sec, size_t nch, int order, uint64_t seed, bool det, Fr* shares, hipStream_t st) {
    dim3 grid((unsigned)((nch + KING_THREADS - 1) / KING_THREADS)), block(KING_THREADS);
    if (det)
      pss_pack_kernel<FrP, L, true><<<grid, block, 0, st>>>(sec, nch, order, rs(0), pmat_, pack2_, shares);
    else
      pss_pack_kernel<FrP, L, false><<<grid, block, 0, st>>>(sec, nch, order, rs(seed), pmat_, pack2_, shares);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int pss_pack(const void* secrets, size_t nchunks, int order, uint64_t seed, bool det, void* shares,
               hipStream_t st) override {
    if (nchunks == 0) return ZK_OK;
    if (!secrets || !shares) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const Fr* s = (const Fr*)secrets;
    Fr* o = (Fr*)shares;
    switch (l) {
      case 1: return pack_l<1>(s, nchunks, order, seed, det, o, st);
      case 2: return pack_l<2>(s, nchunks, order, seed, det, o, st);
      case 4: return pack_l<4>(s, nchunks, order, seed, det, o, st);
      default: return pack_l<8>(s, nchunks, order, seed, det, o, st);
    }
  }

  template <int L>
  int unpack_l(const Fr* sh, int np, size_t nch, const Fr* U, Fr* sec, hipStream_t st) {
    dim3 grid((unsigned)((nch + KING_THREADS - 1) / KING_THREADS)), block(KING_THREADS);
    pss_unpack_kernel<FrP, L><<<grid, block, 0, st>>>(sh, np, nch, U, sec);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int pss_unpack(const void* shares, const uint32_t* parties, int np, size_t nchunks, bool two, void* secrets,
                 hipStream_t st) override {
    if (nchunks == 0) return ZK_OK;
    if (!secrets || !shares) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const Fr* U = nullptr;
    if (!two) {
      if (np != n) return fail(ZK_ERR_BAD_INPUT, "unpack needs all n shares");
      U = umats_[key_u(0xffffffffu, 1)];
    } else {
      int rc = umat_for(parties, np, &U);
      if (rc) return rc;
    }
    const Fr* s = (const Fr*)shares;
    Fr* o = (Fr*)secrets;
    switch (l) {
      case 1: return unpack_l<1>(s, np, nchunks, U, o, st);
      case 2: return unpack_l<2>(s, np, nchunks, U, o, st);
      case 4: return unpack_l<4>(s, np, nchunks, U, o, st);
      default: return unpack_l<8>(s, np, nchunks, U, o, st);
    }
  }

  // ---------------------------------------------------------------- vector helpers
  int bitrev(void* x, int log_len, hipStream_t st) override {
    if (log_len < 0 || log_len > 40) return fail(ZK_ERR_BAD_INPUT, "bad length");
    size_t len = (size_t)1 << log_len;
    bitrev_kernel<Fr><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st>>>((Fr*)x, log_len);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int vec_add(void* x, const void* y, size_t len, hipStream_t st) override {
    if (!len) return ZK_OK;
    vec_add_kernel<Fr><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st>>>((Fr*)x, (const Fr*)y, len);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int vec_scale(void* x, const void* k, size_t len, hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!x || !k) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    vec_scale_kernel<Fr><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st>>>((Fr*)x, Fr::from_limbs((const uint32_t*)k),
                                                                                 len);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int vec_mul_sub(void* out, const void* a, const void* b, const void* c, size_t len, hipStream_t st) override {
    if (!len) return ZK_OK;
    const unsigned bk = (unsigned)king_block(len);       // one-wave groups at proof sizes (see king_block)
    size_t wgs = (len + bk - 1) / bk;
    if (bk == 64 && wgs > 512) wgs = 512;                // grid-stride inside (see the kernel)
    vec_mul_sub_kernel<Fr><<<dim3((unsigned)wgs), dim3(bk), 0, st>>>((Fr*)out, (const Fr*)a, (const Fr*)b, (const Fr*)c, len);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }

  // Base-field primitives of the group kernels, exposed for the parity tests (edge values against Python integers):
  //   op 0: out[i] = a*b - c*d through Fp::mul_sub_mul (ONE reduction; Y3 of every XYZZ formula)
  //   op 1: out[2i], out[2i+1] = (a + b u)(c + d u) through the Fq2 product of the MSM kernels (lazy reduction on 8 limbs)
  //   op 2 + k (k < 16): the lazy-residue operations of the G1 accumulate kernel with operand j entered as x + p when bit j
  //         of k is set: out[5i ..] = a b, a - b, 2a, a b - c d (canonical; all-ones if a result left [0, 2p)), and the
  //         raw word (a == c) | (a == 0) << 1 as is_zero_lazy sees them
  int fq_selftest(int op, const void* a, const void* b, const void* c, const void* d, size_t len, void* out,
                  hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    if (!len) return ZK_OK;
    if (!a || !b || !c || !d || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (op < 0 || op > 17) return fail(ZK_ERR_BAD_INPUT, "op must be 0 .. 17");
    if (op >= 2 && !Fq::LAZY_OK) return fail(ZK_ERR_BAD_INPUT, "this base field has no lazy form");
    if (op == 1 && !Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    fq_selftest_kernel<Fq, (Fq::N == 8)><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st>>>(
        op, (const Fq*)a, (const Fq*)b, (const Fq*)c, (const Fq*)d, len, (Fq*)out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }

  // ---------------------------------------------------------------- fft1 (dfft/mod.rs:178-208)
  int fft1(void* shares, int log_m, int inverse, size_t batch, const void* add, hipStream_t st) override {
    return fft1_src(shares, log_m, inverse, batch, add, st, NttSrc<Fr>{{nullptr, nullptr, nullptr}, 1});
  }
  // src (optional): the first pass reads vector y from src.p[y / src.per] instead of from `shares` (out of place)
  int fft1_src(void* shares, int log_m, int inverse, size_t batch, const void* add, hipStream_t st, NttSrc<Fr> src) {
    int log_l = ilog2(l);
    if (log_m < log_l) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    int log_n = log_m - log_l;
    if (batch == 0) return ZK_OK;
    const Fr* tw = nullptr;
    int rc = gentab(log_m, inverse, st, &tw);
    if (rc) return rc;
    Fr* data = (Fr*)shares;
    size_t nvec = (size_t)1 << log_n;
    if ((log_n < NTT_TILE_BITS_SMALL || force_simple_ntt) && src.p[0]) {
      for (size_t y = 0; y < batch; y += src.per)
        ZK_HIP(hipMemcpyAsync(data + y * nvec, src.p[y / src.per], (size_t)src.per * nvec * sizeof(Fr),
                              hipMemcpyDeviceToDevice, st));
    }
    if (log_n < NTT_TILE_BITS_SMALL || force_simple_ntt) {
      for (int s = 1; s <= log_n; s++) {
        size_t work = (nvec / 2) * batch;
        ntt_stage_simple_kernel<Fr><<<dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st>>>(data, log_n, s, tw,
                                                                                                log_l, batch);
        ZK_HIP(hipGetLastError());
      }
      if (add) return vec_add(shares, add, nvec * batch, st);
      return ZK_OK;
    }
    if (ntt_tile_bits(log_n) == NTT_TILE_BITS_SMALL && (small_groups() || log_n < NTT_TILE_BITS))
      return fft1_tiled<NTT_TILE_BITS_SMALL>(data, log_n, log_l, batch, tw, (const Fr*)add, st, src);
    return fft1_tiled<NTT_TILE_BITS>(data, log_n, log_l, batch, tw, (const Fr*)add, st, src);
  }
  template <int TB>
  int fft1_tiled(Fr* data, int log_n, int log_l, size_t batch, const Fr* tw, const Fr* add, hipStream_t st,
                 NttSrc<Fr> src) {
    constexpr size_t TILE = (size_t)1 << TB;
    const size_t nvec = (size_t)1 << log_n;
    NttPlan plan = make_ntt_plan(log_n, TB);
    for (int p = 0; p < plan.npass; p++) {
      const NttPass& ps = plan.pass[p];
      int rbits = ps.s1 - ps.s0;
      // large tile: every fourth stage twiddle in LDS (72 KB: two workgroups per CU), see ntt_pass_kernel
      const int tws = (TB >= 10 && rbits >= 4) ? 2 : 0;
      const int tws_eff = rbits >= 4 ? tws : 0;      // one value for the LDS size AND the kernel argument
      size_t lds = (size_t)(sizeof(Fr) / 16) * 16 * (TILE + ((((size_t)1 << rbits) / 2) >> tws_eff) + 1);
      bool& attr_set = ntt_attr_set_[TB == NTT_TILE_BITS_SMALL ? 0 : 1];     // per engine, i.e. per device
      if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<Fr, TB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((sizeof(Fr) / 16) * 16 * (TILE + TILE / 2 + 1))));
        attr_set = true;
      }
      dim3 grid((unsigned)(nvec >> TB), (unsigned)batch);
      ProfScope ps_(prof, PROF_NTT_PASS, st, (double)nvec * batch);
      ntt_pass_kernel<Fr, TB><<<grid, dim3((unsigned)(TILE / 4)), lds, st>>>(
          data, log_n, ps.s0, ps.s1, ps.cbits, tw, log_l, p == plan.npass - 1 ? add : nullptr,
          p == 0 ? src : NttSrc<Fr>{{nullptr, nullptr, nullptr}, 1}, tws_eff);
      ZK_HIP(hipGetLastError());
    }
    return ZK_OK;
  }

  // ---------------------------------------------------------------- king of d_fft (dfft/mod.rs:264-304)
  template <int L>
  int king_l(const Fr* in, const KingBatch<Fr>& kb, int batch, int np, int log_lc, const Fr* U, const Fr* gen,
             const GTab* gt, const Fr* in_scale, int rearrange, uint64_t seed_, Fr* out, bool negate, hipStream_t st,
             const KingRange* range = nullptr) {
    size_t Lc = (size_t)1 << log_lc;
    const size_t kbk = (size_t)king_block(Lc);
    size_t Wc = Lc < kbk ? Lc : kbk;
    size_t lds = (size_t)L * Wc * sizeof(Fr);
    const size_t cols = range ? range->cnt : Lc;          // king workgroup columns of this launch
    if (range && (Wc == Lc || range->rs % Wc || range->seg % Wc || range->cnt % Wc || batch != 1))
      return fail(ZK_ERR_GENERIC, "bad king range");
    if (!cols) return ZK_OK;
    dim3 grid((unsigned)(cols / Wc), (unsigned)batch), block((unsigned)kbk);
    ProfScope ps_(prof, PROF_KING, st, (double)cols * batch);
    const RngSeed seed = rs(seed_, kb.seed_off((uint32_t)batch - 1) + 1);
    const uint32_t rs_ = range ? range->rs : 0u, seg_ = range ? range->seg : 0u;
    if (negate)
      king_fft2_kernel<FrP, L, true><<<grid, block, lds, st>>>(in, kb, np, (uint32_t)log_lc, U, pmat_, gen,
                                                               gt ? gt->tab : nullptr, gt ? gt->step : nullptr,
                                                               in_scale, pack2_, rearrange, seed, out, rs_, seg_);
    else
      king_fft2_kernel<FrP, L, false><<<grid, block, lds, st>>>(in, kb, np, (uint32_t)log_lc, U, pmat_, gen,
                                                                gt ? gt->tab : nullptr, gt ? gt->step : nullptr,
                                                                in_scale, pack2_, rearrange, seed, out, rs_, seg_);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  // `batch` vectors at in + y * stride -> out + y * stride, masks per item (all in-masks present or all absent),
  // share randomness seed + y
  int king_dispatch_batch(const Fr* in, const KingBatch<Fr>& kb, int batch, int np, int log_m, int inverse,
                          const Fr* U, const void* g, int scale, int rearrange, uint64_t seed, Fr* out, bool negate,
                          hipStream_t st, const KingRange* range = nullptr) {
    int log_l = ilog2(l);
    if (log_m < log_l) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (batch < 1 || batch > KING_BATCH) return fail(ZK_ERR_BAD_INPUT, "bad king batch");
    const bool has_in_mask = kb.in_mask[0] != nullptr;
    for (int y = 1; y < batch; y++)
      if ((kb.in_mask[y] != nullptr) != has_in_mask) return fail(ZK_ERR_BAD_INPUT, "mixed in-masks in one batch");
    const Fr* gen = nullptr;
    int rc = gentab(log_m, inverse, st, &gen);
    if (rc) return rc;
    Fr gv = g ? Fr::from_limbs((const uint32_t*)g) : Fr::one();
    GTab gt{};
    // 1/m: folded into the g^i table when there is no in-mask, applied to the shares at load otherwise
    bool fold = scale && !has_in_mask;
    const Fr* in_scale = nullptr;
    if (scale && has_in_mask) {
      rc = size_inv_dev(log_m, st, &in_scale);
      if (rc) return rc;
    }
    bool need = fold || gv != Fr::one();
    if (need) {
      rc = gtab(log_m, gv, fold, st, &gt);
      if (rc) return rc;
    }
    int log_lc = log_m - log_l;
    switch (l) {
      case 1: return king_l<1>(in, kb, batch, np, log_lc, U, gen, need ? &gt : nullptr, in_scale, rearrange, seed, out, negate, st, range);
      case 2: return king_l<2>(in, kb, batch, np, log_lc, U, gen, need ? &gt : nullptr, in_scale, rearrange, seed, out, negate, st, range);
      case 4: return king_l<4>(in, kb, batch, np, log_lc, U, gen, need ? &gt : nullptr, in_scale, rearrange, seed, out, negate, st, range);
      default: return king_l<8>(in, kb, batch, np, log_lc, U, gen, need ? &gt : nullptr, in_scale, rearrange, seed, out, negate, st, range);
    }
  }
  int king_dispatch(const Fr* in, const Fr* in_mask, int np, int log_m, int inverse, const Fr* U, const void* g,
                    int scale, int rearrange, uint64_t seed, Fr* out, const Fr* out_mask, bool negate,
                    hipStream_t st, const KingRange* range = nullptr) {
    KingBatch<Fr> kb{};
    kb.in_mask[0] = in_mask;
    kb.out_mask[0] = out_mask;
    kb.stride = 0;
    return king_dispatch_batch(in, kb, 1, np, log_m, inverse, U, g, scale, rearrange, seed, out, negate, st, range);
  }
  int fft2_king(const void* in, const void* in_mask, const uint32_t* parties, int np, int log_m, int inverse,
                const void* g, int scale_size_inv, int rearrange, uint64_t seed, void* out, const void* out_mask,
                hipStream_t st) override {
    if (!in || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const Fr* U = nullptr;
    int rc = umat_for(parties, np, &U);
    if (rc) return rc;
    return king_dispatch((const Fr*)in, (const Fr*)in_mask, np, log_m, inverse, U, g, scale_size_inv, rearrange, seed,
                         (Fr*)out, (const Fr*)out_mask, false, st);
  }

  // d_fft / d_ifft for all n parties on this device (dfft/mod.rs:99-175).  The king kernel exchanges chunks
  // between workgroups, so it never runs in place.
  int d_fft(void* shares, const void* in_mask, const void* out_mask, int rearrange, int log_m, int inverse,
            const void* g, uint64_t seed, void* out, hipStream_t st) override {
    if (!shares) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (out == shares) out = nullptr;
    int log_l = ilog2(l);
    if (log_m < log_l) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    size_t bytes = (size_t)n * (((size_t)1 << log_m) / l) * sizeof(Fr);
    // the local stages go out of place into the caller stream's working vector (the first pass reads `shares`), the king
    // reads that and writes the destination: no copy back, and `shares` is left untouched when `out` is given
    DevBuf& king_tmp_ = ws(st)->king_tmp;
    ZK_HIP(king_tmp_.ensure(bytes));
    NttSrc<Fr> src{};
    src.p[0] = (const Fr*)shares;
    src.per = (uint32_t)n;
    int rc = fft1_src(king_tmp_.p, log_m, inverse, (size_t)n, nullptr, st, src);
    if (rc) return rc;
    return fft2_king(king_tmp_.p, in_mask, nullptr, n, log_m, inverse, g, inverse ? 1 : 0, rearrange, seed,
                     out ? out : shares, out_mask, st);
  }

  // FftMask::sample (dfft/mod.rs:30-85).  Streams: values = seed, in-mask randomness = seed ^ 0x1111,
  // out-mask randomness = seed ^ 0x2222 (same convention as oracle/dist.py).
  int fft_mask_sample(int rearrange, const void* g, int inverse, int log_m, uint64_t seed, void* in_mask,
                      void* out_mask, hipStream_t st) override {
    int log_l = ilog2(l);
    if (log_m < log_l) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    size_t m = (size_t)1 << log_m, Lc = m / l;
    DevBuf& scratch_ = ws(st)->scratch;
    ZK_HIP(scratch_.ensure(m * sizeof(Fr)));
    Fr* vals = (Fr*)scratch_.p;   // layout [l][Lc]: value k*l+s at [s][k]
    rand_fill_kernel<Fr><<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st>>>(vals, rs(seed), m, (uint32_t)l,
                                                                                 (uint32_t)(log_m - log_l));
    ZK_HIP(hipGetLastError());
    int rc = pss_pack(vals, Lc, 1, seed ^ 0x1111, false, in_mask, st);
    if (rc) return rc;
    return king_dispatch(vals, nullptr, l, log_m, inverse, ident_, g, 0, rearrange, seed ^ 0x2222, (Fr*)out_mask,
                         nullptr, true, st);
  }

  // ---------------------------------------------------------------- deg_red (deg_red.rs:80-126)
  template <int L>
  int degred_l(const Fr* in, const DegredBatch<Fr>& db, int batch, int np, size_t len, const Fr* U, uint64_t seed,
               Fr* out, hipStream_t st, size_t stride = 0, size_t j0 = 0, const Fr* mul_b = nullptr,
               const Fr* sub_c = nullptr) {
    if (!stride) stride = len;
    if constexpr (L == 2) {
      if (np == n && n == KING_COOP_LANES && len <= KING_COOP_MAX) {
        const unsigned per = 64 / KING_COOP_LANES;
        dim3 grid((unsigned)((len + per - 1) / per), (unsigned)batch), block(64);
        ProfScope ps_(prof, PROF_DEGRED, st, (double)len * batch);
        const uint64_t span = batch > 1 ? (uint64_t)(batch - 1) * db.seed_step + 1 : 1;
        king_degred_coop_kernel<FrP><<<grid, block, 0, st>>>(in, db, len, U, pmat_, rs(seed, span), out, stride, j0, mul_b,
                                                             sub_c);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
      }
    }
    const size_t kbk = (size_t)king_block(len);
    dim3 grid((unsigned)((len + kbk - 1) / kbk), (unsigned)batch), block((unsigned)kbk);
    ProfScope ps_(prof, PROF_DEGRED, st, (double)len * batch);
    const uint64_t span = batch > 1 ? (uint64_t)(batch - 1) * db.seed_step + 1 : 1;
    king_degred_kernel<FrP, L><<<grid, block, 0, st>>>(in, db, np, len, U, pmat_, pack2_, rs(seed, span), out, stride, j0,
                                                       mul_b, sub_c);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  // `batch` independent vectors (the proofs of a batch): see DegredBatch
  int deg_red_batch(const Fr* in, const DegredBatch<Fr>& db, int batch, const uint32_t* parties, int np, size_t len,
                    uint64_t seed, Fr* out, hipStream_t st, size_t stride = 0, size_t j0 = 0, const Fr* mul_b = nullptr,
                    const Fr* sub_c = nullptr) {
    if (!len || batch < 1) return ZK_OK;
    if (batch > DEGRED_BATCH) return fail(ZK_ERR_BAD_INPUT, "bad deg_red batch");
    const Fr* U = nullptr;
    int rc = umat_for(parties, np, &U);
    if (rc) return rc;
    switch (l) {
      case 1: return degred_l<1>(in, db, batch, np, len, U, seed, out, st, stride, j0, mul_b, sub_c);
      case 2: return degred_l<2>(in, db, batch, np, len, U, seed, out, st, stride, j0, mul_b, sub_c);
      case 4: return degred_l<4>(in, db, batch, np, len, U, seed, out, st, stride, j0, mul_b, sub_c);
      default: return degred_l<8>(in, db, batch, np, len, U, seed, out, st, stride, j0, mul_b, sub_c);
    }
  }
  int deg_red_np(const Fr* in, const Fr* in_mask, const uint32_t* parties, int np, size_t len, uint64_t seed, Fr* out,
                 const Fr* out_mask, hipStream_t st, size_t stride = 0, size_t j0 = 0, const Fr* mul_b = nullptr,
                 const Fr* sub_c = nullptr) {
    DegredBatch<Fr> db{};
    db.in_mask[0] = in_mask;
    db.out_mask[0] = out_mask;
    return deg_red_batch(in, db, 1, parties, np, len, seed, out, st, stride, j0, mul_b, sub_c);
  }
  int deg_red(void* x, const void* in_mask, const void* out_mask, size_t len, uint64_t seed, hipStream_t st) override {
    if (len && !x) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    return deg_red_np((const Fr*)x, (const Fr*)in_mask, nullptr, n, len, seed, (Fr*)x, (const Fr*)out_mask, st);
  }
  int deg_red_parties(const void* x, const uint32_t* parties, int np, const void* in_mask, const void* out_mask,
                      size_t len, uint64_t seed, void* out, hipStream_t st) override {
    if (len && (!x || !out)) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (np != n && x == out) return fail(ZK_ERR_BAD_INPUT, "in-place deg_red needs all n parties");
    return deg_red_np((const Fr*)x, (const Fr*)in_mask, parties, np, len, seed, (Fr*)out, (const Fr*)out_mask, st);
  }
  // d_msm when only the listed parties' contributions reached the king (ser_net.rs:57-94): bases/scalars [np][len]
  int d_msm_parties(int group, const void* bases, const void* scalars, size_t len, const uint32_t* parties, int np,
                    const void* in_mask, const void* out_mask, void* out, hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    std::vector<Fr> coef;
    int rc = coefs_for(parties, np, coef);
    if (rc) return rc;
    if (group == ZK_G1) return msm_.template d_msm_coef_t<Fq>(this, bases, scalars, len, coef, in_mask, out_mask, out, st);
    if (group == ZK_G2 && Cfg::HAS_G2)
      return msm_.template d_msm_coef_t<Fq2>(this, bases, scalars, len, coef, in_mask, out_mask, out, st);
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }
  // DegRedMask::sample with gen = 1 (deg_red.rs:40-66)
  int degred_mask_sample(size_t len, uint64_t seed, void* in_mask, void* out_mask, hipStream_t st) override {
    if (!len) return ZK_OK;
    size_t cnt = len * l;
    DevBuf& scratch_ = ws(st)->scratch;
    ZK_HIP(scratch_.ensure(cnt * sizeof(Fr)));
    Fr* vals = (Fr*)scratch_.p;
    rand_fill_kernel<Fr><<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st>>>(vals, rs(seed), cnt, (uint32_t)l,
                                                                                   0xffffffffu);
    ZK_HIP(hipGetLastError());
    int rc = pss_pack(vals, len, 0, seed ^ 0x1111, false, in_mask, st);
    if (rc) return rc;
    vec_neg_kernel<Fr><<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st>>>(vals, vals, cnt);
    ZK_HIP(hipGetLastError());
    return pss_pack(vals, len, 0, seed ^ 0x2222, false, out_mask, st);
  }

  // ---------------------------------------------------------------- d_pp (dpp/mod.rs:15-87), csrc/dpp.hpp
  // The king's part (dpp/mod.rs:40-73) for the listed parties: prefix products of num / den as fresh shares, three launches
  // (tile scans, carries + the one inversion, finish).  With `fuse_degred` the finish kernel also carries the deg_red round
  // that follows (:86) -- only valid when every party's share lives on this device.  The zero-denominator flag is read
  // once, after the last launch.
  template <int L, int E>
  int dpp_le(const Fr* num, const Fr* den, int np, size_t len, const Fr* U, const Fr* Ufull, const Fr* in_mask,
             const Fr* out_mask, const RngSeed& seed, Fr* out, hipStream_t st) {
    constexpr size_t TILE = DppGeom<E>::TILE;
    const size_t m = len * L, ntiles = (m + TILE - 1) / TILE;
    StreamWs* w_ = ws(st);                              // scratch and error word of THIS stream (two d_pp on two stream ids never meet)
    if (!w_->err) return fail(ZK_ERR_GENERIC, "d_pp: out of device memory");
    DevBuf& scratch_ = w_->scratch;
    int* const err_flag_ = w_->err;
    ZK_HIP(scratch_.ensure((m + 3 * ntiles) * sizeof(Fr)));
    Fr* y = (Fr*)scratch_.p;
    Fr* tile_n = y + m;
    Fr* tile_d = tile_n + ntiles;
    Fr* ctile = tile_d + ntiles;
    constexpr size_t lds = (size_t)(sizeof(Fr) / 16) * 16 * DppGeom<E>::LDS_SLOTS;
    bool& attr_set = dpp_attr_set_[E == DPP_E_LONG ? 0 : 1];
    if (!attr_set) {
      ZK_HIP(hipFuncSetAttribute((const void*)dpp_tile_kernel<FrP, L, E>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      attr_set = true;
    }
    ZK_HIP(hipMemsetAsync(err_flag_, 0, 2 * sizeof(int), st));      // error flag + the carry kernel's ready flag
    {
      ProfScope ps_(prof, PROF_DPP_TILE, st, (double)m);
      dpp_tile_kernel<FrP, L, E><<<dim3((unsigned)ntiles), dim3(DPP_THREADS), lds, st>>>(num, den, np, len, len, U, y,
                                                                                          tile_n, tile_d);
    }
    ZK_HIP(hipGetLastError());
    {
      ProfScope ps_(prof, PROF_DPP_CARRY, st, (double)ntiles);
      // one workgroup; no more waves than there are tiles to own (every wave issues the scans' products)
      const unsigned cthreads = (unsigned)std::min<size_t>(DPP_CARRY_THREADS, std::max<size_t>(64, (ntiles + 63) / 64 * 64));
      dpp_carry_kernel<Fr><<<dim3(2), dim3(cthreads), 0, st>>>(tile_n, tile_d, ntiles, ctile, err_flag_);    // scans | inversion
    }
    ZK_HIP(hipGetLastError());
    {
      ProfScope ps_(prof, PROF_DPP_FINISH, st, (double)m);
      dpp_finish_kernel<FrP, L><<<dim3((unsigned)((len + KING_THREADS - 1) / KING_THREADS)), dim3(KING_THREADS), 0, st>>>(
          y, ctile, len, in_mask, out_mask, Ufull, pmat_, pack2_, seed, out, (uint32_t)TILE);
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  template <int L>
  int dpp_l(const Fr* num, const Fr* den, int np, size_t len, const Fr* U, const Fr* Ufull, const Fr* in_mask,
            const Fr* out_mask, const RngSeed& seed, Fr* out, hipStream_t st) {
    // 2048-element tiles at every size: 1024-element ones (E = 4) were measured on the same box and lose at 2^20
    // (tile 176 vs 169 us, carry 111 vs 93 us with twice the tiles) and tie at 2^24 (3.85 ms both)
    return dpp_le<L, DPP_E_LONG>(num, den, np, len, U, Ufull, in_mask, out_mask, seed, out, st);
  }
  int d_pp_king(const Fr* num, const Fr* den, const uint32_t* parties, int np, size_t len, uint64_t seed, Fr* out,
                hipStream_t st, bool fuse_degred = false, const Fr* in_mask = nullptr, const Fr* out_mask = nullptr) {
    const Fr *U = nullptr, *Ufull = nullptr;
    int rc = umat_for(parties, np, &U);
    if (rc) return rc;
    if (fuse_degred && (rc = umat_for(nullptr, n, &Ufull))) return rc;
    // the shares the caller gets are packed ONCE: by the king round when it stands alone (stream `seed`), by the deg_red
    // round when that one is fused in (stream seed ^ 0x3333, as the separate call draws: oracle/dist.py d_pp)
    const RngSeed r = rs(fuse_degred ? seed ^ 0x3333 : seed);
    if (!fuse_degred) in_mask = out_mask = nullptr;
    switch (l) {
      case 1: rc = dpp_l<1>(num, den, np, len, U, Ufull, in_mask, out_mask, r, out, st); break;
      case 2: rc = dpp_l<2>(num, den, np, len, U, Ufull, in_mask, out_mask, r, out, st); break;
      case 4: rc = dpp_l<4>(num, den, np, len, U, Ufull, in_mask, out_mask, r, out, st); break;
      case 8: rc = dpp_l<8>(num, den, np, len, U, Ufull, in_mask, out_mask, r, out, st); break;
      default: return fail(ZK_ERR_BAD_INPUT, "d_pp is built for packing factors 1, 2, 4 and 8");
    }
    if (rc) return rc;
    int herr = 0;
    ZK_HIP(hipMemcpyAsync(&herr, ws(st)->err, sizeof(int), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    if (herr) return fail(ZK_ERR_GENERIC, "d_pp: zero denominator (reference panics: dpp/mod.rs:55)");
    return ZK_OK;
  }
  int d_pp(const void* num, const void* den, const void* in_mask, const void* out_mask, size_t len, uint64_t seed,
           void* out, hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!num || !den || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    return d_pp_king((const Fr*)num, (const Fr*)den, nullptr, n, len, seed, (Fr*)out, st, true, (const Fr*)in_mask,
                     (const Fr*)out_mask);
  }

