// Part of class Engine<Cfg> (engine_impl.hpp includes this file INSIDE the class body): MSM entry points, the dealer's fixed-base multiplication, PSS / deg_red / unpack over group elements, the circom front end, wire formats, context options.
// Split out of engine_impl.hpp in round 5 (one 3 400-line class body had stopped being navigable); not a stand-alone header.

  // ---------------------------------------------------------------- MSM
  int msm(int group, const void* bases, size_t nb, const void* scalars, size_t ns, void* out,
          hipStream_t st) override {
    if (nb != ns) {   // dmsm/mod.rs:73: G::msm returns Err(min len) -> MpcNetError::Generic(len.to_string())
      return fail(ZK_ERR_GENERIC, std::to_string(nb < ns ? nb : ns));
    }
    return msm_.run(this, group, bases, scalars, nb, nullptr, 1, out, st);
  }
  int d_msm(int group, const void* bases, const void* scalars, size_t len, const void* in_mask, const void* out_mask,
            void* out, hipStream_t st) override {
    return msm_.d_msm(this, group, bases, scalars, len, in_mask, out_mask, out, st);
  }

  // ---------------------------------------------------------------- fixed-base multiplication (dealer)
  static constexpr size_t BASE_MUL_WIDE_FROM = (size_t)1 << 19;      // scalars per call from which 16-bit windows pay
  template <class Fld>
  int base_mul_t(const void* base_affine, const void* scalars, size_t len, void* out_affine, hipStream_t st) {
    if (!len) return ZK_OK;
    if (!base_affine || !scalars || !out_affine) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const int nwin = (FrP::BITS + 7) / 8;
    std::string key((const char*)base_affine, sizeof(Affine<Fld>));
    Affine<Fld>* table = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = base_tables_.find(key);
      if (it != base_tables_.end()) table = (Affine<Fld>*)it->second;
    }
    if (!table) {
      // table[w][d-1] = d * 256^w * Base, built on the host
      Affine<Fld> base;
      memcpy(&base, base_affine, sizeof(base));
      std::vector<Affine<Fld>> h((size_t)nwin * 255);
      XYZZ<Fld> wbase = XYZZ<Fld>::from_affine(base);
      for (int w = 0; w < nwin; w++) {
        XYZZ<Fld> cur = wbase;
        for (int d = 1; d <= 255; d++) {
          h[(size_t)w * 255 + d - 1] = xyzz_to_affine(cur);
          cur = xyzz_add_ni(cur, wbase);
        }
        wbase = cur;   // 256 * previous
      }
      ZK_HIP(hipMalloc((void**)&table, h.size() * sizeof(Affine<Fld>)));
      ZK_HIP(hipMemcpy(table, h.data(), h.size() * sizeof(Affine<Fld>), hipMemcpyHostToDevice));
      std::lock_guard<std::mutex> lk(mu_);
      base_tables_[key] = table;
    }
    if (len >= BASE_MUL_WIDE_FROM) {
      // 16-bit windows (half the mixed additions per scalar), widened on the device from the 8-bit table once per base
      const int nwin16 = (nwin + 1) / 2;
      Affine<Fld>* wide = nullptr;
      {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = base_tables_.find(key + "#16");
        if (it != base_tables_.end()) wide = (Affine<Fld>*)it->second;
      }
      if (!wide) {
        const size_t entries = (size_t)nwin16 * 65535u;
        ZK_HIP(hipMalloc((void**)&wide, entries * sizeof(Affine<Fld>)));
        fixed_base_widen_kernel<Fld><<<dim3((unsigned)((entries + 127) / 128)), dim3(128), 0, st>>>(table, nwin, nwin16, wide);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(st));          // once per base: a later call on another stream finds a finished table
        std::lock_guard<std::mutex> lk(mu_);
        base_tables_[key + "#16"] = wide;
      }
      // extension field: a quad of lanes per scalar (pack_split.hpp, launched from the curve's G2 translation unit)
      if constexpr (IsExtField<Fld>::value) return base_mul_split_launch<FrP, Fld>(this, scalars, len, wide, nwin16, 16, out_affine, st);
      else
        fixed_base_mul_kernel<FrP, Fld, 16><<<dim3((unsigned)((len + 127) / 128)), dim3(128), 0, st>>>(
            (const Fr*)scalars, len, wide, nwin16, (Affine<Fld>*)out_affine);
    } else {
      if constexpr (IsExtField<Fld>::value) return base_mul_split_launch<FrP, Fld>(this, scalars, len, table, nwin, 8, out_affine, st);
      else
        fixed_base_mul_kernel<FrP, Fld, 8><<<dim3((unsigned)((len + 127) / 128)), dim3(128), 0, st>>>(
            (const Fr*)scalars, len, table, nwin, (Affine<Fld>*)out_affine);
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int base_mul(int group, const void* base_affine, const void* scalars, size_t len, void* out_affine,
               hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (group == ZK_G1) return base_mul_t<Fq>(base_affine, scalars, len, out_affine, st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) return base_mul_t<Fq2>(base_affine, scalars, len, out_affine, st);     // (quad-split kernel: G2 units only)
      else return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    }
    return fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
  }

  // ---------------------------------------------------------------- PSS over group elements
  template <class Fld>
  int pack_points_t(const void* points, size_t nchunks, int nv, void* shares, hipStream_t st) {
    if (!nchunks) return ZK_OK;
    if (!points || !shares) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (nv != l && nv != l + t) return fail(ZK_ERR_BAD_INPUT, "points per chunk must be l (det_pack) or l+t (pack)");
    // canonical (non-Montgomery) copies of the first nv columns of P, [n][nv]
    Fr* coef = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = pcoef_.find(nv);
      if (it != pcoef_.end()) coef = it->second;
    }
    if (!coef) {
      std::vector<Fr> h((size_t)n * nv);
      for (int p = 0; p < n; p++)
        for (int i = 0; i < nv; i++) h[(size_t)p * nv + i] = pmat_host_[(size_t)p * (l + t) + i].from_mont();
      int rc = upload(h, &coef);
      if (rc) return rc;
      std::lock_guard<std::mutex> lk(mu_);
      pcoef_[nv] = coef;
    }
    size_t total = nchunks * (size_t)n;
    dim3 grid((unsigned)((total + 127) / 128)), block(128);
    const Affine<Fld>* in = (const Affine<Fld>*)points;
    Affine<Fld>* out = (Affine<Fld>*)shares;
    if (nv == 2) {
      // the party's two fixed scalars in joint sparse form (host, once per context): [n][jlen] bytes, MSB first, pair A in
      // the low nibble and pair B in the high one.  With the curve's endomorphism (glv_params.hpp) each scalar is split
      // k = k1 + lambda k2 into two half-length parts: pair A = the k1 parts over (P0, P1), pair B = the k2 parts over
      // (phi P0, phi P1), signs folded into the digits -- half the doubling chain.  Otherwise pair B is all zero digits.
      uint8_t* dig = nullptr;
      int jlen = 0;
      bool glv = false;
      {
        std::lock_guard<std::mutex> lk(mu_);
        dig = pjsf_;
        jlen = pjsf_len_;
        glv = pjsf_glv_;
      }
      if (!dig) {
        constexpr uint8_t ZZ = (uint8_t)(1 | (1 << 2));                           // the digit pair (0, 0)
        std::vector<uint8_t> h;
        std::vector<int8_t> u0, u1;
        auto put = [&](int p, int shift, const uint32_t* a, bool na, const uint32_t* b, bool nb) {
          jsf_digits<FrP::N>(a, b, u0, u1);
          if ((int)u0.size() > jlen) return false;
          for (size_t q = 0; q < u0.size(); q++) {
            const int d0 = na ? -u0[q] : u0[q], d1 = nb ? -u1[q] : u1[q];
            uint8_t& cell = h[(size_t)p * jlen + (jlen - 1 - q)];
            cell = (uint8_t)((cell & ~(15u << shift)) | (((d0 + 1) | ((d1 + 1) << 2)) << shift));
          }
          return true;
        };
        if (pack_glv_ && Glv<FrP>::OK) {
          if constexpr (Glv<FrP>::OK) {
            glv = true;
            jlen = Glv<FrP>::BITS + 2;
            h.assign((size_t)n * jlen, (uint8_t)(ZZ | (ZZ << 4)));
            for (int p = 0; p < n && glv; p++) {
              uint32_t a1[FrP::N], a2[FrP::N], b1[FrP::N], b2[FrP::N];
              bool na1, na2, nb1, nb2;
              glv = glv_split<FrP>(pmat_host_[(size_t)p * (l + t)], a1, &na1, a2, &na2) &&
                    glv_split<FrP>(pmat_host_[(size_t)p * (l + t) + 1], b1, &nb1, b2, &nb2) &&
                    put(p, 0, a1, na1, b1, nb1) && put(p, 4, a2, na2, b2, nb2);
            }
          }
        }
        if (!glv) {
          jlen = FrP::N * 32 + 1;
          h.assign((size_t)n * jlen, (uint8_t)(ZZ | (ZZ << 4)));
          for (int p = 0; p < n; p++) {
            const Fr a = pmat_host_[(size_t)p * (l + t)].from_mont(), b = pmat_host_[(size_t)p * (l + t) + 1].from_mont();
            if (!put(p, 0, a.v, false, b.v, false)) return fail(ZK_ERR_GENERIC, "joint sparse form longer than the scalar field");
          }
        }
        ZK_HIP(hipMalloc((void**)&dig, h.size()));
        ZK_HIP(hipMemcpy(dig, h.data(), h.size(), hipMemcpyHostToDevice));
        std::lock_guard<std::mutex> lk(mu_);
        if (pjsf_) {
          (void)hipFree(dig);
          dig = pjsf_, jlen = pjsf_len_, glv = pjsf_glv_;
        } else {
          pjsf_ = dig, pjsf_len_ = jlen, pjsf_glv_ = glv;
        }
      }
      using BF = Fp<typename BaseParams<Fld>::type>;
      BF beta = BF::one();
      if constexpr (Glv<FrP>::OK) {
        if (glv) beta = BF::from_limbs(IsExtField<Fld>::value ? Glv<FrP>::BETA_G2 : Glv<FrP>::BETA_G1);
      }
      if constexpr (IsExtField<Fld>::value) {
        // extension field: a quad of lanes per (chunk, party), one base-field value per lane (pack_split.hpp; launched from
        // the curve's G2 translation unit)
        return pack_points_split_launch<FrP, Fld>(this, in, nchunks, n, dig, jlen, &beta, out, st);
      } else {
        // (two waves per SIMD, 197 registers on 8 limbs; the same kernel compiled for three -- 168 registers, 4 spilled
        // dwords -- measured 3.38 against 3.36 ms per vector: no difference, not kept)
        pss_pack_points_jsf_kernel<FrP, Fld><<<dim3((unsigned)((nchunks + 127) / 128), (unsigned)n), block, 0, st>>>(
            in, nchunks, n, dig, jlen, beta, out);
      }
    } else if (nv == 4) pss_pack_points_kernel<FrP, Fld, 4><<<grid, block, 0, st>>>(in, nchunks, n, coef, out);
    else return fail(ZK_ERR_BAD_INPUT, "point packing is built for 2 or 4 points per chunk (l = 2, or det_pack at l = 4)");
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int pss_pack_points(int group, const void* points, size_t nchunks, int nv, void* shares, hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (group == ZK_G1) return pack_points_t<Fq>(points, nchunks, nv, shares, st);
    if constexpr (Cfg::HAS_G2) {             // (the quad-split kernel exists in the G2 translation units only)
      if (group == ZK_G2) return pack_points_t<Fq2>(points, nchunks, nv, shares, st);
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

  // ---------------------------------------------------------------- circom front end (qap.rs:42-89)
  int r1cs_qap(const void* pa, const void* ca, const void* va, const void* pb, const void* cb, const void* vb,
               const void* w, size_t nvars, size_t nc, size_t ni, int log_m, void* a, void* b, void* c,
               hipStream_t st) override {
    if (!pa || !pb || !w || !a || !b || !c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (log_m < 0 || log_m > 30) return fail(ZK_ERR_BAD_INPUT, "bad domain size");
    const size_t m = (size_t)1 << log_m;
    if (nc + ni > m) return fail(ZK_ERR_BAD_INPUT, "domain smaller than num_constraints + num_inputs");   // qap.rs:52-56
    if (ni > nvars || nvars >= ((size_t)1 << 32) || nc >= ((size_t)1 << 32))
      return fail(ZK_ERR_BAD_INPUT, "bad R1CS dimensions");
    ZK_HIP(flag_.ensure(4));
    ZK_HIP(hipMemsetAsync(flag_.p, 0, 4, st));
    r1cs_qap_kernel<Fr><<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st>>>(
        (const uint32_t*)pa, (const uint32_t*)ca, (const Fr*)va, (const uint32_t*)pb, (const uint32_t*)cb,
        (const Fr*)vb, (const Fr*)w, (uint32_t)nvars, (uint32_t)nc, (uint32_t)ni, m, (Fr*)a, (Fr*)b, (Fr*)c,
        (uint32_t*)flag_.p);
    ZK_HIP(hipGetLastError());
    uint32_t bad = 0;
    ZK_HIP(hipMemcpyAsync(&bad, flag_.p, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    if (bad) return fail(ZK_ERR_GENERIC, "R1CS wire index out of range");
    return ZK_OK;
  }
  int fr_bytes(const void* in, size_t len, void* out, int from_bytes, hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!in || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    dim3 g((unsigned)((len + 255) / 256)), b(256);
    if (!from_bytes) {
      fr_to_bytes_kernel<Fr><<<g, b, 0, st>>>((const Fr*)in, len, (Fr*)out);
      ZK_HIP(hipGetLastError());
      return ZK_OK;
    }
    ZK_HIP(flag_.ensure(4));
    ZK_HIP(hipMemsetAsync(flag_.p, 0, 4, st));
    fr_from_bytes_kernel<Fr><<<g, b, 0, st>>>((const Fr*)in, len, (Fr*)out, (uint32_t*)flag_.p);
    ZK_HIP(hipGetLastError());
    uint32_t bad = 0;
    ZK_HIP(hipMemcpyAsync(&bad, flag_.p, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    if (bad) return fail(ZK_ERR_GENERIC, "field element not below the modulus (InvalidData)");
    return ZK_OK;
  }

  int msm_precompute(int group, const void* bases, size_t len, hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (group == ZK_G1) return msm_.template precompute_t<Fq>(this, bases, len, st);
    if (group == ZK_G2 && Cfg::HAS_G2) return msm_.template precompute_t<Fq2>(this, bases, len, st);
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }
  int msm_forget(const void* bases) override {
    return msm_.forget_table(bases) ? ZK_OK : fail(ZK_ERR_BAD_INPUT, "no table registered for this base vector");
  }
  int msm_table_info(int group, const void* bases, int* info) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (!info || !bases || (group != ZK_G1 && group != ZK_G2)) return fail(ZK_ERR_BAD_INPUT, "bad argument");
    msm_.table_info(bases, group == ZK_G2 ? sizeof(Affine<Fq2>) : sizeof(Affine<Fq>), info);
    return ZK_OK;
  }
  int set_option(const char* name, long long value) override {
    if (!name) return fail(ZK_ERR_BAD_INPUT, "null option name");
    if (!strcmp(name, "msm_bigsort_min")) {
      if (value < 0) return fail(ZK_ERR_BAD_INPUT, "msm_bigsort_min must be >= 0");
      msm_.bigsort_min = (size_t)value;
      return ZK_OK;
    }
    if (!strcmp(name, "msm_sort_lo_tab")) {        // A/B only: low bucket bits per bin of a small sort over a fixed-base table
      if (value != 0 && (value < 4 || value > 10)) return fail(ZK_ERR_BAD_INPUT, "msm_sort_lo_tab must be 0 or in 4..10");
      msm_.sort_lo_tab = (int)value;
      return ZK_OK;
    }
    if (!strcmp(name, "msm_skip_kernel")) {        // A/B only: identity bases through a mask kernel of its own (round 5's form)
      msm_.skip_kernel = value != 0;
      return ZK_OK;
    }
    if (!strcmp(name, "msm_acc_lds")) {            // dynamic LDS per accumulate workgroup (MsmTuning::acc_lds); 0 = none
      if (value < 0 || value > 65536) return fail(ZK_ERR_BAD_INPUT, "msm_acc_lds must be in 0..65536");
      msm_.acc_lds = (unsigned)value;
      return ZK_OK;
    }
    if (!strcmp(name, "pack_glv")) {                // det_pack over points: split the scalars by the curve's endomorphism
      std::lock_guard<std::mutex> lk(mu_);
      if (pjsf_) return fail(ZK_ERR_BAD_INPUT, "pack_glv must be set before the first zk_pss_pack_points");
      pack_glv_ = value != 0;
      return ZK_OK;
    }
    if (!strcmp(name, "rng_replay")) {
      rng_replay_ = value != 0;
      if (rng_replay_) warn_replay();
      return ZK_OK;
    }
    if (!strcmp(name, "dist_deadline")) {         // zk_dist_* return only with their data-plane work done, or fail
      dist_deadline_ = value != 0;
      return ZK_OK;
    }
    if (!strcmp(name, "king_alltoall")) {         // every rank of a net must choose alike
      king_a2a_ = value != 0;
      return ZK_OK;
    }
    if (!strcmp(name, "msm_table_c")) {
      if (value != 0 && (value < 8 || value > 22)) return fail(ZK_ERR_BAD_INPUT, "msm_table_c must be 0 (by length) or in 8..22");
      msm_.table_c = msm_.table_c_g2 = (int)value;
      return ZK_OK;
    }
    if (!strcmp(name, "msm_table_c_g2")) {
      if (value != 0 && (value < 8 || value > 22)) return fail(ZK_ERR_BAD_INPUT, "msm_table_c_g2 must be 0 (by length) or in 8..22");
      msm_.table_c_g2 = (int)value;
      return ZK_OK;
    }
    if (!strcmp(name, "wait_deadline_ms")) {      // bound of every host-side wait of the prover (engine.hpp); 0 = unbounded
      if (value < 0 || value > 86400000) return fail(ZK_ERR_BAD_INPUT, "wait_deadline_ms must be in 0..86400000");
      wait_deadline_ms.store(value, std::memory_order_relaxed);
      return ZK_OK;
    }
    if (!strcmp(name, "host_threads")) {          // workers of the host pool (MSM launch / fold tasks, scalar multiples)
      if (value < 0 || value > 256) return fail(ZK_ERR_BAD_INPUT, "host_threads must be in 0..256");
      if (value && value < 4) return fail(ZK_ERR_BAD_INPUT, "host_threads must be 0 (automatic) or at least 4");
      std::lock_guard<std::mutex> lk(mu_);
      if (streams_ready_) return fail(ZK_ERR_BAD_INPUT, "host_threads must be set before the first proof");
      host_threads_ = (int)value;
      return ZK_OK;
    }
    if (!strcmp(name, "msm_c") || !strcmp(name, "msm_c_g2")) {      // window bits of table-free MSMs (0 = cost model)
      if (value != 0 && (value < 2 || value > 20)) return fail(ZK_ERR_BAD_INPUT, "msm_c must be 0 or in 2..20");
      if (name[5] == 0) msm_.c_g1 = msm_.c_g2 = (int)value;
      else msm_.c_g2 = (int)value;
      return ZK_OK;
    }
    if (!strcmp(name, "h_first_log_m")) {         // domains of 2^value and up: circom_h + the U sort ahead of the accumulates
      if (value < 1 || value > 64) return fail(ZK_ERR_BAD_INPUT, "h_first_log_m must be in 1..64");
      h_first_log_m_ = (int)value;
      return ZK_OK;
    }
    return fail(ZK_ERR_BAD_INPUT, "unknown option");
  }
  int msm_plan(int group, size_t len, int* plan) override {
    if (!plan || (group != ZK_G1 && group != ZK_G2)) return fail(ZK_ERR_BAD_INPUT, "bad argument");
    if (group == ZK_G2 && !Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    msm_.plan(len, group == ZK_G2, plan);
    return ZK_OK;
  }
  int group_add(int group, const void* a, const void* b, void* out) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (!a || !b || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (group == ZK_G1) {
      Jacobian<Fq> x, y;
      memcpy(&x, a, sizeof(x));
      memcpy(&y, b, sizeof(y));
      Jacobian<Fq> o = xyzz_to_jacobian(xyzz_add_ni(jacobian_to_xyzz(x), jacobian_to_xyzz(y)));
      memcpy(out, &o, sizeof(o));
      return ZK_OK;
    }
    if (group == ZK_G2 && Cfg::HAS_G2) {
      Jacobian<Fq2> x, y;
      memcpy(&x, a, sizeof(x));
      memcpy(&y, b, sizeof(y));
      Jacobian<Fq2> o = xyzz_to_jacobian(xyzz_add_ni(jacobian_to_xyzz(x), jacobian_to_xyzz(y)));
      memcpy(out, &o, sizeof(o));
      return ZK_OK;
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }
  // A, B, C shares from the five d_msm king outputs (prove.rs:40-56, 99-110, 148-158, 229-235) -- used when the MSMs
  // were computed per rank and summed by the caller; sums[0..4] = S, H, V (G2), W, U.
  int groth16_assemble(const zk_crs_share* crs, const void* r_, const void* s_, const void* const* sums,
                       const zk_groth16_masks* mk, void* pi_a, void* pi_b, void* pi_c) override {
    if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    if (!crs || !r_ || !s_ || !sums || !pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    Fr r = Fr::from_limbs((const uint32_t*)r_), s = Fr::from_limbs((const uint32_t*)s_);
    auto j1 = [](const void* p) {
      Jacobian<Fq_> j;
      memcpy(&j, p, sizeof(j));
      return jacobian_to_xyzz(j);
    };
    Jacobian<Fq2_> jv;
    memcpy(&jv, sums[2], sizeof(jv));
    return assemble_points(crs, r, s, j1(sums[0]), j1(sums[1]), jacobian_to_xyzz(jv), j1(sums[3]), j1(sums[4]), mk, n, pi_a,
                           pi_b, pi_c);
  }
  // the same for `np` parties whose out-masks are rows 0..np-1 of mk->msm_out[*]
  int assemble_points(const zk_crs_share* crs, const Fr& r, const Fr& s, const P1& S, const P1& H, const P2& V, const P1& W,
                      const P1& U, const zk_groth16_masks* mk, int np, void* pi_a, void* pi_b, void* pi_c) {
    const bool r_zero = r.is_zero();
    P1 d1 = aff1(crs->delta_g1);
    P1 rN = host_scalar_mul<FrP, Fq_>(d1, r), sK = host_scalar_mul<FrP, Fq_>(d1, s),
       rsM = host_scalar_mul<FrP, Fq_>(d1, r * s);
    P2 sK2 = host_scalar_mul<FrP, Fq2_>(aff2(crs->delta_g2), s);
    P1 A0 = xyzz_add_ni(xyzz_add_ni(xyzz_add_ni(aff1(crs->a_query0), rN), aff1(crs->alpha_g1)), S);
    P1 B10 = r_zero ? P1::identity()
                    : xyzz_add_ni(xyzz_add_ni(xyzz_add_ni(aff1(crs->b_g1_query0), sK), aff1(crs->beta_g1)), H);
    P2 B20 = xyzz_add_ni(xyzz_add_ni(xyzz_add_ni(aff2(crs->b_g2_query0), sK2), aff2(crs->beta_g2)), V);
    P1 WU = xyzz_add_ni(xyzz_add_ni(W, U), rsM.neg());
    Jacobian<Fq_>* oa = (Jacobian<Fq_>*)pi_a;
    Jacobian<Fq2_>* ob = (Jacobian<Fq2_>*)pi_b;
    Jacobian<Fq_>* oc = (Jacobian<Fq_>*)pi_c;
    auto om1 = [&](int k, int p, const P1& v) {
      if (!mk || !mk->msm_out[k]) return v;
      return xyzz_add_ni(v, jacobian_to_xyzz(((const Jacobian<Fq_>*)mk->msm_out[k])[p]));
    };
    const bool uniform = !mk || (!mk->msm_out[0] && !mk->msm_out[1] && !mk->msm_out[2] && !mk->msm_out[3] &&
                                 !mk->msm_out[4]);
    for (int p = 0; p < np; p++) {
      if (uniform && p > 0) {
        oa[p] = oa[0];
        ob[p] = ob[0];
        oc[p] = oc[0];
        continue;
      }
      P1 A = om1(0, p, A0);
      P1 B1 = r_zero ? P1::identity() : om1(1, p, B10);
      P2 B2 = B20;
      if (mk && mk->msm_out[2]) B2 = xyzz_add_ni(B2, jacobian_to_xyzz(((const Jacobian<Fq2_>*)mk->msm_out[2])[p]));
      P1 C = xyzz_add_ni(host_scalar_mul<FrP, Fq_>(A, s), host_scalar_mul<FrP, Fq_>(B1, r));
      C = xyzz_add_ni(C, om1(4, p, om1(3, p, WU)));
      oa[p] = xyzz_to_jacobian(A);
      ob[p] = xyzz_to_jacobian(B2);
      oc[p] = xyzz_to_jacobian(C);
    }
    return ZK_OK;
  }

  // ---- the five partial d_msm of one rank, concurrently (multi-GPU flow) ---------------------------------
  // begin: S, H, V, W over this rank's parties [first, first + count) start on internal streams (crs vectors and
  // shares are [count][len]; masks, if any: msm_in[k] holds this rank's `count` in-mask points); finish: U (needs h)
  // runs on `stream`, everything joins; out[0..4] = S, H, V(G2), W, U including the in-mask terms.
  int msms_begin(const zk_crs_share* crs, const void* a_share, const void* ax_share, int first, int count, int skip_h,
                 const zk_groth16_masks* mk, hipStream_t st) override {
    if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    if (!crs || !a_share || !ax_share) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (first < 0 || count <= 0 || first + count > n) return fail(ZK_ERR_BAD_INPUT, "bad party range");
    ProveJob& j = jobs_[0];
    if (j.active) return fail(ZK_ERR_BAD_INPUT, "zk_groth16_msms_begin called twice");
    j.slot = 0;
    Fr r = skip_h ? Fr::zero() : Fr::one();          // only r == 0 matters here (H skipped, prove.rs:96-98)
    int rc = prove_begin(j, crs, nullptr, nullptr, nullptr, a_share, ax_share, r, Fr::one(), 0, mk, 0, false, first,
                         count, st);
    if (rc) {
      Status keep = last;
      abort_job(j);
      last = keep;
    }
    return rc;
  }
  int msms_finish(const void* h_share, void* const* out, hipStream_t st) override {
    ProveJob& j = jobs_[0];
    if (!j.active || j.full) return fail(ZK_ERR_BAD_INPUT, "zk_groth16_msms_finish without begin");
    if (!h_share || !out) {
      abort_job(j);
      return fail(ZK_ERR_BAD_INPUT, "null pointer");
    }
    int rc = prove_launch_u(j, h_share, st);
    if (rc) {
      Status keep = last;
      abort_job(j);
      last = keep;
      return rc;
    }
    P1 S, H, W, U;
    P2 V;
    rc = prove_join(j, &S, &H, &V, &W, &U);
    if (rc) return rc;
    Jacobian<Fq_> jj;
    jj = xyzz_to_jacobian(S);
    memcpy(out[0], &jj, sizeof(jj));
    jj = xyzz_to_jacobian(H);
    memcpy(out[1], &jj, sizeof(jj));
    Jacobian<Fq2_> j2 = xyzz_to_jacobian(V);
    memcpy(out[2], &j2, sizeof(j2));
    jj = xyzz_to_jacobian(W);
    memcpy(out[3], &jj, sizeof(jj));
    jj = xyzz_to_jacobian(U);
    memcpy(out[4], &jj, sizeof(jj));
    return ZK_OK;
  }

  // ---------------------------------------------------------------- deg_red over group elements (deg_red.rs:80-126, T = G)
  // x, masks, out: [n][len] affine.  King: unpack2 over the n (masked) points of a chunk, then pack with t fresh
  // random group elements (random multiples of `gen`, the `T::rand` of a group; stream `seed`, element j*t + i) --
  // both are small fixed linear maps, evaluated by points_lincomb_kernel.
  DevBuf ptw_[3];
  Fr* u2c_ = nullptr;      // canonical copies of U2 [l][n] and P [n][l+t]
  Fr* pmc_ = nullptr;
  int ensure_canon_mats() {
    std::lock_guard<std::mutex> lk(mu_);
    if (u2c_) return ZK_OK;
    std::vector<Fr> x, y, z;
    points(x, y, z);
    Fr ninv = Fr::from_u64((uint64_t)n).inverse();
    std::vector<Fr> U2((size_t)l * n), Pc((size_t)n * (l + t));
    for (int kk = 0; kk < l; kk++)
      for (int p = 0; p < n; p++) {
        Fr r2 = z[2 * kk] * x[p].inverse(), a2 = Fr::zero(), c2 = Fr::one();
        for (int d = 0; d < n; d++) {
          a2 = a2 + c2;
          c2 = c2 * r2;
        }
        U2[(size_t)kk * n + p] = (a2 * ninv).from_mont();
      }
    for (size_t i = 0; i < Pc.size(); i++) Pc[i] = pmat_host_[i].from_mont();
    int rc = upload(U2, &u2c_);
    if (rc) return rc;
    return upload(Pc, &pmc_);
  }
  template <class Fld>
  int deg_red_points_t(const void* x, const void* in_mask, const void* out_mask, size_t len, const void* gen_affine,
                       uint64_t seed, void* out, hipStream_t st, const uint32_t* parties = nullptr, int np = 0) {
    if (!len) return ZK_OK;
    if (!x || !out || !gen_affine) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int rc = ensure_canon_mats();
    if (rc) return rc;
    // only some parties' points reached the king (a rank was left out of the round): x / in_mask are [np][len] and the
    // unpack2 is the Lagrange form over the present parties (pss.rs:170-221), as for field elements
    const Fr* u2 = u2c_;
    if (np && np != n) {
      Fr* ud = nullptr;
      rc = ucanon_for(parties, np, 2, &ud);
      if (rc) return rc;
      u2 = ud;
    } else {
      np = n;
    }
    using A = Affine<Fld>;
    ZK_HIP(ptw_[0].ensure(len * t * sizeof(Fr)));
    ZK_HIP(ptw_[1].ensure(len * t * sizeof(A)));
    ZK_HIP(ptw_[2].ensure(len * l * sizeof(A)));
    Fr* rs_ = (Fr*)ptw_[0].p;
    A* rnd = (A*)ptw_[1].p;
    A* sec = (A*)ptw_[2].p;
    const size_t cnt = len * t;
    rand_fill_kernel<Fr><<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st>>>(rs_, rs(seed), cnt, (uint32_t)t,
                                                                                   0xffffffffu);
    ZK_HIP(hipGetLastError());
    rc = base_mul_t<Fld>(gen_affine, rs_, cnt, rnd, st);
    if (rc) return rc;
    // unpack2 of (x + in_mask): rows = l secrets of the chunk, inputs = the n parties' points
    PtGroup<Fld> gx{(const A*)x, 1, len, np, 0}, gm{(const A*)in_mask, 1, len, np, 0};
    size_t total = len * (size_t)l;
    points_lincomb_kernel<FrP, Fld><<<dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st>>>(
        gx, gm, in_mask ? 2 : 1, u2, np, l, len, nullptr, 0, sec, 1, (size_t)l);
    ZK_HIP(hipGetLastError());
    // pack: rows = n parties, inputs = l secrets + t random points, + the party's out-mask
    PtGroup<Fld> gs{sec, (size_t)l, 1, l, 0}, gr{rnd, (size_t)t, 1, t, l};
    total = len * (size_t)n;
    points_lincomb_kernel<FrP, Fld><<<dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st>>>(
        gs, gr, 2, pmc_, l + t, n, len, (const A*)out_mask, len, (A*)out, len, 1);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int deg_red_points(int group, const void* x, const void* in_mask, const void* out_mask, size_t len,
                     const void* gen_affine, uint64_t seed, void* out, hipStream_t st) override {
    if (x == out) return fail(ZK_ERR_BAD_INPUT, "deg_red over points cannot run in place");
    if (group == ZK_G1) return deg_red_points_t<Fq_>(x, in_mask, out_mask, len, gen_affine, seed, out, st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) return deg_red_points_t<Fq2_>(x, in_mask, out_mask, len, gen_affine, seed, out, st);
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }
  // DegRedMask::sample with a group generator (deg_red.rs:40-66): mask values r_i * gen, in_mask = pack(mask),
  // out_mask = pack(-mask); by linearity: pack the scalars, then multiply the generator
  int degred_mask_sample_points(int group, const void* gen_affine, size_t len, uint64_t seed, void* in_mask,
                                void* out_mask, hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!gen_affine || !in_mask || !out_mask) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const size_t cnt = (size_t)n * len;
    ZK_HIP(ptw_[0].ensure(2 * cnt * sizeof(Fr)));
    Fr* si = (Fr*)ptw_[0].p;
    Fr* so = si + cnt;
    int rc = degred_mask_sample(len, seed, si, so, st);
    if (rc) return rc;
    rc = base_mul(group, gen_affine, si, cnt, in_mask, st);
    if (rc) return rc;
    return base_mul(group, gen_affine, so, cnt, out_mask, st);
  }

  // ---------------------------------------------------------------- unpack / unpack2 over group elements
  // secret-sharing/src/pss.rs:125-166 with T = curve point (`T: DomainCoeff<F>`), used by the reference to turn the n
  // parties' proof shares into the proof (groth16/examples/sha256.rs:375-377: pp.unpack2(shares)[0]).  The maps are the
  // same l x np matrices as over Fr (unpack: U1; unpack2: U2, or the Lagrange form for a party subset,
  // pss.rs:170-221), applied by points_lincomb_kernel with canonical scalars.
  std::map<uint64_t, Fr*> ucanon_;
  // host copy (Montgomery) of the l x np matrix: kind 1 = unpack (all n parties), 2 = unpack2 / lagrange_unpack
  int umat_host(const uint32_t* parties, int np, int kind, std::vector<Fr>& U, uint32_t* mask_out) {
    if (np <= 0 || np > n) return fail(ZK_ERR_BAD_INPUT, "bad party count");
    uint32_t mask = 0;
    std::vector<uint32_t> ids((size_t)np);
    for (int i = 0; i < np; i++) {
      ids[i] = parties ? parties[i] : (uint32_t)i;
      if (ids[i] >= (uint32_t)n || (i > 0 && ids[i] <= ids[i - 1]))
        return fail(ZK_ERR_BAD_INPUT, "party ids must be ascending and < n");
      mask |= 1u << ids[i];
    }
    if (kind == 1 && np != n) return fail(ZK_ERR_BAD_INPUT, "unpack needs all n shares");
    if (np < n && np <= 2 * (t + l - 1)) return fail(ZK_ERR_PROTOCOL, "Not enough shares to reconstruct", 0);   // pss.rs:183-186
    *mask_out = mask;
    std::vector<Fr> x, y, z;
    points(x, y, z);
    U.assign((size_t)l * np, Fr::zero());
    if (np == n) {
      Fr ninv = Fr::from_u64((uint64_t)n).inverse();
      const int k = l + t;
      for (int kk = 0; kk < l; kk++)
        for (int p = 0; p < n; p++) {
          Fr ratio = (kind == 1 ? y[kk] : z[2 * kk]) * x[p].inverse(), acc = Fr::zero(), c = Fr::one();
          for (int d = 0; d < (kind == 1 ? k : n); d++) {
            acc = acc + c;
            c = c * ratio;
          }
          U[(size_t)kk * n + p] = acc * ninv;
        }
      return ZK_OK;
    }
    for (int kk = 0; kk < l; kk++)
      for (int i = 0; i < np; i++) {
        Fr num = Fr::one(), den = Fr::one();
        for (int j = 0; j < np; j++)
          if (j != i) {
            num = num * (z[2 * kk] - x[ids[j]]);
            den = den * (x[ids[i]] - x[ids[j]]);
          }
        U[(size_t)kk * np + i] = num * den.inverse();
      }
    return ZK_OK;
  }
  // device copy with CANONICAL entries (what points_lincomb_kernel takes), cached per (party set, kind)
  int ucanon_for(const uint32_t* parties, int np, int kind, Fr** out) {
    std::vector<Fr> U;
    uint32_t mask = 0;
    int rc = umat_host(parties, np, kind, U, &mask);
    if (rc) return rc;
    Fr* Ud = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = ucanon_.find(key_u(mask, kind));
      if (it != ucanon_.end()) Ud = it->second;
    }
    if (!Ud) {
      for (auto& v : U) v = v.from_mont();
      rc = upload(U, &Ud);
      if (rc) return rc;
      std::lock_guard<std::mutex> lk(mu_);
      ucanon_[key_u(mask, kind)] = Ud;
    }
    *out = Ud;
    return ZK_OK;
  }
  template <class Fld>
  int unpack_points_t(const void* shares, const uint32_t* parties, int np, size_t nchunks, int kind, void* out,
                      hipStream_t st) {
    if (!nchunks) return ZK_OK;
    if (!shares || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    Fr* Ud = nullptr;
    int rc = ucanon_for(parties, np, kind, &Ud);
    if (rc) return rc;
    using A = Affine<Fld>;
    PtGroup<Fld> g0{(const A*)shares, 1, nchunks, np, 0};
    const size_t total = nchunks * (size_t)l;
    points_lincomb_kernel<FrP, Fld><<<dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st>>>(
        g0, g0, 1, Ud, np, l, nchunks, nullptr, 0, (A*)out, 1, (size_t)l);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  // shares: [np][nchunks] affine (device); out: [nchunks][l] affine (device).  two = 0: unpack (np must be n).
  int pss_unpack_points(int group, const void* shares, const uint32_t* parties, int np, size_t nchunks, int two, void* out,
                        hipStream_t st) override {
    if (group == ZK_G1) return unpack_points_t<Fq_>(shares, parties, np, nchunks, two ? 2 : 1, out, st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) return unpack_points_t<Fq2_>(shares, parties, np, nchunks, two ? 2 : 1, out, st);
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }
  // sha256.rs:375-377: (a, b, c) = pp.unpack2(shares)[0] for the three proof elements, from the parties' Jacobian
  // outputs (host).  proof_affine (optional): A (G1) | B (G2) | C (G1) affine Montgomery; proof_bytes (optional):
  // ark_groth16::Proof::serialize_compressed (a | b | c; 4 |Fq| bytes).  Three points: evaluated on the host
  // (Straus over the np shares), compressed by the device codec.
  DevBuf recon_;
  int groth16_reconstruct(const void* pi_a, const void* pi_b, const void* pi_c, const uint32_t* parties, int np,
                          void* proof_affine, void* proof_bytes, hipStream_t st) override {
    if constexpr (!Cfg::HAS_G2) {
      return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    } else {
      if (!pi_a || !pi_b || !pi_c || (!proof_affine && !proof_bytes)) return fail(ZK_ERR_BAD_INPUT, "null pointer");
      std::vector<Fr> U;
      uint32_t mask = 0;
      int rc = umat_host(parties, np, 2, U, &mask);
      if (rc) return rc;
      auto one = [&](auto tag, const void* jac) {
        using Fld = decltype(tag);
        const Jacobian<Fld>* j = (const Jacobian<Fld>*)jac;
        std::vector<XYZZ<Fld>> pts((size_t)np);
        for (int p = 0; p < np; p++) pts[p] = jacobian_to_xyzz(j[p]);
        return xyzz_to_affine(host_straus<FrP, Fld>(pts.data(), U.data(), np));      // row 0 of the matrix: secret 0
      };
      struct Out {
        Affine<Fq_> a;
        Affine<Fq2_> b;
        Affine<Fq_> c;
      } o{one(Fq_{}, pi_a), one(Fq2_{}, pi_b), one(Fq_{}, pi_c)};
      static_assert(sizeof(Out) == 4 * sizeof(Affine<Fq_>), "proof layout");
      if (proof_affine) memcpy(proof_affine, &o, sizeof(o));
      if (!proof_bytes) return ZK_OK;
      constexpr size_t NB = (Cfg::FqP::BITS + 7) / 8;
      ZK_HIP(recon_.ensure(sizeof(o) + 4 * NB));
      char* d = (char*)recon_.p;
      Affine<Fq_> g1s[2] = {o.a, o.c};
      ZK_HIP(hipMemcpyAsync(d, g1s, sizeof(g1s), hipMemcpyHostToDevice, st));
      ZK_HIP(hipMemcpyAsync(d + sizeof(g1s), &o.b, sizeof(o.b), hipMemcpyHostToDevice, st));
      char* bytes_d = d + sizeof(o);
      rc = points_codec_t<Fq_>(d, 2, bytes_d, 0, Fq_::zero(), st);                    // a, c -> bytes [0, 2 NB)
      if (rc) return rc;
      rc = points_codec_t<Fq2_>(d + sizeof(g1s), 1, bytes_d + 2 * NB, 0, Fq2_{}, st);  // b -> bytes [2 NB, 4 NB)
      if (rc) return rc;
      std::vector<uint8_t> hb(4 * NB);
      ZK_HIP(hipMemcpyAsync(hb.data(), bytes_d, 4 * NB, hipMemcpyDeviceToHost, st));
      ZK_HIP(hipStreamSynchronize(st));
      uint8_t* ob = (uint8_t*)proof_bytes;
      memcpy(ob, hb.data(), NB);                       // a
      memcpy(ob + NB, hb.data() + 2 * NB, 2 * NB);     // b
      memcpy(ob + 3 * NB, hb.data() + NB, NB);         // c
      return ZK_OK;
    }
  }

  // Tonelli-Shanks parameters of the base field (q - 1 = 2^s t): z = c^t for the least quadratic non-residue c, e = (t-1)/2
  TsParams<typename Cfg::FqP> ts_params() {
    using FqP = typename Cfg::FqP;
    static const TsParams<FqP> cached = [] {
      TsParams<FqP> tp{};
      constexpr int N = FqP::N;
      uint32_t t[N];
      for (int i = 0; i < N; i++) t[i] = FqP::MOD[i];
      t[0] -= 1;                                        // q - 1 (q odd)
      int sh = 0;
      while (!(t[0] & 1u)) {                            // t = (q - 1) >> s
        for (int i = 0; i < N - 1; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
        t[N - 1] >>= 1;
        sh++;
      }
      tp.s = sh;
      uint32_t half[N];                                 // (q - 1) / 2: Euler's criterion
      for (int i = 0; i < N; i++) half[i] = (FqP::MOD[i] >> 1) | (i + 1 < N ? FqP::MOD[i + 1] << 31 : 0u);
      const Fq_ minus_one = Fq_::one().neg();
      for (uint64_t c = 2;; c++) {
        const Fq_ cv = Fq_::from_u64(c);
        if (cv.pow(half, N) == minus_one) {
          tp.z = cv.pow(t, N);
          break;
        }
      }
      for (int i = 0; i < N; i++) tp.e[i] = (t[i] >> 1) | (i + 1 < N ? t[i + 1] << 31 : 0u);      // (t - 1) / 2, t odd
      return tp;
    }();
    return cached;
  }

  // ---------------------------------------------------------------- compressed point vectors (ser_net.rs:111-120)
  template <class Fld>
  int points_codec_t(const void* in, size_t len, void* out, int decompress, const Fld& b, hipStream_t st) {
    dim3 g((unsigned)((len + 127) / 128)), blk(128);
    if (!decompress) {
      points_compress_kernel<Fld><<<g, blk, 0, st>>>((const Affine<Fld>*)in, len, Cfg::ZCASH ? 1 : 0, (uint8_t*)out);
      ZK_HIP(hipGetLastError());
      return ZK_OK;
    }
    ZK_HIP(flag_.ensure(4));
    ZK_HIP(hipMemsetAsync(flag_.p, 0, 4, st));
    if constexpr (Cfg::SQRT_3MOD4) {
      points_decompress_kernel<Fld, NoTs><<<g, blk, 0, st>>>((const uint8_t*)in, len, b, Cfg::ZCASH ? 1 : 0, NoTs{},
                                                            (Affine<Fld>*)out, (uint32_t*)flag_.p);
    } else if constexpr (!IsExtField<Fld>::value) {
      points_decompress_kernel<Fld, TsParams<typename Cfg::FqP>><<<g, blk, 0, st>>>(
          (const uint8_t*)in, len, b, Cfg::ZCASH ? 1 : 0, ts_params(), (Affine<Fld>*)out, (uint32_t*)flag_.p);
    } else {
      return fail(ZK_ERR_BAD_INPUT, "device-side decompression over Fq2 needs q = 3 mod 4");
    }
    ZK_HIP(hipGetLastError());
    uint32_t bad = 0;
    ZK_HIP(hipMemcpyAsync(&bad, flag_.p, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    if (bad) return fail(ZK_ERR_GENERIC, "invalid compressed point at index " + std::to_string(bad - 1) + " (InvalidData)");
    return ZK_OK;
  }
  int points_codec(int group, const void* in, size_t len, void* out, int decompress, hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!in || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (group == ZK_G1) return points_codec_t<Fq_>(in, len, out, decompress, Fq_::from_u64((uint64_t)Cfg::B1), st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) {
        Fq2_ xi{Fq_::from_u64((uint64_t)Cfg::XI0), Fq_::from_u64((uint64_t)Cfg::XI1)};
        Fq2_ b1{Fq_::from_u64((uint64_t)Cfg::B1), Fq_::zero()};
        Fq2_ b2 = Cfg::TWIST_MUL ? b1 * xi : b1 * xi.inverse();
        return points_codec_t<Fq2_>(in, len, out, decompress, b2, st);
      }
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

