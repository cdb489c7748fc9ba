// Part of class Engine<Cfg> (engine_impl.hpp includes this file INSIDE the class body): circom_h / libsnark_h, the Groth16 prover (one proof, two in flight) and the batch prover.
// Split out of engine_impl.hpp in round 5 (one 3 400-line class body had stopped being navigable); not a stand-alone header.

  // ---------------------------------------------------------------- circom_h (ext_wit.rs:104-181)
  // the king step of three d_fft / d_ifft (a, b, c at in + k * per; masks mk->fft_*[first + k]; randomness seed + k):
  // one batched launch when the in-masks are all present or all absent, three launches otherwise
  int king3(const Fr* in, const zk_groth16_masks* mk, int first, int log_m, int inverse, const void* g, int scale,
            int rearrange, uint64_t seed, Fr* out, size_t per, hipStream_t st) {
    const Fr* U = nullptr;
    int rc = umat_for(nullptr, n, &U);
    if (rc) return rc;
    KingBatch<Fr> kb{};
    kb.stride = per;
    bool any = false, all = true;
    for (int k = 0; k < 3; k++) {
      kb.in_mask[k] = mk ? (const Fr*)mk->fft_in[first + k] : nullptr;
      kb.out_mask[k] = mk ? (const Fr*)mk->fft_out[first + k] : nullptr;
      any = any || kb.in_mask[k];
      all = all && kb.in_mask[k];
    }
    if (any == all)
      return king_dispatch_batch(in, kb, 3, n, log_m, inverse, U, g, scale, rearrange, seed, out, false, st);
    for (int k = 0; k < 3; k++) {
      rc = king_dispatch(in + k * per, kb.in_mask[k], n, log_m, inverse, U, g, scale, rearrange, seed + k,
                         out + k * per, kb.out_mask[k], false, st);
      if (rc) return rc;
    }
    return ZK_OK;
  }
  int circom_h(const void* qa, const void* qb, const void* qc, int log_m, const zk_groth16_masks* mk, uint64_t seed,
               void* h, hipStream_t st) override {
    return circom_h_ws(qa, qb, qc, log_m, mk, seed, h, ws(st)->hwork, st);
  }
  int circom_h_ws(const void* qa, const void* qb, const void* qc, int log_m, const zk_groth16_masks* mk, uint64_t seed,
                  void* h, DevBuf& hwork, hipStream_t st) {
    int log_l = ilog2(l);
    if (log_m < log_l) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (log_m + 1 > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "domain too large for this field");
    if (!qa || !qb || !qc || !h) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    size_t Lc = ((size_t)1 << log_m) / l;
    size_t per = (size_t)n * Lc;
    ZK_HIP(hwork.ensure(6 * per * sizeof(Fr)));
    Fr* W0 = (Fr*)hwork.p;
    Fr* W1 = W0 + 3 * per;
    Fr w2m = root_of_unity(log_m + 1);     // Radix2EvaluationDomain::new(2m).element(1), ext_wit.rs:120-125
    // 3 x d_ifft(rearrange = true, g = w_2m)   (ext_wit.rs:127-159); the first pass reads the caller's vectors (no copy)
    int rc = fft1_src(W0, log_m, 1, 3 * (size_t)n, nullptr, st,
                      NttSrc<Fr>{{(const Fr*)qa, (const Fr*)qb, (const Fr*)qc}, (uint32_t)n});
    if (rc) return rc;
    rc = king3(W0, mk, 0, log_m, 1, &w2m, 1, 1, seed, W1, per, st);
    if (rc) return rc;
    // 3 x d_fft(rearrange = false)             (ext_wit.rs:161-170)
    rc = fft1(W1, log_m, 0, 3 * (size_t)n, nullptr, st);
    if (rc) return rc;
    rc = king3(W1, mk, 3, log_m, 0, nullptr, 0, 0, seed + 3, W0, per, st);
    if (rc) return rc;
    // h = a*b - c share-wise, then deg_red     (ext_wit.rs:173-179): the product is formed at deg_red's load
    return deg_red_np(W0, mk ? (const Fr*)mk->degred_in : nullptr, nullptr, n, Lc, seed + 6, (Fr*)h,
                      mk ? (const Fr*)mk->degred_out : nullptr, st, 0, 0, W0 + per, W0 + 2 * per);
  }

  // circom_h of `nb` proofs as ONE launch chain (zk_groth16_prove_batch): the 3 nb vectors go through every NTT pass,
  // king and deg_red launch together (grid.y), so the chain's dozen dependent launches are paid once per batch.
  // q*[b]: [n][Lc]; mk: nb mask sets or nullptr; h: [nb][n][Lc]; proof b draws the randomness a single circom_h with
  // seed + PROOF_SEED_STEP * b draws.
  static constexpr uint32_t PROOF_SEED_STEP = 16;
  int circom_h_batch(int nb, const void* const* qa, const void* const* qb, const void* const* qc, int log_m,
                     const zk_groth16_masks* mk, uint64_t seed, Fr* h, DevBuf& hwork, hipStream_t st) {
    int log_l = ilog2(l);
    if (log_m < log_l) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (log_m + 1 > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "domain too large for this field");
    if (nb < 1 || 3 * nb > KING_BATCH || nb > DEGRED_BATCH) return fail(ZK_ERR_BAD_INPUT, "bad batch size");
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)n * Lc;
    // one launch needs the in-masks of a stage all present or all absent across the batch; otherwise proof by proof
    bool uniform = true;
    if (mk)
      for (int k = 0; k < 6 && uniform; k++)
        for (int b = 1; b < nb; b++)
          if ((mk[b].fft_in[k] != nullptr) != (mk[0].fft_in[k] != nullptr)) uniform = false;
    if (mk && uniform)
      for (int k = 0; k < 6 && uniform; k += 3)
        if ((mk[0].fft_in[k] != nullptr) != (mk[0].fft_in[k + 1] != nullptr) ||
            (mk[0].fft_in[k] != nullptr) != (mk[0].fft_in[k + 2] != nullptr))
          uniform = false;
    if (!uniform) {
      for (int b = 0; b < nb; b++) {
        int rc = circom_h_ws(qa[b], qb[b], qc[b], log_m, &mk[b], seed + (uint64_t)PROOF_SEED_STEP * b, h + b * per, hwork, st);
        if (rc) return rc;
      }
      return ZK_OK;
    }
    ZK_HIP(hwork.ensure(6 * per * nb * sizeof(Fr)));
    Fr* W0 = (Fr*)hwork.p;
    Fr* W1 = W0 + 3 * per * nb;
    Fr w2m = root_of_unity(log_m + 1);
    NttSrc<Fr> src{};
    src.per = (uint32_t)n;
    for (int b = 0; b < nb; b++) {
      if (!qa[b] || !qb[b] || !qc[b]) return fail(ZK_ERR_BAD_INPUT, "null pointer");
      src.p[3 * b] = (const Fr*)qa[b];
      src.p[3 * b + 1] = (const Fr*)qb[b];
      src.p[3 * b + 2] = (const Fr*)qc[b];
    }
    int rc = fft1_src(W0, log_m, 1, 3 * (size_t)n * nb, nullptr, st, src);
    if (rc) return rc;
    const Fr* U = nullptr;
    rc = umat_for(nullptr, n, &U);
    if (rc) return rc;
    auto king = [&](const Fr* in, int first, int inverse, const void* g, int scale, int rearrange, uint64_t sd, Fr* out) {
      KingBatch<Fr> kb{};
      kb.stride = per;
      kb.items_per = 3;
      kb.seed_step = PROOF_SEED_STEP;
      for (int b = 0; b < nb; b++)
        for (int k = 0; k < 3; k++) {
          kb.in_mask[3 * b + k] = mk ? (const Fr*)mk[b].fft_in[first + k] : nullptr;
          kb.out_mask[3 * b + k] = mk ? (const Fr*)mk[b].fft_out[first + k] : nullptr;
        }
      return king_dispatch_batch(in, kb, 3 * nb, n, log_m, inverse, U, g, scale, rearrange, sd, out, false, st);
    };
    rc = king(W0, 0, 1, &w2m, 1, 1, seed, W1);
    if (rc) return rc;
    rc = fft1(W1, log_m, 0, 3 * (size_t)n * nb, nullptr, st);
    if (rc) return rc;
    rc = king(W1, 3, 0, nullptr, 0, 0, seed + 3, W0);
    if (rc) return rc;
    DegredBatch<Fr> db{};
    for (int b = 0; b < nb; b++) {
      db.in_mask[b] = mk ? (const Fr*)mk[b].degred_in : nullptr;
      db.out_mask[b] = mk ? (const Fr*)mk[b].degred_out : nullptr;
    }
    db.in_step = 3 * per;
    db.out_step = per;
    db.seed_step = PROOF_SEED_STEP;
    return deg_red_batch(W0, db, nb, nullptr, n, Lc, seed + 6, h, st, 0, 0, W0 + per, W0 + 2 * per);
  }

  // ---------------------------------------------------------------- prover (prove.rs, sha256.rs:32-129)
  // One proof in flight = one ProveJob: its device scratch, the pending MSMs and the host-side terms.  Everything the
  // host contributes (scalar multiples of CRS constants, of the out-masks and of the in-mask sums, MSM window folds)
  // is a task of the context's persistent pool, submitted when the proof starts and running beside the device work
  // (the overlap the reference gets from tokio::try_join!, prove.rs:209-227); prove_end only adds points.
  static constexpr int NJOBS = 2;                   // proofs in flight per context (zk_groth16_prove_async)
  struct ProveJob {
    bool active = false;
    int slot = 0;
    zk_crs_share crs{};
    zk_groth16_masks mk{};
    bool has_mk = false, r_zero = false, full = true, gate_sorts = false;
    std::atomic<int> sorted_cnt{0};     // witness MSMs of this proof whose sort has been enqueued and recorded (MsmGate)
    int first = 0, count = 0;
    Fr r, s;
    DevBuf hwork, hshare;
    MsmPending pS, pV0, pW, pU;
    P1 S, H, W, U, sS, rH;
    P2 V0;
    P1 rN, sK, rsM, s_cA, r_cB1;
    P2 sK2;
    P1 in1[5], s_in0, r_in1;                        // in-mask sums (index 2 unused) and their multiples
    P2 in2;
    std::vector<P1> s_om0, r_om1;                   // per party: s * out_mask_A[p], r * out_mask_B1[p]
    std::vector<std::future<void>> fut;
    int rc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    Status err;                                     // first failure reported by a task (tasks must not touch `last`)
    std::mutex emu;
  };
  ProveJob jobs_[NJOBS];

  static P1 aff1(const void* p) {
    Affine<Fq_> a;
    memcpy(&a, p, sizeof(a));
    return P1::from_affine(a);
  }
  static P2 aff2(const void* p) {
    Affine<Fq2_> a;
    memcpy(&a, p, sizeof(a));
    return P2::from_affine(a);
  }
  // ---- bounded waits (round 6, VERDICT r5 #10 / ADVICE r5): every wait of a job is bounded by the context's deadline
  // (engine.hpp wait_deadline_ms).  A wait that expires leaves device work or a pool task that still references the
  // job's buffers: the job's slot is then never reused (it stays `active`), the context is marked wedged -- later prover
  // calls fail at once with the first message -- and the state of every gate / event of the job goes to stderr.
  std::atomic<bool> wedged_{false};
  std::string wedged_msg_;
  int wedge(const std::string& what) {
    bool was = wedged_.exchange(true);
    if (!was) {
      std::lock_guard<std::mutex> lk(last_mu);
      wedged_msg_ = what;
    }
    return fail(ZK_ERR_GENERIC, what);
  }
  static const char* ev_state(hipEvent_t e) {
    if (!e) return "none";
    hipError_t q = hipEventQuery(e);
    return q == hipSuccess ? "done" : (q == hipErrorNotReady ? "PENDING" : hipGetErrorString(q));
  }
  static int futs_pending(std::vector<std::future<void>>& fut) {
    int np = 0;
    for (auto& f : fut)
      if (f.valid() && f.wait_for(std::chrono::seconds(0)) != std::future_status::ready) np++;
    return np;
  }
  // joins the pool tasks of a job; false = some did not finish within the deadline (they stay in `fut`)
  bool drain_futs(std::vector<std::future<void>>& fut) {
    const auto dl = deadline_from_now();
    for (auto& f : fut)
      if (f.valid() && f.wait_until(dl) != std::future_status::ready) return false;
    fut.clear();
    return true;
  }
  bool drain(ProveJob& j) { return drain_futs(j.fut); }
  // waits for a launched MSM's device work without folding it (abort paths); false = deadline
  bool settle(MsmPending* p) {
    if (!p->active) return true;
    hipError_t e = event_wait(p->slot->ev);
    if (e == hipErrorNotReady) return false;
    p->active = false;
    p->tab.reset();
    p->tab2.reset();
    return true;
  }
  void dump_job(const char* why, ProveJob& j) {
    fprintf(stderr, "[zksaas] %s: proof slot %d: pool tasks pending %d, sorted_cnt %d; MSM events S+H %s%s, V %s%s, W %s%s, U %s%s\n",
            why, j.slot, futs_pending(j.fut), j.sorted_cnt.load(), j.pS.active ? "" : "(idle) ",
            j.pS.slot ? ev_state(j.pS.slot->ev) : "-", j.pV0.active ? "" : "(idle) ", j.pV0.slot ? ev_state(j.pV0.slot->ev) : "-",
            j.pW.active ? "" : "(idle) ", j.pW.slot ? ev_state(j.pW.slot->ev) : "-", j.pU.active ? "" : "(idle) ",
            j.pU.slot ? ev_state(j.pU.slot->ev) : "-");
  }
  // engine-level failure recorded from a pool task (IEngine::fail is not thread-safe)
  int task_fail(ProveJob& j, int code, const std::string& msg) {
    std::lock_guard<std::mutex> lk(j.emu);
    if (j.err.code == ZK_OK) {
      j.err.code = code;
      j.err.msg = msg;
    }
    return code;
  }

  // Starts one proof (full = all n parties and the assembly; otherwise the five partial d_msm sums of parties
  // [first, first + count) for the multi-GPU flow, where circom_h is driven by the caller and h arrives in finish).
  // A failure after the job has been marked active leaves pool tasks and MSMs in flight that reference the caller's
  // buffers: every such return goes through abort_job (the job is free again, the error message is kept).
  int prove_begin(ProveJob& j, const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                  const void* a_share, const void* ax_share, const Fr& r, const Fr& s, int log_m,
                  const zk_groth16_masks* mk, uint64_t seed, bool full, int first, int count, hipStream_t st,
                  bool gate_sorts = false) {
    if (j.active) return fail(ZK_ERR_BAD_INPUT, "a proof is already in flight on this slot");
    int rc = prove_begin_impl(j, crs, qa, qb, qc, a_share, ax_share, r, s, log_m, mk, seed, full, first, count, st,
                              gate_sorts);
    if (rc && j.active) {
      Status keep = last;
      abort_job(j);
      last = keep;
    }
    return rc;
  }
  void init_job(ProveJob& j, const zk_crs_share* crs, const zk_groth16_masks* mk, const Fr& r, const Fr& s, bool full,
                int first, int count) {
    j.crs = *crs;
    j.has_mk = mk != nullptr;
    j.mk = mk ? *mk : zk_groth16_masks{};
    j.r = r;
    j.s = s;
    j.r_zero = r.is_zero();
    j.full = full;
    j.first = first;
    j.count = count;
    j.err = Status{};
    for (int i = 0; i < 8; i++) j.rc[i] = 0;
    j.S = j.H = j.W = j.U = j.sS = j.rH = j.s_in0 = j.r_in1 = P1::identity();
    j.V0 = j.in2 = P2::identity();
    for (int k = 0; k < 5; k++) j.in1[k] = P1::identity();
    j.s_om0.assign(n, P1::identity());
    j.r_om1.assign(n, P1::identity());
  }
  // the gate of a proof's U-MSM: its sort is event 3 of the "all sorts first" barrier (prove_begin_impl)
  MsmGate u_gate(ProveJob& j) {
    MsmGate g{};
    if (j.gate_sorts) {
      g.sorted_ev = ev_sorted_[j.slot][3];
      g.sorted_cnt = &j.sorted_cnt;
    }
    return g;
  }
  int prove_begin_impl(ProveJob& j, const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                       const void* a_share, const void* ax_share, const Fr& r, const Fr& s, int log_m,
                       const zk_groth16_masks* mk, uint64_t seed, bool full, int first, int count, hipStream_t st,
                       bool gate_sorts) {
    if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    auto t_mark = std::chrono::steady_clock::now();
    auto host_span = [&](int slot) {                 // host:launch.* (zk_profile_read)
      if (!prof.on) return;
      const auto now = std::chrono::steady_clock::now();
      prof.host_add(slot, std::chrono::duration<double, std::milli>(now - t_mark).count());
      t_mark = now;
    };
    int rc = ensure_streams();
    if (rc) return rc;
    init_job(j, crs, mk, r, s, full, first, count);
    const size_t Lc = ((size_t)1 << log_m) / l;
    const int dev = device;
    const int ws0 = j.slot * 6;
    // every internal stream is ordered after the work already queued on the caller's stream (the shares may still
    // be in flight there: found by tools/c5_bls381.py, where the a_share pack kernel of a 2^22 witness was still
    // running when the S/H/V MSMs started reading it)
    // (nothing to order after when the caller's stream has drained -- the case between two proofs of a loop: one query
    // instead of an event and six waits at the head of every chain of the proof)
    if (hipStreamQuery(st) != hipSuccess) {
      (void)hipGetLastError();                       // hipErrorNotReady is not an error here
      ZK_HIP(hipEventRecord(ev_in_[j.slot], st));
      for (hipStream_t is : streams_) ZK_HIP(hipStreamWaitEvent(is, ev_in_[j.slot], 0));
    }
    j.active = true;
    ProveJob* J = &j;
    const size_t cstride = crs->len_a;
    // ---- device pipelines: the four MSMs over the witness shares do not depend on h.  Each is enqueued by a pool task
    // (a launch is a dozen kernel launches), which then waits for the slot's event and folds the windows on the host.
    // V (G2) is the longest chain: issued first, on a high-priority stream, as ONE launch over all parties (two halves
    // on two streams paid a second bucket reduction: 257 vs 279 proofs/s, round 2).
    // Orders that were measured and dropped (rounds 2-4, DESIGN.md "What bounds one proof"): V's accumulate ahead of the
    // G1 accumulates (283-285 vs 291-293 proofs/s), circom_h's launches enqueued ahead of the MSM tasks at the SHA-256
    // size (550-555 vs 571-596), S and H as two MSMs with their own sorts (391 vs 464), raised issue priority or shorter
    // lane ranges for the U-MSM (541-544 vs 591-594).
    const Fr* cf = msm_.coef_d_ + first;
    MsmGate gate{};
    // Large domains (2^h_first_log_m_ and up; 2^20 by default): circom_h and the SORT of the U-MSM that consumes it go
    // first, and the accumulate kernels of the witness MSMs wait for them (they still sort beside circom_h).  A proof of
    // this size is bound by the multiplier throughput of its five accumulate kernels whichever order they run in (0.667
    // vs 0.666 s per proof at 2^22, round 3), but next to four multi-hundred-millisecond accumulates every NTT pass (a
    // 512-thread / 64 KB workgroup needs a whole CU to drain) and every one-workgroup-per-CU sort kernel sat in the
    // dispatcher for most of the proof: the HIP-event spans of those slots then measured the wait, not the kernel
    // (0.43 s per sort against 25 ms of execution, profiles/r04_c5_kernel_stats.csv).
    // The witness MSMs also wait for EACH OTHER's sorts (V's accumulate alone holds every CU for 0.46 s at 2^24): every
    // launch of the proof records an event after its sort and counts itself in; an accumulate kernel is enqueued once all
    // four are on record, behind all four events.  The sharded prover (full == false) does the same: there the U sort is
    // enqueued by prove_launch_u once the king rounds of circom_h have been.
    const bool gated = j.gate_sorts = (full || gate_sorts) && log_m >= h_first_log_m_;
    j.sorted_cnt.store(0, std::memory_order_relaxed);
    if (gated) {
      gate.sorted_cnt = &j.sorted_cnt;
      gate.wait_sorted = ev_sorted_[j.slot];
      gate.n_wait_sorted = gate.sorted_need = 4;
    }
    bool hu_done = false;
    if (full && gated) {
      hipError_t he = j.hshare.ensure((size_t)n * Lc * sizeof(Fr));
      if (he != hipSuccess) return hip_fail(he, "h share buffer");
      rc = circom_h_ws(qa, qb, qc, log_m, mk, seed, j.hshare.p, j.hwork, streams_[5]);
      if (rc) return rc;
      rc = msm_.template launch_t<Fq_>(this, crs->u_d, j.hshare.p, (size_t)n * crs->len_u, msm_.coef_d_, crs->len_u,
                                      streams_[5], ws0 + 0, &j.pU, nullptr, u_gate(j));
      if (rc) return rc;
      hu_done = true;
    }
    auto msm_task = [this, J, dev, gate](auto fld_tag, int which, const void* bases, const void* bases2, const void* scal,
                                         size_t npts, const Fr* coef, size_t plen, hipStream_t stream, int wslot,
                                         MsmPending* pend, auto* out1, auto* out2) {
      using Fld = decltype(fld_tag);
      J->fut.push_back(pool_->submit([=]() {
        (void)hipSetDevice(dev);
        MsmGate g = gate;
        if (g.n_wait_sorted) g.sorted_ev = ev_sorted_[J->slot][which];          // which = 0 (S/H), 1 (W), 2 (V); 3 = U
        int rc2 = msm_.template launch_t<Fld>(this, bases, scal, npts, coef, plen, stream, wslot, pend, bases2, g);
        if (rc2 && g.sorted_cnt) g.sorted_cnt->fetch_add(1, std::memory_order_release);   // never leave the others spinning
        if (!rc2) rc2 = msm_.template finish_t<Fld>(this, pend, out1, out2);
        J->rc[which] = rc2;
        if constexpr (std::is_same<Fld, Fq_>::value) {
          if (!rc2 && J->full && which == 0) {        // s*S and r*H off the tail (prove.rs:229-235, linearity)
            J->sS = host_scalar_mul<FrP, Fq_>(J->S, J->s);
            if (out2 != nullptr && !J->r_zero) J->rH = host_scalar_mul<FrP, Fq_>(J->H, J->r);
          }
        }
      }));
    };
    msm_task(Fq2_{}, 2, crs->v_d, nullptr, a_share, (size_t)count * cstride, cf, cstride, streams_[2], ws0 + 3, &j.pV0,
             &j.V0, (P2*)nullptr);
    // S and H multiply two base vectors by the same witness shares: ONE launch over both vectors, each with its own sort
    // (their identity bases differ: b_query is the identity for every wire no B-row mentions, 59 % in the SHA-256 circuit)
    msm_task(Fq_{}, 0, crs->s_d, j.r_zero ? nullptr : crs->h_d, a_share, (size_t)count * cstride, cf, cstride,
             streams_[0], ws0 + 1, &j.pS, &j.S, j.r_zero ? (P1*)nullptr : &j.H);
    msm_task(Fq_{}, 1, crs->w_d, nullptr, ax_share, (size_t)count * crs->len_w, cf, crs->len_w, streams_[3], ws0 + 4,
             &j.pW, &j.W, (P1*)nullptr);
    // (the host terms are two dozen pool submissions: behind circom_h's launches when those are still to come -- they
    // are the head of the proof's longest chain and the terms are not needed before prove_end)
    if (!(full && !hu_done)) submit_host_terms(J, j.fut, full, first, count);
    // ---- circom_h and the U-MSM that depends on it form a long dependent chain: high-priority internal stream.
    // At the SHA-256 size, holding the other MSM streams (or only their accumulate launches) back until circom_h has
    // finished was measured and rejected: 8.6-8.8 ms per proof against 6.9 ms when everything is issued at once.
    if (full && !hu_done) {
      host_span(PROF_HOST_SUBMIT);
      hipStream_t hs = streams_[5];
      hipError_t he = j.hshare.ensure((size_t)n * Lc * sizeof(Fr));
      if (he != hipSuccess) return hip_fail(he, "h share buffer");
      rc = circom_h_ws(qa, qb, qc, log_m, mk, seed, j.hshare.p, j.hwork, hs);
      if (rc) return rc;
      host_span(PROF_HOST_H);
      rc = msm_.template launch_t<Fq_>(this, crs->u_d, j.hshare.p, (size_t)n * crs->len_u, msm_.coef_d_, crs->len_u, hs,
                                      ws0 + 0, &j.pU);
      if (rc) return rc;
      submit_host_terms(J, j.fut, full, first, count);
      host_span(PROF_HOST_U);
    }
    return ZK_OK;
  }

  // Host-side terms of one proof that depend on nothing but its inputs (scalar multiples of CRS constants, in-mask sums,
  // per-party multiples of the out-masks): tasks of the worker pool, running beside the device work.
  void submit_host_terms(ProveJob* J, std::vector<std::future<void>>& fut, bool full, int first, int count) {
    if (full) {
      fut.push_back(pool_->submit([J]() {
        P1 d1 = aff1(J->crs.delta_g1);
        J->rN = host_scalar_mul<FrP, Fq_>(d1, J->r);
        J->s_cA = host_scalar_mul<FrP, Fq_>(xyzz_add_ni(xyzz_add_ni(aff1(J->crs.a_query0), J->rN), aff1(J->crs.alpha_g1)), J->s);
      }));
      fut.push_back(pool_->submit([J]() {
        P1 d1 = aff1(J->crs.delta_g1);
        J->sK = host_scalar_mul<FrP, Fq_>(d1, J->s);
        J->r_cB1 = host_scalar_mul<FrP, Fq_>(xyzz_add_ni(xyzz_add_ni(aff1(J->crs.b_g1_query0), J->sK), aff1(J->crs.beta_g1)), J->r);
      }));
      fut.push_back(pool_->submit([J]() { J->rsM = host_scalar_mul<FrP, Fq_>(aff1(J->crs.delta_g1), J->r * J->s); }));
      fut.push_back(pool_->submit([J]() { J->sK2 = host_scalar_mul<FrP, Fq2_>(aff2(J->crs.delta_g2), J->s); }));
    }
    const zk_groth16_masks* mk = J->has_mk ? &J->mk : nullptr;
    if (mk) {
      // in-mask terms sum_p coef_p * mask_p (the king's unpack2 + sum over the masked points, dmsm/mod.rs:85-86)
      for (int k = 0; k < 5; k++) {
        if (!mk->msm_in[k] || (k == 1 && J->r_zero)) continue;
        const void* im = mk->msm_in[k];
        fut.push_back(pool_->submit([this, J, k, im, first, count]() {
          if (k == 2) {
            J->in2 = msm_.template mask_term<Fq2_>(im, first, count);
          } else {
            J->in1[k] = msm_.template mask_term<Fq_>(im, first, count);
            if (k == 0 && J->full) J->s_in0 = host_scalar_mul<FrP, Fq_>(J->in1[0], J->s);
            if (k == 1 && J->full) J->r_in1 = host_scalar_mul<FrP, Fq_>(J->in1[1], J->r);
          }
        }));
      }
      // per-party multiples of the out-masks of A and B1 that enter C = s*A + r*B1 + ...
      if (full)
        for (int p = 0; p < n; p++) {
          if (mk->msm_out[0])
            fut.push_back(pool_->submit([J, p]() {
              J->s_om0[p] = host_scalar_mul<FrP, Fq_>(jacobian_to_xyzz(((const Jacobian<Fq_>*)J->mk.msm_out[0])[p]), J->s);
            }));
          if (mk->msm_out[1] && !J->r_zero)
            fut.push_back(pool_->submit([J, p]() {
              J->r_om1[p] = host_scalar_mul<FrP, Fq_>(jacobian_to_xyzz(((const Jacobian<Fq_>*)J->mk.msm_out[1])[p]), J->r);
            }));
        }
    }
  }

  // the U-MSM of a partial job (h comes from the caller's king rounds)
  int prove_launch_u(ProveJob& j, const void* h_share, hipStream_t st) {
    return msm_.template launch_t<Fq_>(this, j.crs.u_d, h_share, (size_t)j.count * j.crs.len_u, msm_.coef_d_ + j.first,
                                      j.crs.len_u, st, j.slot * 6 + 0, &j.pU, nullptr, u_gate(j));
  }

  // joins everything; sums[0..4] = S, H, V, W, U including the in-mask terms
  int prove_join(ProveJob& j, P1* S, P1* H, P2* V, P1* W, P1* U) {
    int rc = msm_.template finish_t<Fq_>(this, &j.pU, &j.U);
    if (rc && j.pU.active) {
      dump_job("prove_join: the U-MSM chain did not finish within the deadline", j);
      Status keep = last;
      return wedge(keep.msg);
    }
    if (!drain(j)) {
      dump_job("prove_join: pool tasks did not finish within the deadline", j);
      return wedge("host tasks of the proof did not finish within the deadline (wait_deadline_ms); the slot is not reused");
    }
    j.active = false;
    if (rc) return rc;
    for (int i = 0; i < 8; i++)
      if (j.rc[i]) {
        // a pool task failed: its message was recorded on the engine by msm_launch (hip_fail); keep it
        return j.rc[i];
      }
    if (j.err.code) return fail(j.err.code, j.err.msg);
    *S = xyzz_add_ni(j.S, j.in1[0]);
    *H = j.r_zero ? P1::identity() : xyzz_add_ni(j.H, j.in1[1]);
    *V = xyzz_add_ni(j.V0, j.in2);
    *W = xyzz_add_ni(j.W, j.in1[3]);
    *U = xyzz_add_ni(j.U, j.in1[4]);
    return ZK_OK;
  }

  // The end of one proof.  The U chain (circom_h, then its MSM) ends last by a margin: everything that does not depend on U
  // -- the join of the pool tasks, A and B of every party, C without its U term -- is done while it still runs; after U's
  // event only its fold and one addition per party are left.  (U is always waited for, whatever failed before: its launch
  // references the job's buffers.)
  int prove_end(ProveJob& j, void* pi_a, void* pi_b, void* pi_c) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!drain(j)) {
      dump_job("zk_groth16_wait: pool tasks did not finish within the deadline", j);
      return wedge("zk_groth16_wait: host tasks of the proof did not finish within the deadline (wait_deadline_ms); the slot is not reused");
    }
    bool early_ok = !j.err.code;
    for (int i = 0; i < 8; i++) early_ok = early_ok && !j.rc[i];
    std::vector<P1> c_part(early_ok ? n : 0);
    if (early_ok) {
      const P1 S = xyzz_add_ni(j.S, j.in1[0]), H = j.r_zero ? P1::identity() : xyzz_add_ni(j.H, j.in1[1]),
               W = xyzz_add_ni(j.W, j.in1[3]);
      const P2 V = xyzz_add_ni(j.V0, j.in2);
      int rc = assemble_job(j, S, H, V, W, j.in1[4], pi_a, pi_b, pi_c, c_part.data());
      if (rc) early_ok = false;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (prof.on && j.pU.active) (void)event_wait(j.pU.slot->ev);                // (finish_t waits again: returns at once)
    const auto t2 = std::chrono::steady_clock::now();
    struct HostSpans {                                                          // host:prove_wait / host:prove_tail
      Profiler& pr;
      std::chrono::steady_clock::time_point a, b;
      ~HostSpans() {
        if (!pr.on) return;
        pr.host_add(PROF_HOST_WAIT, std::chrono::duration<double, std::milli>(b - a).count());
        pr.host_add(PROF_HOST_TAIL, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b).count());
      }
    } spans{prof, t0, t2};
    (void)t1;
    int rc = msm_.template finish_t<Fq_>(this, &j.pU, &j.U);
    if (rc && j.pU.active) {               // the U chain's event did not signal within the deadline: nothing may be reused
      dump_job("zk_groth16_wait: the U-MSM chain did not finish within the deadline", j);
      Status keep = last;
      return wedge(keep.msg);
    }
    j.active = false;
    if (rc) return rc;
    for (int i = 0; i < 8; i++)
      if (j.rc[i]) return j.rc[i];      // a pool task failed: its message was recorded on the engine by msm_launch; keep it
    if (j.err.code) return fail(j.err.code, j.err.msg);
    if (!early_ok) return fail(ZK_ERR_GENERIC, "proof assembly failed");
    Jacobian<Fq_>* oc = (Jacobian<Fq_>*)pi_c;
    for (int p = 0; p < n; p++) oc[p] = xyzz_to_jacobian(xyzz_add_ni(c_part[p], j.U));
    return ZK_OK;
  }
  // the n parties' (A, B, C) shares of one proof from its five MSM totals (in-mask terms included) and the host terms
  // (c_part != nullptr: the C shares stay in extended form there instead of pi_c -- prove_end adds a late term to them)
  int assemble_job(ProveJob& j, const P1& S, const P1& H, const P2& V, const P1& W, const P1& U, void* pi_a, void* pi_b,
                   void* pi_c, P1* c_part = nullptr) {
    const zk_groth16_masks* mk = j.has_mk ? &j.mk : nullptr;
    // prove.rs:40-56 / 99-110 / 148-158 / 229-235 for every party; C = s*A + r*B1 - rs*delta + W + U by linearity:
    //   s*A_p = s*(a0 + r*delta + alpha) + s*S + s*in0 + s*om0_p     (every term was computed beside the device work)
    P1 cA = xyzz_add_ni(xyzz_add_ni(aff1(j.crs.a_query0), j.rN), aff1(j.crs.alpha_g1));
    P1 cB1 = xyzz_add_ni(xyzz_add_ni(aff1(j.crs.b_g1_query0), j.sK), aff1(j.crs.beta_g1));
    P2 cB2 = xyzz_add_ni(xyzz_add_ni(aff2(j.crs.b_g2_query0), j.sK2), aff2(j.crs.beta_g2));
    P1 A0 = xyzz_add_ni(cA, S);
    P2 B0 = xyzz_add_ni(cB2, V);
    P1 C0 = xyzz_add_ni(xyzz_add_ni(j.s_cA, j.sS), j.s_in0);
    if (!j.r_zero) C0 = xyzz_add_ni(C0, xyzz_add_ni(xyzz_add_ni(j.r_cB1, j.rH), j.r_in1));
    C0 = xyzz_add_ni(C0, j.rsM.neg());
    C0 = xyzz_add_ni(C0, xyzz_add_ni(W, U));
    (void)cB1;
    (void)H;
    Jacobian<Fq_>* oa = (Jacobian<Fq_>*)pi_a;
    Jacobian<Fq2_>* ob = (Jacobian<Fq2_>*)pi_b;
    Jacobian<Fq_>* oc = (Jacobian<Fq_>*)pi_c;
    auto om1 = [&](int k, int p) { return jacobian_to_xyzz(((const Jacobian<Fq_>*)mk->msm_out[k])[p]); };
    const bool uniform = !mk || (!mk->msm_out[0] && !mk->msm_out[1] && !mk->msm_out[2] && !mk->msm_out[3] &&
                                 !mk->msm_out[4]);
    for (int p = 0; p < n; p++) {
      if (uniform && p > 0) {
        oa[p] = oa[0];
        ob[p] = ob[0];
        if (c_part) c_part[p] = c_part[0];
        else oc[p] = oc[0];
        continue;
      }
      P1 A = A0, C = C0;
      P2 B2 = B0;
      if (!uniform) {
        if (mk->msm_out[0]) {
          A = xyzz_add_ni(A, om1(0, p));
          C = xyzz_add_ni(C, j.s_om0[p]);
        }
        if (mk->msm_out[1] && !j.r_zero) C = xyzz_add_ni(C, j.r_om1[p]);
        if (mk->msm_out[2]) B2 = xyzz_add_ni(B2, jacobian_to_xyzz(((const Jacobian<Fq2_>*)mk->msm_out[2])[p]));
        if (mk->msm_out[3]) C = xyzz_add_ni(C, om1(3, p));
        if (mk->msm_out[4]) C = xyzz_add_ni(C, om1(4, p));
      }
      oa[p] = xyzz_to_jacobian(A);
      ob[p] = xyzz_to_jacobian(B2);
      if (c_part) c_part[p] = C;
      else oc[p] = xyzz_to_jacobian(C);
    }
    return ZK_OK;
  }

  int check_prove_args(const zk_crs_share* crs, const void* r_, const void* s_, int log_m) {
    if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    if (wedged_.load()) {
      std::string m;
      {
        std::lock_guard<std::mutex> lk(last_mu);
        m = wedged_msg_;
      }
      return fail(ZK_ERR_GENERIC, "the context is wedged by an earlier timed-out wait (destroy it): " + m);
    }
    if (!crs || !r_ || !s_) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int log_l = ilog2(l);
    if (log_m < log_l || log_m + 1 > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (crs->len_u != ((size_t)1 << log_m) / l)
      return fail(ZK_ERR_BAD_INPUT, "h_query share length must be m/l");   // dmsm/mod.rs:71
    return ZK_OK;
  }

  int groth16_prove(const zk_crs_share* crs, const void* qa, const void* qb, const void* qc, const void* a_share,
                    const void* ax_share, const void* r_, const void* s_, int log_m, const zk_groth16_masks* mk,
                    uint64_t seed, void* pi_a, void* pi_b, void* pi_c, hipStream_t st) override {
    if (!pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int h = -1;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = groth16_prove_async(crs, qa, qb, qc, a_share, ax_share, r_, s_, log_m, mk, seed, st, &h);
    if (rc) return rc;
    if (prof.on) prof.host_add(PROF_HOST_LAUNCH, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return groth16_wait(h, pi_a, pi_b, pi_c);
  }
  int groth16_prove_async(const zk_crs_share* crs, const void* qa, const void* qb, const void* qc, const void* a_share,
                          const void* ax_share, const void* r_, const void* s_, int log_m, const zk_groth16_masks* mk,
                          uint64_t seed, hipStream_t st, int* handle) override {
    int rc = check_prove_args(crs, r_, s_, log_m);
    if (rc) return rc;
    if (!qa || !qb || !qc || !a_share || !ax_share || !handle) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int slot = -1;
    for (int i = 0; i < NJOBS; i++)
      if (!jobs_[i].active && !djobs_[i].active) {       // (a sharded proof owns its slot until it is joined or aborted)
        slot = i;
        break;
      }
    if (slot < 0) return fail(ZK_ERR_BAD_INPUT, "too many proofs in flight (zk_groth16_wait one first)");
    ProveJob& j = jobs_[slot];
    j.slot = slot;
    Fr r = Fr::from_limbs((const uint32_t*)r_), s = Fr::from_limbs((const uint32_t*)s_);
    rc = prove_begin(j, crs, qa, qb, qc, a_share, ax_share, r, s, log_m, mk, seed, true, 0, n, st);
    if (rc) {
      Status keep = last;
      abort_job(j);
      last = keep;
      return rc;
    }
    *handle = slot;
    return ZK_OK;
  }
  int groth16_wait(int handle, void* pi_a, void* pi_b, void* pi_c) override {
    if (handle < 0 || handle >= NJOBS || !jobs_[handle].active) return fail(ZK_ERR_BAD_INPUT, "no proof in flight on this handle");
    if (djobs_[handle].active) return fail(ZK_ERR_BAD_INPUT, "a sharded proof is in flight on this handle: zk_dist_groth16_wait joins it");
    if (!pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    return prove_end(jobs_[handle], pi_a, pi_b, pi_c);
  }
  // joins a job's tasks and device work and marks it free (after an error, or zk_groth16_abort)
  void abort_job(ProveJob& j) {
    j.sorted_cnt.fetch_add(1 << 20, std::memory_order_release);      // tasks waiting at the sort barrier go on (and fail or finish)
    bool ok = drain(j);
    MsmPending* ps[4] = {&j.pS, &j.pV0, &j.pW, &j.pU};
    for (MsmPending* p : ps) ok = settle(p) && ok;
    if (!ok) {
      dump_job("abort: work of the proof is still pending after the deadline", j);
      Status keep = last;
      wedge("a proof could not be aborted within the deadline (wait_deadline_ms): its slot is not reused");
      if (keep.code) last = keep;
      return;                                                        // stays active: nothing of it is reused
    }
    j.active = false;
  }
  // also the abort of a sharded proof in flight (zk_dist_groth16_prove_async hands out handles of the same space): the
  // slot is free for either kind afterwards.  The channels' round counters are NOT rewound -- every rank must abort the
  // same proof, as every rank must issue the same sequence of collective calls.
  int groth16_abort(int handle) override {
    if (handle < 0 || handle >= NJOBS) return fail(ZK_ERR_BAD_INPUT, "bad handle");
    abort_job(jobs_[handle]);
    djobs_[handle].active = false;
    return ZK_OK;
  }

  // ---------------------------------------------------------------- a batch of proofs against one CRS
  // zk_groth16_prove_batch: `nb` witnesses proved against the SAME packed CRS in one pass.  The reference runs its
  // parties as concurrent tasks and a service runs proofs concurrently (mpc-net/src/multi.rs:317-327,
  // groth16/examples/sha256.rs:316-360); one proof of the SHA-256 circuit leaves most of the chip idle (a dozen short
  // dependent launches per MSM, 2.3 waves per SIMD in the accumulate), so the batch goes through every stage TOGETHER:
  // each of the five MSMs is ONE sort / accumulate / finalize / reduce chain over nb scalar vectors against one base
  // vector (msm.hpp MsmScalars: bucket sets indexed by (proof, window)), circom_h is one launch chain over 3 nb
  // vectors.  Host terms (scalar multiples of masks and CRS constants) are per proof, on the worker pool as before.
  static constexpr int MAX_PROOF_BATCH = MSM_MAXB;
  struct BatchJob {
    bool active = false;
    int nb = 0;
    std::vector<std::unique_ptr<ProveJob>> pj;      // per proof: host terms and the five MSM totals
    MsmPending pSH, pV, pW, pU;
    DevBuf hwork, hshare;
    std::vector<std::future<void>> fut;
    int rc[4] = {0, 0, 0, 0};
  };
  // Two batches may be in flight (zk_groth16_prove_batch_async): each has its own MSM workspaces, scratch and stream
  // set, so that the sort phase of one batch runs under the accumulate kernels of the other and the reduction tails of one
  // under the other's accumulates -- one batch alone leaves the chip partly idle for ~1 ms at either end.
  static constexpr int NBATCH = 3;
  struct BatchJobX : BatchJob {
    zk_crs_share crs{};
    int slot = 0;
    hipEvent_t ev_in = nullptr;
    bool own_streams = false, full = true;
    int first = 0, count = 0;
    MsmGate gate_u;
    hipEvent_t ev_acc[4] = {nullptr, nullptr, nullptr, nullptr};      // V, S+H, W, U: recorded behind the accumulate kernel
    std::atomic<int> acc_flag[4];                                     // ... once that record has been enqueued (MsmGate)
    hipStream_t st[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  };
  BatchJobX bjobs_[NBATCH];

  void dump_batch(const char* why, BatchJobX& B) {
    const MsmPending* ps[4] = {&B.pV, &B.pSH, &B.pW, &B.pU};
    const char* nm[4] = {"V", "S+H", "W", "U"};
    fprintf(stderr, "[zksaas] %s: batch slot %d (%d proofs): pool tasks pending %d;", why, B.slot, B.nb, futs_pending(B.fut));
    for (int i = 0; i < 4; i++)
      fprintf(stderr, " %s: launched-flag %d, accumulate event %s, chain event %s%s;", nm[i], B.acc_flag[i].load(), ev_state(B.ev_acc[i]),
              ps[i]->slot ? ev_state(ps[i]->slot->ev) : "-", ps[i]->active ? "" : " (idle)");
    fprintf(stderr, "\n");
  }
  void abort_batch(BatchJobX& B) {
    bool ok = drain_futs(B.fut);
    MsmPending* ps[4] = {&B.pSH, &B.pV, &B.pW, &B.pU};
    for (MsmPending* p : ps) ok = settle(p) && ok;
    if (!ok) {
      dump_batch("abort: work of the batch is still pending after the deadline", B);
      Status keep = last;
      wedge("a batch could not be aborted within the deadline (wait_deadline_ms): its slot is not reused");
      if (keep.code) last = keep;
      return;
    }
    B.active = false;
  }

  int groth16_prove_batch(const zk_crs_share* crs, int nb, const void* const* qa, const void* const* qb,
                          const void* const* qc, const void* const* a_share, const void* const* ax_share, const void* r_,
                          const void* s_, int log_m, const zk_groth16_masks* mk, uint64_t seed, void* pi_a, void* pi_b,
                          void* pi_c, hipStream_t st) override {
    if (!pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int h = -1;
    int rc = groth16_prove_batch_async(crs, nb, qa, qb, qc, a_share, ax_share, r_, s_, log_m, mk, seed, st, &h);
    if (rc) return rc;
    return groth16_batch_wait(h, pi_a, pi_b, pi_c);
  }
  int groth16_prove_batch_async(const zk_crs_share* crs_in, int nb, const void* const* qa, const void* const* qb,
                                const void* const* qc, const void* const* a_share, const void* const* ax_share,
                                const void* r_, const void* s_, int log_m, const zk_groth16_masks* mk, uint64_t seed,
                                hipStream_t st, int* handle) override {
    if (!qa || !qb || !qc || !handle) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (nb >= 1 && nb <= MAX_PROOF_BATCH)
      for (int b = 0; b < nb; b++)
        if (!qa[b] || !qb[b] || !qc[b]) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int slot = -1;
    int rc = batch_begin(crs_in, nb, a_share, ax_share, r_, s_, log_m, mk, true, 0, n, st, &slot);
    if (rc) return rc;
    BatchJobX& B = bjobs_[slot];
    // ---- circom_h of the whole batch and the U-MSM behind it
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)n * Lc;
    rc = circom_h_batch(nb, qa, qb, qc, log_m, mk, seed, (Fr*)B.hshare.p, B.hwork, B.st[5]);
    if (!rc) rc = batch_launch_u(B, (const Fr*)B.hshare.p, per, B.st[5]);
    if (rc) {
      Status keep = last;
      abort_batch(B);
      last = keep;
      return rc;
    }
    *handle = slot;
    return ZK_OK;
  }
  // the U-MSM of a batch: h_all = [nb][count * m/l] (proof b at h_all + b * stride)
  int batch_launch_u(BatchJobX& B, const Fr* h_all, size_t stride, hipStream_t st) {
    MsmBatchArg ba_h;
    ba_h.nb = B.nb;
    for (int b = 0; b < B.nb; b++) ba_h.p[b] = h_all + (size_t)b * stride;
    int rc = msm_.template launch_t<Fq_>(this, B.crs.u_d, nullptr, (size_t)B.count * B.crs.len_u, msm_.coef_d_ + B.first,
                                        B.crs.len_u, st, 12 + 6 * B.slot + 0, &B.pU, nullptr, B.gate_u, &ba_h);
    B.acc_flag[3].store(1, std::memory_order_release);
    return rc;
  }
  // Starts a batch: the four witness MSMs of parties [first, first + count) for nb proofs (one launch chain each) and
  // the proofs' host terms.  full: all n parties, assembly included; otherwise the partial sums of one rank (the in-mask
  // terms only).  The caller enqueues circom_h and batch_launch_u, then batch_join.
  int batch_begin(const zk_crs_share* crs_in, int nb, const void* const* a_share, const void* const* ax_share,
                  const void* r_, const void* s_, int log_m, const zk_groth16_masks* mk, bool full, int first, int count,
                  hipStream_t st, int* slot_out) {
    if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
    int rc = check_prove_args(crs_in, r_, s_, log_m);
    if (rc) return rc;
    if (nb < 1 || nb > MAX_PROOF_BATCH || 3 * nb > KING_BATCH || nb > DEGRED_BATCH)
      return fail(ZK_ERR_BAD_INPUT, "batch size must be in 1.." + std::to_string(MAX_PROOF_BATCH));
    if (!a_share || !ax_share) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    for (int b = 0; b < nb; b++)
      if (!a_share[b] || !ax_share[b]) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int slot = -1;
    for (int i = 0; i < NBATCH; i++)
      if (!bjobs_[i].active) {
        slot = i;
        break;
      }
    if (slot < 0) return fail(ZK_ERR_BAD_INPUT, "too many batches in flight (zk_groth16_batch_wait one first)");
    BatchJobX& B = bjobs_[slot];
    B.slot = slot;
    rc = ensure_streams();
    if (rc) return rc;
    {
      // every handle is checked on its own: a failure half-way leaves the rest to be created by the next call
      if (!B.ev_in) ZK_HIP(hipEventCreateWithFlags(&B.ev_in, hipEventDisableTiming));
      for (int i = 0; i < 4; i++)
        if (!B.ev_acc[i]) ZK_HIP(hipEventCreateWithFlags(&B.ev_acc[i], hipEventDisableTiming));
      // (splitting the chip between the G2 MSM and the rest with CU masks was measured in round 3 and dropped: DESIGN.md
      // "batched proving")
      B.own_streams = slot != 0;
      for (int i = 0; i < 6; i++) {
        if (B.st[i]) continue;
        if (!B.own_streams) {
          B.st[i] = streams_[i];                     // batch slot 0 shares the single-proof stream set
        } else {
          int pr = 0;
          ZK_HIP(hipStreamGetPriority(streams_[i], &pr));
          ZK_HIP(hipStreamCreateWithPriority(&B.st[i], hipStreamNonBlocking, pr));
        }
      }
    }
    B.crs = *crs_in;
    const zk_crs_share* crs = &B.crs;
    hipStream_t* const streams_ = B.st;              // this batch's stream set
    hipEvent_t const ev_batch_in_ = B.ev_in;
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)count * Lc;
    ZK_HIP(B.hshare.ensure(per * nb * sizeof(Fr)));
    B.nb = nb;
    B.first = first;
    B.count = count;
    B.full = full;
    const int n = count;                                 // parties of this launch (shadows the context's n below)
    while ((int)B.pj.size() < nb) B.pj.emplace_back(new ProveJob());
    bool all_r_zero = true;
    for (int b = 0; b < nb; b++) {
      Fr r = Fr::from_limbs((const uint32_t*)r_ + (size_t)b * FrP::N), s = Fr::from_limbs((const uint32_t*)s_ + (size_t)b * FrP::N);
      init_job(*B.pj[b], crs, mk ? &mk[b] : nullptr, r, s, full, first, count);
      all_r_zero = all_r_zero && r.is_zero();
    }
    for (int i = 0; i < 4; i++) B.rc[i] = 0;
    ZK_HIP(hipEventRecord(ev_batch_in_, st));
    for (int i = 0; i < 6; i++) ZK_HIP(hipStreamWaitEvent(streams_[i], ev_batch_in_, 0));
    B.active = true;
    BatchJob* BJ = &B;
    const int dev = device;
    const int ws0 = 12 + 6 * slot;                      // the batch's own MSM workspaces (msm.hpp MSM_WS)
    const Fr* cf = msm_.coef_d_ + first;
    MsmBatchArg ba_a, ba_x;
    ba_a.nb = ba_x.nb = nb;
    for (int b = 0; b < nb; b++) {
      ba_a.p[b] = a_share[b];
      ba_x.p[b] = ax_share[b];
    }
    // Order of the accumulate kernels.  Each accumulate of a batch fills the chip by itself (two or more rounds of
    // waves), so running them ONE AFTER THE OTHER loses nothing -- the sorts, finalizes and reductions of the other MSMs
    // run beside it -- and spares them each other: issued together, four kernels' waves interleave on every SIMD, the
    // 256-register G2 waves are crowded out by the 168-register G1 waves (which fit any slot a G1 wave frees) and V's
    // accumulate ends alone long after the others with its reduction tail on an empty chip
    // (profiles/r03_b8_timeline_*.txt).  Letters of the chain in order, V S W U (S = the S + H launch); "-" = no chain.
    // (a small batch does not fill the chip with one accumulate: its kernels run side by side as in a single proof)
    const std::string order = nb >= 4 ? "VSWU" : "-";
    MsmGate gates[4];                                  // V, S, W, U
    {
      int prev = -1;
      for (char ch : order) {
        const int id = ch == 'V' ? 0 : ch == 'S' ? 1 : ch == 'W' ? 2 : ch == 'U' ? 3 : -1;
        if (id < 0 || gates[id].signal_ev) continue;
        B.acc_flag[id].store(0, std::memory_order_relaxed);
        gates[id].signal_ev = B.ev_acc[id];
        gates[id].signal_flag = &B.acc_flag[id];
        if (prev >= 0) {
          gates[id].wait_ev = B.ev_acc[prev];
          gates[id].wait_flag = &B.acc_flag[prev];
        }
        prev = id;
      }
    }
    const MsmGate gate_v = gates[0], gate_s = gates[1], gate_w = gates[2], gate_u = gates[3];
    std::atomic<int>* aflag = B.acc_flag;
    // ---- the witness MSMs: V (G2) first on its high-priority stream, S + H as one launch over both base vectors, W
    B.fut.push_back(pool_->submit([=]() {
      (void)hipSetDevice(dev);
      int rc2 = msm_.template launch_t<Fq2_>(this, crs->v_d, nullptr, (size_t)n * crs->len_a, cf, crs->len_a, streams_[2],
                                            ws0 + 3, &BJ->pV, nullptr, gate_v, &ba_a);
      aflag[0].store(1, std::memory_order_release);        // also when the launch failed early (waiters must not hang)
      std::vector<P2> res((size_t)nb);
      if (!rc2) rc2 = msm_fold_batch<Fq2_>(this, BJ->pV, res.data(), 1);
      BJ->rc[1] = rc2;
      if (!rc2)
        for (int b = 0; b < nb; b++) BJ->pj[b]->V0 = res[b];
    }));
    const void* hd = all_r_zero ? nullptr : crs->h_d;
    B.fut.push_back(pool_->submit([=]() {
      (void)hipSetDevice(dev);
      int rc2 = msm_.template launch_t<Fq_>(this, crs->s_d, nullptr, (size_t)n * crs->len_a, cf, crs->len_a, streams_[0],
                                           ws0 + 1, &BJ->pSH, hd, gate_s, &ba_a);
      aflag[1].store(1, std::memory_order_release);
      std::vector<P1> res((size_t)nb * 2);
      if (!rc2) rc2 = msm_fold_batch<Fq_>(this, BJ->pSH, res.data(), hd ? 2 : 1);
      BJ->rc[0] = rc2;
      if (rc2) return;
      // s*S and r*H per proof (prove.rs:229-235, linearity): spread over the pool, this task takes proof 0
      std::vector<std::future<void>> sub;
      auto fin = [=, &res](int b) {
        ProveJob* J = BJ->pj[b].get();
        J->S = res[b];
        J->sS = host_scalar_mul<FrP, Fq_>(J->S, J->s);
        if (hd && !J->r_zero) {
          J->H = res[(size_t)nb + b];
          J->rH = host_scalar_mul<FrP, Fq_>(J->H, J->r);
        }
      };
      for (int b = 1; b < nb; b++) sub.push_back(pool_->submit([=]() { fin(b); }));
      fin(0);
      // (the sub-tasks reference this frame: the task runs queued work while it waits, HostPool::wait_helping -- a pool
      // whose workers all sit here with their sub-tasks queued behind them cannot starve)
      for (auto& f : sub)
        while (!pool_->wait_helping(f, deadline_from_now())) {}
    }));
    B.fut.push_back(pool_->submit([=]() {
      (void)hipSetDevice(dev);
      int rc2 = msm_.template launch_t<Fq_>(this, crs->w_d, nullptr, (size_t)n * crs->len_w, cf, crs->len_w, streams_[3],
                                           ws0 + 4, &BJ->pW, nullptr, gate_w, &ba_x);
      aflag[2].store(1, std::memory_order_release);
      std::vector<P1> res((size_t)nb);
      if (!rc2) rc2 = msm_fold_batch<Fq_>(this, BJ->pW, res.data(), 1);
      BJ->rc[2] = rc2;
      if (!rc2)
        for (int b = 0; b < nb; b++) BJ->pj[b]->W = res[b];
    }));
    // ---- host terms of every proof
    for (int b = 0; b < nb; b++) submit_host_terms(B.pj[b].get(), B.fut, full, first, count);
    B.gate_u = gate_u;
    *slot_out = slot;
    return ZK_OK;
  }
  // the five MSM totals of every proof of the batch, in-mask terms included (sums[b] = S, H, V, W, U)
  struct BatchSums {
    P1 S, H, W, U;
    P2 V;
  };
  int batch_join(BatchJobX& B, std::vector<BatchSums>& sums) {
    const int nb = B.nb;
    std::vector<P1> ures((size_t)nb);
    int rc = msm_fold_batch<Fq_>(this, B.pU, ures.data(), 1);
    if (rc) {
      if (B.pU.active) dump_batch("zk_groth16_batch_wait: the U-MSM chain did not finish within the deadline", B);
      Status keep = last;
      abort_batch(B);
      last = keep;
      return rc;
    }
    if (!drain_futs(B.fut)) {
      dump_batch("zk_groth16_batch_wait: pool tasks did not finish within the deadline", B);
      return wedge("zk_groth16_batch_wait: host tasks of the batch did not finish within the deadline (wait_deadline_ms); the slot is not reused");
    }
    B.active = false;
    for (int i = 0; i < 4; i++)
      if (B.rc[i]) return B.rc[i];
    sums.resize((size_t)nb);
    for (int b = 0; b < nb; b++) {
      ProveJob& j = *B.pj[b];
      if (j.err.code) return fail(j.err.code, j.err.msg);
      sums[b].S = xyzz_add_ni(j.S, j.in1[0]);
      sums[b].H = j.r_zero ? P1::identity() : xyzz_add_ni(j.H, j.in1[1]);
      sums[b].V = xyzz_add_ni(j.V0, j.in2);
      sums[b].W = xyzz_add_ni(j.W, j.in1[3]);
      sums[b].U = xyzz_add_ni(ures[b], j.in1[4]);
    }
    return ZK_OK;
  }
  int groth16_batch_wait(int handle, void* pi_a, void* pi_b, void* pi_c) override {
    if (handle < 0 || handle >= NBATCH || !bjobs_[handle].active) return fail(ZK_ERR_BAD_INPUT, "no batch in flight on this handle");
    BatchJobX& B = bjobs_[handle];
    if (!pi_a || !pi_b || !pi_c) {
      abort_batch(B);
      return fail(ZK_ERR_BAD_INPUT, "null pointer");
    }
    const int nb = B.nb;
    std::vector<BatchSums> sums;
    int rc = batch_join(B, sums);
    if (rc) return rc;
    for (int b = 0; b < nb; b++) {
      ProveJob& j = *B.pj[b];
      const P1 &S = sums[b].S, &H = sums[b].H, &W = sums[b].W, &U = sums[b].U;
      const P2& V = sums[b].V;
      rc = assemble_job(j, S, H, V, W, U, (char*)pi_a + (size_t)b * n * sizeof(Jacobian<Fq_>),
                        (char*)pi_b + (size_t)b * n * sizeof(Jacobian<Fq2_>), (char*)pi_c + (size_t)b * n * sizeof(Jacobian<Fq_>));
      if (rc) return rc;
    }
    return ZK_OK;
  }

  // zk_msm_batch: G::msm of ONE base vector against `nb` scalar vectors (the sort / accumulate / reduce chain of
  // msm.hpp runs once for the batch); out: nb Jacobian points (host).
  int msm_batch(int group, const void* bases, size_t len, const void* const* scalars, int nb, void* out,
                hipStream_t st) override {
    if (nb < 1 || nb > MSM_MAXB) return fail(ZK_ERR_BAD_INPUT, "batch size must be in 1.." + std::to_string(MSM_MAXB));
    if (!out || !scalars || (len && !bases)) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    MsmBatchArg ba;
    ba.nb = nb;
    for (int b = 0; b < nb; b++) {
      if (len && !scalars[b]) return fail(ZK_ERR_BAD_INPUT, "null pointer");
      ba.p[b] = scalars[b];
    }
    auto run = [&](auto tag) -> int {
      using Fld = decltype(tag);
      MsmPending pend;
      int rc = msm_.template launch_t<Fld>(this, bases, nullptr, len, nullptr, 1, st, 0, &pend, nullptr, MsmGate{}, &ba);
      if (rc) return rc;
      std::vector<XYZZ<Fld>> res((size_t)nb);
      rc = msm_fold_batch<Fld>(this, pend, res.data(), 1);
      if (rc) return rc;
      for (int b = 0; b < nb; b++) {
        Jacobian<Fld> j = xyzz_to_jacobian(res[b]);
        memcpy((char*)out + (size_t)b * sizeof(j), &j, sizeof(j));
      }
      return ZK_OK;
    };
    if (group == ZK_G1) return run(Fq_{});
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) return run(Fq2_{});
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

  // sum over the listed local parties of coef_p * (msm_p + in_mask_p): one rank's contribution to the king's
  // unpack2 + sum (dmsm/mod.rs:85-86) when the n parties are spread over several GPUs.
  int d_msm_local(int group, const void* bases, const void* scalars, size_t len, int first_party, int nparties,
                  const void* in_mask, void* out, hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (first_party < 0 || nparties <= 0 || first_party + nparties > n) return fail(ZK_ERR_BAD_INPUT, "bad party range");
    if (!out) return fail(ZK_ERR_BAD_INPUT, "null output");
    if (group == ZK_G1) {
      XYZZ<Fq> r;
      int rc = msm_.template d_msm_range_t<Fq>(this, bases, scalars, len, first_party, nparties, in_mask, &r, st);
      if (rc) return rc;
      Jacobian<Fq> j = xyzz_to_jacobian(r);
      memcpy(out, &j, sizeof(j));
      return ZK_OK;
    }
    if (group == ZK_G2 && Cfg::HAS_G2) {
      XYZZ<Fq2> r;
      int rc = msm_.template d_msm_range_t<Fq2>(this, bases, scalars, len, first_party, nparties, in_mask, &r, st);
      if (rc) return rc;
      Jacobian<Fq2> j = xyzz_to_jacobian(r);
      memcpy(out, &j, sizeof(j));
      return ZK_OK;
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }
  // MsmMask::sample (dmsm/mod.rs:21-47); scalar form: shares of (x_0..x_{l-1} | t random) times the generator
  template <class Fld>
  int msm_mask_sample_t(const void* gen_affine, uint64_t seed, void* in_mask, void* out_mask) {
    Affine<Fld> g;
    memcpy(&g, gen_affine, sizeof(g));
    XYZZ<Fld> gen = XYZZ<Fld>::from_affine(g);
    const int k = l + t;
    std::vector<Fr> sec_in(k), sec_out(k);
    Fr sum = Fr::zero();
    const RngSeed r0 = rs_host(seed), r1 = rs_host(seed ^ 0x1111ull), r2 = rs_host(seed ^ 0x2222ull);
    for (int i = 0; i < l; i++) {
      sec_in[i] = rand_fp<FrP>(r0, (uint64_t)i);
      sum = sum + sec_in[i];
    }
    for (int i = 0; i < l; i++) sec_out[i] = sum.neg();
    for (int i = 0; i < t; i++) {
      sec_in[l + i] = rand_fp<FrP>(r1, (uint64_t)i);
      sec_out[l + i] = rand_fp<FrP>(r2, (uint64_t)i);
    }
    Jacobian<Fld>* oi = (Jacobian<Fld>*)in_mask;
    Jacobian<Fld>* oo = (Jacobian<Fld>*)out_mask;
    for (int p = 0; p < n; p++) {
      Fr si = Fr::zero(), so = Fr::zero();
      for (int j = 0; j < k; j++) {
        si = si + pmat_host_[(size_t)p * k + j] * sec_in[j];
        so = so + pmat_host_[(size_t)p * k + j] * sec_out[j];
      }
      oi[p] = xyzz_to_jacobian(host_scalar_mul<FrP, Fld>(gen, si));
      oo[p] = xyzz_to_jacobian(host_scalar_mul<FrP, Fld>(gen, so));
    }
    return ZK_OK;
  }
  int msm_mask_sample(int group, const void* gen_affine, uint64_t seed, void* in_mask, void* out_mask) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (!gen_affine || !in_mask || !out_mask) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (group == ZK_G1) return msm_mask_sample_t<Fq>(gen_affine, seed, in_mask, out_mask);
    if (group == ZK_G2 && Cfg::HAS_G2) return msm_mask_sample_t<Fq2>(gen_affine, seed, in_mask, out_mask);
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

  // ---------------------------------------------------------------- libsnark_h (ext_wit.rs:14-102)
  // 3 x d_ifft with the coset shift g = F::GENERATOR (rearranged) -> 3 x d_fft (rearranged) -> (a*b - c) / Z(g) ->
  // d_ifft with g^-1.  Seven masks (or NULL arrays for FftMask::zero).
  int libsnark_h(const void* qa, const void* qb, const void* qc, int log_m, const void* const* fft_in,
                 const void* const* fft_out, uint64_t seed, void* h, hipStream_t st) override {
    if (log_m < ilog2(l) || log_m > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (!qa || !qb || !qc || !h) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)n * Lc;
    DevBuf& hwork_ = ws(st)->hwork;
    ZK_HIP(hwork_.ensure(6 * per * sizeof(Fr)));
    Fr* W0 = (Fr*)hwork_.p;
    Fr* W1 = W0 + 3 * per;
    const void* q[3] = {qa, qb, qc};
    for (int k = 0; k < 3; k++) ZK_HIP(hipMemcpyAsync(W0 + k * per, q[k], per * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    Fr g = generator();
    auto mi = [&](int k) { return fft_in ? fft_in[k] : nullptr; };
    auto mo = [&](int k) { return fft_out ? fft_out[k] : nullptr; };
    int rc;
    for (int k = 0; k < 3; k++) {
      rc = d_fft(W0 + k * per, mi(k), mo(k), 1, log_m, 1, &g, seed + k, W1 + k * per, st);
      if (rc) return rc;
      rc = d_fft(W1 + k * per, mi(3 + k), mo(3 + k), 1, log_m, 0, nullptr, seed + 3 + k, W0 + k * per, st);
      if (rc) return rc;
    }
    rc = vec_mul_sub(W1, W0, W0 + per, W0 + 2 * per, per, st);
    if (rc) return rc;
    // 1 / Z(g), Z(x) = x^m - 1  (ext_wit.rs:78-81)
    Fr zinv = (g.pow_u64((uint64_t)1 << log_m) - Fr::one()).inverse();
    rc = vec_scale(W1, &zinv, per, st);
    if (rc) return rc;
    Fr ginv = g.inverse();
    return d_fft(W1, mi(6), mo(6), 0, log_m, 1, &ginv, seed + 6, h, st);
  }

