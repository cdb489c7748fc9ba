// Engine<Cfg>: per-curve implementation of IEngine (included by exactly one .hip file per curve).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>

#include "engine.hpp"
#include "groth16.hpp"
#include "qap.hpp"
#include "msm.hpp"
#include "hostpool.hpp"
#include "points.hpp"
#include "dpp.hpp"

namespace zk {

template <class F>
__global__ void rand_fill_kernel(F* __restrict__ out, RngSeed seed, size_t count, uint32_t L, uint32_t transpose_lc_log) {
  // out[i] = rand(seed, i); with transpose: value index v = k*L + s is stored at [s][k] (k < 2^log)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  using P = typename F::Params;
  F v = rand_fp<P>(seed, i);
  size_t o = i;
  if (transpose_lc_log != 0xffffffffu) {
    size_t k = i / L, s = i % L;
    o = (s << transpose_lc_log) + k;
  }
  store_elem(out + o, v);
}

template <class F>
__global__ void vec_neg_kernel(F* __restrict__ out, const F* __restrict__ in, size_t len) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) store_elem(out + i, load_elem(in + i).neg());
}

template <class Cfg>
class Engine : public IEngine {
 public:
  using FrP = typename Cfg::FrP;
  using Fr = Fp<FrP>;
  static constexpr int MAXL = 8;

  Engine(int l_, int device_) {
    l = l_;
    t = l_;
    n = 4 * l_;
    device = device_;
  }
  ~Engine() override {
    for (auto& j : jobs_)
      if (j.active) abort_job(j);
    for (auto& b : bjobs_) {
      if (b.active) abort_batch(b);
      if (b.ev_in) (void)hipEventDestroy(b.ev_in);
      for (hipEvent_t e_ : b.ev_acc)
        if (e_) (void)hipEventDestroy(e_);
      if (b.own_streams)
        for (hipStream_t s_ : b.st)
          if (s_) (void)hipStreamDestroy(s_);
    }
    pool_.reset();                                   // joins the host workers before anything they use goes away
    TableRegistry::inst().forget_owner(this);
    for (auto& kv : gentabs_) (void)hipFree(kv.second);
    for (auto& kv : gtabs_) {
      (void)hipFree(kv.second.tab);
      (void)hipFree(kv.second.step);
    }
    for (auto& kv : base_tables_) (void)hipFree(kv.second);
    for (auto& kv : umats_) (void)hipFree(kv.second);
    for (auto& kv : sizeinv_) (void)hipFree(kv.second);
    if (pmat_) (void)hipFree(pmat_);
    for (auto& kv : pcoef_) (void)hipFree(kv.second);
    if (pack2_) (void)hipFree(pack2_);
    if (ident_) (void)hipFree(ident_);
    if (err_flag_) (void)hipFree(err_flag_);
    if (rng_key_d_) (void)hipFree(rng_key_d_);
    if (u2c_) (void)hipFree(u2c_);
    for (auto& kv : ucanon_) (void)hipFree(kv.second);
    if (pmc_) (void)hipFree(pmc_);
  }

  size_t fr_bytes() const override { return sizeof(Fr); }
  size_t fq_bytes() const override { return sizeof(Fp<typename Cfg::FqP>); }

  int init() {
    ZK_HIP(hipSetDevice(device));
    if (l != 1 && l != 2 && l != 4 && l != 8) return fail(ZK_ERR_BAD_INPUT, "packing factor l must be 1, 2, 4 or 8");
    if (ilog2(n) > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "domain too large");
    build_matrices();
    ZK_HIP(hipMalloc(&err_flag_, sizeof(int)));
    // share randomness: ChaCha20 keyed from the operating system's generator (prng.hpp); the context option "rng_replay"
    // selects the documented replayable stream the parity tests compare shares with
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(rng_key_h_, 1, sizeof(rng_key_h_), f) != sizeof(rng_key_h_)) {
      if (f) fclose(f);
      return fail(ZK_ERR_GENERIC, "cannot read /dev/urandom for the share-randomness key");
    }
    fclose(f);
    ZK_HIP(hipMalloc((void**)&rng_key_d_, sizeof(rng_key_h_)));
    ZK_HIP(hipMemcpy(rng_key_d_, rng_key_h_, sizeof(rng_key_h_), hipMemcpyHostToDevice));
    rng_replay_ = false;        // only zk_ctx_set_option("rng_replay", 1) turns the replay stream on (never the environment)
    return ZK_OK;
  }
  // Replay mode trades the hiding property for reproducible shares (two packs with the same seed reuse their random
  // points): it is a test / parity setting, so say so once per process wherever it gets switched on.
  static void warn_replay() {
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
      fprintf(stderr, "zksaas: share randomness is in REPLAY mode (option rng_replay): shares are "
                      "reproducible from the caller's seeds and NOT hiding -- tests and parity runs only\n");
  }
  // the randomness of one launch: `span` consecutive stream ids (batch items).  Replay mode: the caller's seed;
  // otherwise the seed is IGNORED and the launch gets fresh nonces of this context's ChaCha20 stream.
  RngSeed rs(uint64_t seed, uint64_t span = 1) {
    if (rng_replay_) return RngSeed{seed, nullptr};
    return RngSeed{rng_nonce_.fetch_add(span), rng_key_d_};
  }
  RngSeed rs_host(uint64_t seed) {
    if (rng_replay_) return RngSeed{seed, nullptr};
    return RngSeed{rng_nonce_.fetch_add(1), rng_key_h_};
  }
  uint32_t rng_key_h_[8] = {0};
  uint32_t* rng_key_d_ = nullptr;
  std::atomic<uint64_t> rng_nonce_{1};
  bool rng_replay_ = false;

  // ---------------------------------------------------------------- field helpers (host)
  static Fr root_of_unity(int log_size) {
    Fr r = Fr::from_limbs(FrP::TWO_ADIC_ROOT);
    for (int i = log_size; i < FrP::TWO_ADICITY; i++) r = r.sqr();
    return r;
  }
  static Fr generator() { return Fr::from_limbs(FrP::GENERATOR); }

  // ---------------------------------------------------------------- PSS matrices
  // Points: share x_p = w_n^p; secret y_j = g w_{l+t}^j; secret2 z_j = g w_{2(l+t)}^j
  // (secret-sharing/src/pss.rs:44-52).
  void points(std::vector<Fr>& x, std::vector<Fr>& y, std::vector<Fr>& z) const {
    Fr wn = root_of_unity(ilog2(n)), ws = root_of_unity(ilog2(l + t)), w2 = root_of_unity(ilog2(2 * (l + t)));
    Fr g = generator();
    x.resize(n);
    y.resize(l + t);
    z.resize(2 * (l + t));
    Fr c = Fr::one();
    for (int i = 0; i < n; i++) x[i] = c, c = c * wn;
    c = g;
    for (int i = 0; i < l + t; i++) y[i] = c, c = c * ws;
    c = g;
    for (int i = 0; i < 2 * (l + t); i++) z[i] = c, c = c * w2;
  }

  int upload(const std::vector<Fr>& h, Fr** d) {
    ZK_HIP(hipMalloc((void**)d, h.size() * sizeof(Fr)));
    ZK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice));
    return ZK_OK;
  }

  void build_matrices() {
    std::vector<Fr> x, y, z;
    points(x, y, z);
    int k = l + t;
    // pack: P[p][j] = prod_{i != j} (x_p - y_i) / (y_j - y_i)       (pss.rs:90-122)
    std::vector<Fr> Pm((size_t)n * k);
    for (int p = 0; p < n; p++)
      for (int j = 0; j < k; j++) {
        Fr num = Fr::one(), den = Fr::one();
        for (int i = 0; i < k; i++)
          if (i != j) {
            num = num * (x[p] - y[i]);
            den = den * (y[j] - y[i]);
          }
        Pm[(size_t)p * k + j] = num * den.inverse();
      }
    (void)upload(Pm, &pmat_);
    pmat_host_ = Pm;
    if (l == 2) {
      // constants of the FFT-structured pack (pss.hpp pack_chunk): secret domain g*H_4, share domain H_8
      Fr w4 = root_of_unity(2), w8 = root_of_unity(3), g = generator();
      PackL2<Fr> k;
      k.w4inv = w4.inverse();
      Fr ginv = g.inverse(), quarter = Fr::from_u64(4).inverse(), cur = quarter;
      for (int i = 0; i < 4; i++) k.kc[i] = cur, cur = cur * ginv;
      k.w8 = w8;
      k.w4 = w4;
      k.w8_3 = w8 * w8 * w8;
      if (hipMalloc((void**)&pack2_, sizeof(k)) == hipSuccess)
        (void)hipMemcpy(pack2_, &k, sizeof(k), hipMemcpyHostToDevice);
    }
    // unpack (pss.rs:125-138): IFFT_n, truncate to l+t coefficients, evaluate at y_k:
    //   U1[k][p] = 1/n sum_{d < l+t} (y_k / x_p)^d
    // unpack2 (pss.rs:141-166): U2[k][p] = 1/n sum_{d < n} (z_{2k} / x_p)^d
    Fr ninv = Fr::from_u64((uint64_t)n).inverse();
    std::vector<Fr> U1((size_t)l * n), U2((size_t)l * n);
    for (int kk = 0; kk < l; kk++)
      for (int p = 0; p < n; p++) {
        Fr xi = x[p].inverse();
        Fr r1 = y[kk] * xi, r2 = z[2 * kk] * xi;
        Fr a1 = Fr::zero(), a2 = Fr::zero(), c1 = Fr::one(), c2 = Fr::one();
        for (int d = 0; d < n; d++) {
          if (d < k) a1 = a1 + c1;
          a2 = a2 + c2;
          c1 = c1 * r1;
          c2 = c2 * r2;
        }
        U1[(size_t)kk * n + p] = a1 * ninv;
        U2[(size_t)kk * n + p] = a2 * ninv;
      }
    Fr* d = nullptr;
    (void)upload(U1, &d);
    umats_[key_u(0xffffffffu, 1)] = d;
    (void)upload(U2, &d);
    umats_[key_u(full_mask(), 2)] = d;
    // d_msm: king output = sum_k unpack2(shares)[k] = sum_p (sum_k U2[k][p]) * share_p  (dmsm/mod.rs:85-86)
    std::vector<Fr> coef(n, Fr::zero());
    for (int p = 0; p < n; p++)
      for (int kk = 0; kk < l; kk++) coef[p] = coef[p] + U2[(size_t)kk * n + p];
    (void)msm_.set_coefs(this, coef);
    std::vector<Fr> I((size_t)l * l, Fr::zero());
    for (int i = 0; i < l; i++) I[(size_t)i * l + i] = Fr::one();
    (void)upload(I, &ident_);
  }

  uint32_t full_mask() const { return n >= 32 ? 0xffffffffu : ((1u << n) - 1); }
  static uint64_t key_u(uint32_t mask, int kind) { return ((uint64_t)kind << 32) | mask; }

  // unpack_missing_shares (pss.rs:210-221): unpack2 when all n present, else lagrange_unpack (:170-205):
  //   U_S[k][i] = prod_{j in S, j != i} (z_{2k} - x_j) / (x_i - x_j)
  int umat_for(const uint32_t* parties, int np, const Fr** out) {
    if (np <= 0 || np > n) return fail(ZK_ERR_BAD_INPUT, "bad party count");
    uint32_t mask = 0;
    for (int i = 0; i < np; i++) {
      uint32_t p = parties ? parties[i] : (uint32_t)i;
      if (p >= (uint32_t)n || (i > 0 && parties && parties[i] <= parties[i - 1]))
        return fail(ZK_ERR_BAD_INPUT, "party ids must be ascending and < n");
      mask |= 1u << p;
    }
    if (np < n && np <= 2 * (t + l - 1))
      return fail(ZK_ERR_PROTOCOL, "Not enough shares to reconstruct", 0);   // pss.rs:183-186
    std::lock_guard<std::mutex> g(mu_);
    auto it = umats_.find(key_u(mask, 2));
    if (it != umats_.end()) {
      *out = it->second;
      return ZK_OK;
    }
    std::vector<Fr> x, y, z;
    points(x, y, z);
    std::vector<Fr> U((size_t)l * np);
    for (int kk = 0; kk < l; kk++)
      for (int i = 0; i < np; i++) {
        Fr num = Fr::one(), den = Fr::one();
        for (int j = 0; j < np; j++)
          if (j != i) {
            num = num * (z[2 * kk] - x[parties[j]]);
            den = den * (x[parties[i]] - x[parties[j]]);
          }
        U[(size_t)kk * np + i] = num * den.inverse();
      }
    Fr* d = nullptr;
    int rc = upload(U, &d);
    if (rc) return rc;
    umats_[key_u(mask, 2)] = d;
    *out = d;
    return ZK_OK;
  }

  // coef_i = sum_k U_S[k][i] for a party subset S (the king's unpack_missing_shares + sum of d_msm as one
  // linear form over the surviving parties, dmsm/mod.rs:85-86 with pss.rs:170-221)
  int coefs_for(const uint32_t* parties, int np, std::vector<Fr>& coef) {
    if (np <= 0 || np > n || !parties) return fail(ZK_ERR_BAD_INPUT, "bad party list");
    for (int i = 0; i < np; i++)
      if (parties[i] >= (uint32_t)n || (i > 0 && parties[i] <= parties[i - 1]))
        return fail(ZK_ERR_BAD_INPUT, "party ids must be ascending and < n");
    if (np < n && np <= 2 * (t + l - 1)) return fail(ZK_ERR_PROTOCOL, "Not enough shares to reconstruct", 0);
    std::vector<Fr> x, y, z;
    points(x, y, z);
    coef.assign(np, Fr::zero());
    if (np == n) {
      coef.assign(msm_.coef_h_.begin(), msm_.coef_h_.begin() + n);
      return ZK_OK;
    }
    for (int kk = 0; kk < l; kk++)
      for (int i = 0; i < np; i++) {
        Fr num = Fr::one(), den = Fr::one();
        for (int j = 0; j < np; j++)
          if (j != i) {
            num = num * (z[2 * kk] - x[parties[j]]);
            den = den * (x[parties[i]] - x[parties[j]]);
          }
        coef[i] = coef[i] + num * den.inverse();
      }
    return ZK_OK;
  }

  // ---------------------------------------------------------------- cached tables
  // gentab(log_m, inverse)[e] = gen^e, e in [0, m]; gen = w_m or w_m^-1.
  int gentab(int log_m, int inverse, hipStream_t st, const Fr** out) {
    if (log_m > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "domain too large for this field");
    std::lock_guard<std::mutex> g(mu_);
    int key = log_m * 2 + (inverse ? 1 : 0);
    auto it = gentabs_.find(key);
    if (it != gentabs_.end()) {
      *out = it->second;
      return ZK_OK;
    }
    Fr w = root_of_unity(log_m);
    if (inverse) w = w.inverse();
    size_t count = ((size_t)1 << log_m) + 1;
    Fr* d = nullptr;
    ZK_HIP(hipMalloc((void**)&d, count * sizeof(Fr)));
    powers_kernel<Fr><<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st>>>(d, w, count, Fr::one());
    ZK_HIP(hipGetLastError());
    gentabs_[key] = d;
    *out = d;
    return ZK_OK;
  }

  // device copy of 1/m
  int size_inv_dev(int log_m, hipStream_t st, const Fr** out) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = sizeinv_.find(log_m);
    if (it != sizeinv_.end()) {
      *out = it->second;
      return ZK_OK;
    }
    Fr c = Fr::from_u64((uint64_t)1 << log_m).inverse();
    Fr* d = nullptr;
    ZK_HIP(hipMalloc((void**)&d, sizeof(Fr)));
    ZK_HIP(hipMemcpy(d, &c, sizeof(Fr), hipMemcpyHostToDevice));
    sizeinv_[log_m] = d;
    *out = d;
    return ZK_OK;
  }

  struct GTab {
    Fr* tab;
    Fr* step;
  };
  // gtab[e] = c * g^e for e in [0, Lc]; step[e] = g^(Lc*e), e < l.  c = 1/m when scale.
  int gtab(int log_m, const Fr& g, bool scale, hipStream_t st, GTab* out) {
    std::string key((const char*)g.v, sizeof(Fr));
    key.push_back((char)log_m);
    key.push_back(scale ? 1 : 0);
    std::lock_guard<std::mutex> lk(mu_);
    auto it = gtabs_.find(key);
    if (it != gtabs_.end()) {
      *out = it->second;
      return ZK_OK;
    }
    size_t Lc = ((size_t)1 << log_m) / l;
    Fr c = Fr::one();
    if (scale) c = Fr::from_u64((uint64_t)1 << log_m).inverse();
    GTab gt{};
    ZK_HIP(hipMalloc((void**)&gt.tab, (Lc + 1) * sizeof(Fr)));
    powers_kernel<Fr><<<dim3((unsigned)((Lc + 1 + 255) / 256)), dim3(256), 0, st>>>(gt.tab, g, Lc + 1, c);
    ZK_HIP(hipGetLastError());
    std::vector<Fr> step(l);
    Fr gl = g.pow_u64(Lc), cur = Fr::one();
    for (int e = 0; e < l; e++) step[e] = cur, cur = cur * gl;
    ZK_HIP(hipMalloc((void**)&gt.step, l * sizeof(Fr)));
    ZK_HIP(hipMemcpyAsync(gt.step, step.data(), l * sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipStreamSynchronize(st));   // `step` is a stack vector
    gtabs_[key] = gt;
    *out = gt;
    return ZK_OK;
  }

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
    // the local stages go out of place into a context-owned vector (the first pass reads `shares`), the king reads
    // that and writes the destination: no copy back, and `shares` is left untouched when `out` is given
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
  template <int L>
  int dpp_l(const Fr* num, const Fr* den, int np, size_t len, const Fr* U, const Fr* Ufull, const Fr* in_mask,
            const Fr* out_mask, const RngSeed& seed, Fr* out, hipStream_t st) {
    const size_t m = len * L, ntiles = (m + DPP_TILE - 1) / DPP_TILE;
    ZK_HIP(scratch_.ensure((m + 3 * ntiles) * sizeof(Fr)));
    Fr* y = (Fr*)scratch_.p;
    Fr* tile_n = y + m;
    Fr* tile_d = tile_n + ntiles;
    Fr* ctile = tile_d + ntiles;
    constexpr size_t lds = (size_t)(sizeof(Fr) / 16) * 16 * DPP_LDS_SLOTS;
    if (!dpp_attr_set_) {
      ZK_HIP(hipFuncSetAttribute((const void*)dpp_tile_kernel<FrP, L>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      dpp_attr_set_ = true;
    }
    ZK_HIP(hipMemsetAsync(err_flag_, 0, sizeof(int), st));
    {
      ProfScope ps_(prof, PROF_DPP_TILE, st, (double)m);
      dpp_tile_kernel<FrP, L><<<dim3((unsigned)ntiles), dim3(DPP_THREADS), lds, st>>>(num, den, np, len, len, U, y,
                                                                                       tile_n, tile_d);
    }
    ZK_HIP(hipGetLastError());
    {
      ProfScope ps_(prof, PROF_DPP_CARRY, st, (double)ntiles);
      dpp_carry_kernel<Fr><<<dim3(1), dim3(DPP_CARRY_THREADS), 0, st>>>(tile_n, tile_d, ntiles, ctile, err_flag_);
    }
    ZK_HIP(hipGetLastError());
    {
      ProfScope ps_(prof, PROF_DPP_FINISH, st, (double)m);
      dpp_finish_kernel<FrP, L><<<dim3((unsigned)((len + KING_THREADS - 1) / KING_THREADS)), dim3(KING_THREADS), 0, st>>>(
          y, ctile, len, in_mask, out_mask, Ufull, pmat_, pack2_, seed, out);
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
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
      default: rc = dpp_l<8>(num, den, np, len, U, Ufull, in_mask, out_mask, r, out, st); break;
    }
    if (rc) return rc;
    int herr = 0;
    ZK_HIP(hipMemcpyAsync(&herr, err_flag_, sizeof(int), hipMemcpyDeviceToHost, st));
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
    fixed_base_mul_kernel<FrP, Fld><<<dim3((unsigned)((len + 127) / 128)), dim3(128), 0, st>>>(
        (const Fr*)scalars, len, table, nwin, (Affine<Fld>*)out_affine);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int base_mul(int group, const void* base_affine, const void* scalars, size_t len, void* out_affine,
               hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (group == ZK_G1) return base_mul_t<Fq>(base_affine, scalars, len, out_affine, st);
    if (group == ZK_G2) {
      if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
      return base_mul_t<Fq2>(base_affine, scalars, len, out_affine, st);
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
    if (nv == 2) pss_pack_points_kernel<FrP, Fld, 2><<<grid, block, 0, st>>>(in, nchunks, n, coef, out);
    else if (nv == 4) pss_pack_points_kernel<FrP, Fld, 4><<<grid, block, 0, st>>>(in, nchunks, n, coef, out);
    else return fail(ZK_ERR_BAD_INPUT, "point packing is built for 2 or 4 points per chunk (l = 2, or det_pack at l = 4)");
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int pss_pack_points(int group, const void* points, size_t nchunks, int nv, void* shares, hipStream_t st) override {
    using Fq = Fp<typename Cfg::FqP>;
    using Fq2 = Fp2<typename Cfg::FqP>;
    if (group == ZK_G1) return pack_points_t<Fq>(points, nchunks, nv, shares, st);
    if (group == ZK_G2 && Cfg::HAS_G2) return pack_points_t<Fq2>(points, nchunks, nv, shares, st);
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

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
    return circom_h_ws(qa, qb, qc, log_m, mk, seed, h, hwork_, st);
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
  using Fq_ = Fp<typename Cfg::FqP>;
  using Fq2_ = Fp2<typename Cfg::FqP>;
  using P1 = XYZZ<Fq_>;
  using P2 = XYZZ<Fq2_>;
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
  void drain(ProveJob& j) {
    for (auto& f : j.fut)
      if (f.valid()) f.wait();
    j.fut.clear();
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
    int rc = ensure_streams();
    if (rc) return rc;
    init_job(j, crs, mk, r, s, full, first, count);
    const size_t Lc = ((size_t)1 << log_m) / l;
    const int dev = device;
    const int ws0 = j.slot * 6;
    // every internal stream is ordered after the work already queued on the caller's stream (the shares may still
    // be in flight there: found by tools/c5_bls381.py, where the a_share pack kernel of a 2^22 witness was still
    // running when the S/H/V MSMs started reading it)
    ZK_HIP(hipEventRecord(ev_in_[j.slot], st));
    for (hipStream_t is : streams_) ZK_HIP(hipStreamWaitEvent(is, ev_in_[j.slot], 0));
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
    submit_host_terms(J, j.fut, full, first, count);
    // ---- circom_h and the U-MSM that depends on it form a long dependent chain: high-priority internal stream.
    // At the SHA-256 size, holding the other MSM streams (or only their accumulate launches) back until circom_h has
    // finished was measured and rejected: 8.6-8.8 ms per proof against 6.9 ms when everything is issued at once.
    if (full && !hu_done) {
      hipStream_t hs = streams_[5];
      hipError_t he = j.hshare.ensure((size_t)n * Lc * sizeof(Fr));
      if (he != hipSuccess) return hip_fail(he, "h share buffer");
      rc = circom_h_ws(qa, qb, qc, log_m, mk, seed, j.hshare.p, j.hwork, hs);
      if (rc) return rc;
      rc = msm_.template launch_t<Fq_>(this, crs->u_d, j.hshare.p, (size_t)n * crs->len_u, msm_.coef_d_, crs->len_u, hs,
                                      ws0 + 0, &j.pU);
      if (rc) return rc;
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
    drain(j);
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

  int prove_end(ProveJob& j, void* pi_a, void* pi_b, void* pi_c) {
    P1 S, H, W, U;
    P2 V;
    int rc = prove_join(j, &S, &H, &V, &W, &U);
    if (rc) return rc;
    return assemble_job(j, S, H, V, W, U, pi_a, pi_b, pi_c);
  }
  // the n parties' (A, B, C) shares of one proof from its five MSM totals (in-mask terms included) and the host terms
  int assemble_job(ProveJob& j, const P1& S, const P1& H, const P2& V, const P1& W, const P1& U, void* pi_a, void* pi_b,
                   void* pi_c) {
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
        oc[p] = oc[0];
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
      oc[p] = xyzz_to_jacobian(C);
    }
    return ZK_OK;
  }

  int check_prove_args(const zk_crs_share* crs, const void* r_, const void* s_, int log_m) {
    if (!Cfg::HAS_G2) return fail(ZK_ERR_BAD_INPUT, "G2 is not available for this curve");
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
    int rc = groth16_prove_async(crs, qa, qb, qc, a_share, ax_share, r_, s_, log_m, mk, seed, st, &h);
    if (rc) return rc;
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
    drain(j);
    MsmPending* ps[4] = {&j.pS, &j.pV0, &j.pW, &j.pU};
    for (MsmPending* p : ps)
      if (p->active) {
        (void)hipEventSynchronize(p->slot->ev);
        p->active = false;
        p->tab.reset();
        p->tab2.reset();
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

  void abort_batch(BatchJob& B) {
    for (auto& f : B.fut)
      if (f.valid()) f.wait();
    B.fut.clear();
    MsmPending* ps[4] = {&B.pSH, &B.pV, &B.pW, &B.pU};
    for (MsmPending* p : ps)
      if (p->active) {
        (void)hipEventSynchronize(p->slot->ev);
        p->active = false;
        p->tab.reset();
        p->tab2.reset();
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
      for (auto& f : sub) f.get();
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
      Status keep = last;
      abort_batch(B);
      last = keep;
      return rc;
    }
    for (auto& f : B.fut)
      if (f.valid()) f.wait();
    B.fut.clear();
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

  // ---------------------------------------------------------------- libsnark_h (ext_wit.rs:14-102)
  // 3 x d_ifft with the coset shift g = F::GENERATOR (rearranged) -> 3 x d_fft (rearranged) -> (a*b - c) / Z(g) ->
  // d_ifft with g^-1.  Seven masks (or NULL arrays for FftMask::zero).
  int libsnark_h(const void* qa, const void* qb, const void* qc, int log_m, const void* const* fft_in,
                 const void* const* fft_out, uint64_t seed, void* h, hipStream_t st) override {
    if (log_m < ilog2(l) || log_m > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (!qa || !qb || !qc || !h) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)n * Lc;
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

  // ---------------------------------------------------------------- per-rank collective forms (net.hpp)
  // Every rank calls these collectively with the rows of ITS k = n / world parties ([k][len] buffers, masks
  // likewise); rank 0 hosts the king.  A call first enters the round on the control plane (net.enter): ranks that do
  // not show up within the timeout are left out and the king goes through lagrange_unpack (pss.rs:170-221) like
  // ser_net.rs:57-94 does with `Partial` results.
  DevBuf dist_in_[NET_NSID], dist_out_[NET_NSID], dist_coef_[NET_NSID];
  DevBuf dist_w0_, dist_w1_, dist_h_;

  int net_err(Net* net, int rc) {
    if (rc == ZK_OK) return rc;
    return fail(rc, "net: " + net->err, net->err_party);
  }
  // Option "dist_deadline": the reference's collectives return Err after their timeout (mpc-net/src/ser_net.rs:122-125);
  // here a zk_dist_* call has only ENQUEUED its data-plane work when it returns, and a hung RCCL collective would
  // surface at the caller's next stream synchronisation, without a deadline.  With the option every zk_dist_* entry
  // point waits for the channels' streams before returning, with the net's timeout as the deadline; on expiry the
  // communicators are aborted, the call fails with ZK_ERR_PROTOCOL and the net refuses every later round (the process
  // then opens a new net, or hands over to a fresh child process).
  bool dist_deadline_ = false;
  int dist_finish(Net* net, int rc) override {
    if (rc != ZK_OK || !dist_deadline_) return rc;
    for (int sid = 0; sid < NET_NSID; sid++) {
      const int r = net->sync_deadline(sid);
      if (r) return net_err(net, r);
    }
    return ZK_OK;
  }
  // parties of the ranks in `mask`, ascending (the row order the king sees, net.hpp)
  std::vector<uint32_t> parties_of(const Net* net, uint32_t mask) const {
    std::vector<uint32_t> ps;
    const int k = net->parties_per_rank();
    for (int r = 0; r < net->world; r++)
      if (mask & (1u << r))
        for (int p = 0; p < k; p++) ps.push_back((uint32_t)net->party(r, p));
    std::sort(ps.begin(), ps.end());
    return ps;
  }
  // Index of this rank's first party in the d_msm coefficient table.  The parties of a contiguous map are a range of
  // the table; those of a general party_to_rank map are copied to the window behind its n entries (msm.hpp set_window),
  // so that everything taking (first, count) works on either.
  int local_window(Net* net, int* first) {
    if (net->contiguous()) {
      *first = net->first_party(net->rank);
      return ZK_OK;
    }
    const int k = net->parties_per_rank();
    std::vector<int> mine((size_t)k);
    for (int i = 0; i < k; i++) mine[i] = net->party(net->rank, i);
    for (int i = 0; i < NJOBS; i++)
      if (jobs_[i].active && mine != msm_.window_) return fail(ZK_ERR_BAD_INPUT, "a proof over another party map is in flight");
    int rc = msm_.set_window(this, mine);
    if (rc) return rc;
    *first = n;
    return ZK_OK;
  }
  hipStream_t net_stream(Net* net, int sid, hipStream_t st) { return net->stream(sid) ? net->stream(sid) : st; }

  // one king round on channel sid: gather the local rows -> king step on rank 0 -> scatter.  `king` is called on rank 0
  // with (in [np][len], parties, np, out [n][len], stream).
  // `out_mask` (optional, this rank's rows): added to the scattered result.  With the LOCAL transport (one rank holds all
  // parties) the king reads the caller's rows directly and the result comes back through ONE pass that adds the mask,
  // instead of a gather copy, a scatter copy and an addition (d_fft 2^20 on one GPU: 0.96 -> 0.87 ms).
  template <class KingFn>
  int king_round(Net* net, int sid, uint32_t mask, Fr* local, size_t len, KingFn king, const Fr* out_mask = nullptr,
                 const Fr* king_in = nullptr, bool in_place_ok = false) {
    const int k = net->parties_per_rank();
    const size_t bytes = (size_t)k * len * sizeof(Fr);
    hipStream_t ks = net_stream(net, sid, nullptr);
    if (net->transport == ZK_NET_LOCAL) {
      std::vector<uint32_t> ps = parties_of(net, mask);
      // one rank holds all parties: no gather / scatter.  The king writes the caller's rows directly (out-mask added in
      // its store) when its input lives elsewhere (king_in: d_fft's local stages went out of place) or when it only
      // touches its own column (in_place_ok: deg_red); otherwise through a scratch vector and one pass back.
      if (king_in || in_place_ok) return king(king_in ? king_in : local, ps.data(), (int)ps.size(), local, out_mask, ks);
      ZK_HIP(dist_out_[sid].ensure((size_t)n * len * sizeof(Fr)));
      Fr* fout = (Fr*)dist_out_[sid].p;
      int rc = king(local, ps.data(), (int)ps.size(), fout, nullptr, ks);
      if (rc) return rc;
      const size_t cnt = (size_t)k * len;
      if (out_mask) {
        vec_sum_kernel<Fr><<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ks>>>(local, fout, out_mask, cnt);
        ZK_HIP(hipGetLastError());
      } else {
        ZK_HIP(hipMemcpyAsync(local, fout, bytes, hipMemcpyDeviceToDevice, ks));
      }
      return ZK_OK;
    }
    Fr *fin = nullptr, *fout = nullptr;
    if (net->rank == 0) {
      ZK_HIP(dist_in_[sid].ensure((size_t)n * len * sizeof(Fr)));
      ZK_HIP(dist_out_[sid].ensure((size_t)n * len * sizeof(Fr)));
      fin = (Fr*)dist_in_[sid].p;
      fout = (Fr*)dist_out_[sid].p;
    }
    int rc = net_err(net, net->gather(sid, mask, local, bytes, fin));
    if (rc) return rc;
    if (net->rank == 0) {
      std::vector<uint32_t> ps = parties_of(net, mask);
      rc = king(fin, ps.data(), (int)ps.size(), fout, nullptr, ks);
      if (rc) return rc;
    }
    rc = net_err(net, net->scatter(sid, mask, fout, bytes, local));
    if (rc) return rc;
    if (out_mask) return vec_add(local, out_mask, (size_t)k * len, ks);
    return ZK_OK;
  }

  // ---- second-stage king: every present rank is king of a contiguous range of chunks.  One all-to-all
  // brings a range's input chunks of all parties to its owner, the king kernel runs on the range, a second all-to-all
  // returns every party's output shares.  Per link a round carries 1/W of the star's bytes and no rank does more than
  // 1/W of the king's arithmetic.  Option "king_alltoall" (or ZK_KING_ALLTOALL=1), set alike on every rank.
  bool king_a2a_ = false;        // zk_ctx_set_option("king_alltoall")
  DevBuf a2a_send_[NET_NSID], a2a_back_[NET_NSID];
  struct A2aPlan {
    A2aMap map;
    int me;
    uint32_t seg, rs, cnt;
  };
  // ranges of `gran`-aligned columns over the present ranks; false when there is less than one granule per rank
  bool a2a_plan(const Net* net, uint32_t mask, size_t len, size_t gran, A2aPlan* pl) const {
    pl->map.nranks = net->world;
    pl->map.npresent = 0;
    pl->me = -1;
    for (int r = 0; r < 16; r++) pl->map.idx_of_rank[r] = -1;
    for (int r = 0; r < net->world; r++)
      if (mask & (1u << r)) {
        if (r == net->rank) pl->me = pl->map.npresent;
        pl->map.idx_of_rank[r] = pl->map.npresent++;
      }
    const size_t np_r = (size_t)pl->map.npresent;
    if (pl->me < 0 || np_r < 2 || len / gran < np_r || len % gran) return false;
    const size_t seg = ((len / gran + np_r - 1) / np_r) * gran;
    if (seg > 0xffffffffull) return false;
    pl->seg = (uint32_t)seg;
    const size_t rs0 = (size_t)pl->me * seg;
    pl->rs = (uint32_t)(rs0 < len ? rs0 : len);
    pl->cnt = (uint32_t)(rs0 >= len ? 0 : (len - rs0 < seg ? len - rs0 : seg));
    return true;
  }
  static constexpr int A2A_NOT_APPLICABLE = -1000;
  // king(in [np][seg], parties, np, out [n][seg], plan, stream); shift / unpack: see pss.hpp a2a_*_kernel
  template <class KingFn, class UnpackFn>
  int king_round_a2a(Net* net, int sid, uint32_t mask, Fr* local, size_t len, size_t gran, uint32_t shift, KingFn king,
                     UnpackFn unpack) {
    A2aPlan pl;
    if (!a2a_plan(net, mask, len, gran, &pl)) return A2A_NOT_APPLICABLE;
    const int k = net->parties_per_rank();
    const size_t blk = (size_t)k * pl.seg;                       // elements per (rank, rank) block
    const int npres = pl.map.npresent;
    hipStream_t s = net_stream(net, sid, nullptr);
    ZK_HIP(a2a_send_[sid].ensure((size_t)net->world * blk * sizeof(Fr)));
    ZK_HIP(a2a_back_[sid].ensure((size_t)npres * blk * sizeof(Fr)));
    ZK_HIP(dist_in_[sid].ensure((size_t)npres * blk * sizeof(Fr)));
    ZK_HIP(dist_out_[sid].ensure((size_t)n * pl.seg * sizeof(Fr)));
    Fr* send = (Fr*)a2a_send_[sid].p;
    Fr* back = (Fr*)a2a_back_[sid].p;
    Fr* fin = (Fr*)dist_in_[sid].p;
    Fr* fout = (Fr*)dist_out_[sid].p;
    {
      const size_t tot = (size_t)net->world * blk;
      a2a_pack_kernel<Fr><<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(local, k, len, pl.seg, shift, pl.map, send);
      ZK_HIP(hipGetLastError());
    }
    int rc = net_err(net, net->alltoall(sid, mask, send, blk * sizeof(Fr), fin));
    if (rc) return rc;
    std::vector<uint32_t> ps = parties_of(net, mask);
    rc = king(fin, ps.data(), (int)ps.size(), fout, pl, s);
    if (rc) return rc;
    rc = net_err(net, net->alltoall(sid, mask, fout, blk * sizeof(Fr), back));
    if (rc) return rc;
    return unpack(back, pl, s);
  }

  int dist_d_fft_on(Net* net, int sid, uint32_t mask, Fr* shares, const Fr* in_mask, const Fr* out_mask, int rearrange,
                    int log_m, int inverse, const void* g, uint64_t seed, bool do_fft1) {
    const int k = net->parties_per_rank();
    const size_t Lc = ((size_t)1 << log_m) / l;
    hipStream_t s = net_stream(net, sid, nullptr);
    int rc;
    const Fr* king_in = nullptr;
    if (do_fft1) {
      // d_ifft scales by 1/m before anything else (dfft/mod.rs:159); without an in-mask the king folds the factor
      // into its g^i table (same field values), with one it has to come before the mask is added (:254-258)
      if (inverse && in_mask) {
        Fr c = Fr::from_u64((uint64_t)1 << log_m).inverse();
        rc = vec_scale(shares, &c, (size_t)k * Lc, s);
        if (rc) return rc;
      }
      if (net->transport == ZK_NET_LOCAL) {
        // one rank: the first NTT pass reads the caller's rows and writes a work vector, the king reads that and writes
        // the caller's rows -- no copy back (d_fft 2^20: one 60 us pass less)
        ZK_HIP(dist_in_[sid].ensure((size_t)k * Lc * sizeof(Fr)));
        NttSrc<Fr> src{};
        src.p[0] = shares;
        src.per = (uint32_t)k;
        rc = fft1_src(dist_in_[sid].p, log_m, inverse, (size_t)k, in_mask, s, src);
        king_in = (const Fr*)dist_in_[sid].p;
      } else {
        rc = fft1(shares, log_m, inverse, (size_t)k, in_mask, s);
      }
      if (rc) return rc;
    } else if (in_mask) {
      rc = vec_add(shares, in_mask, (size_t)k * Lc, s);
      if (rc) return rc;
    }
    const int scale = (inverse && !in_mask) ? 1 : 0;
    rc = A2A_NOT_APPLICABLE;
    if (king_a2a_ && net->contiguous() && net->transport != ZK_NET_LOCAL) {
      const size_t kbk = (size_t)king_block(Lc);
      const size_t Wc = Lc < kbk ? Lc : kbk;
      const int log_lc = log_m - ilog2(l);
      if (Wc < Lc)
        rc = king_round_a2a(
            net, sid, mask, shares, Lc, Wc, 1u,
            [&](const Fr* in, const uint32_t* ps, int np, Fr* out, const A2aPlan& pl, hipStream_t ks) {
              const Fr* U = nullptr;
              int r2 = umat_for(ps, np, &U);
              if (r2) return r2;
              KingRange rg{pl.rs, pl.seg, pl.cnt};
              return king_dispatch(in, nullptr, np, log_m, inverse, U, g, scale, rearrange, seed, out, nullptr, false, ks, &rg);
            },
            [&](const Fr* back, const A2aPlan& pl, hipStream_t ks) {
              const size_t tot = (size_t)pl.map.npresent * k * pl.seg;
              a2a_unpack_fft_kernel<Fr><<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ks>>>(
                  back, k, (uint32_t)log_lc, pl.seg, l, ilog2(l), rearrange, pl.map.npresent, shares);
              ZK_HIP(hipGetLastError());
              return (int)ZK_OK;
            });
    }
    if (rc == A2A_NOT_APPLICABLE)
    rc = king_round(net, sid, mask, shares, Lc, [&](const Fr* in, const uint32_t* ps, int np, Fr* out, const Fr* om, hipStream_t ks) {
      const Fr* U = nullptr;
      int r2 = umat_for(ps, np, &U);
      if (r2) return r2;
      return king_dispatch(in, nullptr, np, log_m, inverse, U, g, scale, rearrange, seed, out, om, false, ks);
    }, out_mask, king_in);
    else if (!rc && out_mask) rc = vec_add(shares, out_mask, (size_t)k * Lc, s);       // after the all-to-all king
    return rc;
  }
  int dist_d_fft(Net* net, int sid, void* shares, const void* in_mask, const void* out_mask, int rearrange, int log_m,
                 int inverse, const void* g, uint64_t seed, hipStream_t st) override {
    if (!shares) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (log_m < ilog2(l) || log_m > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    uint32_t mask = 0;
    int rc = net_err(net, net->enter(sid, &mask));
    if (rc) return rc;
    rc = net_err(net, net->begin(sid, st));
    if (rc) return rc;
    rc = dist_d_fft_on(net, sid, mask, (Fr*)shares, (const Fr*)in_mask, (const Fr*)out_mask, rearrange, log_m, inverse, g,
                       seed, true);
    if (rc) return rc;
    return net_err(net, net->end(sid, st));
  }

  int dist_deg_red_on(Net* net, int sid, uint32_t mask, Fr* x, const Fr* in_mask, const Fr* out_mask, size_t len,
                      uint64_t seed) {
    const int k = net->parties_per_rank();
    hipStream_t s = net_stream(net, sid, nullptr);
    int rc;
    if (in_mask) {
      rc = vec_add(x, in_mask, (size_t)k * len, s);
      if (rc) return rc;
    }
    rc = A2A_NOT_APPLICABLE;
    if (king_a2a_ && net->contiguous() && net->transport != ZK_NET_LOCAL)
      rc = king_round_a2a(
          net, sid, mask, x, len, 1, 0u,
          [&](const Fr* in, const uint32_t* ps, int np, Fr* out, const A2aPlan& pl, hipStream_t ks) {
            return deg_red_np(in, nullptr, ps, np, pl.cnt, seed, out, nullptr, ks, pl.seg, pl.rs);
          },
          [&](const Fr* back, const A2aPlan& pl, hipStream_t ks) {
            const size_t tot = (size_t)pl.map.npresent * k * pl.seg;
            a2a_unpack_rows_kernel<Fr><<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ks>>>(back, k, len, pl.seg,
                                                                                                pl.map.npresent, x);
            ZK_HIP(hipGetLastError());
            return (int)ZK_OK;
          });
    if (rc == A2A_NOT_APPLICABLE)
      return king_round(net, sid, mask, x, len, [&](const Fr* in, const uint32_t* ps, int np, Fr* out, const Fr* om, hipStream_t ks) {
        return deg_red_np(in, nullptr, ps, np, len, seed, out, om, ks);
      }, out_mask, nullptr, true);
    if (rc) return rc;
    if (out_mask) return vec_add(x, out_mask, (size_t)k * len, s);        // after the all-to-all king
    return ZK_OK;
  }
  int dist_deg_red(Net* net, int sid, void* x, const void* in_mask, const void* out_mask, size_t len, uint64_t seed,
                   hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!x) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    uint32_t mask = 0;
    int rc = net_err(net, net->enter(sid, &mask));
    if (rc) return rc;
    rc = net_err(net, net->begin(sid, st));
    if (rc) return rc;
    rc = dist_deg_red_on(net, sid, mask, (Fr*)x, (const Fr*)in_mask, (const Fr*)out_mask, len, seed);
    if (rc) return rc;
    return net_err(net, net->end(sid, st));
  }

  // d_pp (dpp/mod.rs:15-87): round 1 gathers num || den, the king divides, scans and packs fresh shares of the prefix
  // products, round 2 is deg_red (:84-86)
  int dist_d_pp(Net* net, int sid, const void* num, const void* den, const void* in_mask, const void* out_mask, size_t len,
                uint64_t seed, void* out, hipStream_t st) override {
    if (!len) return ZK_OK;
    if (!num || !den || !out) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const int k = net->parties_per_rank();
    uint32_t mask = 0;
    int rc = net_err(net, net->enter(sid, &mask));
    if (rc) return rc;
    rc = net_err(net, net->begin(sid, st));
    if (rc) return rc;
    hipStream_t s = net_stream(net, sid, nullptr);
    const size_t bytes = (size_t)k * len * sizeof(Fr);
    Fr *fin = nullptr, *fout = nullptr;
    const std::vector<uint32_t> ps = parties_of(net, mask);
    const int np = (int)ps.size();
    if (net->rank == 0) {
      ZK_HIP(dist_in_[sid].ensure((size_t)2 * n * len * sizeof(Fr)));
      ZK_HIP(dist_out_[sid].ensure((size_t)n * len * sizeof(Fr)));
      fin = (Fr*)dist_in_[sid].p;
      fout = (Fr*)dist_out_[sid].p;
    }
    rc = net_err(net, net->gather(sid, mask, num, bytes, fin));
    if (rc) return rc;
    rc = net_err(net, net->gather(sid, mask, den, bytes, fin ? fin + (size_t)np * len : nullptr));
    if (rc) return rc;
    if (net->rank == 0) {
      rc = d_pp_king(fin, fin + (size_t)np * len, ps.data(), np, len, seed, fout, s);
      if (rc) return rc;
    }
    rc = net_err(net, net->scatter(sid, mask, fout, bytes, out));
    if (rc) return rc;
    rc = dist_deg_red_on(net, sid, mask, (Fr*)out, (const Fr*)in_mask, (const Fr*)out_mask, len, seed ^ 0x3333);
    if (rc) return rc;
    return net_err(net, net->end(sid, st));
  }

  // d_msm (dmsm/mod.rs:59-102): this rank's fused contribution sum_p coef_p (msm_p + in_mask_p) goes to the king as ONE
  // point; the king sums the ranks' points (= unpack2 + sum over all parties) and sends the result to everyone.
  template <class Fld>
  int dist_d_msm_t(Net* net, int sid, const void* bases, const void* scalars, size_t len, const void* in_mask,
                   const void* out_mask, void* out, hipStream_t st) {
    const int k = net->parties_per_rank();
    int first = 0;
    uint32_t mask = 0;
    int rc = net_err(net, net->enter(sid, &mask));
    if (rc) return rc;
    rc = local_window(net, &first);
    if (rc) return rc;
    // coefficients of my parties in the king's linear form; they depend on who takes part (pss.rs:170-221)
    const Fr* cd = msm_.coef_d_ + first;
    std::vector<Fr> csub;
    if (mask != net->full_mask()) {
      std::vector<uint32_t> ps = parties_of(net, mask);
      std::vector<Fr> coef;
      rc = coefs_for(ps.data(), (int)ps.size(), coef);
      if (rc) return rc;
      csub.resize((size_t)k);
      for (int i = 0; i < k; i++) {
        size_t pos = 0;
        while (pos < ps.size() && ps[pos] != (uint32_t)net->party(net->rank, i)) pos++;
        csub[i] = coef[pos];
      }
      ZK_HIP(dist_coef_[sid].ensure(k * sizeof(Fr)));
      ZK_HIP(hipMemcpy(dist_coef_[sid].p, csub.data(), k * sizeof(Fr), hipMemcpyHostToDevice));
      cd = (const Fr*)dist_coef_[sid].p;
    }
    MsmPending pend;
    rc = msm_.template launch_t<Fld>(this, bases, scalars, (size_t)k * len, cd, len, st, MSM_WS - 1 - sid, &pend);
    if (rc) return rc;
    XYZZ<Fld> mine = XYZZ<Fld>::identity();
    if (in_mask) mine = msm_.template mask_term<Fld>(in_mask, first, k, csub.empty() ? nullptr : csub.data());
    XYZZ<Fld> r;
    rc = msm_.template finish_t<Fld>(this, &pend, &r);
    if (rc) return rc;
    mine = xyzz_add_ni(mine, r);
    std::vector<XYZZ<Fld>> all((size_t)net->world);
    rc = net_err(net, net->gather_host(sid, mask, &mine, sizeof(mine), all.data()));
    if (rc) return rc;
    XYZZ<Fld> total = XYZZ<Fld>::identity();
    if (net->rank == 0) {
      int cnt = 0;
      for (int rr = 0; rr < net->world; rr++)
        if (mask & (1u << rr)) total = xyzz_add_ni(total, all[cnt++]);
    }
    rc = net_err(net, net->bcast_host(sid, mask, &total, sizeof(total)));
    if (rc) return rc;
    const Jacobian<Fld>* om = (const Jacobian<Fld>*)out_mask;
    Jacobian<Fld>* o = (Jacobian<Fld>*)out;
    for (int p = 0; p < k; p++) {
      XYZZ<Fld> v = total;
      if (om) v = xyzz_add_ni(v, jacobian_to_xyzz(om[p]));
      o[p] = xyzz_to_jacobian(v);
    }
    return ZK_OK;
  }
  int dist_d_msm(Net* net, int sid, int group, const void* bases, const void* scalars, size_t len, const void* in_mask,
                 const void* out_mask, void* out, hipStream_t st) override {
    if (!out) return fail(ZK_ERR_BAD_INPUT, "null output");
    if (len && (!bases || !scalars)) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (group == ZK_G1) return dist_d_msm_t<Fq_>(net, sid, bases, scalars, len, in_mask, out_mask, out, st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2) return dist_d_msm_t<Fq2_>(net, sid, bases, scalars, len, in_mask, out_mask, out, st);
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

  // circom_h (ext_wit.rs:104-181): the three d_ifft, then the three d_fft, each triple in flight together on channels
  // 0..2 (ext_wit.rs:158-170 joins them); then a*b - c and deg_red on channel 0.  masks: LOCAL rows.
  int dist_circom_h_on(Net* net, const uint32_t* cmask, const void* qa, const void* qb, const void* qc, int log_m,
                       const zk_groth16_masks* mk, uint64_t seed, void* h, hipStream_t st, DevBuf* wb = nullptr) {
    const int k = net->parties_per_rank();
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)k * Lc;
    if (!wb) wb = &dist_w0_;
    ZK_HIP(wb->ensure(3 * per * sizeof(Fr)));
    Fr* W = (Fr*)wb->p;
    const void* q[3] = {qa, qb, qc};
    for (int j = 0; j < 3; j++) ZK_HIP(hipMemcpyAsync(W + j * per, q[j], per * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    Fr w2m = root_of_unity(log_m + 1);
    const bool has_in = mk && mk->fft_in[0];
    int rc;
    for (int phase = 0; phase < 2; phase++) {
      const int inverse = phase == 0 ? 1 : 0;
      // the local stages of the three vectors as one batched launch on the caller's stream
      if (inverse && has_in) {
        Fr c = Fr::from_u64((uint64_t)1 << log_m).inverse();
        rc = vec_scale(W, &c, 3 * per, st);
        if (rc) return rc;
      }
      rc = fft1(W, log_m, inverse, 3 * (size_t)k, nullptr, st);
      if (rc) return rc;
      for (int j = 0; j < 3; j++) {
        rc = net_err(net, net->begin(j, st));
        if (rc) return rc;
      }
      for (int j = 0; j < 3; j++) {
        const int mi = phase * 3 + j;
        const bool masked = mk && mk->fft_in[mi];
        if ((mk && mk->fft_in[mi] != nullptr) != has_in) return fail(ZK_ERR_BAD_INPUT, "mixed in-masks in circom_h");
        (void)masked;
        rc = dist_d_fft_on(net, j, cmask[j], W + j * per, mk ? (const Fr*)mk->fft_in[mi] : nullptr,
                           mk ? (const Fr*)mk->fft_out[mi] : nullptr, phase == 0 ? 1 : 0, log_m, inverse,
                           phase == 0 ? (const void*)&w2m : nullptr, seed + mi, false);
        if (rc) return rc;
      }
      for (int j = 0; j < 3; j++) {
        rc = net_err(net, net->end(j, st));
        if (rc) return rc;
      }
    }
    rc = vec_mul_sub(h, W, W + per, W + 2 * per, per, st);
    if (rc) return rc;
    rc = net_err(net, net->begin(0, st));
    if (rc) return rc;
    rc = dist_deg_red_on(net, 0, cmask[0], (Fr*)h, mk ? (const Fr*)mk->degred_in : nullptr,
                         mk ? (const Fr*)mk->degred_out : nullptr, Lc, seed + 6);
    if (rc) return rc;
    return net_err(net, net->end(0, st));
  }
  // ---- deg_red over GROUP elements as the reference calls it (deg_red.rs:80-126 is generic over T: DomainCoeff<F> and
  // takes net, sid): this rank's k parties' points x [k][len] (affine), masks likewise.  The parties add their in-mask
  // (point additions), the king unpack2's and re-packs over points (points_lincomb_kernel), the parties add their out-mask.
  Fr* one_canon_ = nullptr;
  DevBuf dist_pt_[NET_NSID];
  template <class Fld>
  int points_add_rows(Affine<Fld>* dst, const Affine<Fld>* a, const Affine<Fld>* b, size_t count, hipStream_t st) {
    if (!one_canon_) {
      std::vector<Fr> one(1, Fr::one().from_mont());
      int rc = upload(one, &one_canon_);
      if (rc) return rc;
    }
    PtGroup<Fld> g{a, 1, 0, 1, 0};
    points_lincomb_kernel<FrP, Fld><<<dim3((unsigned)((count + 127) / 128)), dim3(128), 0, st>>>(g, g, 1, one_canon_, 1, 1, count,
                                                                                              b, 0, dst, 0, 1);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  template <class Fld>
  int dist_deg_red_points_t(Net* net, int sid, const void* x, const void* in_mask, const void* out_mask, size_t len,
                            const void* gen_affine, uint64_t seed, void* out, hipStream_t st) {
    using A = Affine<Fld>;
    if (!len) return ZK_OK;
    if (!x || !out || !gen_affine) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (x == out) return fail(ZK_ERR_BAD_INPUT, "deg_red over points cannot run in place");
    const int k = net->parties_per_rank();
    const size_t cnt = (size_t)k * len, bytes = cnt * sizeof(A);
    uint32_t mask = 0;
    int rc = net_err(net, net->enter(sid, &mask));
    if (rc) return rc;
    rc = net_err(net, net->begin(sid, st));
    if (rc) return rc;
    hipStream_t s = net_stream(net, sid, st);
    const A* send = (const A*)x;
    if (in_mask) {
      ZK_HIP(dist_pt_[sid].ensure(bytes));
      rc = points_add_rows<Fld>((A*)dist_pt_[sid].p, (const A*)x, (const A*)in_mask, cnt, s);
      if (rc) return rc;
      send = (const A*)dist_pt_[sid].p;
    }
    A *fin = nullptr, *fout = nullptr;
    if (net->rank == 0) {
      ZK_HIP(dist_in_[sid].ensure((size_t)n * len * sizeof(A)));
      ZK_HIP(dist_out_[sid].ensure((size_t)n * len * sizeof(A)));
      fin = (A*)dist_in_[sid].p;
      fout = (A*)dist_out_[sid].p;
    }
    rc = net_err(net, net->gather(sid, mask, send, bytes, fin));
    if (rc) return rc;
    if (net->rank == 0) {
      std::vector<uint32_t> ps = parties_of(net, mask);
      rc = deg_red_points_t<Fld>(fin, nullptr, nullptr, len, gen_affine, seed, fout, s, ps.data(), (int)ps.size());
      if (rc) return rc;
    }
    rc = net_err(net, net->scatter(sid, mask, fout, bytes, out));
    if (rc) return rc;
    if (out_mask) {
      rc = points_add_rows<Fld>((A*)out, (const A*)out, (const A*)out_mask, cnt, s);
      if (rc) return rc;
    }
    return net_err(net, net->end(sid, st));
  }
  int dist_deg_red_points(Net* net, int sid, int group, const void* x, const void* in_mask, const void* out_mask, size_t len,
                          const void* gen_affine, uint64_t seed, void* out, hipStream_t st) override {
    if (group == ZK_G1) return dist_deg_red_points_t<Fq_>(net, sid, x, in_mask, out_mask, len, gen_affine, seed, out, st);
    if (group == ZK_G2) {
      if constexpr (Cfg::HAS_G2)
        return dist_deg_red_points_t<Fq2_>(net, sid, x, in_mask, out_mask, len, gen_affine, seed, out, st);
    }
    return fail(ZK_ERR_BAD_INPUT, "bad group");
  }

  // ---- libsnark_h as the reference calls it (ext_wit.rs:14-102): three d_ifft with the coset shift g = F::GENERATOR on
  // channels 0..2 (joined), three d_fft likewise, (a b - c) / Z(g) locally, d_ifft with g^-1 on channel 0.  Buffers and
  // the seven masks: this rank's k parties' rows.
  int dist_libsnark_h(Net* net, const void* qa, const void* qb, const void* qc, int log_m, const void* const* fft_in,
                      const void* const* fft_out, uint64_t seed, void* h, hipStream_t st) override {
    if (log_m < ilog2(l) || log_m > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (!qa || !qb || !qc || !h) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    uint32_t cmask[3];
    for (int j = 0; j < 3; j++) {
      int rc = net_err(net, net->enter(j, &cmask[j]));
      if (rc) return rc;
    }
    if (cmask[1] != cmask[0] || cmask[2] != cmask[0])
      return fail(ZK_ERR_PROTOCOL, "the three channels of libsnark_h saw different parties", -1);
    const int k = net->parties_per_rank();
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)k * Lc;
    ZK_HIP(dist_w0_.ensure(3 * per * sizeof(Fr)));
    Fr* W = (Fr*)dist_w0_.p;
    const void* q[3] = {qa, qb, qc};
    for (int j = 0; j < 3; j++) ZK_HIP(hipMemcpyAsync(W + j * per, q[j], per * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    const Fr g = generator();
    auto mi = [&](int i) { return fft_in ? (const Fr*)fft_in[i] : nullptr; };
    auto mo = [&](int i) { return fft_out ? (const Fr*)fft_out[i] : nullptr; };
    int rc;
    for (int phase = 0; phase < 2; phase++) {
      const int inverse = phase == 0 ? 1 : 0;
      bool any = false, all = true;
      for (int j = 0; j < 3; j++) {
        any = any || mi(phase * 3 + j);
        all = all && mi(phase * 3 + j);
      }
      if (any && !all) return fail(ZK_ERR_BAD_INPUT, "mixed in-masks in libsnark_h");
      if (inverse && any) {
        Fr c = Fr::from_u64((uint64_t)1 << log_m).inverse();
        rc = vec_scale(W, &c, 3 * per, st);
        if (rc) return rc;
      }
      rc = fft1(W, log_m, inverse, 3 * (size_t)k, nullptr, st);
      if (rc) return rc;
      for (int j = 0; j < 3; j++) {
        rc = net_err(net, net->begin(j, st));
        if (rc) return rc;
      }
      for (int j = 0; j < 3; j++) {
        const int i = phase * 3 + j;
        rc = dist_d_fft_on(net, j, cmask[j], W + j * per, mi(i), mo(i), 1, log_m, inverse,
                           phase == 0 ? (const void*)&g : nullptr, seed + i, false);
        if (rc) return rc;
      }
      for (int j = 0; j < 3; j++) {
        rc = net_err(net, net->end(j, st));
        if (rc) return rc;
      }
    }
    rc = vec_mul_sub(h, W, W + per, W + 2 * per, per, st);
    if (rc) return rc;
    Fr zinv = (g.pow_u64((uint64_t)1 << log_m) - Fr::one()).inverse();      // 1 / Z(g), Z(x) = x^m - 1 (ext_wit.rs:78-81)
    rc = vec_scale(h, &zinv, per, st);
    if (rc) return rc;
    const Fr ginv = g.inverse();
    rc = net_err(net, net->begin(0, st));
    if (rc) return rc;
    rc = dist_d_fft_on(net, 0, cmask[0], (Fr*)h, mi(6), mo(6), 0, log_m, 1, &ginv, seed + 6, true);
    if (rc) return rc;
    return net_err(net, net->end(0, st));
  }

  // circom_h of a whole BATCH of proofs with ONE king round per phase and channel (round 4; round 3 sent the proofs'
  // rounds over the channels one proof after the other: 7 nb rounds per batch, the star's serial rounds bounded the
  // sharded throughput mode).  The three joined d_ifft / d_fft of ext_wit.rs:127-170 stay three channels in flight; a
  // channel's round now carries the vectors of all nb proofs: party rows are [nb][m/l] (KingBatch::row_pitch,
  // DegredBatch strides), so gather, king kernel and scatter move nb vectors per party at once.  Proof b draws the share
  // randomness of a single proof with seed + 16 b.  h_all: [nb][k][m/l].  mk: nb mask sets (local rows) or nullptr.
  DevBuf dist_wb_, dist_hb_;
  int vec_add2d(Fr* x, size_t xpitch, const Fr* y, size_t ypitch, size_t width, size_t rows, hipStream_t st) {
    const size_t tot = width * rows;
    if (!tot) return ZK_OK;
    vec_add2d_kernel<Fr><<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(x, xpitch, y, ypitch, width, rows);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  }
  int dist_circom_h_batch_on(Net* net, const uint32_t* cmask, int nb, const void* const* qa, const void* const* qb,
                             const void* const* qc, int log_m, const zk_groth16_masks* mk, uint64_t seed, Fr* h_all,
                             hipStream_t st) {
    const int k = net->parties_per_rank();
    const size_t Lc = ((size_t)1 << log_m) / l, row = (size_t)nb * Lc, per = (size_t)k * row;   // per channel: [k][nb][Lc]
    if (nb < 1 || nb > KING_BATCH || nb > DEGRED_BATCH) return fail(ZK_ERR_BAD_INPUT, "bad circom_h batch");
    ZK_HIP(dist_wb_.ensure(3 * per * sizeof(Fr)));
    ZK_HIP(dist_hb_.ensure(per * sizeof(Fr)));
    Fr* W = (Fr*)dist_wb_.p;
    Fr* H = (Fr*)dist_hb_.p;
    const void* const* q[3] = {qa, qb, qc};
    for (int j = 0; j < 3; j++)
      for (int b = 0; b < nb; b++)
        ZK_HIP(hipMemcpy2DAsync(W + j * per + (size_t)b * Lc, row * sizeof(Fr), q[j][b], Lc * sizeof(Fr), Lc * sizeof(Fr),
                                (size_t)k, hipMemcpyDeviceToDevice, st));
    Fr w2m = root_of_unity(log_m + 1);
    const bool has_in = mk && mk[0].fft_in[0];
    int rc;
    for (int phase = 0; phase < 2; phase++) {
      const int inverse = phase == 0 ? 1 : 0;
      if (inverse && has_in) {
        Fr c = Fr::from_u64((uint64_t)1 << log_m).inverse();
        rc = vec_scale(W, &c, 3 * per, st);
        if (rc) return rc;
      }
      rc = fft1(W, log_m, inverse, 3 * (size_t)k * nb, nullptr, st);
      if (rc) return rc;
      for (int j = 0; j < 3; j++) {
        rc = net_err(net, net->begin(j, st));
        if (rc) return rc;
      }
      for (int j = 0; j < 3; j++) {
        const int mi = phase * 3 + j;
        hipStream_t s = net_stream(net, j, nullptr);
        Fr* Wj = W + j * per;
        for (int b = 0; b < nb; b++) {
          if ((mk && mk[b].fft_in[mi] != nullptr) != has_in) return fail(ZK_ERR_BAD_INPUT, "mixed in-masks in circom_h");
          if (has_in) {
            rc = vec_add2d(Wj + (size_t)b * Lc, row, (const Fr*)mk[b].fft_in[mi], Lc, Lc, (size_t)k, s);
            if (rc) return rc;
          }
        }
        const int scale = (inverse && !has_in) ? 1 : 0;
        const void* g = phase == 0 ? (const void*)&w2m : nullptr;
        rc = king_round(net, j, cmask[j], Wj, row,
                        [&](const Fr* in, const uint32_t* ps, int np, Fr* out, const Fr*, hipStream_t ks) {
                          const Fr* U = nullptr;
                          int r2 = umat_for(ps, np, &U);
                          if (r2) return r2;
                          KingBatch<Fr> kb{};
                          kb.stride = Lc;
                          kb.row_pitch = row;
                          kb.items_per = 1;
                          kb.seed_step = PROOF_SEED_STEP;
                          return king_dispatch_batch(in, kb, nb, np, log_m, inverse, U, g, scale, phase == 0 ? 1 : 0,
                                                     seed + (uint64_t)mi, out, false, ks);
                        });
        if (rc) return rc;
        for (int b = 0; b < nb; b++)
          if (mk && mk[b].fft_out[mi]) {
            rc = vec_add2d(Wj + (size_t)b * Lc, row, (const Fr*)mk[b].fft_out[mi], Lc, Lc, (size_t)k, s);
            if (rc) return rc;
          }
      }
      for (int j = 0; j < 3; j++) {
        rc = net_err(net, net->end(j, st));
        if (rc) return rc;
      }
    }
    rc = vec_mul_sub(H, W, W + per, W + 2 * per, per, st);
    if (rc) return rc;
    rc = net_err(net, net->begin(0, st));
    if (rc) return rc;
    {
      hipStream_t s = net_stream(net, 0, nullptr);
      for (int b = 0; b < nb; b++)
        if (mk && mk[b].degred_in) {
          rc = vec_add2d(H + (size_t)b * Lc, row, (const Fr*)mk[b].degred_in, Lc, Lc, (size_t)k, s);
          if (rc) return rc;
        }
      rc = king_round(net, 0, cmask[0], H, row,
                      [&](const Fr* in, const uint32_t* ps, int np, Fr* out, const Fr*, hipStream_t ks) {
                        DegredBatch<Fr> db{};
                        db.in_step = db.out_step = Lc;
                        db.seed_step = PROOF_SEED_STEP;
                        return deg_red_batch(in, db, nb, ps, np, Lc, seed + 6, out, ks, row);
                      },
                      nullptr, nullptr, true);
      if (rc) return rc;
      for (int b = 0; b < nb; b++)
        if (mk && mk[b].degred_out) {
          rc = vec_add2d(H + (size_t)b * Lc, row, (const Fr*)mk[b].degred_out, Lc, Lc, (size_t)k, s);
          if (rc) return rc;
        }
    }
    rc = net_err(net, net->end(0, st));
    if (rc) return rc;
    // [k][nb][Lc] -> the batch prover's [nb][k][Lc]
    for (int b = 0; b < nb; b++)
      ZK_HIP(hipMemcpy2DAsync(h_all + (size_t)b * k * Lc, Lc * sizeof(Fr), H + (size_t)b * Lc, row * sizeof(Fr),
                              Lc * sizeof(Fr), (size_t)k, hipMemcpyDeviceToDevice, st));
    return ZK_OK;
  }
  int dist_circom_h(Net* net, const void* qa, const void* qb, const void* qc, int log_m, const zk_groth16_masks* mk,
                    uint64_t seed, void* h, hipStream_t st) override {
    if (log_m < ilog2(l) || log_m + 1 > FrP::TWO_ADICITY) return fail(ZK_ERR_BAD_INPUT, "Mismatch of size in FFT");
    if (!qa || !qb || !qc || !h) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    uint32_t cmask[3];
    for (int j = 0; j < 3; j++) {
      int rc = net_err(net, net->enter(j, &cmask[j]));
      if (rc) return rc;
    }
    if (cmask[1] != cmask[0] || cmask[2] != cmask[0])
      return fail(ZK_ERR_PROTOCOL, "the three channels of circom_h saw different parties", -1);
    return dist_circom_h_on(net, cmask, qa, qb, qc, log_m, mk, seed, h, st);
  }

  // dsha256 per rank (sha256.rs:32-129): all shares and masks are this rank's k parties' rows; pi_*: k Jacobian points.
  // Two halves, so that a rank can keep NJOBS proofs in flight (the reference's parties are concurrent tasks,
  // mpc-net/src/multi.rs:317-327; prove.rs:209-227 joins the W and U d_msm):
  //   dist_prove_async  admits the proof on the control plane, starts the four witness MSMs on the job's streams, runs
  //                     circom_h's king rounds over channels 0..2 and queues the U-MSM behind them; returns with all of
  //                     that ENQUEUED (the host only waited for the peers' staging ticks);
  //   dist_prove_wait   joins the MSMs, sends this rank's five partial sums to the king on channel 3 (d_msm's king step
  //                     for all five products in one message, dmsm/mod.rs:76-92) and assembles the k proof shares.
  // Every rank issues the same sequence of calls (channels are ordered, multi.rs:418-445): async(A), async(B), wait(A),
  // wait(B) overlaps B's king rounds with A's MSMs.
  struct DistJob {
    bool active = false;
    int k = 0;
    uint32_t cmask3 = 0;
  };
  DistJob djobs_[NJOBS];
  DevBuf dist_wj_[NJOBS], dist_hj_[NJOBS];
  int dist_prove_async(Net* net, const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                       const void* a_share, const void* ax_share, const void* r_, const void* s_, int log_m,
                       const zk_groth16_masks* mk, uint64_t seed, hipStream_t st, int* handle) override {
    int rc = check_prove_args(crs, r_, s_, log_m);
    if (rc) return rc;
    if (!qa || !qb || !qc || !a_share || !ax_share || !handle) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int slot = -1;
    for (int i = 0; i < NJOBS; i++)
      if (!jobs_[i].active && !djobs_[i].active) {
        slot = i;
        break;
      }
    if (slot < 0) return fail(ZK_ERR_BAD_INPUT, "too many proofs in flight (zk_dist_groth16_wait one first)");
    const int k = net->parties_per_rank();
    int first = 0;
    rc = local_window(net, &first);
    if (rc) return rc;
    const size_t Lc = ((size_t)1 << log_m) / l;
    uint32_t cmask[NET_NSID];
    for (int j = 0; j < NET_NSID; j++) {
      rc = net_err(net, net->enter(j, &cmask[j]));
      if (rc) return rc;
    }
    for (int j = 0; j < NET_NSID; j++)
      if (cmask[j] != net->full_mask())
        return fail(ZK_ERR_PROTOCOL, "a party did not show up for the proof (timed out)", -1);
    Fr r = Fr::from_limbs((const uint32_t*)r_), s = Fr::from_limbs((const uint32_t*)s_);
    ProveJob& j = jobs_[slot];
    j.slot = slot;
    // the four MSMs over the witness shares start now and overlap the king rounds of circom_h (prove.rs try_join!)
    rc = prove_begin(j, crs, nullptr, nullptr, nullptr, a_share, ax_share, r, s, log_m, mk, seed, false, first, k, st, true);
    auto bail = [&](int code) {
      Status keep = last;
      abort_job(j);
      last = keep;
      return code;
    };
    if (rc) return bail(rc);
    {
      hipError_t he = dist_hj_[slot].ensure((size_t)k * Lc * sizeof(Fr));
      if (he != hipSuccess) return bail(hip_fail(he, "h share buffer"));
    }
    rc = dist_circom_h_on(net, cmask, qa, qb, qc, log_m, mk, seed, dist_hj_[slot].p, st, &dist_wj_[slot]);
    if (rc) return bail(rc);
    rc = prove_launch_u(j, dist_hj_[slot].p, st);
    if (rc) return bail(rc);
    djobs_[slot].active = true;
    djobs_[slot].k = k;
    djobs_[slot].cmask3 = cmask[3];
    *handle = slot;
    return ZK_OK;
  }
  int dist_prove_wait(Net* net, int handle, void* pi_a, void* pi_b, void* pi_c) override {
    if (handle < 0 || handle >= NJOBS || !djobs_[handle].active) return fail(ZK_ERR_BAD_INPUT, "no sharded proof in flight on this handle");
    if (!pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    ProveJob& j = jobs_[handle];
    DistJob& d = djobs_[handle];
    d.active = false;
    struct Sums {
      P1 S, H, W, U;
      P2 V;
    } mine, total;
    int rc = prove_join(j, &mine.S, &mine.H, &mine.V, &mine.W, &mine.U);
    if (rc) return rc;
    static_assert(sizeof(Sums) <= NET_PAYLOAD, "payload");
    std::vector<Sums> all((size_t)net->world);
    rc = net_err(net, net->gather_host(3, d.cmask3, &mine, sizeof(mine), all.data()));
    if (rc) return rc;
    total = mine;
    if (net->rank == 0)
      for (int rr = 1; rr < net->world; rr++) {
        total.S = xyzz_add_ni(total.S, all[rr].S);
        total.H = xyzz_add_ni(total.H, all[rr].H);
        total.V = xyzz_add_ni(total.V, all[rr].V);
        total.W = xyzz_add_ni(total.W, all[rr].W);
        total.U = xyzz_add_ni(total.U, all[rr].U);
      }
    rc = net_err(net, net->bcast_host(3, d.cmask3, &total, sizeof(total)));
    if (rc) return rc;
    return assemble_points(&j.crs, j.r, j.s, total.S, total.H, total.V, total.W, total.U, j.has_mk ? &j.mk : nullptr, d.k, pi_a,
                           pi_b, pi_c);
  }
  int dist_prove(Net* net, const zk_crs_share* crs, const void* qa, const void* qb, const void* qc, const void* a_share,
                 const void* ax_share, const void* r_, const void* s_, int log_m, const zk_groth16_masks* mk,
                 uint64_t seed, void* pi_a, void* pi_b, void* pi_c, hipStream_t st) override {
    if (!pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    int h = -1;
    int rc = dist_prove_async(net, crs, qa, qb, qc, a_share, ax_share, r_, s_, log_m, mk, seed, st, &h);
    if (rc) return rc;
    return dist_prove_wait(net, h, pi_a, pi_b, pi_c);
  }

  // A batch of proofs per rank (zk_dist_groth16_prove_batch): the throughput mode of the sharded prover.  One round of
  // the control plane admits the whole batch; every rank runs each of its five MSMs ONCE over the nb witnesses (the
  // batched Pippenger of msm.hpp) while the king rounds of the proofs' circom_h go over the channels one proof after the
  // other; the partial sums of the whole batch cross in ONE host message per rank and one answer.  Shares and masks: this
  // rank's k parties' rows of every proof; pi_*: [nb][k] Jacobian points.  Proof b draws the share randomness
  // zk_dist_groth16_prove draws with seed + 16 b.
  int dist_prove_batch(Net* net, const zk_crs_share* crs, int nb, const void* const* qa, const void* const* qb,
                       const void* const* qc, const void* const* a_share, const void* const* ax_share, const void* r_,
                       const void* s_, int log_m, const zk_groth16_masks* mk, uint64_t seed, void* pi_a, void* pi_b,
                       void* pi_c, hipStream_t st) override {
    if (!qa || !qb || !qc || !pi_a || !pi_b || !pi_c) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    if (nb < 1 || nb > MAX_PROOF_BATCH) return fail(ZK_ERR_BAD_INPUT, "batch size must be in 1.." + std::to_string(MAX_PROOF_BATCH));
    for (int b = 0; b < nb; b++)
      if (!qa[b] || !qb[b] || !qc[b]) return fail(ZK_ERR_BAD_INPUT, "null pointer");
    const int k = net->parties_per_rank();
    int first = 0;
    const size_t Lc = ((size_t)1 << log_m) / l, per = (size_t)k * Lc;
    uint32_t cmask[NET_NSID];
    int rc = local_window(net, &first);
    if (rc) return rc;
    for (int j = 0; j < NET_NSID; j++) {
      rc = net_err(net, net->enter(j, &cmask[j]));
      if (rc) return rc;
    }
    for (int j = 0; j < NET_NSID; j++)
      if (cmask[j] != net->full_mask())
        return fail(ZK_ERR_PROTOCOL, "a party did not show up for the batch (timed out)", -1);
    int slot = -1;
    rc = batch_begin(crs, nb, a_share, ax_share, r_, s_, log_m, mk, false, first, k, st, &slot);
    if (rc) return rc;
    BatchJobX& B = bjobs_[slot];
    auto bail = [&](int code) {
      Status keep = last;
      abort_batch(B);
      last = keep;
      return code;
    };
    // circom_h of the whole batch: one king round per phase and channel carries all nb proofs (7 rounds per batch; the MSMs
    // of the batch run beside them; round 3 ran one proof's rounds after the other's).
    rc = dist_circom_h_batch_on(net, cmask, nb, qa, qb, qc, log_m, mk, seed, (Fr*)B.hshare.p, st);
    if (rc) return bail(rc);
    // the U-MSM runs on its own stream of the batch's set, behind everything queued on the caller's stream
    {
      hipError_t he = hipEventRecord(B.ev_in, st);
      if (he == hipSuccess) he = hipStreamWaitEvent(B.st[5], B.ev_in, 0);
      if (he != hipSuccess) return bail(hip_fail(he, "batch U-MSM ordering"));
    }
    rc = batch_launch_u(B, (const Fr*)B.hshare.p, per, B.st[5]);
    if (rc) return bail(rc);
    std::vector<BatchSums> mine;
    rc = batch_join(B, mine);
    if (rc) return rc;
    // d_msm's king step for the 5 nb products (dmsm/mod.rs:76-92): host messages of as many proofs as fit the payload
    std::vector<BatchSums> total = mine;
    const int per_msg = (int)(NET_PAYLOAD / sizeof(BatchSums));
    static_assert(sizeof(BatchSums) <= NET_PAYLOAD, "payload");
    std::vector<BatchSums> all((size_t)net->world * per_msg);
    for (int b0 = 0; b0 < nb; b0 += per_msg) {
      const int cnt = nb - b0 < per_msg ? nb - b0 : per_msg;
      const size_t bytes = (size_t)cnt * sizeof(BatchSums);
      rc = net_err(net, net->gather_host(3, cmask[3], mine.data() + b0, bytes, all.data()));
      if (rc) return rc;
      if (net->rank == 0)
        for (int rr = 1; rr < net->world; rr++)
          for (int i = 0; i < cnt; i++) {
            const BatchSums& o = *(const BatchSums*)((const char*)all.data() + (size_t)rr * bytes + (size_t)i * sizeof(BatchSums));
            BatchSums& t_ = total[b0 + i];
            t_.S = xyzz_add_ni(t_.S, o.S);
            t_.H = xyzz_add_ni(t_.H, o.H);
            t_.V = xyzz_add_ni(t_.V, o.V);
            t_.W = xyzz_add_ni(t_.W, o.W);
            t_.U = xyzz_add_ni(t_.U, o.U);
          }
      rc = net_err(net, net->bcast_host(3, cmask[3], total.data() + b0, bytes));
      if (rc) return rc;
    }
    for (int b = 0; b < nb; b++) {
      const ProveJob& j = *B.pj[b];
      rc = assemble_points(crs, j.r, j.s, total[b].S, total[b].H, total[b].V, total[b].W, total[b].U, mk ? &mk[b] : nullptr, k,
                           (char*)pi_a + (size_t)b * k * sizeof(Jacobian<Fq_>), (char*)pi_b + (size_t)b * k * sizeof(Jacobian<Fq2_>),
                           (char*)pi_c + (size_t)b * k * sizeof(Jacobian<Fq_>));
      if (rc) return rc;
    }
    return ZK_OK;
  }

  int ensure_streams() {
    std::lock_guard<std::mutex> lk(mu_);
    if (streams_ready_) return ZK_OK;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);     // hi = numerically lowest = highest priority
    // high priority: the G2 stream(s) and the circom_h -> U chain (short kernels the bulk accumulates would otherwise
    // starve; measured 291 vs 242 proofs/s without priorities, a third level gains nothing; keeping CUs free for the
    // chain with CU masks was measured in round 4 and dropped).  Streams 0..5: S+H, (spare), V, W, (spare), circom_h -> U.
    const char* pe = "llhlhh";
    for (int i = 0; i < 6; i++)
      ZK_HIP(hipStreamCreateWithPriority(&streams_[i], hipStreamNonBlocking, pe[i] == 'h' ? hi : lo));
    for (int i = 0; i < NJOBS; i++) {
      ZK_HIP(hipEventCreateWithFlags(&ev_in_[i], hipEventDisableTiming));
      ZK_HIP(hipEventCreateWithFlags(&ev_gate_[i], hipEventDisableTiming));
      for (int k = 0; k < 4; k++) ZK_HIP(hipEventCreateWithFlags(&ev_sorted_[i][k], hipEventDisableTiming));
    }
    // host workers: the MSM tasks block on their events while the scalar-multiple tasks run
    int nthreads = host_threads_;                        // zk_ctx_set_option("host_threads"), before the first proof
    if (nthreads <= 0) {
      unsigned hc = std::thread::hardware_concurrency();
      nthreads = hc >= 32 ? 16 : (hc >= 8 ? 8 : 4);
    }
    const int dev = device;
    pool_.reset(new HostPool(nthreads, [dev]() { (void)hipSetDevice(dev); }));
    streams_ready_ = true;
    return ZK_OK;
  }
  std::unique_ptr<HostPool> pool_;
  HostPool* host_pool() override { return pool_.get(); }
  hipEvent_t ev_in_[NJOBS] = {nullptr, nullptr};
  hipEvent_t ev_gate_[NJOBS] = {nullptr, nullptr};
  hipEvent_t ev_sorted_[NJOBS][4] = {};
  int h_first_log_m_ = 20;      // zk_ctx_set_option("h_first_log_m"): see prove_begin_impl
  int host_threads_ = 0;        // zk_ctx_set_option("host_threads"): workers of the host pool (0 = by the core count)
  hipStream_t streams_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool streams_ready_ = false;
  bool force_simple_ntt = false;
  bool ntt_attr_set_[2] = {false, false};
  bool dpp_attr_set_ = false;
  std::map<std::string, void*> base_tables_;
  DevBuf hwork_;
  Fr* pmat_ = nullptr;
  std::vector<Fr> pmat_host_;
  DevBuf flag_;   // 4-byte device flag for the validating kernels
  std::map<int, Fr*> pcoef_;
  PackL2<Fr>* pack2_ = nullptr;
  Fr* ident_ = nullptr;
  int* err_flag_ = nullptr;
  std::map<uint64_t, Fr*> umats_;
  std::map<int, Fr*> gentabs_;
  std::map<int, Fr*> sizeinv_;
  std::map<std::string, GTab> gtabs_;
  std::mutex mu_;
  DevBuf scratch_;
  DevBuf king_tmp_;
  MsmRunner<Cfg> msm_;
};

}  // namespace zk
