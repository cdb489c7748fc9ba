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
    if (pjsf_) (void)hipFree(pjsf_);
    if (pack2_) (void)hipFree(pack2_);
    if (ident_) (void)hipFree(ident_);
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

  // group-element types the fragments share
  using Fq_ = Fp<typename Cfg::FqP>;
  using Fq2_ = Fp2<typename Cfg::FqP>;
  using P1 = XYZZ<Fq_>;
  using P2 = XYZZ<Fq2_>;
  // ---- the class body continues in four fragments, by primitive (round 5) ----
#include "engine_fft.inc.hpp"        // PSS, fft1, king of d_fft, deg_red, d_pp
#include "engine_points.inc.hpp"     // MSM entry points, group-element PSS / deg_red, front end, wire formats, options
#include "engine_groth16.inc.hpp"    // circom_h / libsnark_h, prover, batch prover
#include "engine_dist.inc.hpp"       // zk_dist_*: per-rank collective forms

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
  bool dpp_attr_set_[2] = {false, false};
  std::map<std::string, void*> base_tables_;
  Fr* pmat_ = nullptr;
  std::vector<Fr> pmat_host_;
  DevBuf flag_;   // 4-byte device flag for the validating kernels
  std::map<int, Fr*> pcoef_;
  uint8_t* pjsf_ = nullptr;     // joint-sparse-form digits of the first two pack columns (pss_pack_points at 2 points per chunk)
  int pjsf_len_ = 0;            // ... their columns per party, and whether they are the endomorphism-split form
  bool pjsf_glv_ = false, pack_glv_ = true;     // zk_ctx_set_option("pack_glv")
  PackL2<Fr>* pack2_ = nullptr;
  Fr* ident_ = nullptr;
  std::map<uint64_t, Fr*> umats_;
  std::map<int, Fr*> gentabs_;
  std::map<int, Fr*> sizeinv_;
  std::map<std::string, GTab> gtabs_;
  std::mutex mu_;
  // Working memory of the all-parties-on-one-GPU entry points, ONE SET PER CALLER STREAM (round 6; round 5 had one per context
  // and documented "concurrent transforms are undefined", against SURVEY.md 8b: up to three d_ifft / d_fft run at once on
  // three stream ids, ext_wit.rs:127-159).  Calls on one stream are ordered by the stream, so they may share a set; calls on
  // different streams -- the three channels of a rank, a host that overlaps two transforms -- never meet.  `err`: the
  // zero-denominator word of d_pp.  Growth frees the old buffer with hipFree, which waits for the device.
  struct StreamWs {
    DevBuf scratch, king_tmp, hwork;
    int* err = nullptr;
    ~StreamWs() {
      if (err) (void)hipFree(err);
    }
  };
  std::mutex ws_mu_;
  std::map<hipStream_t, std::unique_ptr<StreamWs>> ws_;
  StreamWs* ws(hipStream_t st) {
    std::lock_guard<std::mutex> lk(ws_mu_);
    std::unique_ptr<StreamWs>& w = ws_[st];
    if (!w) {
      w.reset(new StreamWs());
      // [0] zero-denominator flag, [1] ready flag, [8 .. 8 + N) the inverse's limbs (dpp_carry_kernel's hand-off)
      if (hipMalloc((void**)&w->err, 64 * sizeof(int)) != hipSuccess) w->err = nullptr;
      else (void)hipMemset(w->err, 0, 64 * sizeof(int));
    }
    return w.get();
  }
  MsmRunner<Cfg> msm_;
};

}  // namespace zk
