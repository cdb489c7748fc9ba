// Host side of the library: context, cached tables, PSS matrices, kernel launches.
// One Engine<Cfg> instantiation per curve (compiled in its own translation unit).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zksaas.h"
#include "field.hpp"
#include "ntt.hpp"
#include "pss.hpp"
#include "hostpool.hpp"
#include "net.hpp"

namespace zk {

struct Status {
  int code = ZK_OK;
  int party = -1;
  std::string msg;
};

// Per-kernel timing slots: HIP events recorded on the launching stream around selected kernels, so that a
// host program can read a kernel's average launch duration over exactly its own timed region (bench.py).
enum ProfSlot { PROF_NTT_PASS = 0, PROF_KING, PROF_MSM_ACC_G1, PROF_MSM_ACC_G2, PROF_MSM_SORT, PROF_MSM_REDUCE,
                PROF_DEGRED, PROF_MSM_REDUCE_G2, PROF_DPP_TILE, PROF_DPP_CARRY, PROF_DPP_FINISH,
                // host spans of zk_groth16_prove (steady_clock, not HIP events): entry -> everything launched; then -> the event of the
                // last chain (the U-MSM); then -> return
                PROF_HOST_LAUNCH, PROF_HOST_WAIT, PROF_HOST_TAIL,
                // parts of the first: entry -> MSM tasks handed to the pool; circom_h's launches; the U-MSM's launches + host terms
                PROF_HOST_SUBMIT, PROF_HOST_H, PROF_HOST_U, PROF_NSLOTS };

struct Profiler {
  bool on = false;
  std::mutex mu;
  struct Rec {
    hipEvent_t a, b;
    int slot;
    double units;
  };
  std::vector<Rec> recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  double ms[PROF_NSLOTS] = {0};
  double units[PROF_NSLOTS] = {0};
  long calls[PROF_NSLOTS] = {0};
  hipEvent_t begin(hipStream_t st, hipEvent_t* end_out) {
    std::lock_guard<std::mutex> lk(mu);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!pool.empty()) {
      ev = pool.back();
      pool.pop_back();
    } else {
      (void)hipEventCreate(&ev.first);
      (void)hipEventCreate(&ev.second);
    }
    (void)hipEventRecord(ev.first, st);
    *end_out = ev.second;
    return ev.first;
  }
  void push(const Rec& r) {
    std::lock_guard<std::mutex> lk(mu);
    recs.push_back(r);
  }
  void host_add(int slot, double t_ms) {
    std::lock_guard<std::mutex> lk(mu);
    ms[slot] += t_ms;
    units[slot] += 1;
    calls[slot]++;
  }
  void collect() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& r : recs) {
      (void)hipEventSynchronize(r.b);
      float t = 0;
      if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
        ms[r.slot] += t;
        units[r.slot] += r.units;
        calls[r.slot]++;
      }
      pool.push_back({r.a, r.b});
    }
    recs.clear();
  }
  void reset() {
    collect();
    for (int i = 0; i < PROF_NSLOTS; i++) ms[i] = 0, units[i] = 0, calls[i] = 0;
  }
};

// RAII scope: records the two events around the launches made inside it.
struct ProfScope {
  Profiler* p;
  hipStream_t st;
  hipEvent_t a, b;
  int slot;
  double units;
  ProfScope(Profiler& pr, int slot_, hipStream_t st_, double units_) : p(pr.on ? &pr : nullptr), st(st_), slot(slot_), units(units_) {
    if (p) a = p->begin(st, &b);
  }
  ~ProfScope() {
    if (p) {
      (void)hipEventRecord(b, st);
      p->push({a, b, slot, units});
    }
  }
};

// Fixed-base tables (zk_msm_precompute) are found by the ADDRESS of the registered base vector, so the registry is
// process-wide: a zk_free issued through ANY context (or a NULL one) invalidates every table over the freed range,
// whichever context built it.  Device addresses are unique across the GPUs of a process (unified addressing).
// Lookups hand out shared ownership, so a table dropped while an MSM is in flight stays alive until that MSM ends.
// A registered vector must not be modified in place while its table exists (call zk_msm_forget first).
struct MsmTable {
  const char* base = nullptr;     // the registered affine vector [len]
  size_t len = 0, elem = 0;
  int c = 0, nwin = 0, wide = 0, bits = 0, device = 0;
  const void* owner = nullptr;    // the engine that built it (dropped with it)
  void* data = nullptr;           // [nwin][len] affine
  bool any_identity = true;       // false: no base of the vector is the identity (MSMs over it need no skip mask)
  ~MsmTable() {
    if (data) (void)hipFree(data);
  }
};
class TableRegistry {
 public:
  static TableRegistry& inst() {
    static TableRegistry r;
    return r;
  }
  std::shared_ptr<const MsmTable> find(const void* p, size_t npts, size_t elem, int bits, size_t* offset) {
    std::lock_guard<std::mutex> g(mu_);
    const char* q = (const char*)p;
    for (auto& t : v_)
      if (t->elem == elem && t->bits == bits && q >= t->base && q + npts * elem <= t->base + t->len * elem &&
          (size_t)(q - t->base) % elem == 0) {
        *offset = (size_t)(q - t->base) / elem;
        return t;
      }
    return nullptr;
  }
  void add(std::shared_ptr<MsmTable> t) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& o : v_)                              // the same registration again (another context): keep the first
      if (o->base == t->base && o->len == t->len && o->elem == t->elem && o->c == t->c && o->bits == t->bits) return;
    for (auto it = v_.begin(); it != v_.end();)     // otherwise a new table replaces every table it overlaps
      if ((*it)->base < t->base + t->len * t->elem && t->base < (*it)->base + (*it)->len * (*it)->elem) it = v_.erase(it);
      else ++it;
    v_.push_back(std::move(t));
  }
  int forget(const void* base) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto it = v_.begin(); it != v_.end(); ++it)
      if ((*it)->base == (const char*)base) {
        v_.erase(it);
        return 1;
      }
    return 0;
  }
  // drop every table whose base vector overlaps [lo, lo + bytes) (the allocation is being freed)
  void forget_range(const void* lo_, size_t bytes) {
    std::lock_guard<std::mutex> g(mu_);
    const char* lo = (const char*)lo_;
    for (auto it = v_.begin(); it != v_.end();)
      if ((*it)->base < lo + bytes && lo < (*it)->base + (*it)->len * (*it)->elem) it = v_.erase(it);
      else ++it;
  }
  void forget_owner(const void* owner) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto it = v_.begin(); it != v_.end();)
      if ((*it)->owner == owner) it = v_.erase(it);
      else ++it;
  }

 private:
  std::mutex mu_;
  std::vector<std::shared_ptr<MsmTable>> v_;
};

// Abstract interface the C ABI dispatches to (one implementation per curve).
class IEngine {
 public:
  virtual ~IEngine() {}
  // the context's host worker pool (nullptr before the first prover call): msm_fold splits a table-free fold over it
  virtual HostPool* host_pool() { return nullptr; }
  // zk_ctx_set_option("wait_deadline_ms"): no host-side wait of the library (an MSM's completion event, a gate between two
  // launches, a pool task) outlasts this; on expiry the call returns ZK_ERR_GENERIC naming what it waited for and the
  // job is aborted.  0 = unbounded.  Default 120 s (a 2^24-constraint BLS12-381 proof takes 1.3 s).
  std::atomic<int64_t> wait_deadline_ms{120000};
  std::chrono::steady_clock::time_point deadline_from_now() const {
    const int64_t ms = wait_deadline_ms.load(std::memory_order_relaxed);
    return ms > 0 ? std::chrono::steady_clock::now() + std::chrono::milliseconds(ms)
                  : std::chrono::steady_clock::time_point::max();
  }
  // hipEventSynchronize with the deadline: polls (yielding for the first 2 ms, the span of a proof's chain; then in 100 us
  // sleeps).  hipErrorNotReady = the deadline passed.
  hipError_t event_wait(hipEvent_t ev) const {
    using namespace std::chrono;
    hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    const auto t0 = steady_clock::now(), dl = deadline_from_now();
    for (;;) {
      e = hipEventQuery(ev);
      if (e != hipErrorNotReady) return e;
      const auto now = steady_clock::now();
      if (now > dl) return hipErrorNotReady;
      if (now - t0 < milliseconds(2)) std::this_thread::yield();
      else std::this_thread::sleep_for(microseconds(100));
    }
  }
  // a host-side gate (a flag another launching thread raises): false = the deadline passed
  template <class Pred>
  bool spin_until(Pred ready) const {
    if (ready()) return true;
    const auto dl = deadline_from_now();
    for (unsigned i = 0;; i++) {
      if (ready()) return true;
      std::this_thread::yield();
      if ((i & 1023) == 1023 && std::chrono::steady_clock::now() > dl) return false;
    }
  }
  Profiler prof;
  // MSM statistics since creation (zk_msm_stats): mixed additions performed and (point, window) pairs offered, G1 / G2
  std::atomic<uint64_t> msm_adds[2] = {{0}, {0}}, msm_offered[2] = {{0}, {0}};
  int l = 0, n = 0, t = 0, device = 0;
  Status last;
  std::mutex last_mu;               // host worker tasks report failures too
  int fail(int code, const std::string& m, int party = -1) {
    std::lock_guard<std::mutex> lk(last_mu);
    last.code = code;
    last.msg = m;
    last.party = party;
    return code;
  }
  int hip_fail(hipError_t e, const char* where) {
    return fail(ZK_ERR_GENERIC, std::string(where) + ": " + hipGetErrorString(e));
  }
  virtual size_t fr_bytes() const = 0;
  virtual size_t fq_bytes() const = 0;
  virtual int pss_pack(const void* secrets, size_t nchunks, int order, uint64_t seed, bool det, void* shares,
                       hipStream_t st) = 0;
  virtual int pss_unpack(const void* shares, const uint32_t* parties, int np, size_t nchunks, bool two, void* secrets,
                         hipStream_t st) = 0;
  virtual int bitrev(void* x, int log_len, hipStream_t st) = 0;
  virtual int vec_add(void* x, const void* y, size_t len, hipStream_t st) = 0;
  virtual int vec_scale(void* x, const void* k, size_t len, hipStream_t st) = 0;
  virtual int vec_mul_sub(void* out, const void* a, const void* b, const void* c, size_t len, hipStream_t st) = 0;
  virtual int fq_selftest(int op, const void* a, const void* b, const void* c, const void* d, size_t len, void* out,
                          hipStream_t st) = 0;
  virtual int fft1(void* shares, int log_m, int inverse, size_t batch, const void* add, hipStream_t st) = 0;
  virtual int fft2_king(const void* in, const void* in_mask, const uint32_t* parties, int np, int log_m, int inverse,
                        const void* g, int scale_size_inv, int rearrange, uint64_t seed, void* out,
                        const void* out_mask, hipStream_t st) = 0;
  virtual int d_fft(void* shares, const void* in_mask, const void* out_mask, int rearrange, int log_m, int inverse,
                    const void* g, uint64_t seed, void* out, hipStream_t st) = 0;
  virtual int fft_mask_sample(int rearrange, const void* g, int inverse, int log_m, uint64_t seed, void* in_mask,
                              void* out_mask, hipStream_t st) = 0;
  virtual int deg_red(void* x, const void* in_mask, const void* out_mask, size_t len, uint64_t seed,
                      hipStream_t st) = 0;
  virtual int deg_red_parties(const void* x, const uint32_t* parties, int np, const void* in_mask,
                              const void* out_mask, size_t len, uint64_t seed, void* out, hipStream_t st) = 0;
  virtual int d_msm_parties(int group, const void* bases, const void* scalars, size_t len, const uint32_t* parties,
                            int np, const void* in_mask, const void* out_mask, void* out, hipStream_t st) = 0;
  virtual int degred_mask_sample(size_t len, uint64_t seed, void* in_mask, void* out_mask, hipStream_t st) = 0;
  virtual int d_pp(const void* num, const void* den, const void* in_mask, const void* out_mask, size_t len,
                   uint64_t seed, void* out, hipStream_t st) = 0;
  virtual int msm(int group, const void* bases, size_t nb, const void* scalars, size_t ns, void* out,
                  hipStream_t st) = 0;
  virtual int d_msm(int group, const void* bases, const void* scalars, size_t len, const void* in_mask,
                    const void* out_mask, void* out, hipStream_t st) = 0;
  virtual int d_msm_local(int group, const void* bases, const void* scalars, size_t len, int first_party, int nparties,
                          const void* in_mask, void* out, hipStream_t st) = 0;
  virtual int group_add(int group, const void* a, const void* b, void* out) = 0;
  virtual int msm_plan(int group, size_t len, int* plan) = 0;
  virtual int set_option(const char* name, long long value) = 0;
  // exit of every zk_dist_* entry point: with the option "dist_deadline" the call returns only when the channels'
  // data-plane work has completed, or fails with ZK_ERR_PROTOCOL once the net's timeout has passed (ser_net.rs:122-125)
  virtual int dist_finish(Net* net, int rc) = 0;
  virtual int dist_deg_red_points(Net* net, int sid, int group, const void* x, const void* in_mask, const void* out_mask,
                                  size_t len, const void* gen_affine, uint64_t seed, void* out, hipStream_t st) = 0;
  virtual int dist_libsnark_h(Net* net, const void* qa, const void* qb, const void* qc, int log_m, const void* const* fft_in,
                              const void* const* fft_out, uint64_t seed, void* h, hipStream_t st) = 0;
  virtual int msm_precompute(int group, const void* bases, size_t len, hipStream_t st) = 0;
  virtual int msm_forget(const void* bases) = 0;
  virtual int msm_table_info(int group, const void* bases, int* info) = 0;
  virtual int msm_mask_sample(int group, const void* gen_affine, uint64_t seed, void* in_mask, void* out_mask) = 0;
  virtual int r1cs_qap(const void* pa, const void* ca, const void* va, const void* pb, const void* cb, const void* vb,
                       const void* w, size_t nvars, size_t nc, size_t ni, int log_m, void* a, void* b, void* c,
                       hipStream_t st) = 0;
  virtual int fr_bytes(const void* in, size_t len, void* out, int from_bytes, hipStream_t st) = 0;
  virtual int groth16_assemble(const zk_crs_share* crs, const void* r, const void* s, const void* const* sums,
                               const zk_groth16_masks* masks, void* pi_a, void* pi_b, void* pi_c) = 0;
  virtual int msms_begin(const zk_crs_share* crs, const void* a_share, const void* ax_share, int first, int count,
                         int skip_h, const zk_groth16_masks* masks, hipStream_t st) = 0;
  virtual int msms_finish(const void* h_share, void* const* out, hipStream_t st) = 0;
  virtual int groth16_prove_async(const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                                  const void* a_share, const void* ax_share, const void* r, const void* s, int log_m,
                                  const zk_groth16_masks* masks, uint64_t seed, hipStream_t st, int* handle) = 0;
  virtual int groth16_wait(int handle, void* pi_a, void* pi_b, void* pi_c) = 0;
  virtual int groth16_prove_batch(const zk_crs_share* crs, int nb, const void* const* qa, const void* const* qb,
                                  const void* const* qc, const void* const* a_share, const void* const* ax_share,
                                  const void* r, const void* s, int log_m, const zk_groth16_masks* masks, uint64_t seed,
                                  void* pi_a, void* pi_b, void* pi_c, hipStream_t st) = 0;
  virtual int pss_unpack_points(int group, const void* shares, const uint32_t* parties, int np, size_t nchunks, int two,
                                void* out, hipStream_t st) = 0;
  virtual int groth16_reconstruct(const void* pi_a, const void* pi_b, const void* pi_c, const uint32_t* parties, int np,
                                  void* proof_affine, void* proof_bytes, hipStream_t st) = 0;
  virtual int groth16_prove_batch_async(const zk_crs_share* crs, int nb, const void* const* qa, const void* const* qb,
                                        const void* const* qc, const void* const* a_share,
                                        const void* const* ax_share, const void* r, const void* s, int log_m,
                                        const zk_groth16_masks* masks, uint64_t seed, hipStream_t st, int* handle) = 0;
  virtual int groth16_batch_wait(int handle, void* pi_a, void* pi_b, void* pi_c) = 0;
  virtual int msm_batch(int group, const void* bases, size_t len, const void* const* scalars, int nb, void* out,
                        hipStream_t st) = 0;
  virtual int groth16_abort(int handle) = 0;
  virtual int deg_red_points(int group, const void* x, const void* in_mask, const void* out_mask, size_t len,
                             const void* gen_affine, uint64_t seed, void* out, hipStream_t st) = 0;
  virtual int degred_mask_sample_points(int group, const void* gen_affine, size_t len, uint64_t seed, void* in_mask,
                                        void* out_mask, hipStream_t st) = 0;
  virtual int points_codec(int group, const void* in, size_t len, void* out, int decompress, hipStream_t st) = 0;
  virtual int libsnark_h(const void* qa, const void* qb, const void* qc, int log_m, const void* const* fft_in,
                         const void* const* fft_out, uint64_t seed, void* h, hipStream_t st) = 0;
  // per-rank collective forms (net.hpp): this rank's k = n / world parties, king = rank 0
  virtual int dist_d_fft(Net* net, int sid, void* shares, const void* in_mask, const void* out_mask, int rearrange,
                         int log_m, int inverse, const void* g, uint64_t seed, hipStream_t st) = 0;
  virtual int dist_deg_red(Net* net, int sid, void* x, const void* in_mask, const void* out_mask, size_t len,
                           uint64_t seed, hipStream_t st) = 0;
  virtual int dist_d_pp(Net* net, int sid, const void* num, const void* den, const void* in_mask, const void* out_mask,
                        size_t len, uint64_t seed, void* out, hipStream_t st) = 0;
  virtual int dist_d_msm(Net* net, int sid, int group, const void* bases, const void* scalars, size_t len,
                         const void* in_mask, const void* out_mask, void* out, hipStream_t st) = 0;
  virtual int dist_circom_h(Net* net, const void* qa, const void* qb, const void* qc, int log_m,
                            const zk_groth16_masks* masks, uint64_t seed, void* h, hipStream_t st) = 0;
  virtual int dist_prove(Net* net, const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                         const void* a_share, const void* ax_share, const void* r, const void* s, int log_m,
                         const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c,
                         hipStream_t st) = 0;
  virtual int dist_prove_async(Net* net, const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                               const void* a_share, const void* ax_share, const void* r, const void* s, int log_m,
                               const zk_groth16_masks* masks, uint64_t seed, hipStream_t st, int* handle) = 0;
  virtual int dist_prove_wait(Net* net, int handle, void* pi_a, void* pi_b, void* pi_c) = 0;
  virtual int dist_prove_batch(Net* net, const zk_crs_share* crs, int nb, const void* const* qa, const void* const* qb,
                               const void* const* qc, const void* const* a_share, const void* const* ax_share,
                               const void* r, const void* s, int log_m, const zk_groth16_masks* masks, uint64_t seed,
                               void* pi_a, void* pi_b, void* pi_c, hipStream_t st) = 0;
  virtual int pss_pack_points(int group, const void* points, size_t nchunks, int nv, void* shares, hipStream_t st) = 0;
  virtual int base_mul(int group, const void* base_affine, const void* scalars, size_t len, void* out_affine,
                       hipStream_t st) = 0;
  virtual int circom_h(const void* qa, const void* qb, const void* qc, int log_m, const zk_groth16_masks* masks,
                       uint64_t seed, void* h, hipStream_t st) = 0;
  virtual int groth16_prove(const zk_crs_share* crs, const void* qa, const void* qb, const void* qc,
                            const void* a_share, const void* ax_share, const void* r, const void* s, int log_m,
                            const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c,
                            hipStream_t st) = 0;
};

IEngine* make_engine_bn254(int l, int device);
IEngine* make_engine_bls381(int l, int device);
IEngine* make_engine_bls377(int l, int device);

#define ZK_HIP(expr)                                  \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return hip_fail(_e, #expr); \
  } while (0)

inline int ilog2(size_t x) {
  int r = 0;
  while (((size_t)1 << r) < x) r++;
  return r;
}

// Simple device buffer owner.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t ensure(size_t b) {
    if (b <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, b);
    if (e == hipSuccess) bytes = b;
    return e;
  }
};

}  // namespace zk
