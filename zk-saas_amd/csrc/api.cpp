// C ABI (include/zksaas.h) -> IEngine dispatch.  No compute lives here.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/zksaas.h"
#include "engine.hpp"
#include "prng.hpp"

struct zk_ctx {
  zk::IEngine* eng;
};
struct zk_net {
  zk::Net net;
  std::string last;      // message of the last failure of a raw verb
};

using zk::IEngine;

// Kernel arguments in device memory (the HIP runtime's HIP_FORCE_DEV_KERNARG): a dependent launch then starts without the
// packet processor fetching its arguments from host memory.  A proof of the SHA-256 circuit is a chain of ~17 dependent
// launches: 600.6 against 591.2 proofs/s (four same-box pairs, round 4).  Set when the library is loaded, before the first
// HIP call of an ordinary host program initialises the runtime; never overrides the caller's own setting.
__attribute__((constructor(101))) static void zk_runtime_defaults() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

extern "C" {

const char* zk_version(void) { return "zksaas-hip 0.2 (gfx950)"; }

// the block function behind the share-randomness stream (prng.hpp), for known-answer tests of the host build
void zk_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint32_t out[16]) {
  zk::chacha20_block(key, counter, nonce, out);
}

int zk_ctx_create(int curve, int l, int device, zk_ctx** out) {
  if (!out) return ZK_ERR_BAD_INPUT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZK_ERR_NOT_CONNECTED;   // no GPU: fail loudly
  if (device < 0 || device >= ndev) return ZK_ERR_BAD_INPUT;
  IEngine* e = nullptr;
  switch (curve) {
    case ZK_BN254: e = zk::make_engine_bn254(l, device); break;
    case ZK_BLS12_381: e = zk::make_engine_bls381(l, device); break;
    case ZK_BLS12_377: e = zk::make_engine_bls377(l, device); break;
    default: return ZK_ERR_BAD_INPUT;
  }
  if (!e) return ZK_ERR_GENERIC;
  zk_ctx* c = new (std::nothrow) zk_ctx{e};
  if (!c) {
    delete e;
    return ZK_ERR_GENERIC;
  }
  *out = c;
  return e->last.code;
}

void zk_ctx_destroy(zk_ctx* ctx) {
  if (!ctx) return;
  delete ctx->eng;
  delete ctx;
}

const char* zk_last_error(zk_ctx* ctx, int* party) {
  if (!ctx) return "null context";
  if (party) *party = ctx->eng->last.party;
  return ctx->eng->last.msg.c_str();
}
int zk_ctx_n(const zk_ctx* ctx) { return ctx ? ctx->eng->n : 0; }
int zk_ctx_l(const zk_ctx* ctx) { return ctx ? ctx->eng->l : 0; }
size_t zk_fr_bytes(const zk_ctx* ctx) { return ctx ? ctx->eng->fr_bytes() : 0; }
size_t zk_fq_bytes(const zk_ctx* ctx) { return ctx ? ctx->eng->fq_bytes() : 0; }

#define CTX_OR_FAIL() \
  if (!ctx) return ZK_ERR_BAD_INPUT; \
  IEngine* e = ctx->eng; \
  (void)hipSetDevice(e->device)

int zk_malloc(zk_ctx* ctx, size_t bytes, void** out_d) {
  CTX_OR_FAIL();
  if (!out_d) return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  hipError_t h = hipMalloc(out_d, bytes ? bytes : 1);
  return h == hipSuccess ? ZK_OK : e->hip_fail(h, "hipMalloc");
}
int zk_free(zk_ctx* ctx, void* p_d) {
  // ctx may be NULL (a buffer outliving its context is still released)
  IEngine* e = ctx ? ctx->eng : nullptr;
  if (e) (void)hipSetDevice(e->device);
  // a fixed-base table is found by the ADDRESS of its base vector: drop the tables (of every context) over vectors
  // inside this allocation, or a later allocation at the same address would be multiplied through a stale table
  if (p_d) {
    hipDeviceptr_t abase = nullptr;
    size_t asize = 0;
    if (hipMemGetAddressRange(&abase, &asize, (hipDeviceptr_t)p_d) == hipSuccess && asize)
      zk::TableRegistry::inst().forget_range(abase, asize);
    else
      zk::TableRegistry::inst().forget_range(p_d, 1);
  }
  hipError_t h = hipFree(p_d);
  if (h == hipSuccess) return ZK_OK;
  return e ? e->hip_fail(h, "hipFree") : ZK_ERR_GENERIC;
}
int zk_memcpy_h2d(zk_ctx* ctx, void* dst_d, const void* src, size_t bytes, void* stream) {
  CTX_OR_FAIL();
  hipError_t h = hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, S(stream));
  if (h == hipSuccess) h = hipStreamSynchronize(S(stream));
  return h == hipSuccess ? ZK_OK : e->hip_fail(h, "hipMemcpy h2d");
}
int zk_memcpy_d2h(zk_ctx* ctx, void* dst, const void* src_d, size_t bytes, void* stream) {
  CTX_OR_FAIL();
  hipError_t h = hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, S(stream));
  if (h == hipSuccess) h = hipStreamSynchronize(S(stream));
  return h == hipSuccess ? ZK_OK : e->hip_fail(h, "hipMemcpy d2h");
}
int zk_stream_sync(zk_ctx* ctx, void* stream) {
  CTX_OR_FAIL();
  hipError_t h = hipStreamSynchronize(S(stream));
  return h == hipSuccess ? ZK_OK : e->hip_fail(h, "hipStreamSynchronize");
}

int zk_pss_pack(zk_ctx* ctx, const void* secrets_d, size_t nchunks, int order, uint64_t seed, void* shares_d,
                void* stream) {
  CTX_OR_FAIL();
  return e->pss_pack(secrets_d, nchunks, order, seed, false, shares_d, S(stream));
}
int zk_pss_det_pack(zk_ctx* ctx, const void* secrets_d, size_t nchunks, int order, void* shares_d, void* stream) {
  CTX_OR_FAIL();
  return e->pss_pack(secrets_d, nchunks, order, 0, true, shares_d, S(stream));
}
int zk_pss_unpack(zk_ctx* ctx, const void* shares_d, size_t nchunks, void* secrets_d, void* stream) {
  CTX_OR_FAIL();
  return e->pss_unpack(shares_d, nullptr, e->n, nchunks, false, secrets_d, S(stream));
}
int zk_pss_unpack2(zk_ctx* ctx, const void* shares_d, const uint32_t* parties, int nparties, size_t nchunks,
                   void* secrets_d, void* stream) {
  CTX_OR_FAIL();
  if (!parties && nparties != e->n) return e->fail(ZK_ERR_BAD_INPUT, "parties list required when some are missing");
  return e->pss_unpack(shares_d, parties, nparties, nchunks, true, secrets_d, S(stream));
}

int zk_bitrev(zk_ctx* ctx, void* x_d, int log2_len, void* stream) {
  CTX_OR_FAIL();
  return e->bitrev(x_d, log2_len, S(stream));
}
int zk_vec_add(zk_ctx* ctx, void* x_d, const void* y_d, size_t len, void* stream) {
  CTX_OR_FAIL();
  return e->vec_add(x_d, y_d, len, S(stream));
}
int zk_vec_scale(zk_ctx* ctx, void* x_d, const void* k, size_t len, void* stream) {
  CTX_OR_FAIL();
  return e->vec_scale(x_d, k, len, S(stream));
}
int zk_deg_red_parties(zk_ctx* ctx, const void* x_d, const uint32_t* parties, int nparties, const void* in_mask_d,
                       const void* out_mask_d, size_t len, uint64_t seed, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->deg_red_parties(x_d, parties, nparties, in_mask_d, out_mask_d, len, seed, out_d, S(stream));
}
int zk_d_msm_parties(zk_ctx* ctx, int group, const void* bases_d, const void* scalars_d, size_t len,
                     const uint32_t* parties, int nparties, const void* in_mask, const void* out_mask, void* out,
                     void* stream) {
  CTX_OR_FAIL();
  return e->d_msm_parties(group, bases_d, scalars_d, len, parties, nparties, in_mask, out_mask, out, S(stream));
}
int zk_vec_mul_sub(zk_ctx* ctx, void* out_d, const void* a_d, const void* b_d, const void* c_d, size_t len,
                   void* stream) {
  CTX_OR_FAIL();
  return e->vec_mul_sub(out_d, a_d, b_d, c_d, len, S(stream));
}

int zk_fq_selftest(zk_ctx* ctx, int op, const void* a_d, const void* b_d, const void* c_d, const void* d_d, size_t len,
                   void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->fq_selftest(op, a_d, b_d, c_d, d_d, len, out_d, S(stream));
}

int zk_fft1(zk_ctx* ctx, void* shares_d, int log2_m, int inverse, size_t batch, const void* add_d, void* stream) {
  CTX_OR_FAIL();
  return e->fft1(shares_d, log2_m, inverse, batch, add_d, S(stream));
}
int zk_fft2_king(zk_ctx* ctx, const void* in_d, const uint32_t* parties, int nparties, int log2_m, int inverse,
                 const void* g, int scale_size_inv, int rearrange, uint64_t seed, void* out_d,
                 const void* out_mask_d, void* stream) {
  CTX_OR_FAIL();
  if (!parties && nparties != e->n) return e->fail(ZK_ERR_BAD_INPUT, "parties list required when some are missing");
  if (in_d == out_d) return e->fail(ZK_ERR_BAD_INPUT, "zk_fft2_king cannot run in place (workgroups exchange chunks)");
  return e->fft2_king(in_d, nullptr, parties, nparties, log2_m, inverse, g, scale_size_inv, rearrange, seed, out_d,
                      out_mask_d, S(stream));
}
// d_fft (dfft/mod.rs:99-134): fft1 on every party's vector, then the king closure with the mask adds fused.
int zk_d_fft(zk_ctx* ctx, void* shares_d, const void* in_mask_d, const void* out_mask_d, int rearrange, int log2_m,
             uint64_t seed, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->d_fft(shares_d, in_mask_d, out_mask_d, rearrange, log2_m, 0, nullptr, seed, out_d, S(stream));
}
// d_ifft (dfft/mod.rs:137-175): the 1/m scaling of :159 is applied by the king together with g^i.
int zk_d_ifft(zk_ctx* ctx, void* shares_d, const void* in_mask_d, const void* out_mask_d, int rearrange, int log2_m,
              const void* g, uint64_t seed, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->d_fft(shares_d, in_mask_d, out_mask_d, rearrange, log2_m, 1, g, seed, out_d, S(stream));
}
int zk_fft_mask_sample(zk_ctx* ctx, int rearrange, const void* g, int inverse, int log2_m, uint64_t seed,
                       void* in_mask_d, void* out_mask_d, void* stream) {
  CTX_OR_FAIL();
  return e->fft_mask_sample(rearrange, g, inverse, log2_m, seed, in_mask_d, out_mask_d, S(stream));
}

int zk_deg_red(zk_ctx* ctx, void* x_d, const void* in_mask_d, const void* out_mask_d, size_t len, uint64_t seed,
               void* stream) {
  CTX_OR_FAIL();
  return e->deg_red(x_d, in_mask_d, out_mask_d, len, seed, S(stream));
}
int zk_degred_mask_sample(zk_ctx* ctx, size_t len, uint64_t seed, void* in_mask_d, void* out_mask_d, void* stream) {
  CTX_OR_FAIL();
  return e->degred_mask_sample(len, seed, in_mask_d, out_mask_d, S(stream));
}
int zk_d_pp(zk_ctx* ctx, const void* num_d, const void* den_d, const void* in_mask_d, const void* out_mask_d,
            size_t len, uint64_t seed, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->d_pp(num_d, den_d, in_mask_d, out_mask_d, len, seed, out_d, S(stream));
}

int zk_msm(zk_ctx* ctx, int group, const void* bases_d, size_t len_bases, const void* scalars_d, size_t len_scalars,
           void* out, void* stream) {
  CTX_OR_FAIL();
  return e->msm(group, bases_d, len_bases, scalars_d, len_scalars, out, S(stream));
}
int zk_d_msm(zk_ctx* ctx, int group, const void* bases_d, const void* scalars_d, size_t len, const void* in_mask,
             const void* out_mask, void* out, void* stream) {
  CTX_OR_FAIL();
  return e->d_msm(group, bases_d, scalars_d, len, in_mask, out_mask, out, S(stream));
}

// ---- host-pointer forms: the reference's functions take and return host vectors (Vec<F>, Vec<G::Affine>); these
// stage them through device scratch owned by the call (H2D, compute, D2H) so that a shim can be written without managing
// device buffers.  The device-pointer forms above are the fast path (shares resident in HBM across calls).
namespace {
struct HostStage {
  std::vector<void*> bufs;
  ~HostStage() {
    // on an error path kernels that read the staged buffers may still be enqueued: make their end explicit instead of
    // leaning on hipFree's implicit synchronisation
    if (!bufs.empty()) (void)hipDeviceSynchronize();
    for (void* p : bufs) (void)hipFree(p);
  }
  void* up(IEngine* e, const void* h, size_t bytes, hipStream_t st, int* rc) {
    void* d = nullptr;
    if (*rc) return nullptr;
    hipError_t he = hipMalloc(&d, bytes ? bytes : 1);
    if (he == hipSuccess) bufs.push_back(d);
    if (he == hipSuccess && h && bytes) he = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
    if (he != hipSuccess) *rc = e->hip_fail(he, "host staging");
    return d;
  }
};
}  // namespace
int zk_d_fft_host(zk_ctx* ctx, void* shares /* host [n][m/l], in place */, const void* in_mask, const void* out_mask,
                  int rearrange, int log2_m, int inverse, const void* g, uint64_t seed, void* stream) {
  CTX_OR_FAIL();
  if (!shares || log2_m < 0 || log2_m > 40) return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  const size_t bytes = (size_t)e->n * (((size_t)1 << log2_m) / (size_t)e->l) * e->fr_bytes();
  HostStage hs;
  int rc = ZK_OK;
  void* sd = hs.up(e, shares, bytes, S(stream), &rc);
  void* od = hs.up(e, nullptr, bytes, S(stream), &rc);
  void* im = in_mask ? hs.up(e, in_mask, bytes, S(stream), &rc) : nullptr;
  void* om = out_mask ? hs.up(e, out_mask, bytes, S(stream), &rc) : nullptr;
  if (rc) return rc;
  rc = e->d_fft(sd, im, om, rearrange, log2_m, inverse, g, seed, od, S(stream));
  if (rc) return rc;
  hipError_t he = hipMemcpyAsync(shares, od, bytes, hipMemcpyDeviceToHost, S(stream));
  if (he == hipSuccess) he = hipStreamSynchronize(S(stream));
  return he == hipSuccess ? ZK_OK : e->hip_fail(he, "host staging");
}
int zk_deg_red_host(zk_ctx* ctx, void* x /* host [n][len], in place */, const void* in_mask, const void* out_mask, size_t len,
                    uint64_t seed, void* stream) {
  CTX_OR_FAIL();
  if (!len) return ZK_OK;
  if (!x) return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  const size_t bytes = (size_t)e->n * len * e->fr_bytes();
  HostStage hs;
  int rc = ZK_OK;
  void* xd = hs.up(e, x, bytes, S(stream), &rc);
  void* im = in_mask ? hs.up(e, in_mask, bytes, S(stream), &rc) : nullptr;
  void* om = out_mask ? hs.up(e, out_mask, bytes, S(stream), &rc) : nullptr;
  if (rc) return rc;
  rc = e->deg_red(xd, im, om, len, seed, S(stream));
  if (rc) return rc;
  hipError_t he = hipMemcpyAsync(x, xd, bytes, hipMemcpyDeviceToHost, S(stream));
  if (he == hipSuccess) he = hipStreamSynchronize(S(stream));
  return he == hipSuccess ? ZK_OK : e->hip_fail(he, "host staging");
}
int zk_d_pp_host(zk_ctx* ctx, const void* num, const void* den, const void* in_mask, const void* out_mask, size_t len,
                 uint64_t seed, void* out, void* stream) {
  CTX_OR_FAIL();
  if (!len) return ZK_OK;
  if (!num || !den || !out) return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  const size_t bytes = (size_t)e->n * len * e->fr_bytes();
  HostStage hs;
  int rc = ZK_OK;
  void* nd = hs.up(e, num, bytes, S(stream), &rc);
  void* dd = hs.up(e, den, bytes, S(stream), &rc);
  void* od = hs.up(e, nullptr, bytes, S(stream), &rc);
  void* im = in_mask ? hs.up(e, in_mask, bytes, S(stream), &rc) : nullptr;
  void* om = out_mask ? hs.up(e, out_mask, bytes, S(stream), &rc) : nullptr;
  if (rc) return rc;
  rc = e->d_pp(nd, dd, im, om, len, seed, od, S(stream));
  if (rc) return rc;
  hipError_t he = hipMemcpyAsync(out, od, bytes, hipMemcpyDeviceToHost, S(stream));
  if (he == hipSuccess) he = hipStreamSynchronize(S(stream));
  return he == hipSuccess ? ZK_OK : e->hip_fail(he, "host staging");
}
namespace {
// the Fr-vector masks of a zk_groth16_masks given as HOST pointers -> device copies (the MsmMasks are host values anyway)
int stage_masks(IEngine* e, HostStage& hs, const zk_groth16_masks* in, size_t bytes, hipStream_t st, zk_groth16_masks* out) {
  int rc = ZK_OK;
  *out = *in;
  for (int i = 0; i < 6; i++) {
    out->fft_in[i] = in->fft_in[i] ? hs.up(e, in->fft_in[i], bytes, st, &rc) : nullptr;
    out->fft_out[i] = in->fft_out[i] ? hs.up(e, in->fft_out[i], bytes, st, &rc) : nullptr;
  }
  out->degred_in = in->degred_in ? hs.up(e, in->degred_in, bytes, st, &rc) : nullptr;
  out->degred_out = in->degred_out ? hs.up(e, in->degred_out, bytes, st, &rc) : nullptr;
  return rc;
}
}  // namespace
int zk_circom_h_host(zk_ctx* ctx, const void* qap_a, const void* qap_b, const void* qap_c, int log2_m,
                     const zk_groth16_masks* masks, uint64_t seed, void* h, void* stream) {
  CTX_OR_FAIL();
  if (!qap_a || !qap_b || !qap_c || !h || log2_m < 0 || log2_m > 40) return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  const size_t bytes = (size_t)e->n * (((size_t)1 << log2_m) / (size_t)e->l) * e->fr_bytes();
  HostStage hs;
  int rc = ZK_OK;
  void* a = hs.up(e, qap_a, bytes, S(stream), &rc);
  void* b = hs.up(e, qap_b, bytes, S(stream), &rc);
  void* c = hs.up(e, qap_c, bytes, S(stream), &rc);
  void* hd = hs.up(e, nullptr, bytes, S(stream), &rc);
  zk_groth16_masks md;
  if (!rc && masks) rc = stage_masks(e, hs, masks, bytes, S(stream), &md);
  if (rc) return rc;
  rc = e->circom_h(a, b, c, log2_m, masks ? &md : nullptr, seed, hd, S(stream));
  if (rc) return rc;
  hipError_t he = hipMemcpyAsync(h, hd, bytes, hipMemcpyDeviceToHost, S(stream));
  if (he == hipSuccess) he = hipStreamSynchronize(S(stream));
  return he == hipSuccess ? ZK_OK : e->hip_fail(he, "host staging");
}
int zk_groth16_prove_host(zk_ctx* ctx, const zk_crs_share* crs, const void* qap_a, const void* qap_b, const void* qap_c,
                          const void* a_share, const void* ax_share, const void* r, const void* s, int log2_m,
                          const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c, void* stream) {
  CTX_OR_FAIL();
  if (!crs || !qap_a || !qap_b || !qap_c || !a_share || !ax_share || log2_m < 0 || log2_m > 40)
    return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  // a NULL query vector with a non-zero length would be staged as uninitialised device memory
  if ((crs->len_a && (!crs->s_d || !crs->h_d || !crs->v_d)) || (crs->len_w && !crs->w_d) || (crs->len_u && !crs->u_d))
    return e->fail(ZK_ERR_BAD_INPUT, "null CRS query vector");
  const size_t n = (size_t)e->n, fr = e->fr_bytes(), g1 = 2 * e->fq_bytes(), g2 = 2 * g1;
  const size_t qbytes = n * (((size_t)1 << log2_m) / (size_t)e->l) * fr;
  HostStage hs;
  int rc = ZK_OK;
  zk_crs_share cd = *crs;
  cd.s_d = hs.up(e, crs->s_d, n * crs->len_a * g1, S(stream), &rc);
  cd.h_d = hs.up(e, crs->h_d, n * crs->len_a * g1, S(stream), &rc);
  cd.v_d = hs.up(e, crs->v_d, n * crs->len_a * g2, S(stream), &rc);
  cd.w_d = hs.up(e, crs->w_d, n * crs->len_w * g1, S(stream), &rc);
  cd.u_d = hs.up(e, crs->u_d, n * crs->len_u * g1, S(stream), &rc);
  void* a = hs.up(e, qap_a, qbytes, S(stream), &rc);
  void* b = hs.up(e, qap_b, qbytes, S(stream), &rc);
  void* c = hs.up(e, qap_c, qbytes, S(stream), &rc);
  void* as = hs.up(e, a_share, n * crs->len_a * fr, S(stream), &rc);
  void* ax = hs.up(e, ax_share, n * crs->len_w * fr, S(stream), &rc);
  zk_groth16_masks md;
  if (!rc && masks) rc = stage_masks(e, hs, masks, qbytes, S(stream), &md);
  if (rc) return rc;
  // (groth16_prove returns with the proof on the host and every kernel that read the staged buffers finished)
  return e->groth16_prove(&cd, a, b, c, as, ax, r, s, log2_m, masks ? &md : nullptr, seed, pi_a, pi_b, pi_c, S(stream));
}
int zk_msm_host(zk_ctx* ctx, int group, const void* bases /* host affine [len] */, size_t len_bases, const void* scalars,
                size_t len_scalars, void* out, void* stream) {
  CTX_OR_FAIL();
  if (group != ZK_G1 && group != ZK_G2) return e->fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
  const size_t pb = 2 * e->fq_bytes() * (group == ZK_G2 ? 2 : 1);
  HostStage hs;
  int rc = ZK_OK;
  void* bd = hs.up(e, bases, len_bases * pb, S(stream), &rc);
  void* sd = hs.up(e, scalars, len_scalars * e->fr_bytes(), S(stream), &rc);
  if (rc) return rc;
  return e->msm(group, bd, len_bases, sd, len_scalars, out, S(stream));
}
int zk_d_msm_host(zk_ctx* ctx, int group, const void* bases /* host affine [n][len] */, const void* scalars /* [n][len] */,
                  size_t len, const void* in_mask, const void* out_mask, void* out, void* stream) {
  CTX_OR_FAIL();
  if (group != ZK_G1 && group != ZK_G2) return e->fail(ZK_ERR_BAD_INPUT, "group must be ZK_G1 or ZK_G2");
  const size_t pb = 2 * e->fq_bytes() * (group == ZK_G2 ? 2 : 1);
  HostStage hs;
  int rc = ZK_OK;
  void* bd = hs.up(e, bases, (size_t)e->n * len * pb, S(stream), &rc);
  void* sd = hs.up(e, scalars, (size_t)e->n * len * e->fr_bytes(), S(stream), &rc);
  if (rc) return rc;
  return e->d_msm(group, bd, sd, len, in_mask, out_mask, out, S(stream));
}

int zk_d_msm_local(zk_ctx* ctx, int group, const void* bases_d, const void* scalars_d, size_t len, int first_party,
                   int nparties, const void* in_mask, void* out, void* stream) {
  CTX_OR_FAIL();
  return e->d_msm_local(group, bases_d, scalars_d, len, first_party, nparties, in_mask, out, S(stream));
}
int zk_group_add(zk_ctx* ctx, int group, const void* a, const void* b, void* out) {
  CTX_OR_FAIL();
  return e->group_add(group, a, b, out);
}
int zk_msm_precompute(zk_ctx* ctx, int group, const void* bases_d, size_t len, void* stream) {
  CTX_OR_FAIL();
  return e->msm_precompute(group, bases_d, len, S(stream));
}
int zk_msm_forget(zk_ctx* ctx, const void* bases_d) {
  CTX_OR_FAIL();
  return e->msm_forget(bases_d);
}
int zk_msm_table_info(zk_ctx* ctx, int group, const void* bases_d, int info[2]) {
  CTX_OR_FAIL();
  return e->msm_table_info(group, bases_d, info);
}
int zk_ctx_set_option(zk_ctx* ctx, const char* name, long long value) {
  CTX_OR_FAIL();
  return e->set_option(name, value);
}
int zk_msm_plan(zk_ctx* ctx, int group, size_t len, int plan[4]) {
  CTX_OR_FAIL();
  return e->msm_plan(group, len, plan);
}
int zk_msm_mask_sample(zk_ctx* ctx, int group, const void* gen_affine, uint64_t seed, void* in_mask, void* out_mask) {
  CTX_OR_FAIL();
  return e->msm_mask_sample(group, gen_affine, seed, in_mask, out_mask);
}
int zk_r1cs_qap(zk_ctx* ctx, const void* a_row_ptr_d, const void* a_col_d, const void* a_val_d,
                const void* b_row_ptr_d, const void* b_col_d, const void* b_val_d, const void* w_d,
                size_t num_variables, size_t num_constraints, size_t num_instance, int log_m, void* a_out_d,
                void* b_out_d, void* c_out_d, void* stream) {
  CTX_OR_FAIL();
  return e->r1cs_qap(a_row_ptr_d, a_col_d, a_val_d, b_row_ptr_d, b_col_d, b_val_d, w_d, num_variables,
                     num_constraints, num_instance, log_m, a_out_d, b_out_d, c_out_d, S(stream));
}
int zk_fr_to_bytes(zk_ctx* ctx, const void* x_d, size_t len, void* bytes_out_d, void* stream) {
  CTX_OR_FAIL();
  return e->fr_bytes(x_d, len, bytes_out_d, 0, S(stream));
}
int zk_fr_from_bytes(zk_ctx* ctx, const void* bytes_d, size_t len, void* x_out_d, void* stream) {
  CTX_OR_FAIL();
  return e->fr_bytes(bytes_d, len, x_out_d, 1, S(stream));
}
int zk_groth16_assemble(zk_ctx* ctx, const zk_crs_share* crs, const void* r, const void* s, const void* const* sums,
                        const zk_groth16_masks* masks, void* pi_a, void* pi_b, void* pi_c) {
  CTX_OR_FAIL();
  return e->groth16_assemble(crs, r, s, sums, masks, pi_a, pi_b, pi_c);
}

int zk_groth16_msms_begin(zk_ctx* ctx, const zk_crs_share* crs, const void* a_share_d, const void* ax_share_d,
                          int first_party, int nparties, int skip_h, const zk_groth16_masks* masks, void* stream) {
  CTX_OR_FAIL();
  return e->msms_begin(crs, a_share_d, ax_share_d, first_party, nparties, skip_h, masks, S(stream));
}
int zk_groth16_msms_finish(zk_ctx* ctx, const void* h_share_d, void* const* out, void* stream) {
  CTX_OR_FAIL();
  return e->msms_finish(h_share_d, out, S(stream));
}
int zk_groth16_prove_async(zk_ctx* ctx, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                           const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r,
                           const void* s, int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* stream,
                           int* handle) {
  CTX_OR_FAIL();
  return e->groth16_prove_async(crs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m, masks, seed,
                                S(stream), handle);
}
int zk_groth16_prove_batch(zk_ctx* ctx, const zk_crs_share* crs, int nproofs, const void* const* qap_a_d,
                           const void* const* qap_b_d, const void* const* qap_c_d, const void* const* a_share_d,
                           const void* const* ax_share_d, const void* r, const void* s, int log2_m,
                           const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c, void* stream) {
  CTX_OR_FAIL();
  return e->groth16_prove_batch(crs, nproofs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m, masks, seed,
                                pi_a, pi_b, pi_c, S(stream));
}
int zk_groth16_prove_batch_async(zk_ctx* ctx, const zk_crs_share* crs, int nproofs, const void* const* qap_a_d,
                                 const void* const* qap_b_d, const void* const* qap_c_d, const void* const* a_share_d,
                                 const void* const* ax_share_d, const void* r, const void* s, int log2_m,
                                 const zk_groth16_masks* masks, uint64_t seed, void* stream, int* handle) {
  CTX_OR_FAIL();
  return e->groth16_prove_batch_async(crs, nproofs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m, masks,
                                      seed, S(stream), handle);
}
int zk_groth16_batch_wait(zk_ctx* ctx, int handle, void* pi_a, void* pi_b, void* pi_c) {
  CTX_OR_FAIL();
  return e->groth16_batch_wait(handle, pi_a, pi_b, pi_c);
}
int zk_msm_batch(zk_ctx* ctx, int group, const void* bases_d, size_t len, const void* const* scalars_d, int nvec,
                 void* out, void* stream) {
  CTX_OR_FAIL();
  return e->msm_batch(group, bases_d, len, scalars_d, nvec, out, S(stream));
}
int zk_pss_unpack_points(zk_ctx* ctx, int group, const void* shares_d, size_t nchunks, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->pss_unpack_points(group, shares_d, nullptr, e->n, nchunks, 0, out_d, S(stream));
}
int zk_pss_unpack2_points(zk_ctx* ctx, int group, const void* shares_d, const uint32_t* parties, int nparties,
                          size_t nchunks, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->pss_unpack_points(group, shares_d, parties, nparties, nchunks, 1, out_d, S(stream));
}
int zk_groth16_reconstruct(zk_ctx* ctx, const void* pi_a, const void* pi_b, const void* pi_c, const uint32_t* parties,
                           int nparties, void* proof_affine, void* proof_bytes, void* stream) {
  CTX_OR_FAIL();
  return e->groth16_reconstruct(pi_a, pi_b, pi_c, parties, parties ? nparties : e->n, proof_affine, proof_bytes, S(stream));
}
int zk_groth16_wait(zk_ctx* ctx, int handle, void* pi_a, void* pi_b, void* pi_c) {
  CTX_OR_FAIL();
  return e->groth16_wait(handle, pi_a, pi_b, pi_c);
}
int zk_groth16_abort(zk_ctx* ctx, int handle) {
  CTX_OR_FAIL();
  return e->groth16_abort(handle);
}

int zk_deg_red_points(zk_ctx* ctx, int group, const void* x_d, const void* in_mask_d, const void* out_mask_d, size_t len,
                      const void* gen_affine, uint64_t seed, void* out_d, void* stream) {
  CTX_OR_FAIL();
  return e->deg_red_points(group, x_d, in_mask_d, out_mask_d, len, gen_affine, seed, out_d, S(stream));
}
int zk_degred_mask_sample_points(zk_ctx* ctx, int group, const void* gen_affine, size_t len, uint64_t seed,
                                 void* in_mask_d, void* out_mask_d, void* stream) {
  CTX_OR_FAIL();
  return e->degred_mask_sample_points(group, gen_affine, len, seed, in_mask_d, out_mask_d, S(stream));
}
int zk_points_decompress(zk_ctx* ctx, int group, const void* bytes_d, size_t len, void* out_affine_d, void* stream) {
  CTX_OR_FAIL();
  return e->points_codec(group, bytes_d, len, out_affine_d, 1, S(stream));
}
int zk_points_compress(zk_ctx* ctx, int group, const void* affine_d, size_t len, void* bytes_out_d, void* stream) {
  CTX_OR_FAIL();
  return e->points_codec(group, affine_d, len, bytes_out_d, 0, S(stream));
}
int zk_libsnark_h(zk_ctx* ctx, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                  const void* const* fft_in, const void* const* fft_out, uint64_t seed, void* h_d, void* stream) {
  CTX_OR_FAIL();
  return e->libsnark_h(qap_a_d, qap_b_d, qap_c_d, log2_m, fft_in, fft_out, seed, h_d, S(stream));
}

// ---- the star network and the per-rank collective forms (net.hpp) ----
int zk_net_unique_id(void* id_out) {
  if (!id_out) return ZK_ERR_BAD_INPUT;
  unsigned char* p = (unsigned char*)id_out;
  memset(p, 0, ZK_NET_ID_BYTES);
  std::string err;
  zk::Rccl& R = zk::Rccl::inst();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && R.load(&err)) {
    for (int s = 0; s < zk::NET_NSID; s++) {
      zk::Rccl::UniqueId uid;
      if (R.GetUniqueId(&uid) != 0) return ZK_ERR_GENERIC;
      memcpy(p + (size_t)s * 128, &uid, 128);
    }
    return ZK_OK;
  }
  // no GPU / no RCCL (host-memory tests of the protocol): a random tag names the shared-memory block
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f) return ZK_ERR_GENERIC;
  size_t got = fread(p, 1, ZK_NET_ID_BYTES, f);
  fclose(f);
  return got == ZK_NET_ID_BYTES ? ZK_OK : ZK_ERR_GENERIC;
}
int zk_net_create(zk_ctx* ctx, int transport, int rank, int world, int n_parties, const int* party_to_rank,
                  const void* id, size_t shm_bytes, zk_net** out) {
  if (!out) return ZK_ERR_BAD_INPUT;
  *out = nullptr;
  const int n = ctx ? ctx->eng->n : n_parties;
  if (n <= 0 || world <= 0 || n % world) return ZK_ERR_BAD_INPUT;
  if (ctx) (void)hipSetDevice(ctx->eng->device);
  zk_net* z = new (std::nothrow) zk_net();
  if (!z) return ZK_ERR_GENERIC;
  int rc = z->net.open(transport, rank, world, n, ctx ? ctx->eng->device : -1, ctx == nullptr, (const unsigned char*)id,
                       shm_bytes, party_to_rank);
  *out = z;               // kept on failure so that zk_net_last_error can be read; the caller destroys it
  return rc;
}
void zk_net_destroy(zk_net* net) { delete net; }
const char* zk_net_last_error(zk_net* net, int* party) {
  if (!net) return "null net";
  if (party) *party = net->net.err_party;
  return net->net.err.c_str();
}
int zk_net_set_timeout_ms(zk_net* net, uint64_t ms) {
  if (!net) return ZK_ERR_BAD_INPUT;
  net->net.timeout_ms = ms;
  return ZK_OK;
}
int zk_net_info(const zk_net* net, int info[4]) {
  if (!net || !info) return ZK_ERR_BAD_INPUT;
  info[0] = net->net.rank;
  info[1] = net->net.world;
  info[2] = net->net.first_party(net->net.rank);
  info[3] = net->net.parties_per_rank();
  return ZK_OK;
}
int zk_net_parties(const zk_net* net, int rank, int* parties) {
  if (!net || !parties || rank < 0 || rank >= net->net.world) return ZK_ERR_BAD_INPUT;
  for (int i = 0; i < net->net.parties_per_rank(); i++) parties[i] = net->net.party(rank, i);
  return ZK_OK;
}
int zk_net_enter(zk_net* net, int sid, uint32_t* mask) {
  if (!net || !mask) return ZK_ERR_BAD_INPUT;
  return net->net.enter(sid, mask);
}
int zk_net_gather(zk_net* net, int sid, uint32_t mask, const void* local, size_t bytes_per_rank, void* full) {
  if (!net || sid < 0 || sid >= zk::NET_NSID) return ZK_ERR_BAD_INPUT;
  return net->net.gather(sid, mask, local, bytes_per_rank, full);
}
int zk_net_scatter(zk_net* net, int sid, uint32_t mask, const void* full, size_t bytes_per_rank, void* local) {
  if (!net || sid < 0 || sid >= zk::NET_NSID) return ZK_ERR_BAD_INPUT;
  return net->net.scatter(sid, mask, full, bytes_per_rank, local);
}
int zk_net_stats(const zk_net* net, uint64_t stats[4]) {
  if (!net || !stats) return ZK_ERR_BAD_INPUT;
  for (int i = 0; i < 4; i++) stats[i] = net->net.stats[i].load(std::memory_order_relaxed);
  return ZK_OK;
}
int zk_net_alltoall(zk_net* net, int sid, uint32_t mask, const void* send, size_t bytes_per_peer, void* recv) {
  if (!net || sid < 0 || sid >= zk::NET_NSID) return ZK_ERR_BAD_INPUT;
  return net->net.alltoall(sid, mask, send, bytes_per_peer, recv);
}
int zk_net_gather_host(zk_net* net, int sid, uint32_t mask, const void* mine, size_t bytes, void* all) {
  if (!net || sid < 0 || sid >= zk::NET_NSID) return ZK_ERR_BAD_INPUT;
  return net->net.gather_host(sid, mask, mine, bytes, all);
}
int zk_net_bcast_host(zk_net* net, int sid, uint32_t mask, void* buf, size_t bytes) {
  if (!net || sid < 0 || sid >= zk::NET_NSID) return ZK_ERR_BAD_INPUT;
  return net->net.bcast_host(sid, mask, buf, bytes);
}
int zk_net_sync(zk_net* net, int sid) {
  if (!net || sid < 0 || sid >= zk::NET_NSID) return ZK_ERR_BAD_INPUT;
  return net->net.sync_deadline(sid);
}
#define NET_OR_FAIL()                                                         \
  CTX_OR_FAIL();                                                              \
  if (!net) return e->fail(ZK_ERR_NOT_CONNECTED, "null net");                 \
  if (net->net.host_mode) return e->fail(ZK_ERR_BAD_INPUT, "host-mode net");  \
  if (sid < 0 || sid >= zk::NET_NSID) return e->fail(ZK_ERR_BAD_INPUT, "bad channel id")
int zk_dist_d_fft(zk_ctx* ctx, zk_net* net, int sid, void* shares_d, const void* in_mask_d, const void* out_mask_d,
                  int rearrange, int log2_m, uint64_t seed, void* stream) {
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_d_fft(&net->net, sid, shares_d, in_mask_d, out_mask_d, rearrange, log2_m, 0, nullptr, seed, S(stream)));
}
int zk_dist_d_ifft(zk_ctx* ctx, zk_net* net, int sid, void* shares_d, const void* in_mask_d, const void* out_mask_d,
                   int rearrange, int log2_m, const void* g, uint64_t seed, void* stream) {
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_d_fft(&net->net, sid, shares_d, in_mask_d, out_mask_d, rearrange, log2_m, 1, g, seed, S(stream)));
}
int zk_dist_deg_red(zk_ctx* ctx, zk_net* net, int sid, void* x_d, const void* in_mask_d, const void* out_mask_d,
                    size_t len, uint64_t seed, void* stream) {
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_deg_red(&net->net, sid, x_d, in_mask_d, out_mask_d, len, seed, S(stream)));
}
int zk_dist_d_pp(zk_ctx* ctx, zk_net* net, int sid, const void* num_d, const void* den_d, const void* in_mask_d,
                 const void* out_mask_d, size_t len, uint64_t seed, void* out_d, void* stream) {
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_d_pp(&net->net, sid, num_d, den_d, in_mask_d, out_mask_d, len, seed, out_d, S(stream)));
}
int zk_dist_d_msm(zk_ctx* ctx, zk_net* net, int sid, int group, const void* bases_d, const void* scalars_d, size_t len,
                  const void* in_mask, const void* out_mask, void* out, void* stream) {
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_d_msm(&net->net, sid, group, bases_d, scalars_d, len, in_mask, out_mask, out, S(stream)));
}
int zk_dist_deg_red_points(zk_ctx* ctx, zk_net* net, int sid, int group, const void* x_d, const void* in_mask_d,
                           const void* out_mask_d, size_t len, const void* gen_affine, uint64_t seed, void* out_d,
                           void* stream) {
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_deg_red_points(&net->net, sid, group, x_d, in_mask_d, out_mask_d, len, gen_affine,
                                                          seed, out_d, S(stream)));
}
int zk_dist_libsnark_h(zk_ctx* ctx, zk_net* net, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                       const void* const* fft_in_masks, const void* const* fft_out_masks, uint64_t seed, void* h_d,
                       void* stream) {
  const int sid = 0;
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_libsnark_h(&net->net, qap_a_d, qap_b_d, qap_c_d, log2_m, fft_in_masks,
                                                      fft_out_masks, seed, h_d, S(stream)));
}
int zk_dist_circom_h(zk_ctx* ctx, zk_net* net, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                     const zk_groth16_masks* masks, uint64_t seed, void* h_d, void* stream) {
  const int sid = 0;
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_circom_h(&net->net, qap_a_d, qap_b_d, qap_c_d, log2_m, masks, seed, h_d, S(stream)));
}
int zk_dist_groth16_prove(zk_ctx* ctx, zk_net* net, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                          const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r, const void* s,
                          int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c,
                          void* stream) {
  const int sid = 0;
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_prove(&net->net, crs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m, masks, seed, pi_a,
                       pi_b, pi_c, S(stream)));
}

int zk_dist_groth16_prove_async(zk_ctx* ctx, zk_net* net, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                                const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r,
                                const void* s, int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* stream,
                                int* handle) {
  const int sid = 0;
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_prove_async(&net->net, crs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m, masks, seed,
                             S(stream), handle));
}
int zk_dist_groth16_wait(zk_ctx* ctx, zk_net* net, int handle, void* pi_a, void* pi_b, void* pi_c) {
  const int sid = 0;
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_prove_wait(&net->net, handle, pi_a, pi_b, pi_c));
}

int zk_dist_groth16_prove_batch(zk_ctx* ctx, zk_net* net, const zk_crs_share* crs, int nproofs,
                                const void* const* qap_a_d, const void* const* qap_b_d, const void* const* qap_c_d,
                                const void* const* a_share_d, const void* const* ax_share_d, const void* r, const void* s,
                                int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b,
                                void* pi_c, void* stream) {
  const int sid = 0;
  NET_OR_FAIL();
  return e->dist_finish(&net->net, e->dist_prove_batch(&net->net, crs, nproofs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m,
                             masks, seed, pi_a, pi_b, pi_c, S(stream)));
}

// ---- profiling slots (bench.py roofline leg) ----
static const char* const kSlotNames[zk::PROF_NSLOTS] = {"ntt_pass_kernel", "king_fft2_kernel", "msm_accumulate_kernel<G1>",
                                                         "msm_accumulate_kernel<G2>", "msm_digits+scan+expand",
                                                         "msm_finalize+reduce<G1>", "king_degred_kernel",
                                                         "msm_finalize+reduce<G2>", "dpp_tile_kernel", "dpp_carry_kernel",
                                                         "dpp_finish_kernel", "host:prove_launch", "host:prove_wait",
                                                         "host:prove_tail", "host:launch.submit", "host:launch.circom_h",
                                                         "host:launch.u_msm"};
int zk_profile_enable(zk_ctx* ctx, int on) {
  CTX_OR_FAIL();
  e->prof.reset();
  e->prof.on = on != 0;
  return ZK_OK;
}
int zk_msm_stats(zk_ctx* ctx, uint64_t stats[4]) {
  CTX_OR_FAIL();
  if (!stats) return e->fail(ZK_ERR_BAD_INPUT, "null pointer");
  stats[0] = e->msm_adds[0].load();
  stats[1] = e->msm_adds[1].load();
  stats[2] = e->msm_offered[0].load();
  stats[3] = e->msm_offered[1].load();
  return ZK_OK;
}
int zk_profile_slots(void) { return zk::PROF_NSLOTS; }
const char* zk_profile_name(int slot) { return slot >= 0 && slot < zk::PROF_NSLOTS ? kSlotNames[slot] : ""; }
int zk_profile_read(zk_ctx* ctx, int slot, double* total_ms, double* units, long* calls) {
  CTX_OR_FAIL();
  if (slot < 0 || slot >= zk::PROF_NSLOTS) return e->fail(ZK_ERR_BAD_INPUT, "bad slot");
  e->prof.collect();
  if (total_ms) *total_ms = e->prof.ms[slot];
  if (units) *units = e->prof.units[slot];
  if (calls) *calls = e->prof.calls[slot];
  return ZK_OK;
}

int zk_pss_pack_points(zk_ctx* ctx, int group, const void* points_d, size_t nchunks, int points_per_chunk,
                       void* shares_d, void* stream) {
  CTX_OR_FAIL();
  return e->pss_pack_points(group, points_d, nchunks, points_per_chunk, shares_d, S(stream));
}
int zk_base_mul(zk_ctx* ctx, int group, const void* base_affine, const void* scalars_d, size_t len,
                void* out_affine_d, void* stream) {
  CTX_OR_FAIL();
  return e->base_mul(group, base_affine, scalars_d, len, out_affine_d, S(stream));
}
int zk_circom_h(zk_ctx* ctx, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                const zk_groth16_masks* masks, uint64_t seed, void* h_d, void* stream) {
  CTX_OR_FAIL();
  return e->circom_h(qap_a_d, qap_b_d, qap_c_d, log2_m, masks, seed, h_d, S(stream));
}
int zk_groth16_prove(zk_ctx* ctx, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                     const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r, const void* s,
                     int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c,
                     void* stream) {
  CTX_OR_FAIL();
  return e->groth16_prove(crs, qap_a_d, qap_b_d, qap_c_d, a_share_d, ax_share_d, r, s, log2_m, masks, seed, pi_a,
                          pi_b, pi_c, S(stream));
}

}  // extern "C"
