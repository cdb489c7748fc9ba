/* zksaas.h -- C ABI of the MI355X-native zkSaaS hot path (libzksaas_hip.so).
 *
 * The reference (tangle-network/zk-SaaS, pure Rust) has no FFI seam; the seam is the set of generic
 * Rust functions listed below.  Every entry point here cites the reference function it
 * replaces, so that a thin Rust shim can forward the same arguments.
 *
 * Data layout at the boundary = arkworks in-memory representation:
 *   Fr / Fq      : little-endian u64 limbs in MONTGOMERY form, fully reduced
 *                  (4 limbs: BN254 Fr/Fq, BLS12-381 Fr, BLS12-377 Fr; 6 limbs: BLS12-381/377 Fq).
 *   affine point : x || y (Fq2: c0 || c1), (0,0) = identity sentinel.
 *   group value  : Jacobian X || Y || Z, Z = 0 = identity.
 *   share vectors: what one party holds, `Vec<F>` of length m/l; all-party buffers are [n][m/l]
 *                  (party-major, party p's vector contiguous).
 *
 * Pointers whose name ends in _d are DEVICE pointers (hipMalloc / torch tensors); everything else is
 * host memory.  `stream` is a hipStream_t (NULL = default stream).  All functions return a zk_status;
 * zk_last_error() returns the message and, for ZK_ERR_PROTOCOL, the offending party
 * (mpc-net/src/lib.rs:19-24 MpcNetError).  Nothing here falls back to a CPU implementation.
 */
#ifndef ZKSAAS_H
#define ZKSAAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;

enum zk_curve { ZK_BN254 = 0, ZK_BLS12_381 = 1, ZK_BLS12_377 = 2 };
enum zk_group { ZK_G1 = 1, ZK_G2 = 2 };
/* mpc-net/src/lib.rs:19-24 */
enum zk_status { ZK_OK = 0, ZK_ERR_GENERIC = 1, ZK_ERR_PROTOCOL = 2, ZK_ERR_NOT_CONNECTED = 3, ZK_ERR_BAD_INPUT = 4 };

/* ---- context ------------------------------------------------------------------------------------
 * PackedSharingParams::new(l) (secret-sharing/src/pss.rs:39-66): n = 4l parties, t = l, share domain
 * H_n, secret domain g*H_{l+t}, secret2 domain g*H_{2(l+t)}.  `device` is the HIP device ordinal. */
int zk_ctx_create(int curve, int l, int device, zk_ctx** out);
void zk_ctx_destroy(zk_ctx* ctx);
const char* zk_last_error(zk_ctx* ctx, int* party);
int zk_ctx_n(const zk_ctx* ctx);        /* pp.n */
int zk_ctx_l(const zk_ctx* ctx);        /* pp.l */
size_t zk_fr_bytes(const zk_ctx* ctx);  /* 32 */
size_t zk_fq_bytes(const zk_ctx* ctx);  /* 32 or 48 */
const char* zk_version(void);

/* ---- device memory helpers (so that non-torch hosts can drive the library) ---------------------- */
int zk_malloc(zk_ctx* ctx, size_t bytes, void** out_d);
int zk_free(zk_ctx* ctx, void* p_d);
int zk_memcpy_h2d(zk_ctx* ctx, void* dst_d, const void* src, size_t bytes, void* stream);
int zk_memcpy_d2h(zk_ctx* ctx, void* dst, const void* src_d, size_t bytes, void* stream);
int zk_stream_sync(zk_ctx* ctx, void* stream);

/* ---- share randomness ----------------------------------------------------------------------------------------
 * The t random points of every `pack` (pss.rs:90-122; the reference draws them from thread_rng / test_rng:
 * dfft/mod.rs:251, pack.rs:14, deg_red.rs:108) come from a ChaCha20 stream keyed per context from the operating
 * system's generator; every launch that packs gets fresh nonces, so NO `seed` argument below influences them and two
 * calls never share randomness.  zk_ctx_set_option("rng_replay", 1) (a context option only: nothing in the
 * environment can turn it on) switches to the documented replayable generator (DESIGN.md "Randomness": SplitMix64 over
 * (seed, index)) that the parity tests use to compare shares bit for bit with the oracle -- only then do the `seed`
 * arguments matter.  zk_chacha20_block: the block function (RFC 7539 2.3, words 12..15 = counter, nonce), exported
 * for known-answer tests. */
void zk_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint32_t out[16]);

/* ---- packed secret sharing over Fr (secret-sharing/src/pss.rs) ------------------------------------
 * `order`: 0 = chunk j packs secrets[j*l .. j*l+l-1]  (pack_vec, dist-primitives/src/utils/pack.rs:8-20)
 *          1 = chunk j packs secrets[j], secrets[j+nchunks], ...  (stride packing, dfft/mod.rs:286-299,
 *              groth16/src/qap.rs:103-112)
 * Share randomness: counter-based PRNG documented in DESIGN.md (chunk j uses indices j*t..j*t+t-1 of
 * stream `seed`).  Output shares_d is [n][nchunks]. */
int zk_pss_pack(zk_ctx* ctx, const void* secrets_d, size_t nchunks, int order, uint64_t seed, void* shares_d,
                void* stream);                                                   /* pss.rs:90-122 pack     */
int zk_pss_det_pack(zk_ctx* ctx, const void* secrets_d, size_t nchunks, int order, void* shares_d,
                    void* stream);                                               /* pss.rs:69-87 det_pack  */
/* shares_d is [nparties][nchunks] for the listed parties (ascending ids); secrets_d gets nchunks*l values in
 * order 0.  unpack requires all n parties (pss.rs:125-138); unpack2 falls back to lagrange_unpack when
 * nparties < n (pss.rs:141-221 unpack2 / lagrange_unpack / unpack_missing_shares). */
int zk_pss_unpack(zk_ctx* ctx, const void* shares_d, size_t nchunks, void* secrets_d, void* stream);
int zk_pss_unpack2(zk_ctx* ctx, const void* shares_d, const uint32_t* parties, int nparties, size_t nchunks,
                   void* secrets_d, void* stream);

/* ---- vector helpers ------------------------------------------------------------------------------ */
int zk_bitrev(zk_ctx* ctx, void* x_d, int log2_len, void* stream);  /* dfft/mod.rs:322-335 fft_in_place_rearrange */
int zk_vec_add(zk_ctx* ctx, void* x_d, const void* y_d, size_t len, void* stream);              /* x += y      */
int zk_vec_scale(zk_ctx* ctx, void* x_d, const void* k, size_t len, void* stream);   /* x *= k (k: Montgomery Fr, host);
                                                                    the 1/Z(g) factor of libsnark_h, ext_wit.rs:78-87 */
int zk_vec_mul_sub(zk_ctx* ctx, void* out_d, const void* a_d, const void* b_d, const void* c_d, size_t len,
                   void* stream);                                   /* out = a*b - c, groth16/src/ext_wit.rs:173-177 */

/* ---- d_fft / d_ifft (dist-primitives/src/dfft/mod.rs) --------------------------------------------
 * zk_fft1: fft1_in_place (:178-208) on `batch` share vectors of length m/l each, in place; `inverse`
 *          selects gen = group_gen_inv.  If add_d != NULL it is added element-wise afterwards (the
 *          `share + in_mask` of :254-258 fused into the last pass).
 * zk_fft2_king: the king closure of fft2_with_rearrange (:264-304): unpack_missing_shares per chunk ->
 *          fft2_in_place (:210-237) -> distribute_powers(g) (:278-280) -> (bit-reverse + stride) pack.
 *          in_d is [nparties][m/l], out_d is [n][m/l] and must not alias in_d.  g (Montgomery Fr, host pointer) may be NULL for 1.
 *          scale_size_inv != 0 additionally multiplies by 1/m (d_ifft's :159 folded in).
 * zk_d_fft / zk_d_ifft: :99-175 for all n parties resident on this device: shares_d [n][m/l], masks [n][m/l] each or
 *          NULL for FftMask::zero.  The result is written to out_d [n][m/l] (shares_d is then left as it was);
 *          out_d == NULL or == shares_d returns it in shares_d.  The local stages run out of place into a
 *          working vector and the king step writes the destination, so neither form costs a copy.
 *          RE-ENTRANT ACROSS STREAMS (round 6): the working memory of zk_d_fft / zk_d_ifft / zk_libsnark_h / zk_circom_h /
 *          zk_d_pp / zk_fft_mask_sample / zk_degred_mask_sample (and of the zk_dist_* forms, per channel) belongs to the
 *          `stream` the call is issued on -- calls on one stream are ordered by it and share a set, calls on different
 *          streams never meet (the reference runs three d_ifft at once on three stream ids, groth16/src/ext_wit.rs:127-159;
 *          tests/test_gpu_hardening.py runs two streams against the serial results).  d_pp's zero-denominator word is per
 *          stream as well.  The point kernels (zk_pss_*_points, zk_deg_red_points, zk_groth16_reconstruct) keep
 *          per-context scratch: order them on one stream. */
int zk_fft1(zk_ctx* ctx, void* shares_d, int log2_m, int inverse, size_t batch, const void* add_d, void* stream);
/* Parity-test access to the BASE-field primitives of the group kernels (arkworks' Fq / Fq2 arithmetic, a22; Montgomery Fq
 * elements on the device): op 0: out[i] = a[i] b[i] - c[i] d[i] (the one-reduction form used for Y3 of every XYZZ
 * formula); op 1: out[2i], out[2i+1] = (a[i] + b[i] u)(c[i] + d[i] u), u^2 = -1 (the Fq2 product of the G2 kernels: three
 * unreduced products and two reductions on 8-limb curves); op 2 + k, k < 16: the lazy-residue forms ([0, 2p), field.hpp)
 * the G1 accumulate kernel keeps its running sums in, operand j entered as x + p when bit j of k is set: out[5i .. 5i+4] =
 * a b, a - b, 2a, a b - c d (canonical; all-ones if a result left [0, 2p)) and the raw word
 * (a == c mod p) | (a == 0 mod p) << 1 (out_d holds 5 len elements). */
int zk_fq_selftest(zk_ctx* ctx, int op, const void* a_d, const void* b_d, const void* c_d, const void* d_d, size_t len,
                   void* out_d, void* stream);
int zk_fft2_king(zk_ctx* ctx, const void* in_d, const uint32_t* parties, int nparties, int log2_m, int inverse,
                 const void* g, int scale_size_inv, int rearrange, uint64_t seed, void* out_d,
                 const void* out_mask_d, void* stream);
int zk_d_fft(zk_ctx* ctx, void* shares_d, const void* in_mask_d, const void* out_mask_d, int rearrange,
             int log2_m, uint64_t seed, void* out_d, void* stream);
int zk_d_ifft(zk_ctx* ctx, void* shares_d, const void* in_mask_d, const void* out_mask_d, int rearrange,
              int log2_m, const void* g, uint64_t seed, void* out_d, void* stream);
/* FftMask::sample (:30-85): in_mask_d/out_mask_d are [n][m/l]. */
int zk_fft_mask_sample(zk_ctx* ctx, int rearrange, const void* g, int inverse, int log2_m, uint64_t seed,
                       void* in_mask_d, void* out_mask_d, void* stream);

/* ---- deg_red / d_pp over Fr (dist-primitives/src/utils/deg_red.rs:80-126, dpp/mod.rs:15-87) ------
 * x_d [n][len] in place; masks [n][len] or NULL. */
int zk_deg_red(zk_ctx* ctx, void* x_d, const void* in_mask_d, const void* out_mask_d, size_t len, uint64_t seed,
               void* stream);
/* The same when only `nparties` parties' vectors reached the king (mpc-net/src/ser_net.rs:57-94): x_d is
 * [nparties][len] for the listed ascending ids (masks, if any, in the same order for in_mask and [n][len] for
 * out_mask); the king reconstructs through lagrange_unpack (pss.rs:170-221) and every one of the n parties gets a
 * fresh share in out_d [n][len]. */
int zk_deg_red_parties(zk_ctx* ctx, const void* x_d, const uint32_t* parties, int nparties, const void* in_mask_d,
                       const void* out_mask_d, size_t len, uint64_t seed, void* out_d, void* stream);
int zk_degred_mask_sample(zk_ctx* ctx, size_t len, uint64_t seed, void* in_mask_d, void* out_mask_d,
                          void* stream);                              /* deg_red.rs:40-66 with gen = 1 */
/* num_d, den_d [n][len]; out_d [n][len].  Returns ZK_ERR_GENERIC if a reconstructed denominator is zero
 * (the reference panics on inverse().unwrap(), dpp/mod.rs:55). */
int zk_d_pp(zk_ctx* ctx, const void* num_d, const void* den_d, const void* in_mask_d, const void* out_mask_d,
            size_t len, uint64_t seed, void* out_d, void* stream);

/* deg_red over GROUP elements (deg_red.rs:80-126 with T = G; DegRedMask::sample with a group generator, :40-66):
 * x_d, masks, out_d are [n][len] affine points (out_d must not alias x_d); gen_affine (host) = the generator whose
 * random multiples serve as the t fresh random points of the king's re-pack and as mask values. */
int zk_deg_red_points(zk_ctx* ctx, int group, const void* x_d, const void* in_mask_d, const void* out_mask_d, size_t len,
                      const void* gen_affine, uint64_t seed, void* out_d, void* stream);
int zk_degred_mask_sample_points(zk_ctx* ctx, int group, const void* gen_affine, size_t len, uint64_t seed,
                                 void* in_mask_d, void* out_mask_d, void* stream);

/* ---- MSM / d_msm (dist-primitives/src/dmsm/mod.rs) ---------------------------------------------------
 * zk_msm: G::msm(bases, scalars) (:73, ark-ec VariableBaseMSM): bases affine, scalars Montgomery Fr;
 *         out (HOST pointer) receives one Jacobian point.  `len_bases != len_scalars` -> ZK_ERR_GENERIC
 *         carrying min(len) like arkworks' Err(usize).
 * zk_d_msm: :59-102 for all n parties on this device: bases_d [n][len], scalars_d [n][len];
 *         in_mask/out_mask: n Jacobian points each (host) or NULL for MsmMask::zero; out: n Jacobian points. */
int zk_msm(zk_ctx* ctx, int group, const void* bases_d, size_t len_bases, const void* scalars_d,
           size_t len_scalars, void* out, void* stream);
int zk_d_msm(zk_ctx* ctx, int group, const void* bases_d, const void* scalars_d, size_t len, const void* in_mask,
             const void* out_mask, void* out, void* stream);

/* ---- dealer: fixed-base multiplication -----------------------------------------------------------------
 * out_affine_d[i] = scalars_d[i] * Base for one affine base point (host pointer).  With the trapdoor this is how
 * CRS elements are produced, and -- because det_pack is linear -- how PackedProvingKeyShare::
 * pack_from_arkworks_proving_key (groth16/src/proving_key.rs:47-123) is evaluated: det_pack the discrete logs
 * with zk_pss_det_pack, then multiply the base. */
/* pack / det_pack over GROUP elements (pss.rs:69-122 with T = curve point): points_d is [nchunks][points_per_chunk]
 * affine with points_per_chunk = l (det_pack: PackedProvingKeyShare::pack_from_arkworks_proving_key,
 * proving_key.rs:72-86) or l + t (pack with caller-supplied random points: MsmMask::sample, dmsm/mod.rs:34-38);
 * shares_d is [n][nchunks] affine.  No trapdoor is needed (compare zk_base_mul). */
int zk_pss_pack_points(zk_ctx* ctx, int group, const void* points_d, size_t nchunks, int points_per_chunk,
                       void* shares_d, void* stream);
int zk_base_mul(zk_ctx* ctx, int group, const void* base_affine, const void* scalars_d, size_t len,
                void* out_affine_d, void* stream);

/* ---- Groth16 composition (groth16/src/ext_wit.rs, prove.rs, examples/sha256.rs:32-129) -------------------
 * All n parties' shares live on this device.  Mask slots may be NULL (the *::zero() masks).
 * fft masks 0..2: the three d_ifft (a, b, c), 3..5: the three d_fft (ext_wit.rs:127-170), each [n][m/l];
 * degred: [n][m/l]; msm masks in the order A (S), B-in-G1 (H), B-in-G2 (V), C.w (W), C.u (U): n Jacobian
 * points each, host memory (sha256.rs:226-291). */
typedef struct zk_groth16_masks {
  const void* fft_in[6];
  const void* fft_out[6];
  const void* degred_in;
  const void* degred_out;
  const void* msm_in[5];
  const void* msm_out[5];
} zk_groth16_masks;

/* PackedProvingKeyShare for all parties (groth16/src/proving_key.rs:18-37): share vectors are device
 * buffers [n][len] of affine points, the single elements are host affine points. */
typedef struct zk_crs_share {
  const void* s_d;   /* a_query[1..]      G1 [n][len_a] */
  const void* h_d;   /* b_g1_query[1..]   G1 [n][len_a] */
  const void* v_d;   /* b_g2_query[1..]   G2 [n][len_a] */
  const void* w_d;   /* l_query           G1 [n][len_w] */
  const void* u_d;   /* h_query           G1 [n][len_u], len_u = m/l */
  size_t len_a, len_w, len_u;
  const void *a_query0, *b_g1_query0, *delta_g1, *alpha_g1, *beta_g1; /* G1 */
  const void *b_g2_query0, *delta_g2, *beta_g2;                       /* G2 */
} zk_crs_share;

/* circom_h (ext_wit.rs:104-181): qap_*_d [n][m/l] are not modified; h_d [n][m/l]. */
int zk_circom_h(zk_ctx* ctx, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                const zk_groth16_masks* masks, uint64_t seed, void* h_d, void* stream);
/* dsha256 (sha256.rs:32-129) for all parties: a_share_d [n][len_a] = shares of assignment[1..], ax_share_d
 * [n][len_w] = shares of the aux assignment; r, s: Montgomery Fr (host) -- every party holds r and s in the
 * clear.  pi_a / pi_c: n Jacobian G1 points, pi_b: n Jacobian G2 points (host). */
int zk_groth16_prove(zk_ctx* ctx, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                     const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r, const void* s,
                     int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c,
                     void* stream);

/* ---- multi-GPU building blocks (the n parties spread over several ranks, king = rank 0) ------------------
 * zk_d_msm_local: this rank's share of d_msm's king step (dmsm/mod.rs:85-86): sum over its `nparties` parties
 *   (ids first_party ..) of coef_p * (G::msm(bases_p, scalars_p) + in_mask_p), coef_p = sum_k unpack2 row k.
 *   bases_d / scalars_d are [nparties][len]; in_mask: nparties Jacobian points (host) or NULL; out: one
 *   Jacobian point (host).  Summing the ranks' outputs (zk_group_add) gives the king's broadcast value.
 * zk_groth16_assemble: prove.rs:40-56, 99-110, 148-158, 229-235 from the five summed d_msm values
 *   sums[0..4] = S, H, V (G2), W, U (Jacobian, host). */
int zk_d_msm_local(zk_ctx* ctx, int group, const void* bases_d, const void* scalars_d, size_t len, int first_party,
                   int nparties, const void* in_mask, void* out, void* stream);
int zk_group_add(zk_ctx* ctx, int group, const void* a, const void* b, void* out);
/* Introspection (no reference counterpart; ark-ec picks its window inside VariableBaseMSM::msm): the Pippenger plan
 * zk_msm uses for `len` points of `group`: plan[0] = widest window in bits, plan[1] = number of windows,
 * plan[2] = points per accumulate lane, plan[3] = base-field multiplications per mixed addition (10 for G1,
 * 28 for G2 over Fq2).  Benchmarks use it to turn a launch duration into multiplications per second. */
int zk_msm_plan(zk_ctx* ctx, int group, size_t len, int plan[4]);
/* Fixed-base tables (no reference counterpart; arkworks has FixedBase::msm for setup only).  A prover service proves
 * many witnesses against one CRS: zk_msm_precompute builds, for the affine vector bases_d [len] (e.g. one query of
 * zk_crs_share, all parties: len = n * len_a), the multiples 2^(16 j) * P_i, j < 16, owned by the context (16 x the
 * size of the vector).  Every later zk_msm / zk_d_msm / zk_groth16_prove whose base pointer lies inside a registered
 * vector uses its table: all windows then share one bucket set (one bucket reduction, no doublings in the final
 * fold, 16 instead of 20-22 mixed additions per point).  Results are the same group elements.
 * zk_free drops the tables of vectors inside the freed allocation; for memory the caller frees itself (e.g. a torch
 * tensor) call zk_msm_forget first -- tables are found by address.
 * zk_msm_forget drops the table of bases_d (returns ZK_ERR_BAD_INPUT if there is none); zk_msm_table_info writes
 * info[0] = window bits, info[1] = digit windows of the table that covers bases_d (zeros if none). */
int zk_msm_precompute(zk_ctx* ctx, int group, const void* bases_d, size_t len, void* stream);
int zk_msm_forget(zk_ctx* ctx, const void* bases_d);
int zk_msm_table_info(zk_ctx* ctx, int group, const void* bases_d, int info[2]);
/* Tunables of this context (no reference counterpart).  "rng_replay": see "share randomness".  "msm_bigsort_min": point count from which zk_msm sorts with
 * the two-level LDS counting sort instead of global atomics (default 196608: below that the tiles of the two-level sort are too few to fill the chip; tests force both paths with it).
 * "msm_table_c": window bits (8..22, 0 = default; "msm_table_c_g2" sets G2 alone) of tables built by later zk_msm_precompute calls.
 *   Default: by the vector's length -- G1 15 bits up to 2^17 points, 16 below 2^19, 17 below 2^22, 20 from there; G2 15 below 2^20, 19 from
 *   there (a table folds all windows into one bucket set: too few buckets for a long vector sends every bucket through the
 *   heavy-bucket path; csrc/msm.hpp table_c_auto).
 * "king_alltoall": 1 = the zk_dist_* king rounds of d_fft / d_ifft / deg_red (and everything composed of them) run as
 * all-to-all: every present rank is king of a contiguous chunk range (zk_net_alltoall twice per round) instead of
 * gather -> rank 0 -> scatter; identical results; every rank of a net must choose alike (default 0).
 * "msm_c" / "msm_c_g2": window bits (2..20, 0 = the cost model) of table-free MSMs (tests force widths with it).
 * "h_first_log_m": domains of 2^value and up run circom_h and every MSM's sort ahead of the accumulate kernels (default 20).
 * "host_threads": workers of the context's host pool (0 = by the core count, else >= 4; before the first proof).
 * "wait_deadline_ms": bound of every host-side wait inside the prover entry points (an MSM chain's completion event, a gate
 *   between two launching threads, a pool task; default 120000, 0 = unbounded).  On expiry the call returns ZK_ERR_GENERIC
 *   naming what it waited for, the state of the job's gates and events goes to stderr, the job's slot is never reused and
 *   the context refuses further proofs ("wedged": destroy it).  zk_groth16_wait / zk_groth16_batch_wait / zk_msm and the
 *   zk_dist_* prover calls therefore cannot block for ever (the net's own rounds have zk_net_set_timeout_ms).
 * "dist_deadline": 1 = the zk_dist_* calls return only with their data-plane work done (zk_net_sync inside).
 * "pack_glv": 0 = zk_pss_pack_points at two points per chunk walks the parties' full-length scalars; 1 (default) = the
 *   scalars are split by the curve's endomorphism phi(x, y) = (beta x, y) = lambda (x, y) into two half-length parts (half
 *   the doubling chain; the same points).  Before the first zk_pss_pack_points of the context.
 * "msm_skip_kernel": 1 = identity bases are found by a mask kernel of its own ahead of every sort (round 5's form) instead of
 *   by the first sort kernel looking at the bases (default 0; same results, measured equal: profiles/r06_skipfold_ab.txt).
 * "msm_sort_lo_tab": low bucket bits per bin (4..10, 0 = the default 7) of a small sort over a fixed-base table, i.e. 2^(c-1-value)
 *   bins / bin-sort workgroups (A/B only; 6 and 8 measured against 7 on the SHA-256 proof: no gain, profiles/r06_sort_bins_ab.txt).
 * "msm_acc_lds": dynamic LDS bytes (0..65536, default 0) launched with every accumulate workgroup, which caps how many of
 *   them a CU holds (measured on the SHA-256 proof: a loss at every size, profiles/r06_acc_lds_sweep.txt; kept for A/B runs).
 * Nothing is read from the environment.  Unknown name or value out of range -> ZK_ERR_BAD_INPUT. */
int zk_ctx_set_option(zk_ctx* ctx, const char* name, long long value);
/* Memory-model note (csrc/msm.hpp msm_hist): the last workgroup of a sort's histogram finds out that it is the last through a
 * RELAXED device-scope ticket (an acquire fence follows in that workgroup; the release side would write back an XCD's L2 per
 * workgroup: -3.8 % of a SHA-256 proof, measured).  Correctness rests on gfx950 behaviour, not on the HIP memory model: the
 * counts are device-scope atomic read-modify-writes performed at the memory side, every wave holds the return values of its
 * adds before the barrier that precedes the ticket, and the scanning workgroup reads them with device-scope atomic loads.
 * -DZK_HIST_TICKET_ACQ_REL=1 builds the formally ordered form; tests/test_gpu_msm.py::test_concurrent_sorts_stress runs
 * hundreds of concurrent sorts against serial results on the shipped build. */
/* MsmMask::sample (dmsm/mod.rs:21-47): l random scalars x_i (stream `seed`), mask values x_i * gen, out value
 * -(sum), both packed with t random group elements each (streams seed^0x1111, seed^0x2222; a random group element
 * is a random multiple of gen).  gen_affine: the group generator (host, affine Montgomery); in_mask / out_mask
 * (HOST) receive n Jacobian points each. */
int zk_msm_mask_sample(zk_ctx* ctx, int group, const void* gen_affine, uint64_t seed, void* in_mask, void* out_mask);

/* ---- circom front end (what the reference takes from ark-circom + groth16/src/qap.rs) ------------------
 * zk_r1cs_qap: qap.rs:42-89 (`qap::<F, D>()` for the circom reduction): A and B as CSR on the device
 *   (row_ptr [num_constraints+1] u32, col u32 wire indices, val Montgomery Fr), w_d the full assignment
 *   [num_variables]; writes a, b, c [2^log_m] in natural order: a_i = <A_i,w>, b_i = <B_i,w>, c_i = a_i*b_i,
 *   a[nc .. nc+ni) = w[0 .. ni), zero padding.  2^log_m < nc + ni -> ZK_ERR_BAD_INPUT; a wire index
 *   >= num_variables -> ZK_ERR_GENERIC.
 * zk_fr_to_bytes / zk_fr_from_bytes: ark-serialize CanonicalSerialize / CanonicalDeserialize of Fr elements
 *   (little-endian canonical integers, the payload of mpc-net frames, ser_net.rs:24-25) for device vectors;
 *   from_bytes returns ZK_ERR_GENERIC when an element is >= the modulus (arkworks: InvalidData). */
int zk_r1cs_qap(zk_ctx* ctx, const void* a_row_ptr_d, const void* a_col_d, const void* a_val_d,
                const void* b_row_ptr_d, const void* b_col_d, const void* b_val_d, const void* w_d,
                size_t num_variables, size_t num_constraints, size_t num_instance, int log_m, void* a_out_d,
                void* b_out_d, void* c_out_d, void* stream);
int zk_fr_to_bytes(zk_ctx* ctx, const void* x_d, size_t len, void* bytes_out_d, void* stream);
int zk_fr_from_bytes(zk_ctx* ctx, const void* bytes_d, size_t len, void* x_out_d, void* stream);
/* Vectors of group elements in ark-serialize's COMPRESSED form (what CRS shares / MSM results look like on an mpc-net
 * wire, ser_net.rs:111-120) <-> affine Montgomery points on the device: len x (|Fq| bytes for G1, 2 |Fq| for G2).
 * BN254 / BLS12-377: ark-ec's default flags (little-endian x, bit 7 / 6 of the last byte = y is the larger root /
 * infinity); BLS12-381: the zcash encoding ark-bls12-381 uses (big-endian, c1 || c0, flags in the first byte).
 * Decompression validates like arkworks (x below the modulus, on the curve, consistent flags) and returns
 * ZK_ERR_GENERIC naming the first bad index.  Square roots: the (q + 1) / 4 power for q = 3 mod 4 (BN254, BLS12-381),
 * Tonelli-Shanks for BLS12-377 (q - 1 = 2^46 t). */
int zk_points_decompress(zk_ctx* ctx, int group, const void* bytes_d, size_t len, void* out_affine_d, void* stream);
int zk_points_compress(zk_ctx* ctx, int group, const void* affine_d, size_t len, void* bytes_out_d, void* stream);
/* libsnark_h (ext_wit.rs:14-102) for all parties on this device: fft_in / fft_out: 7 mask pointers each ([n][m/l]
 * device buffers, NULL entries or NULL arrays = FftMask::zero) in the order 3 x d_ifft, 3 x d_fft, final d_ifft. */
int zk_libsnark_h(zk_ctx* ctx, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                  const void* const* fft_in, const void* const* fft_out, uint64_t seed, void* h_d, void* stream);
/* The five zk_d_msm_local of the prover for this rank's parties, overlapped: _begin starts S, H, V, W (they only
 * need the witness shares; crs vectors are [nparties][len] here) on internal streams and returns; _finish runs U
 * on `stream` once h_share_d [nparties][m/l] is available, joins, and writes out[0..4] = S, H, V(G2), W, U
 * (Jacobian, host).  skip_h != 0 when r = 0 (prove.rs:96-98).  masks (optional): msm_in[k] = this rank's nparties
 * in-mask points of MSM k (the other members are ignored here).  The internal streams wait for everything already
 * queued on `stream` (the shares may still be in flight there).  crs must stay valid until _finish returns. */
int zk_groth16_msms_begin(zk_ctx* ctx, const zk_crs_share* crs, const void* a_share_d, const void* ax_share_d,
                          int first_party, int nparties, int skip_h, const zk_groth16_masks* masks, void* stream);
int zk_groth16_msms_finish(zk_ctx* ctx, const void* h_share_d, void* const* out, void* stream);
/* Asynchronous form of zk_groth16_prove (what `tokio::spawn(dsha256(..))` is to the reference, multi.rs:317-327):
 * _async enqueues the whole proof -- device pipelines on internal streams, host-side terms on the context's worker
 * pool -- and returns a handle without waiting; _wait joins and writes the shares.  Up to two proofs may be in
 * flight per context (each has its own scratch); inputs must stay valid and unmodified until _wait returns.
 * _abort joins and discards a proof in flight (also safe after a failed call); it also releases a handle handed out
 * by zk_dist_groth16_prove_async (same handle space), which every rank must then abort alike. */
int zk_groth16_prove_async(zk_ctx* ctx, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                           const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r,
                           const void* s, int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* stream,
                           int* handle);
int zk_groth16_wait(zk_ctx* ctx, int handle, void* pi_a, void* pi_b, void* pi_c);
int zk_groth16_abort(zk_ctx* ctx, int handle);
int zk_groth16_assemble(zk_ctx* ctx, const zk_crs_share* crs, const void* r, const void* s, const void* const* sums,
                        const zk_groth16_masks* masks, void* pi_a, void* pi_b, void* pi_c);
/* A BATCH of proofs against one packed CRS in one pass -- what a proving service does with the reference by spawning
 * concurrent dsha256 runs (mpc-net/src/multi.rs:317-327 runs the parties as concurrent tasks; groth16/examples/
 * sha256.rs:316-360 is one such run): nproofs (1..16) witnesses, each with its own QAP shares, witness shares, (r, s)
 * and masks, same CRS and domain.  Every one of the five d_msm is ONE sort / accumulate / reduce chain over nproofs
 * scalar vectors against the shared base vector, circom_h one launch chain over 3 nproofs vectors; the results are the
 * group elements nproofs calls of zk_groth16_prove give.  Pointer arrays are host arrays of device pointers; r, s:
 * nproofs Montgomery Fr each (host, contiguous); masks: nproofs entries or NULL; pi_a / pi_c: [nproofs][n] Jacobian G1,
 * pi_b: [nproofs][n] Jacobian G2 (host).  Replay mode: proof b draws the share randomness zk_groth16_prove draws with
 * seed + 16 b. */
int zk_groth16_prove_batch(zk_ctx* ctx, const zk_crs_share* crs, int nproofs, const void* const* qap_a_d,
                           const void* const* qap_b_d, const void* const* qap_c_d, const void* const* a_share_d,
                           const void* const* ax_share_d, const void* r, const void* s, int log2_m,
                           const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c, void* stream);
/* Asynchronous form: _async enqueues the batch (device pipelines on the batch's own stream set, host terms on the worker
 * pool) and returns a handle; zk_groth16_batch_wait joins it and writes the shares.  Up to THREE batches may be in flight
 * per context, each with its own scratch: the sorts of one run under the accumulate kernels of the other (two already
 * saturate the chip: 814 / 813 proofs/s with two / three batches of 8 in flight).  Inputs must
 * stay valid and unmodified until _wait returns (the pointer ARRAYS are copied and may be reused at once). */
int zk_groth16_prove_batch_async(zk_ctx* ctx, const zk_crs_share* crs, int nproofs, const void* const* qap_a_d,
                                 const void* const* qap_b_d, const void* const* qap_c_d, const void* const* a_share_d,
                                 const void* const* ax_share_d, const void* r, const void* s, int log2_m,
                                 const zk_groth16_masks* masks, uint64_t seed, void* stream, int* handle);
int zk_groth16_batch_wait(zk_ctx* ctx, int handle, void* pi_a, void* pi_b, void* pi_c);
/* G::msm of ONE base vector bases_d [len] against nvec (1..16) scalar vectors scalars_d[v] [len] (dmsm/mod.rs:73 called
 * once per witness): one Pippenger pass with bucket sets per scalar vector.  out: nvec Jacobian points (host). */
int zk_msm_batch(zk_ctx* ctx, int group, const void* bases_d, size_t len, const void* const* scalars_d, int nvec,
                 void* out, void* stream);
/* unpack / unpack2 over GROUP elements (secret-sharing/src/pss.rs:125-166 with T = curve point; lagrange_unpack
 * :170-221 for a party subset): shares_d [nparties][nchunks] affine points of the listed parties (ascending ids, NULL =
 * 0..nparties-1) -> out_d [nchunks][l] affine.  zk_pss_unpack_points needs all n parties. */
int zk_pss_unpack_points(zk_ctx* ctx, int group, const void* shares_d, size_t nchunks, void* out_d, void* stream);
int zk_pss_unpack2_points(zk_ctx* ctx, int group, const void* shares_d, const uint32_t* parties, int nparties,
                          size_t nchunks, void* out_d, void* stream);
/* The last step of the reference's run (groth16/examples/sha256.rs:375-377): (a, b, c) = pp.unpack2(shares)[0] over the
 * parties' proof shares.  pi_a / pi_c: nparties Jacobian G1, pi_b: nparties Jacobian G2 (host) of the listed parties
 * (ascending ids, NULL = all n; a subset goes through lagrange_unpack and needs more than 2 (t + l - 1) of them).
 * proof_affine (optional, host): A (G1) | B (G2) | C (G1) as affine Montgomery points; proof_bytes (optional, host):
 * ark_groth16::Proof::serialize_compressed, 4 |Fq| bytes (128 for BN254; BLS12-381: 192 in the zcash form). */
int zk_groth16_reconstruct(zk_ctx* ctx, const void* pi_a, const void* pi_b, const void* pi_c, const uint32_t* parties,
                           int nparties, void* proof_affine, void* proof_bytes, void* stream);

/* ---- the star network on one multi-GPU node (mpc-net/src/lib.rs:43-53, 89-176 `MpcNet`; ser_net.rs:16-120) -------
 * One process per GPU; rank rho drives k = n / world parties: with party_to_rank == NULL the block [rho*k, (rho+1)*k),
 * otherwise the parties p with party_to_rank[p] == rho (MpcNet ids are arbitrary, lib.rs:43-53; any map that gives every
 * rank k parties is accepted; a rank's rows are in ascending party order, zk_net_parties lists them).  The king's work is
 * done by rank 0.  The king kernels always see the present parties' rows in ascending party order: with a
 * non-contiguous map gather / scatter move party rows one by one, and the all-to-all king falls back to the star.
 * `sid` = MultiplexedStreamID: channels 0..2 may be in flight at once (ext_wit.rs:158-170), calls on one channel are
 * ordered (multi.rs:418-445); channel 3 is used internally by zk_dist_groth16_prove.
 * Transports: ZK_NET_RCCL (ncclSend / ncclRecv over xGMI on device buffers), ZK_NET_SHM (staged through POSIX shared
 * memory: tests, several ranks on one GPU; with ctx == NULL the buffers of the raw verbs are HOST memory),
 * ZK_NET_LOCAL (world = 1).
 * Timeouts (lib.rs:98-135, ser_net.rs:57-94,122-125): a rank that does not enter a round within the timeout (default
 * 30 s, zk_net_set_timeout_ms) is left out of it; the king continues with the remaining parties through
 * lagrange_unpack when enough remain (else ZK_ERR_PROTOCOL "Not enough shares"), the late rank's call returns
 * ZK_ERR_PROTOCOL with its first party id.  zk_dist_groth16_prove needs every party.
 * id: ZK_NET_ID_BYTES made by zk_net_unique_id on rank 0 and handed to every rank by the launcher. */
typedef struct zk_net zk_net;
enum zk_transport { ZK_NET_LOCAL = 0, ZK_NET_RCCL = 1, ZK_NET_SHM = 2 };
#define ZK_NET_ID_BYTES 512
int zk_net_unique_id(void* id_out);
int zk_net_create(zk_ctx* ctx, int transport, int rank, int world, int n_parties /* used when ctx == NULL */,
                  const int* party_to_rank, const void* id, size_t shm_bytes_per_channel, zk_net** out);
void zk_net_destroy(zk_net* net);
const char* zk_net_last_error(zk_net* net, int* party);
int zk_net_set_timeout_ms(zk_net* net, uint64_t ms);
int zk_net_info(const zk_net* net, int info[4]);          /* rank, world, first party, parties per rank */
int zk_net_parties(const zk_net* net, int rank, int* parties /* k ids, ascending */);
int zk_net_stats(const zk_net* net, uint64_t stats[4]);   /* since creation: gathers, scatters, all-to-alls, bytes this rank sent */
/* raw verbs (what the primitives below are made of; exposed for hosts that compose their own rounds).
 * zk_net_enter: join the next round on `sid`; *mask = ranks taking part.  gather = client_send_or_king_receive
 * (lib.rs:89-135): bytes_per_rank from every rank, the king's `full` receives the present ranks' blocks compacted in
 * rank order.  scatter = client_receive_or_king_send (:137-176): rank r receives block r of the king's `full`.
 * *_host move small host values (<= 4096 bytes) through the control block.  Transfers on a channel are enqueued on
 * the channel's stream; zk_net_sync waits for it with the timeout as deadline.
 * zk_net_alltoall (no counterpart in mpc-net, whose topology is a star: the exchange of the all-to-all king, option
 * "king_alltoall"): the block for rank r is read at send + r*bytes_per_peer (indexed by RANK), the block from the i-th
 * PRESENT rank lands at recv + i*bytes_per_peer (compacted, like gather).
 * Waiting: the zk_dist_* calls return with their device work enqueued and make the caller's stream wait for the channel
 * streams.  A host that wants the timeout to hold for the data plane as well (a peer that dies mid-round, a hung RCCL
 * collective) calls zk_net_sync(net, sid) BEFORE synchronising its own stream: zk_net_sync waits with the timeout as a
 * deadline, aborts the communicators on expiry and returns ZK_ERR_PROTOCOL; a plain stream synchronise would block
 * behind the hung collective forever.  After a PROTOCOL / NOT_CONNECTED result the net must be destroyed and rebuilt. */
int zk_net_enter(zk_net* net, int sid, uint32_t* mask);
int zk_net_gather(zk_net* net, int sid, uint32_t mask, const void* local, size_t bytes_per_rank, void* full);
int zk_net_scatter(zk_net* net, int sid, uint32_t mask, const void* full, size_t bytes_per_rank, void* local);
int zk_net_alltoall(zk_net* net, int sid, uint32_t mask, const void* send, size_t bytes_per_peer, void* recv);
int zk_net_gather_host(zk_net* net, int sid, uint32_t mask, const void* mine, size_t bytes, void* all);
int zk_net_bcast_host(zk_net* net, int sid, uint32_t mask, void* buf, size_t bytes);
int zk_net_sync(zk_net* net, int sid);

/* ---- the reference's entry points as they are CALLED there: per party (here: per rank = its k parties), with a net
 * and a stream id, collectively by all ranks.  Buffers hold THIS RANK's rows: shares [k][m/l], masks [k][m/l] (NULL =
 * the *::zero() mask), MSM masks / outputs k Jacobian points (host).
 *   d_fft(pd_shares, rearrange, &FftMask, pp, &net, sid)            dfft/mod.rs:99-134   -> zk_dist_d_fft
 *   d_ifft(pd_shares, rearrange, &FftMask, dom, g, pp, &net, sid)   dfft/mod.rs:137-175  -> zk_dist_d_ifft
 *   deg_red(px, pp, &DegRedMask, &net, sid)                         deg_red.rs:80-126    -> zk_dist_deg_red
 *   d_pp(num, den, &DegRedMask, pp, &net, sid)                      dpp/mod.rs:15-87     -> zk_dist_d_pp
 *   d_msm(bases, scalars, &MsmMask, pp, &net, sid)                  dmsm/mod.rs:59-102   -> zk_dist_d_msm
 *   circom_h(qap_share, pp, masks.., &net)  (channels 0..2)         ext_wit.rs:104-181   -> zk_dist_circom_h
 *   dsha256(pp, crs_share, qap_share, a_share, ax_share, .., &net)  sha256.rs:32-129     -> zk_dist_groth16_prove */
int zk_dist_d_fft(zk_ctx* ctx, zk_net* net, int sid, void* shares_d, const void* in_mask_d, const void* out_mask_d,
                  int rearrange, int log2_m, uint64_t seed, void* stream);
int zk_dist_d_ifft(zk_ctx* ctx, zk_net* net, int sid, void* shares_d, const void* in_mask_d, const void* out_mask_d,
                   int rearrange, int log2_m, const void* g, uint64_t seed, void* stream);
int zk_dist_deg_red(zk_ctx* ctx, zk_net* net, int sid, void* x_d, const void* in_mask_d, const void* out_mask_d,
                    size_t len, uint64_t seed, void* stream);
int zk_dist_d_pp(zk_ctx* ctx, zk_net* net, int sid, const void* num_d, const void* den_d, const void* in_mask_d,
                 const void* out_mask_d, size_t len, uint64_t seed, void* out_d, void* stream);
int zk_dist_d_msm(zk_ctx* ctx, zk_net* net, int sid, int group, const void* bases_d, const void* scalars_d, size_t len,
                  const void* in_mask, const void* out_mask, void* out, void* stream);
/* deg_red over GROUP elements per rank (dist-primitives/src/utils/deg_red.rs:80-126 is generic over T: DomainCoeff<F> = F or
 * G and takes net, sid): x_d, masks, out_d are this rank's k parties' rows [k][len] of affine points (out_d must not alias
 * x_d); gen_affine as in zk_deg_red_points.  A rank the round left out is handled like for field elements (the king's
 * unpack2 becomes the Lagrange form over the present parties). */
int zk_dist_deg_red_points(zk_ctx* ctx, zk_net* net, int sid, int group, const void* x_d, const void* in_mask_d,
                           const void* out_mask_d, size_t len, const void* gen_affine, uint64_t seed, void* out_d,
                           void* stream);
/* libsnark_h per rank (groth16/src/ext_wit.rs:14-102): three d_ifft with the coset shift F::GENERATOR joined on channels
 * 0..2, three d_fft likewise, (a b - c) / Z(g), d_ifft with g^-1 on channel 0.  qap_*_d, h_d and the seven masks
 * (arrays of 7 pointers, or NULL for FftMask::zero) are this rank's rows [k][m/l]. */
int zk_dist_libsnark_h(zk_ctx* ctx, zk_net* net, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                       const void* const* fft_in_masks, const void* const* fft_out_masks, uint64_t seed, void* h_d,
                       void* stream);
int zk_dist_circom_h(zk_ctx* ctx, zk_net* net, const void* qap_a_d, const void* qap_b_d, const void* qap_c_d, int log2_m,
                     const zk_groth16_masks* masks, uint64_t seed, void* h_d, void* stream);
int zk_dist_groth16_prove(zk_ctx* ctx, zk_net* net, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                          const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r, const void* s,
                          int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c,
                          void* stream);

/* The sharded prover in two halves (the reference's parties are concurrent tasks, mpc-net/src/multi.rs:317-327, and
 * prove.rs:209-227 joins the W and U d_msm): zk_dist_groth16_prove_async admits the proof, starts this rank's four witness
 * MSMs, runs circom_h's king rounds on channels 0..2 and queues the U-MSM behind them -- it returns with that work
 * enqueued; zk_dist_groth16_wait joins it, exchanges the five partial sums with the king on channel 3 and writes the k
 * proof shares.  Up to TWO proofs may be in flight per rank; every rank must issue the same sequence of calls (the
 * channels are ordered): async(A), async(B), wait(A), async(C), wait(B) ... overlaps a proof's king rounds with the
 * previous proof's MSMs.  zk_dist_groth16_prove = async + wait.  Shares stay valid until the wait; masks are read at
 * async time (the struct is copied, the rows it points to must stay valid until the wait). */
int zk_dist_groth16_prove_async(zk_ctx* ctx, zk_net* net, const zk_crs_share* crs, const void* qap_a_d, const void* qap_b_d,
                                const void* qap_c_d, const void* a_share_d, const void* ax_share_d, const void* r,
                                const void* s, int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* stream,
                                int* handle);
int zk_dist_groth16_wait(zk_ctx* ctx, zk_net* net, int handle, void* pi_a, void* pi_b, void* pi_c);

/* Throughput mode of the sharded prover: nproofs (1..16) proofs per collective call -- what a service does with the
 * reference by running several dsha256 instances over one mesh (mpc-net/src/multi.rs:317-327; groth16/examples/
 * sha256.rs:316-360).  ONE round of the control plane admits the batch; every rank runs each of its five d_msm once
 * over the nproofs witnesses (zk_groth16_prove_batch's batched Pippenger) beside the king rounds of the proofs'
 * circom_h; the partial sums of the batch cross in one host message per rank.  Pointer arrays (host arrays of device
 * pointers) hold THIS RANK's rows of every proof; r, s: nproofs Montgomery Fr each; masks: nproofs entries (this rank's
 * rows) or NULL; pi_a / pi_c: [nproofs][k] Jacobian G1, pi_b: [nproofs][k] Jacobian G2 (host).  Results equal nproofs
 * calls of zk_dist_groth16_prove; in replay mode proof b draws the randomness of seed + 16 b. */
int zk_dist_groth16_prove_batch(zk_ctx* ctx, zk_net* net, const zk_crs_share* crs, int nproofs,
                                const void* const* qap_a_d, const void* const* qap_b_d, const void* const* qap_c_d,
                                const void* const* a_share_d, const void* const* ax_share_d, const void* r, const void* s,
                                int log2_m, const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b,
                                void* pi_c, void* stream);

/* ---- host-pointer forms of the primitives (the reference's own signatures take and return host vectors: dfft/mod.rs:99,
 * dmsm/mod.rs:59): the call stages its operands through device scratch (H2D, compute, D2H) and returns when the result is
 * in host memory.  zk_d_fft_host: shares [n][m/l] in place, masks host [n][m/l] or NULL, inverse != 0 = d_ifft with the
 * coset element g (Montgomery Fr, host, or NULL).  zk_msm_host / zk_d_msm_host: affine bases and scalars in host memory,
 * results as zk_msm / zk_d_msm.  The device-pointer forms stay the fast path: a 2^20 d_fft moves 2 x 128 MiB over PCIe
 * (~4.3 ms at 63 GB/s) around 0.87 ms of compute. */
int zk_d_fft_host(zk_ctx* ctx, void* shares, const void* in_mask, const void* out_mask, int rearrange, int log2_m,
                  int inverse, const void* g, uint64_t seed, void* stream);
/* The same staging for the remaining functions whose reference signatures take host vectors (round 4):
 * zk_deg_red_host      dist-primitives/src/utils/deg_red.rs:80 (px: Vec<T>): x [n][len] in place, masks host [n][len] or NULL
 * zk_d_pp_host         dist-primitives/src/dpp/mod.rs:15 (num, den: Vec<F>): out [n][len] host
 * zk_circom_h_host     groth16/src/ext_wit.rs:104 (PackedQAPShare of Vec<F>): qap_* and h host [n][m/l]; the Fr-vector members
 *                      of `masks` (fft_in / fft_out / degred_*) are HOST pointers here
 * zk_groth16_prove_host  groth16/examples/sha256.rs:32 (dsha256 over host vectors): the five share vectors of `crs`, the QAP and
 *                      witness shares and the Fr-vector masks are HOST pointers; the single CRS elements, r, s and the MsmMasks are
 *                      host values as in zk_groth16_prove.  PCIe-inclusive: never what bench.py times. */
int zk_deg_red_host(zk_ctx* ctx, void* x, const void* in_mask, const void* out_mask, size_t len, uint64_t seed, void* stream);
int zk_d_pp_host(zk_ctx* ctx, const void* num, const void* den, const void* in_mask, const void* out_mask, size_t len,
                 uint64_t seed, void* out, void* stream);
int zk_circom_h_host(zk_ctx* ctx, const void* qap_a, const void* qap_b, const void* qap_c, int log2_m,
                     const zk_groth16_masks* masks, uint64_t seed, void* h, void* stream);
int zk_groth16_prove_host(zk_ctx* ctx, const zk_crs_share* crs, const void* qap_a, const void* qap_b, const void* qap_c,
                          const void* a_share, const void* ax_share, const void* r, const void* s, int log2_m,
                          const zk_groth16_masks* masks, uint64_t seed, void* pi_a, void* pi_b, void* pi_c, void* stream);
int zk_msm_host(zk_ctx* ctx, int group, const void* bases, size_t len_bases, const void* scalars, size_t len_scalars,
                void* out, void* stream);
int zk_d_msm_host(zk_ctx* ctx, int group, const void* bases, const void* scalars, size_t len, const void* in_mask,
                  const void* out_mask, void* out, void* stream);

/* ---- per-kernel timing (measurement only) -----------------------------------------------------------------
 * When enabled, HIP events are recorded on the launching stream around the kernels of each slot; zk_profile_read
 * synchronises them and returns the summed duration, the summed work units (elements / chunks / points) and the
 * number of timed launches since zk_profile_enable.  Slots whose name starts with "host:" are host spans of
 * zk_groth16_prove taken with the steady clock instead (entry -> everything launched -> the event of the last chain ->
 * return; units = calls). */
int zk_profile_enable(zk_ctx* ctx, int on);
/* MSM work counters of this context since creation (measurement only): stats[0] / [1] = mixed additions the G1 / G2
 * accumulate kernels performed (= sorted (point, window) entries: identity bases and zero digits leave none), stats[2] /
 * [3] = (point, window) pairs offered to the sorts.  Counted when an MSM's result is collected. */
int zk_msm_stats(zk_ctx* ctx, uint64_t stats[4]);
int zk_profile_slots(void);
const char* zk_profile_name(int slot);
int zk_profile_read(zk_ctx* ctx, int slot, double* total_ms, double* units, long* calls);

/* d_msm when some parties dropped out: bases_d / scalars_d / in_mask hold the `nparties` surviving parties (ascending
 * ids); out (n Jacobian points) as zk_d_msm. */
int zk_d_msm_parties(zk_ctx* ctx, int group, const void* bases_d, const void* scalars_d, size_t len,
                     const uint32_t* parties, int nparties, const void* in_mask, const void* out_mask, void* out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZKSAAS_H */
