/* zkref.c -- plain-C CPU restatement of the reference's hot path (TEST INFRASTRUCTURE + CPU BASELINE ONLY).
 *
 * Nothing in the product (zk-saas_amd/) may link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / the timed CPU "port".
 *
 * The reference is Rust on arkworks (ark-ff/ark-ec/ark-poly ^0.4, not vendored: SURVEY.md F2) and cannot
 * be built here (no Rust toolchain: SURVEY.md F3), so this restates the algorithms the reference's call
 * sites reach, keeping arkworks' algorithmic choices where they decide the CPU cost:
 *   - 4 x u64 Montgomery Fp (CIOS with 128-bit products), as ark-ff's MontBackend
 *   - radix-2 in-place FFT for the small PSS domains (ark-poly Radix2EvaluationDomain)
 *   - fft1_in_place / fft2_in_place loop-for-loop as dist-primitives/src/dfft/mod.rs:178-237
 *   - per-chunk pack/unpack through domain (i)FFTs as secret-sharing/src/pss.rs:90-166
 *   - the serial king closure of dfft/mod.rs:264-304
 *   - signed-digit Pippenger with arkworks' window rule (ark-ec VariableBaseMSM::msm_bigint_wnaf)
 * It is pinned against the Python big-int oracle (tests/test_oracle_c.py), which in turn is pinned by the
 * reference's own tests.
 *
 * All field elements cross this API as 4 little-endian u64 limbs in Montgomery form (arkworks layout).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

#ifndef NL
#define NL 4   /* 64-bit limbs per field element; the Makefile also builds libzkref6.so with -DNL=6 (BLS12-381 / -377 Fq) */
#endif
typedef struct { u64 v[NL]; } fe;

typedef struct {
  u64 mod[NL];
  u64 n0inv;      /* -p^-1 mod 2^64 */
  u64 r1[NL];     /* R mod p */
  u64 r2[NL];     /* R^2 mod p */
} field_t;

/* ------------------------------------------------------------------------------------------------ field */
static inline int fe_is_zero(const fe* a) {
  u64 o = 0;
  for (int i = 0; i < NL; i++) o |= a->v[i];
  return o == 0;
}
static inline int fe_eq(const fe* a, const fe* b) {
  u64 o = 0;
  for (int i = 0; i < NL; i++) o |= a->v[i] ^ b->v[i];
  return o == 0;
}
static inline int ge_mod(const u64* a, const field_t* F) {
  for (int i = NL - 1; i >= 0; i--) {
    if (a[i] > F->mod[i]) return 1;
    if (a[i] < F->mod[i]) return 0;
  }
  return 1;
}
static inline void sub_mod_raw(u64* a, const field_t* F) {
  u64 borrow = 0;
  for (int i = 0; i < NL; i++) {
    u128 t = (u128)a[i] - F->mod[i] - borrow;
    a[i] = (u64)t;
    borrow = (u64)(t >> 127);
  }
}
static inline void fe_add(fe* r, const fe* a, const fe* b, const field_t* F) {
  u64 c = 0;
  u64 t[NL];
  for (int i = 0; i < NL; i++) {
    u128 s = (u128)a->v[i] + b->v[i] + c;
    t[i] = (u64)s;
    c = (u64)(s >> 64);
  }
  if (c || ge_mod(t, F)) sub_mod_raw(t, F);
  memcpy(r->v, t, sizeof t);
}
static inline void fe_sub(fe* r, const fe* a, const fe* b, const field_t* F) {
  u64 borrow = 0;
  u64 t[NL];
  for (int i = 0; i < NL; i++) {
    u128 s = (u128)a->v[i] - b->v[i] - borrow;
    t[i] = (u64)s;
    borrow = (u64)(s >> 127);
  }
  if (borrow) {
    u64 c = 0;
    for (int i = 0; i < NL; i++) {
      u128 s = (u128)t[i] + F->mod[i] + c;
      t[i] = (u64)s;
      c = (u64)(s >> 64);
    }
  }
  memcpy(r->v, t, sizeof t);
}
static inline void fe_neg(fe* r, const fe* a, const field_t* F) {
  fe z = {{0}};
  fe_sub(r, &z, a, F);
}
static inline void fe_dbl(fe* r, const fe* a, const field_t* F) { fe_add(r, a, a, F); }

static inline void fe_mul(fe* r, const fe* a, const fe* b, const field_t* F) {
  u64 t[NL + 2] = {0};
  for (int i = 0; i < NL; i++) {
    u64 c = 0;
    for (int j = 0; j < NL; j++) {
      u128 x = (u128)a->v[i] * b->v[j] + t[j] + c;
      t[j] = (u64)x;
      c = (u64)(x >> 64);
    }
    u128 x = (u128)t[NL] + c;
    t[NL] = (u64)x;
    t[NL + 1] = (u64)(x >> 64);
    u64 m = t[0] * F->n0inv;
    x = (u128)m * F->mod[0] + t[0];
    c = (u64)(x >> 64);
    for (int j = 1; j < NL; j++) {
      x = (u128)m * F->mod[j] + t[j] + c;
      t[j - 1] = (u64)x;
      c = (u64)(x >> 64);
    }
    x = (u128)t[NL] + c;
    t[NL - 1] = (u64)x;
    t[NL] = t[NL + 1] + (u64)(x >> 64);
  }
  if (t[NL] || ge_mod(t, F)) sub_mod_raw(t, F);
  memcpy(r->v, t, NL * sizeof(u64));
}
static inline void fe_sqr(fe* r, const fe* a, const field_t* F) { fe_mul(r, a, a, F); }
static void fe_pow(fe* r, const fe* a, const u64* e, int nl, const field_t* F) {
  fe acc;
  memcpy(acc.v, F->r1, sizeof acc.v);
  for (int i = nl - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      fe_sqr(&acc, &acc, F);
      if ((e[i] >> b) & 1) fe_mul(&acc, &acc, a, F);
    }
  *r = acc;
}
static void fe_inv(fe* r, const fe* a, const field_t* F) {
  u64 e[NL];
  u64 borrow = 2;
  for (int i = 0; i < NL; i++) {
    u128 t = (u128)F->mod[i] - borrow;
    e[i] = (u64)t;
    borrow = (u64)(t >> 127);
  }
  fe_pow(r, a, e, NL, F);
}
static void fe_pow_u64(fe* r, const fe* a, u64 e, const field_t* F) { fe_pow(r, a, &e, 1, F); }
static void fe_one(fe* r, const field_t* F) { memcpy(r->v, F->r1, sizeof r->v); }
static void fe_from_mont(fe* r, const fe* a, const field_t* F) {
  fe one = {{1}};
  fe_mul(r, a, &one, F);
}

/* exported single ops (used by the tests to pin the C field against Python ints) */
void zkref_mul(const field_t* F, const fe* a, const fe* b, fe* r) { fe_mul(r, a, b, F); }
void zkref_add(const field_t* F, const fe* a, const fe* b, fe* r) { fe_add(r, a, b, F); }
void zkref_sub(const field_t* F, const fe* a, const fe* b, fe* r) { fe_sub(r, a, b, F); }
void zkref_inv(const field_t* F, const fe* a, fe* r) { fe_inv(r, a, F); }
/* timing helper: `iters` dependent multiplications */
void zkref_mul_chain(const field_t* F, fe* a, const fe* b, long iters) {
  fe x = *a;
  for (long i = 0; i < iters; i++) fe_mul(&x, &x, b, F);
  *a = x;
}

/* ------------------------------------------------------------------------------------------------ domains */
/* Radix2EvaluationDomain with optional coset offset (ark-poly). */
typedef struct {
  int log_size;
  size_t size;
  fe gen, gen_inv, size_inv, offset, offset_inv;
} domain_t;

static int ilog2(size_t x) {
  int r = 0;
  while (((size_t)1 << r) < x) r++;
  return r;
}

static void bitrev_inplace(fe* a, size_t n) { /* dfft/mod.rs:322-335 */
  size_t target = 0;
  for (size_t pos = 0; pos < n; pos++) {
    if (target > pos) {
      fe t = a[target];
      a[target] = a[pos];
      a[pos] = t;
    }
    size_t mask = n >> 1;
    while (target & mask) {
      target &= ~mask;
      mask >>= 1;
    }
    target |= mask;
  }
}
void zkref_bitrev(fe* a, size_t n) { bitrev_inplace(a, n); }

/* in-order radix-2 NTT (natural in, natural out) with root w of order n */
static void ntt_inplace(fe* a, size_t n, const fe* w, const field_t* F) {
  bitrev_inplace(a, n);
  int logn = ilog2(n);
  for (int s = 1; s <= logn; s++) {
    size_t len = (size_t)1 << s, half = len >> 1;
    fe wlen;
    fe_pow_u64(&wlen, w, n / len, F);
    for (size_t start = 0; start < n; start += len) {
      fe tw;
      fe_one(&tw, F);
      for (size_t k = 0; k < half; k++) {
        fe u = a[start + k], v;
        fe_mul(&v, &a[start + k + half], &tw, F);
        fe_add(&a[start + k], &u, &v, F);
        fe_sub(&a[start + k + half], &u, &v, F);
        fe_mul(&tw, &tw, &wlen, F);
      }
    }
  }
}
static void distribute_powers(fe* a, size_t n, const fe* g, const fe* c, const field_t* F) {
  fe cur = *c;
  for (size_t i = 0; i < n; i++) {
    fe_mul(&a[i], &a[i], &cur, F);
    fe_mul(&cur, &cur, g, F);
  }
}
/* fft_in_place / ifft_in_place on a vector already resized to the domain size */
static void dom_fft(const domain_t* D, fe* a, const field_t* F) {
  fe one;
  fe_one(&one, F);
  if (!fe_eq(&D->offset, &one)) distribute_powers(a, D->size, &D->offset, &one, F);
  ntt_inplace(a, D->size, &D->gen, F);
}
static void dom_ifft(const domain_t* D, fe* a, const field_t* F) {
  ntt_inplace(a, D->size, &D->gen_inv, F);
  distribute_powers(a, D->size, &D->offset_inv, &D->size_inv, F);
}

/* The LOCAL (non-distributed) witness map, BASELINE configs[0]: ark-circom's CircomReduction::witness_map_from_matrices as the
 * reference restates it in-tree (groth16/src/ext_wit.rs:239-285 circom_ref; reached from groth16/examples/sha256.rs:191-199
 * through create_proof_with_reduction_and_matrices): per vector ifft on the domain, coefficient i times w_2m^i, fft; then
 * a b - c pointwise.  a, b, c: D->size values each (overwritten), h: D->size values.  nthreads >= 3: the three vectors run
 * in three threads (arkworks' `parallel` feature splits the FFTs with rayon: groth16/Cargo.toml:37-38). */
typedef struct {
  const field_t* F;
  const domain_t* D;
  fe* v;
  const fe* w2m;
} circom_ref_job;
static void* circom_ref_run(void* arg) {
  circom_ref_job* j = (circom_ref_job*)arg;
  fe one;
  fe_one(&one, j->F);
  dom_ifft(j->D, j->v, j->F);
  distribute_powers(j->v, j->D->size, j->w2m, &one, j->F);
  dom_fft(j->D, j->v, j->F);
  return NULL;
}
void zkref_circom_ref(const field_t* F, const domain_t* D, fe* a, fe* b, fe* c, const fe* w2m, int nthreads, fe* h) {
  circom_ref_job jobs[3] = {{F, D, a, w2m}, {F, D, b, w2m}, {F, D, c, w2m}};
  if (nthreads >= 3) {
    pthread_t th[3];
    for (int i = 0; i < 3; i++) pthread_create(&th[i], NULL, circom_ref_run, &jobs[i]);
    for (int i = 0; i < 3; i++) pthread_join(th[i], NULL);
  } else {
    for (int i = 0; i < 3; i++) circom_ref_run(&jobs[i]);
  }
  for (size_t i = 0; i < D->size; i++) {
    fe t;
    fe_mul(&t, &a[i], &b[i], F);
    fe_sub(&h[i], &t, &c[i], F);
  }
}

/* ------------------------------------------------------------------------------------------------ PSS */
typedef struct {
  field_t F;
  int l, t, n;
  domain_t share, secret, secret2;
} pss_t;

/* Filled from Python (domain constants computed there from the curve parameters). */
size_t zkref_sizeof_pss(void) { return sizeof(pss_t); }
size_t zkref_sizeof_domain(void) { return sizeof(domain_t); }

#define MAXN 64
/* pss.rs:90-122 pack: secrets (l) ++ rand (t) -> n shares */
static void pss_pack(const pss_t* P, const fe* secrets, const fe* rnd, fe* shares) {
  fe buf[MAXN];
  memset(buf, 0, sizeof(fe) * P->n);
  memcpy(buf, secrets, sizeof(fe) * P->l);
  if (rnd) memcpy(buf + P->l, rnd, sizeof(fe) * P->t);
  dom_ifft(&P->secret, buf, &P->F);          /* l+t coefficients; rest stays zero */
  dom_fft(&P->share, buf, &P->F);
  memcpy(shares, buf, sizeof(fe) * P->n);
}
/* pss.rs:125-138 */
static void pss_unpack(const pss_t* P, const fe* shares, fe* secrets) {
  fe buf[MAXN];
  memcpy(buf, shares, sizeof(fe) * P->n);
  dom_ifft(&P->share, buf, &P->F);
  dom_fft(&P->secret, buf, &P->F);           /* truncation to l+t coefficients: uses buf[0..l+t) */
  memcpy(secrets, buf, sizeof(fe) * P->l);
}
/* pss.rs:141-166 */
static void pss_unpack2(const pss_t* P, const fe* shares, fe* secrets) {
  fe buf[MAXN];
  memcpy(buf, shares, sizeof(fe) * P->n);
  dom_ifft(&P->share, buf, &P->F);
  dom_fft(&P->secret2, buf, &P->F);
  for (int i = 0; i < P->l; i++) secrets[i] = buf[2 * i];
}
/* ---- the TUNED king (cpu_baseline only; off by default): pack and unpack2 are fixed linear maps, so a CPU prover that
 * cared would apply them as precomputed matrices (n x (l+t) and l x n: 48 multiplications per chunk at l = 2 instead of
 * two size-8 and two size-4/16 FFTs with their bit reversals and scalings) and split the chunks over threads.  The
 * reference does neither (its king is the serial FFT form above: dfft/mod.rs:264-304); bench.py reports both. */
static int g_fast_king = 0, g_king_threads = 1;
void zkref_set_fast_king(int on, int threads) {
  g_fast_king = on;
  g_king_threads = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
}
typedef struct {
  fe pack[MAXN][MAXN];     /* shares[p] = sum_j pack[p][j] * v[j], j < l + t */
  fe unpack2[MAXN][MAXN];  /* secrets[i] = sum_p unpack2[i][p] * shares[p] */
} king_mats;
static void king_mats_build(const pss_t* P, king_mats* M) {
  fe unit[MAXN], col[MAXN];
  for (int j = 0; j < P->l + P->t; j++) {
    memset(unit, 0, sizeof unit);
    fe_one(&unit[j], &P->F);
    pss_pack(P, unit, unit + P->l, col);
    for (int p = 0; p < P->n; p++) M->pack[p][j] = col[p];
  }
  for (int p = 0; p < P->n; p++) {
    memset(unit, 0, sizeof unit);
    fe_one(&unit[p], &P->F);
    pss_unpack2(P, unit, col);
    for (int i = 0; i < P->l; i++) M->unpack2[i][p] = col[i];
  }
}
static inline void mat_unpack2(const pss_t* P, const king_mats* M, const fe* sh, fe* sec) {
  for (int i = 0; i < P->l; i++) {
    fe acc = {{0}}, t;
    for (int p = 0; p < P->n; p++) {
      fe_mul(&t, &M->unpack2[i][p], &sh[p], &P->F);
      fe_add(&acc, &acc, &t, &P->F);
    }
    sec[i] = acc;
  }
}
static inline void mat_pack(const pss_t* P, const king_mats* M, const fe* sec, const fe* rnd, fe* sh) {
  for (int p = 0; p < P->n; p++) {
    fe acc = {{0}}, t;
    for (int j = 0; j < P->l; j++) {
      fe_mul(&t, &M->pack[p][j], &sec[j], &P->F);
      fe_add(&acc, &acc, &t, &P->F);
    }
    for (int j = 0; j < P->t; j++) {
      fe_mul(&t, &M->pack[p][P->l + j], &rnd[j], &P->F);
      fe_add(&acc, &acc, &t, &P->F);
    }
    sh[p] = acc;
  }
}
/* run fn(arg, lo, hi) over [0, count) on g_king_threads threads */
typedef struct {
  void (*fn)(void*, size_t, size_t);
  void* arg;
  size_t lo, hi;
} king_span;
static void* king_span_run(void* a) {
  king_span* s = (king_span*)a;
  s->fn(s->arg, s->lo, s->hi);
  return NULL;
}
static void king_parallel(void (*fn)(void*, size_t, size_t), void* arg, size_t count) {
  int nt = g_king_threads;
  if (nt <= 1 || count < 1024) {
    fn(arg, 0, count);
    return;
  }
  pthread_t th[64];
  king_span sp[64];
  for (int t = 0; t < nt; t++) {
    sp[t] = (king_span){fn, arg, count * t / nt, count * (t + 1) / nt};
    pthread_create(&th[t], NULL, king_span_run, &sp[t]);
  }
  for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
}

void zkref_pack(const pss_t* P, const fe* secrets, const fe* rnd, fe* shares) { pss_pack(P, secrets, rnd, shares); }
void zkref_unpack(const pss_t* P, const fe* shares, fe* secrets) { pss_unpack(P, shares, secrets); }
void zkref_unpack2(const pss_t* P, const fe* shares, fe* secrets) { pss_unpack2(P, shares, secrets); }

/* ------------------------------------------------------------------------------------------------ PRNG */
static inline u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static void rand_fe(fe* r, u64 seed, u64 idx, int bits, const field_t* F) { /* oracle/prng.py, Montgomery out */
  u64 state = mix64(seed ^ mix64(idx + 0x632BE59BD9B4E019ull));
  const int nd = (bits + 63) / 64;          /* limbs drawn: ceil(bits / 64), as oracle/prng.py; higher limbs are zero */
  const int top = bits - 64 * (nd - 1);
  for (;;) {
    fe x = {{0}};
    for (int i = 0; i < nd; i++) {
      state += 0x9E3779B97F4A7C15ull;
      x.v[i] = mix64(state);
    }
    if (top < 64) x.v[nd - 1] &= (((u64)1 << top) - 1);
    if (!ge_mod(x.v, F)) {
      fe r2;
      memcpy(r2.v, F->r2, sizeof r2.v);
      fe_mul(r, &x, &r2, F);
      return;
    }
  }
}

/* ------------------------------------------------------------------------------------------------ d_fft */
/* dfft/mod.rs:178-208, loop for loop */
void zkref_fft1(const pss_t* P, fe* px, size_t len, const fe* gen) {
  const field_t* F = &P->F;
  size_t dom_size = len * P->l;
  int logm = ilog2(dom_size), logl = ilog2(P->l);
  for (int i = logm; i >= logl + 1; i--) {
    size_t poly_size = dom_size >> i;
    fe stride, factor;
    fe_pow_u64(&stride, gen, (u64)1 << (i - 1), F);
    factor = stride;
    size_t jmax = ((size_t)1 << (i - 1)) / P->l;
    for (size_t k = 0; k < poly_size; k++) {
      for (size_t j = 0; j < jmax; j++) {
        fe x = px[(2 * j) * poly_size + k], y;
        fe_mul(&y, &px[(2 * j + 1) * poly_size + k], &factor, F);
        fe_add(&px[j * (2 * poly_size) + k], &x, &y, F);
        fe_sub(&px[j * (2 * poly_size) + k + poly_size], &x, &y, F);
      }
      fe_mul(&factor, &factor, &stride, F);
    }
  }
}
/* dfft/mod.rs:210-237 */
static void fft2(const pss_t* P, fe** s1p, fe** s2p, size_t dom_size, const fe* gen) {
  const field_t* F = &P->F;
  fe* s1 = *s1p;
  fe* s2 = *s2p;
  int logl = ilog2(P->l);
  for (int i = logl; i >= 1; i--) {
    size_t poly_size = dom_size >> i;
    fe stride, factor;
    fe_pow_u64(&stride, gen, (u64)1 << (i - 1), F);
    factor = stride;
    size_t w = (size_t)1 << i, hw = w >> 1;
    for (size_t k = 0; k < poly_size; k++) {
      for (size_t j = 0; j < hw; j++) {
        fe x = s1[k * w + 2 * j], y;
        fe_mul(&y, &s1[k * w + 2 * j + 1], &factor, F);
        fe_add(&s2[k * hw + j], &x, &y, F);
        fe_sub(&s2[(k + poly_size) * hw + j], &x, &y, F);
      }
      fe_mul(&factor, &factor, &stride, F);
    }
    fe* t = s1;
    s1 = s2;
    s2 = t;
  }
  /* rotate_right(1) */
  fe last = s1[dom_size - 1];
  memmove(s1 + 1, s1, sizeof(fe) * (dom_size - 1));
  s1[0] = last;
  *s1p = s1;
  *s2p = s2;
}

/* chunk loops of the tuned king (g_fast_king) */
typedef struct {
  const pss_t* P;
  const king_mats* M;
  const fe* in;
  fe* s1;
  fe* out;
  size_t mbyl;
  int rearrange;
  u64 seed;
  int fr_bits;
} kf_ctx;
static void kf_unpack(void* a, size_t lo, size_t hi) {
  kf_ctx* c = (kf_ctx*)a;
  fe sh[MAXN], sec[MAXN];
  for (size_t i = lo; i < hi; i++) {
    for (int p = 0; p < c->P->n; p++) sh[p] = c->in[(size_t)p * c->mbyl + i];
    mat_unpack2(c->P, c->M, sh, sec);
    for (int j = 0; j < c->P->l; j++) c->s1[i * c->P->l + j] = sec[j];
  }
}
static void kf_pack(void* a, size_t lo, size_t hi) {
  kf_ctx* c = (kf_ctx*)a;
  const int l = c->P->l, t = c->P->t;
  fe sh[MAXN], sec[MAXN], rnd[MAXN];
  for (size_t j = lo; j < hi; j++) {
    for (int i = 0; i < l; i++) sec[i] = c->rearrange ? c->s1[j + (size_t)i * c->mbyl] : c->s1[j * l + i];
    for (int i = 0; i < t; i++) rand_fe(&rnd[i], c->seed, (u64)j * t + i, c->fr_bits, &c->P->F);
    mat_pack(c->P, c->M, sec, rnd, sh);
    for (int p = 0; p < c->P->n; p++) c->out[(size_t)p * c->mbyl + j] = sh[p];
  }
}

/* King closure of fft2_with_rearrange (dfft/mod.rs:264-304), all n parties present.
 * in: [n][mbyl] (already masked); out: [n][mbyl]; g == NULL means 1; fr_bits for the PRNG. */
void zkref_king_fft2(const pss_t* P, const fe* in, size_t mbyl, const fe* gen, const fe* g, int rearrange, u64 seed,
                     int fr_bits, fe* out) {
  const field_t* F = &P->F;
  int n = P->n, l = P->l, t = P->t;
  size_t m = mbyl * l;
  fe* s1 = malloc(sizeof(fe) * m);
  fe* s2 = malloc(sizeof(fe) * m);
  fe sh[MAXN], sec[MAXN], rnd[MAXN];
  king_mats* KM = NULL;
  kf_ctx kc;
  if (g_fast_king) {
    KM = malloc(sizeof(king_mats));
    king_mats_build(P, KM);
    kc = (kf_ctx){P, KM, in, s1, out, mbyl, rearrange, seed, fr_bits};
    king_parallel(kf_unpack, &kc, mbyl);
  } else
  for (size_t i = 0; i < mbyl; i++) {          /* transpose + unpack per chunk */
    for (int p = 0; p < n; p++) sh[p] = in[(size_t)p * mbyl + i];
    pss_unpack2(P, sh, sec);
    for (int j = 0; j < l; j++) s1[i * l + j] = sec[j];
  }
  fft2(P, &s1, &s2, m, gen);
  fe one;
  fe_one(&one, F);
  if (g && !fe_eq(g, &one)) distribute_powers(s1, m, g, &one, F);
  if (rearrange) bitrev_inplace(s1, m);
  if (g_fast_king) {
    kc.s1 = s1;                                 /* fft2 may have swapped its two buffers */
    king_parallel(kf_pack, &kc, mbyl);
    free(KM);
  } else
  for (size_t j = 0; j < mbyl; j++) {
    for (int i = 0; i < l; i++) sec[i] = rearrange ? s1[j + (size_t)i * mbyl] : s1[j * l + i];
    for (int i = 0; i < t; i++) rand_fe(&rnd[i], seed, (u64)j * t + i, fr_bits, F);
    pss_pack(P, sec, rnd, sh);
    for (int p = 0; p < n; p++) out[(size_t)p * mbyl + j] = sh[p];
  }
  free(s1);
  free(s2);
}

/* d_fft / d_ifft for all parties (dfft/mod.rs:99-175), zero or given masks ([n][mbyl] or NULL).
 * shares [n][mbyl] in place.  gen = group_gen or group_gen_inv; size_inv != NULL => d_ifft scaling. */
void zkref_d_fft(const pss_t* P, fe* shares, size_t mbyl, const fe* gen, const fe* size_inv, const fe* g,
                 int rearrange, const fe* in_mask, const fe* out_mask, u64 seed, int fr_bits) {
  const field_t* F = &P->F;
  int n = P->n;
  for (int p = 0; p < n; p++) {
    fe* px = shares + (size_t)p * mbyl;
    if (size_inv)
      for (size_t i = 0; i < mbyl; i++) fe_mul(&px[i], &px[i], size_inv, F);
    zkref_fft1(P, px, mbyl, gen);
    if (in_mask)
      for (size_t i = 0; i < mbyl; i++) fe_add(&px[i], &px[i], &in_mask[(size_t)p * mbyl + i], F);
  }
  fe* out = malloc(sizeof(fe) * n * mbyl);
  zkref_king_fft2(P, shares, mbyl, gen, g, rearrange, seed, fr_bits, out);
  for (size_t i = 0; i < (size_t)n * mbyl; i++) {
    if (out_mask) fe_add(&shares[i], &out[i], &out_mask[i], F);
    else shares[i] = out[i];
  }
  free(out);
}

/* The same with the n parties' local stages on their own threads (the reference's parties ARE concurrent tasks,
 * mpc-net/src/multi.rs:317-327; same values): what the full-size checks of tests/test_gpu_configs.py run at m = 2^24. */
typedef struct {
  const pss_t* P;
  fe* px;
  size_t mbyl;
  const fe *gen, *size_inv, *in_mask;
} dfft_party_job;
static void* dfft_party_run(void* a) {
  dfft_party_job* j = (dfft_party_job*)a;
  const field_t* F = &j->P->F;
  if (j->size_inv)
    for (size_t i = 0; i < j->mbyl; i++) fe_mul(&j->px[i], &j->px[i], j->size_inv, F);
  zkref_fft1(j->P, j->px, j->mbyl, j->gen);
  if (j->in_mask)
    for (size_t i = 0; i < j->mbyl; i++) fe_add(&j->px[i], &j->px[i], &j->in_mask[i], F);
  return NULL;
}
void zkref_d_fft_mt(const pss_t* P, fe* shares, size_t mbyl, const fe* gen, const fe* size_inv, const fe* g,
                    int rearrange, const fe* in_mask, const fe* out_mask, u64 seed, int fr_bits) {
  const field_t* F = &P->F;
  int n = P->n;
  pthread_t th[MAXN];
  dfft_party_job jobs[MAXN];
  for (int p = 0; p < n; p++) {
    jobs[p] = (dfft_party_job){P, shares + (size_t)p * mbyl, mbyl, gen, size_inv, in_mask ? in_mask + (size_t)p * mbyl : NULL};
    pthread_create(&th[p], NULL, dfft_party_run, &jobs[p]);
  }
  for (int p = 0; p < n; p++) pthread_join(th[p], NULL);
  fe* out = malloc(sizeof(fe) * n * mbyl);
  zkref_king_fft2(P, shares, mbyl, gen, g, rearrange, seed, fr_bits, out);
  for (size_t i = 0; i < (size_t)n * mbyl; i++) {
    if (out_mask) fe_add(&shares[i], &out[i], &out_mask[i], F);
    else shares[i] = out[i];
  }
  free(out);
}

/* ------------------------------------------------------------------------------------------------ curves */
/* y^2 = x^3 + b over Fq (G1) and Fq2 = Fq[u]/(u^2+1) (G2); Jacobian coordinates, Z = 0 identity. */
typedef struct { fe x, y, z; } g1j;
typedef struct { fe x, y; } g1a;          /* (0,0) identity */
typedef struct { fe c0, c1; } fe2;
typedef struct { fe2 x, y, z; } g2j;
typedef struct { fe2 x, y; } g2a;

#define DEF_FIELD_OPS(T, PRE)
/* ---- Fq2 */
static inline void f2_add(fe2* r, const fe2* a, const fe2* b, const field_t* F) { fe_add(&r->c0, &a->c0, &b->c0, F); fe_add(&r->c1, &a->c1, &b->c1, F); }
static inline void f2_sub(fe2* r, const fe2* a, const fe2* b, const field_t* F) { fe_sub(&r->c0, &a->c0, &b->c0, F); fe_sub(&r->c1, &a->c1, &b->c1, F); }
static inline void f2_dbl(fe2* r, const fe2* a, const field_t* F) { f2_add(r, a, a, F); }
static inline void f2_neg(fe2* r, const fe2* a, const field_t* F) { fe_neg(&r->c0, &a->c0, F); fe_neg(&r->c1, &a->c1, F); }
static inline int f2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static inline int f2_eq(const fe2* a, const fe2* b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
static inline void f2_mul(fe2* r, const fe2* a, const fe2* b, const field_t* F) {
  fe v0, v1, s, t, u;
  fe_mul(&v0, &a->c0, &b->c0, F);
  fe_mul(&v1, &a->c1, &b->c1, F);
  fe_add(&s, &a->c0, &a->c1, F);
  fe_add(&t, &b->c0, &b->c1, F);
  fe_mul(&u, &s, &t, F);
  fe_sub(&r->c0, &v0, &v1, F);
  fe_sub(&u, &u, &v0, F);
  fe_sub(&r->c1, &u, &v1, F);
}
static inline void f2_sqr(fe2* r, const fe2* a, const field_t* F) { f2_mul(r, a, a, F); }

/* Generic Jacobian formulas via macros over (element type, op prefix) */
#define DEF_CURVE(NAME, ET, JT, AT, ADD, SUB, DBL, NEG, MUL, SQR, ISZ, EQ)                                   \
  static void NAME##_set_identity(JT* p) { memset(p, 0, sizeof *p); }                                        \
  static int NAME##_is_identity(const JT* p) { return ISZ(&p->z); }                                          \
  static void NAME##_double(JT* r, const JT* p, const field_t* F) {                                          \
    if (ISZ(&p->z)) { *r = *p; return; }                                                                      \
    ET A, B, C, D, E, Fv, t, X3, Y3, Z3;                                                                      \
    SQR(&A, &p->x, F); SQR(&B, &p->y, F); SQR(&C, &B, F);                                                     \
    ADD(&t, &p->x, &B, F); SQR(&t, &t, F); SUB(&t, &t, &A, F); SUB(&t, &t, &C, F); DBL(&D, &t, F);            \
    DBL(&E, &A, F); ADD(&E, &E, &A, F); SQR(&Fv, &E, F);                                                      \
    DBL(&t, &D, F); SUB(&X3, &Fv, &t, F);                                                                     \
    SUB(&t, &D, &X3, F); MUL(&Y3, &E, &t, F); DBL(&t, &C, F); DBL(&t, &t, F); DBL(&t, &t, F);                 \
    SUB(&Y3, &Y3, &t, F);                                                                                     \
    MUL(&Z3, &p->y, &p->z, F); DBL(&Z3, &Z3, F);                                                              \
    r->x = X3; r->y = Y3; r->z = Z3;                                                                          \
  }                                                                                                           \
  static void NAME##_add(JT* r, const JT* p, const JT* q, const field_t* F) {                                \
    if (ISZ(&p->z)) { *r = *q; return; }                                                                      \
    if (ISZ(&q->z)) { *r = *p; return; }                                                                      \
    ET Z1Z1, Z2Z2, U1, U2, S1, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;                                           \
    SQR(&Z1Z1, &p->z, F); SQR(&Z2Z2, &q->z, F);                                                               \
    MUL(&U1, &p->x, &Z2Z2, F); MUL(&U2, &q->x, &Z1Z1, F);                                                     \
    MUL(&t, &q->z, &Z2Z2, F); MUL(&S1, &p->y, &t, F);                                                         \
    MUL(&t, &p->z, &Z1Z1, F); MUL(&S2, &q->y, &t, F);                                                         \
    if (EQ(&U1, &U2)) {                                                                                       \
      if (EQ(&S1, &S2)) { NAME##_double(r, p, F); return; }                                                   \
      NAME##_set_identity(r); return;                                                                         \
    }                                                                                                         \
    SUB(&H, &U2, &U1, F); SUB(&R, &S2, &S1, F); SQR(&HH, &H, F); MUL(&HHH, &H, &HH, F); MUL(&V, &U1, &HH, F); \
    SQR(&X3, &R, F); SUB(&X3, &X3, &HHH, F); DBL(&t, &V, F); SUB(&X3, &X3, &t, F);                            \
    SUB(&t, &V, &X3, F); MUL(&Y3, &R, &t, F); MUL(&t, &S1, &HHH, F); SUB(&Y3, &Y3, &t, F);                    \
    MUL(&Z3, &p->z, &q->z, F); MUL(&Z3, &Z3, &H, F);                                                          \
    r->x = X3; r->y = Y3; r->z = Z3;                                                                          \
  }                                                                                                           \
  /* mixed addition with an affine point (madd-2007-bl), as ark-ec's AddAssign<&Affine> */                    \
  static void NAME##_madd(JT* r, const JT* p, const AT* q, int negate, const field_t* F) {                    \
    if (ISZ(&q->x) && ISZ(&q->y)) { *r = *p; return; }                                                        \
    ET qy = q->y;                                                                                             \
    if (negate) NEG(&qy, &qy, F);                                                                             \
    if (ISZ(&p->z)) { r->x = q->x; r->y = qy; memset(&r->z, 0, sizeof r->z); memcpy(&r->z, F->r1, sizeof(fe)); return; } \
    ET Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;                                                       \
    SQR(&Z1Z1, &p->z, F); MUL(&U2, &q->x, &Z1Z1, F); MUL(&t, &p->z, &Z1Z1, F); MUL(&S2, &qy, &t, F);          \
    if (EQ(&U2, &p->x)) {                                                                                     \
      if (EQ(&S2, &p->y)) { NAME##_double(r, p, F); return; }                                                 \
      NAME##_set_identity(r); return;                                                                         \
    }                                                                                                         \
    SUB(&H, &U2, &p->x, F); SQR(&HH, &H, F); DBL(&I, &HH, F); DBL(&I, &I, F); MUL(&J, &H, &I, F);             \
    SUB(&rr, &S2, &p->y, F); DBL(&rr, &rr, F); MUL(&V, &p->x, &I, F);                                         \
    SQR(&X3, &rr, F); SUB(&X3, &X3, &J, F); DBL(&t, &V, F); SUB(&X3, &X3, &t, F);                             \
    SUB(&t, &V, &X3, F); MUL(&Y3, &rr, &t, F); MUL(&t, &p->y, &J, F); DBL(&t, &t, F); SUB(&Y3, &Y3, &t, F);   \
    ADD(&Z3, &p->z, &H, F); SQR(&Z3, &Z3, F); SUB(&Z3, &Z3, &Z1Z1, F); SUB(&Z3, &Z3, &HH, F);                 \
    r->x = X3; r->y = Y3; r->z = Z3;                                                                          \
  }

DEF_CURVE(g1, fe, g1j, g1a, fe_add, fe_sub, fe_dbl, fe_neg, fe_mul, fe_sqr, fe_is_zero, fe_eq)
DEF_CURVE(g2, fe2, g2j, g2a, f2_add, f2_sub, f2_dbl, f2_neg, f2_mul, f2_sqr, f2_is_zero, f2_eq)

/* ------------------------------------------------------------------------------------------------ MSM */
/* ark-ec 0.4 VariableBaseMSM::msm_bigint_wnaf: c = 3 if n < 32 else ln_without_floats(n) + 2 with
 * ln_without_floats(a) = log2(a) * 69 / 100 (ceil log2); signed radix-2^c digits; per window: buckets of size
 * 2^(c-1), running-sum reduction; windows folded high -> low with c doublings.  Windows are independent and
 * `nthreads` > 1 splits them across threads (arkworks' `parallel` feature does the same with rayon). */
#include <pthread.h>

static int ark_window(size_t n) {
  if (n < 32) return 3;
  int lg = ilog2(n);
  return lg * 69 / 100 + 2;
}

static void make_digits(const u64* a /* canonical, NL limbs */, int w, int num_bits, int64_t* digits, int ndig) {
  u64 radix = (u64)1 << w, window_mask = radix - 1;
  u64 carry = 0;
  for (int i = 0; i < ndig; i++) {
    int bit_offset = i * w;
    int u64_idx = bit_offset / 64, bit_idx = bit_offset % 64;
    u64 bit_buf;
    if (bit_idx < 64 - w || u64_idx == NL - 1)
      bit_buf = u64_idx < NL ? a[u64_idx] >> bit_idx : 0;
    else
      bit_buf = (a[u64_idx] >> bit_idx) | (a[u64_idx + 1] << (64 - bit_idx));
    u64 coef = carry + (bit_buf & window_mask);
    carry = (coef + radix / 2) >> w;
    int64_t d = (int64_t)coef - (int64_t)(carry << w);
    digits[i] = d;
  }
  digits[ndig - 1] += (int64_t)(carry << w);
  (void)num_bits;
}

#define DEF_MSM(NAME, JT, AT)                                                                                 \
  typedef struct {                                                                                            \
    const field_t* Fq;                                                                                        \
    const AT* bases;                                                                                          \
    const int64_t* digits;                                                                                    \
    size_t n;                                                                                                 \
    int c, ndig, w0, w1;                                                                                      \
    JT* window_sums;                                                                                          \
  } NAME##_job;                                                                                               \
  static void* NAME##_worker(void* arg) {                                                                     \
    NAME##_job* J = (NAME##_job*)arg;                                                                         \
    /* 2^c buckets: the top digit is not re-centred (make_digits adds the last carry back), so for a modulus whose  \
     * top window is nearly full (BLS12-381 Fr with c = 15: up to 29 678) it exceeds 2^(c-1) */                      \
    size_t nb = (size_t)1 << J->c;                                                                            \
    JT* buckets = malloc(sizeof(JT) * nb);                                                                    \
    for (int w = J->w0; w < J->w1; w++) {                                                                     \
      for (size_t b = 0; b < nb; b++) NAME##_set_identity(&buckets[b]);                                       \
      for (size_t i = 0; i < J->n; i++) {                                                                     \
        int64_t d = J->digits[i * J->ndig + w];                                                               \
        if (d > 0) NAME##_madd(&buckets[d - 1], &buckets[d - 1], &J->bases[i], 0, J->Fq);                     \
        else if (d < 0) NAME##_madd(&buckets[-d - 1], &buckets[-d - 1], &J->bases[i], 1, J->Fq);              \
      }                                                                                                       \
      JT run, res;                                                                                            \
      NAME##_set_identity(&run);                                                                              \
      NAME##_set_identity(&res);                                                                              \
      for (size_t b = nb; b-- > 0;) {                                                                         \
        NAME##_add(&run, &run, &buckets[b], J->Fq);                                                           \
        NAME##_add(&res, &res, &run, J->Fq);                                                                  \
      }                                                                                                       \
      J->window_sums[w] = res;                                                                                \
    }                                                                                                         \
    free(buckets);                                                                                            \
    return NULL;                                                                                              \
  }                                                                                                           \
  void zkref_msm_##NAME(const field_t* Fr, const field_t* Fq, int fr_bits, const AT* bases, const fe* scalars, \
                        size_t n, int nthreads, JT* out) {                                                    \
    NAME##_set_identity(out);                                                                                 \
    if (n == 0) return;                                                                                       \
    int c = ark_window(n);                                                                                    \
    int ndig = (fr_bits + c - 1) / c;                                                                         \
    int64_t* digits = malloc(sizeof(int64_t) * n * ndig);                                                     \
    for (size_t i = 0; i < n; i++) {                                                                          \
      fe canon;                                                                                               \
      fe_from_mont(&canon, &scalars[i], Fr);                                                                  \
      make_digits(canon.v, c, fr_bits, digits + i * ndig, ndig);                                              \
    }                                                                                                         \
    JT* ws = malloc(sizeof(JT) * ndig);                                                                       \
    if (nthreads < 1) nthreads = 1;                                                                           \
    if (nthreads > ndig) nthreads = ndig;                                                                     \
    pthread_t th[64];                                                                                         \
    NAME##_job jobs[64];                                                                                      \
    if (nthreads > 64) nthreads = 64;                                                                         \
    for (int t = 0; t < nthreads; t++) {                                                                      \
      jobs[t] = (NAME##_job){Fq, bases, digits, n, c, ndig, t * ndig / nthreads, (t + 1) * ndig / nthreads, ws}; \
      if (nthreads == 1) NAME##_worker(&jobs[t]);                                                             \
      else pthread_create(&th[t], NULL, NAME##_worker, &jobs[t]);                                             \
    }                                                                                                         \
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);                           \
    JT total = ws[ndig - 1];                                                                                  \
    for (int w = ndig - 2; w >= 0; w--) {                                                                     \
      for (int k = 0; k < c; k++) NAME##_double(&total, &total, Fq);                                          \
      NAME##_add(&total, &total, &ws[w], Fq);                                                                 \
    }                                                                                                         \
    *out = total;                                                                                             \
    free(ws);                                                                                                 \
    free(digits);                                                                                             \
  }                                                                                                           \
  void zkref_##NAME##_add(const field_t* Fq, const JT* a, const JT* b, JT* r) { NAME##_add(r, a, b, Fq); }    \
  void zkref_##NAME##_double(const field_t* Fq, const JT* a, JT* r) { NAME##_double(r, a, Fq); }

DEF_MSM(g1, g1j, g1a)
DEF_MSM(g2, g2j, g2a)

/* k * P (double-and-add) and doubling chains used to build synthetic bases cheaply
 * (groth16/examples/local_groth_bench.rs:25-49 uses the same trick). */
void zkref_g1_mul(const field_t* Fr, const field_t* Fq, const g1j* p, const fe* k_mont, g1j* r) {
  fe k;
  fe_from_mont(&k, k_mont, Fr);
  g1j acc;
  g1_set_identity(&acc);
  for (int i = NL - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      g1_double(&acc, &acc, Fq);
      if ((k.v[i] >> b) & 1) g1_add(&acc, &acc, p, Fq);
    }
  *r = acc;
}
void zkref_g2_mul(const field_t* Fr, const field_t* Fq, const g2j* p, const fe* k_mont, g2j* r) {
  fe k;
  fe_from_mont(&k, k_mont, Fr);
  g2j acc;
  g2_set_identity(&acc);
  for (int i = NL - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      g2_double(&acc, &acc, Fq);
      if ((k.v[i] >> b) & 1) g2_add(&acc, &acc, p, Fq);
    }
  *r = acc;
}
/* batch normalisation to affine (Montgomery's trick) */
void zkref_g1_to_affine(const field_t* Fq, const g1j* p, size_t n, g1a* out) {
  fe* pre = malloc(sizeof(fe) * (n + 1));
  fe acc;
  fe_one(&acc, Fq);
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!fe_is_zero(&p[i].z)) fe_mul(&acc, &acc, &p[i].z, Fq);
  }
  fe inv;
  fe_inv(&inv, &acc, Fq);
  for (size_t i = n; i-- > 0;) {
    if (fe_is_zero(&p[i].z)) {
      memset(&out[i], 0, sizeof out[i]);
      continue;
    }
    fe zi, zi2, zi3;
    fe_mul(&zi, &inv, &pre[i], Fq);
    fe_mul(&inv, &inv, &p[i].z, Fq);
    fe_sqr(&zi2, &zi, Fq);
    fe_mul(&zi3, &zi2, &zi, Fq);
    fe_mul(&out[i].x, &p[i].x, &zi2, Fq);
    fe_mul(&out[i].y, &p[i].y, &zi3, Fq);
  }
  free(pre);
}
/* out[i] = 2^i * p (affine), i < n */
void zkref_g1_doubling_chain(const field_t* Fq, const g1a* p, size_t n, g1a* out) {
  g1j* js = malloc(sizeof(g1j) * n);
  g1j cur;
  g1_set_identity(&cur);
  g1_madd(&cur, &cur, p, 0, Fq);
  for (size_t i = 0; i < n; i++) {
    js[i] = cur;
    g1_double(&cur, &cur, Fq);
  }
  zkref_g1_to_affine(Fq, js, n, out);
  free(js);
}

/* ------------------------------------------------------------------------------------------------ misc */
/* a*b - c element-wise (ext_wit.rs:173-177) */
void zkref_mul_sub(const field_t* F, const fe* a, const fe* b, const fe* c, size_t n, fe* out) {
  for (size_t i = 0; i < n; i++) {
    fe t;
    fe_mul(&t, &a[i], &b[i], F);
    fe_sub(&out[i], &t, &c[i], F);
  }
}
/* deg_red king + masks for all parties (deg_red.rs:80-126); x [n][len] in place */
typedef struct {
  const pss_t* P;
  const king_mats* M;
  fe* x;
  size_t len;
  const fe *in_mask, *out_mask;
  u64 seed;
  int fr_bits;
} dr_ctx;
static void dr_span(void* a, size_t lo, size_t hi) {
  dr_ctx* c = (dr_ctx*)a;
  const field_t* F = &c->P->F;
  const int n = c->P->n;
  fe sh[MAXN], sec[MAXN], rnd[MAXN];
  for (size_t j = lo; j < hi; j++) {
    for (int p = 0; p < n; p++) {
      sh[p] = c->x[(size_t)p * c->len + j];
      if (c->in_mask) fe_add(&sh[p], &sh[p], &c->in_mask[(size_t)p * c->len + j], F);
    }
    mat_unpack2(c->P, c->M, sh, sec);
    for (int i = 0; i < c->P->t; i++) rand_fe(&rnd[i], c->seed, (u64)j * c->P->t + i, c->fr_bits, F);
    mat_pack(c->P, c->M, sec, rnd, sh);
    for (int p = 0; p < n; p++) {
      if (c->out_mask) fe_add(&sh[p], &sh[p], &c->out_mask[(size_t)p * c->len + j], F);
      c->x[(size_t)p * c->len + j] = sh[p];
    }
  }
}
void zkref_deg_red(const pss_t* P, fe* x, size_t len, const fe* in_mask, const fe* out_mask, u64 seed, int fr_bits) {
  const field_t* F = &P->F;
  int n = P->n;
  fe sh[MAXN], sec[MAXN], rnd[MAXN];
  if (g_fast_king) {
    king_mats* KM = malloc(sizeof(king_mats));
    king_mats_build(P, KM);
    dr_ctx c = {P, KM, x, len, in_mask, out_mask, seed, fr_bits};
    king_parallel(dr_span, &c, len);
    free(KM);
    return;
  }
  for (size_t j = 0; j < len; j++) {
    for (int p = 0; p < n; p++) {
      sh[p] = x[(size_t)p * len + j];
      if (in_mask) fe_add(&sh[p], &sh[p], &in_mask[(size_t)p * len + j], F);
    }
    pss_unpack2(P, sh, sec);
    for (int i = 0; i < P->t; i++) rand_fe(&rnd[i], seed, (u64)j * P->t + i, fr_bits, F);
    pss_pack(P, sec, rnd, sh);
    for (int p = 0; p < n; p++) {
      if (out_mask) fe_add(&sh[p], &sh[p], &out_mask[(size_t)p * len + j], F);
      x[(size_t)p * len + j] = sh[p];
    }
  }
}

/* ------------------------------------------------------------------------------------------------ d_pp */
/* dpp/mod.rs:15-87 for all n parties in one address space: num, den, out [n][len].  The king unpacks every chunk of
 * num || den (:47-52; unpack_missing_shares = unpack2 with all n present), multiplies each numerator by inverse() of its
 * denominator -- ONE inversion per element, as the reference does (:54-57) --, runs the serial prefix product (:62-65),
 * packs (pack_vec, :70; randomness stream `seed`), and every party finishes with deg_red (:86; stream seed ^ 0x3333 as
 * oracle/dist.py d_pp).  s = 1 (:25-26).  Returns 1 when a denominator is zero (the reference panics, :55).
 * With zkref_set_fast_king the unpack, the inversions and the pack are split over threads (the prefix product stays
 * serial: it is a chain); the values are the same. */
typedef struct {
  const pss_t* P;
  const king_mats* M;
  const fe *num, *den;
  size_t len;
  fe *x, *dn;   /* [len * l] */
  fe* out;
  u64 seed;
  int fr_bits;
  volatile int zero;
} dpp_ctx;
static void dpp_unpack_span(void* a, size_t lo, size_t hi) {
  dpp_ctx* c = (dpp_ctx*)a;
  const int n = c->P->n, l = c->P->l;
  fe sh[MAXN];
  for (size_t j = lo; j < hi; j++) {
    for (int p = 0; p < n; p++) sh[p] = c->num[(size_t)p * c->len + j];
    if (c->M) mat_unpack2(c->P, c->M, sh, &c->x[j * l]); else pss_unpack2(c->P, sh, &c->x[j * l]);
    for (int p = 0; p < n; p++) sh[p] = c->den[(size_t)p * c->len + j];
    if (c->M) mat_unpack2(c->P, c->M, sh, &c->dn[j * l]); else pss_unpack2(c->P, sh, &c->dn[j * l]);
  }
}
static void dpp_div_span(void* a, size_t lo, size_t hi) {
  dpp_ctx* c = (dpp_ctx*)a;
  const field_t* F = &c->P->F;
  for (size_t i = lo; i < hi; i++) {
    if (fe_is_zero(&c->dn[i])) {
      c->zero = 1;
      continue;
    }
    fe inv;
    fe_inv(&inv, &c->dn[i], F);
    fe_mul(&c->x[i], &c->x[i], &inv, F);
  }
}
static void dpp_pack_span(void* a, size_t lo, size_t hi) {
  dpp_ctx* c = (dpp_ctx*)a;
  const field_t* F = &c->P->F;
  const int n = c->P->n, l = c->P->l, t = c->P->t;
  fe sh[MAXN], rnd[MAXN];
  for (size_t j = lo; j < hi; j++) {
    for (int i = 0; i < t; i++) rand_fe(&rnd[i], c->seed, (u64)j * t + i, c->fr_bits, F);
    if (c->M) mat_pack(c->P, c->M, &c->x[j * l], rnd, sh); else pss_pack(c->P, &c->x[j * l], rnd, sh);
    for (int p = 0; p < n; p++) c->out[(size_t)p * c->len + j] = sh[p];
  }
}
int zkref_d_pp(const pss_t* P, const fe* num, const fe* den, size_t len, const fe* in_mask, const fe* out_mask, u64 seed,
               int fr_bits, fe* out) {
  const field_t* F = &P->F;
  const size_t m = len * P->l;
  king_mats* KM = NULL;
  if (g_fast_king) {
    KM = malloc(sizeof(king_mats));
    king_mats_build(P, KM);
  }
  dpp_ctx c = {P, KM, num, den, len, malloc(m * sizeof(fe)), malloc(m * sizeof(fe)), out, seed, fr_bits, 0};
  if (g_fast_king) {
    king_parallel(dpp_unpack_span, &c, len);
    king_parallel(dpp_div_span, &c, m);
  } else {
    dpp_unpack_span(&c, 0, len);
    dpp_div_span(&c, 0, m);
  }
  if (!c.zero) {
    for (size_t i = 1; i < m; i++) fe_mul(&c.x[i], &c.x[i], &c.x[i - 1], F);
    if (g_fast_king) king_parallel(dpp_pack_span, &c, len); else dpp_pack_span(&c, 0, len);
  }
  free(c.x);
  free(c.dn);
  free(KM);
  if (c.zero) return 1;
  zkref_deg_red(P, out, len, in_mask, out_mask, seed ^ 0x3333ull, fr_bits);
  return 0;
}
