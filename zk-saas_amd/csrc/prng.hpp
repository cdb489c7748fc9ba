// Counter-based PRNG for share randomness (DESIGN.md "Randomness").
//
// The reference draws the t random points of every `pack` from ark_std::test_rng()/thread_rng()
// (dist-primitives/src/dfft/mod.rs:251, utils/pack.rs:14, utils/deg_red.rs:108).  Reconstructed values
// do not depend on them; so that shares can still be compared bit-for-bit across implementations, the
// stream is defined as: SplitMix64 with initial state mix(seed ^ mix(idx + C0)); draw ceil(bits/64)
// little-endian u64 limbs, mask the top limb to the modulus' bit length, redraw while >= p.  The value is
// a canonical integer; callers convert it to Montgomery form.
#pragma once
#include "field.hpp"
#include "params.hpp"

namespace zk {

ZK_HD uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class P>
ZK_HD Fp<P> rand_fp(uint64_t seed, uint64_t idx) {
  constexpr int N = P::N;
  constexpr int NL = (P::BITS + 63) / 64;
  constexpr int TOPBITS = P::BITS - 64 * (NL - 1);
  uint64_t state = mix64(seed ^ mix64(idx + 0x632BE59BD9B4E019ull));
  for (;;) {
    Fp<P> r = Fp<P>::zero();
#pragma unroll
    for (int i = 0; i < NL; i++) {
      state += 0x9E3779B97F4A7C15ull;
      uint64_t limb = mix64(state);
      if (i == NL - 1 && TOPBITS < 64) limb &= (((uint64_t)1 << TOPBITS) - 1);
      r.v[2 * i] = (uint32_t)limb;
      if (2 * i + 1 < N) r.v[2 * i + 1] = (uint32_t)(limb >> 32);
    }
    // accept if r < p
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)r.v[i] - P::MOD[i] - borrow;
      borrow = (uint32_t)(t >> 63);
    }
    if (borrow) return r.to_mont();
  }
}

// ---- production stream: ChaCha20 (RFC 7539 block function, original 64-bit counter / 64-bit nonce layout) keyed per
// context from the operating system's generator.  The t random points of a `pack` are the only thing that hides the
// secrets, so outside of replay mode (tests that compare shares bit for bit with the oracle) they come from here:
// value (nonce, idx) = the first accepted candidate among the 8-word halves of the blocks with counters idx | attempt << 56
// (rand_fp_secure below).  `nonce` is a per-context call counter, never a caller-supplied seed, so no two
// pack streams of a context ever share randomness.
ZK_HD uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
// DR = double rounds: 10 = ChaCha20 (RFC 7539, pinned on the RFC's vector through zk_chacha20_block), 6 = ChaCha12 -- what
// the production stream runs on: the reference draws its share randomness from rand's StdRng / ThreadRng, which IS
// ChaCha12 (dist-primitives/src/dfft/mod.rs:251, utils/deg_red.rs:108), same block function with fewer rounds
template <int DR>
ZK_HD void chacha_block(const uint32_t* key, uint64_t counter, uint64_t nonce, uint32_t* out) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                    key[4],      key[5],      key[6],      key[7],      (uint32_t)counter, (uint32_t)(counter >> 32),
                    (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = s[i];
#define ZK_QR(a, b, c, d)                   \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
  for (int r = 0; r < DR; r++) {
    ZK_QR(0, 4, 8, 12) ZK_QR(1, 5, 9, 13) ZK_QR(2, 6, 10, 14) ZK_QR(3, 7, 11, 15)
    ZK_QR(0, 5, 10, 15) ZK_QR(1, 6, 11, 12) ZK_QR(2, 7, 8, 13) ZK_QR(3, 4, 9, 14)
  }
#undef ZK_QR
#pragma unroll
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
ZK_HD void chacha20_block(const uint32_t* key, uint64_t counter, uint64_t nonce, uint32_t* out) {
  chacha_block<10>(key, counter, nonce, out);
}
constexpr int RNG_DOUBLE_ROUNDS = 6;      // the production stream: ChaCha12

// Candidate = a whole 32 N-bit half of a block, accepted when it is below K p, K = floor(2^(32 N) / p), and then reduced
// mod p by conditional subtractions of 2^j p: x uniform on [0, K p) makes x mod p exactly uniform.  Acceptance is K p /
// 2^(32 N) -- 0.945 for BN254 Fr (K = 5), 0.906 for BLS12-381 Fr (K = 2), 0.948 for BLS12-377 Fr (K = 13) -- against
// 0.756 for BN254 when candidates are masked to the modulus' bit length: what matters on the GPU is that a WAVE repeats the
// block function until all 64 lanes hold a value (1.2 blocks per draw on average instead of 2.2; the king kernels spent
// 15 % of their issue cycles here, profiles/r03_c2_sq_counters.json).
template <class P>
struct RandWide {
  static constexpr int N = P::N;
  struct Mult {
    uint32_t v[N];
  };
  // floor(2^(32 N) / p), at most 15 (the scalar fields here have at most 3 spare bits + slack)
  static constexpr int K = []() constexpr {
    int k = 0;
    uint64_t acc[N + 1] = {};
    for (;;) {
      uint64_t c = 0;
      for (int i = 0; i < N; i++) {
        uint64_t t = acc[i] + P::MOD[i] + c;
        acc[i] = t & 0xffffffffull;
        c = t >> 32;
      }
      acc[N] += c;
      if (acc[N]) return k;        // (k + 1) p overflowed: K = k
      k++;
      if (k == 15) return k;
    }
  }();
  static constexpr Mult times(int m) {
    Mult r{};
    for (int j = 0; j < m; j++) {
      uint64_t c = 0;
      for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)r.v[i] + P::MOD[i] + c;
        r.v[i] = (uint32_t)t;
        c = t >> 32;
      }
    }
    return r;
  }
};
static_assert(RandWide<Bn254Fr>::K == 5 && RandWide<Bls381Fr>::K == 2 && RandWide<Bls377Fr>::K == 13, "acceptance multiples");
template <class P, int M>
ZK_HD bool rand_lt_mult(const Fp<P>& x) {            // x < M p
  constexpr typename RandWide<P>::Mult m = RandWide<P>::times(M);
  unsigned bw = 0;
  for (int i = 0; i < P::N; i++) (void)__builtin_subc(x.v[i], m.v[i], bw, &bw);
  return bw != 0;
}
template <class P, int M>
ZK_HD void rand_sub_mult_if_ge(Fp<P>& x) {           // x -= M p when x >= M p
  constexpr typename RandWide<P>::Mult m = RandWide<P>::times(M);
  Fp<P> d;
  unsigned bw = 0;
  for (int i = 0; i < P::N; i++) d.v[i] = __builtin_subc(x.v[i], m.v[i], bw, &bw);
  if (!bw) x = d;
}
// candidate `half` (0 / 1) of a block: accepted below K p, then reduced mod p.  The result is a uniform canonical integer
// in [0, p), used AS IT STANDS as the Montgomery representation of the (equally uniform) element r / R: the conversion
// r -> r R of the first three rounds (one Montgomery product per draw) bought nothing
template <class P>
ZK_HD bool rand_candidate_half(const uint32_t* blk, int half, Fp<P>* out) {
  constexpr int N = P::N;
  constexpr int K = RandWide<P>::K;
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = blk[8 * half + i];
  if (!rand_lt_mult<P, K>(r)) return false;
  if constexpr (K >= 8) rand_sub_mult_if_ge<P, 8>(r);
  if constexpr (K >= 4) rand_sub_mult_if_ge<P, 4>(r);
  if constexpr (K >= 2) rand_sub_mult_if_ge<P, 2>(r);
  rand_sub_mult_if_ge<P, 1>(r);
  *out = r;
  return true;
}
template <class P>
ZK_HD Fp<P> rand_fp_secure(const uint32_t* key, uint64_t nonce, uint64_t idx, uint64_t first_attempt = 0) {
  constexpr int N = P::N;
  static_assert(N <= 8, "scalar fields have at most 8 limbs");
  constexpr int K = RandWide<P>::K;
  for (uint64_t attempt = first_attempt;; attempt++) {
    uint32_t blk[16];
    // counter: idx in the low 56 bits, the attempt number above (2^8 attempts never happen)
    chacha_block<RNG_DOUBLE_ROUNDS>(key, idx | (attempt << 56), nonce, blk);
    for (int half = 0; half < 2; half++) {
      Fp<P> r;
      if (rand_candidate_half<P>(blk, half, &r)) return r;
    }
  }
}

// What kernels receive in place of a bare seed: replay stream `seed` (key == nullptr) or the context's ChaCha20 key
// with `seed` = the call's nonce.
struct RngSeed {
  uint64_t seed;
  const uint32_t* key;
  ZK_HD RngSeed plus(uint64_t k) const { return RngSeed{seed + k, key}; }
};
template <class P>
ZK_HD Fp<P> rand_fp(const RngSeed& rs, uint64_t idx) {
  return rs.key ? rand_fp_secure<P>(rs.key, rs.seed, idx) : rand_fp<P>(rs.seed, idx);
}
// The two draws idx, idx + 1 of a chunk (t = 2).  Replay stream: the same values as two rand_fp calls.  Production
// stream: the two 256-bit halves of ONE block (counter idx, attempt 0) -- a block has exactly two candidates, each accepted
// with probability 0.91-0.95; a rejected half is redrawn from blocks of its own (counter idx or idx + 1, attempts >= 1),
// so no key-stream word is used twice.  (Rounds 2-3 spent a block of its own on every draw: two ChaCha20 blocks per chunk
// where this is one ChaCha12 block, 30 % of the instructions.)
template <class P>
ZK_HD void rand_fp_pair(const RngSeed& rs, uint64_t idx, Fp<P>* a, Fp<P>* b) {
  if (!rs.key) {
    *a = rand_fp<P>(rs.seed, idx);
    *b = rand_fp<P>(rs.seed, idx + 1);
    return;
  }
  uint32_t blk[16];
  chacha_block<RNG_DOUBLE_ROUNDS>(rs.key, idx, rs.seed, blk);
  const bool oa = rand_candidate_half<P>(blk, 0, a), ob = rand_candidate_half<P>(blk, 1, b);
  if (!oa) *a = rand_fp_secure<P>(rs.key, rs.seed, idx, 1);
  if (!ob) *b = rand_fp_secure<P>(rs.key, rs.seed, idx + 1, 1);
}

}  // namespace zk
