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
// value (nonce, idx) = the first candidate below p among the 8-word halves of the blocks with counters 2 idx' ...,
// masked to the modulus' bit length.  `nonce` is a per-context call counter, never a caller-supplied seed, so no two
// pack streams of a context ever share randomness.
ZK_HD uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
ZK_HD void chacha20_block(const uint32_t* key, uint64_t counter, uint64_t nonce, uint32_t* out) {
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
  for (int r = 0; r < 10; r++) {
    ZK_QR(0, 4, 8, 12) ZK_QR(1, 5, 9, 13) ZK_QR(2, 6, 10, 14) ZK_QR(3, 7, 11, 15)
    ZK_QR(0, 5, 10, 15) ZK_QR(1, 6, 11, 12) ZK_QR(2, 7, 8, 13) ZK_QR(3, 4, 9, 14)
  }
#undef ZK_QR
#pragma unroll
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

template <class P>
ZK_HD Fp<P> rand_fp_secure(const uint32_t* key, uint64_t nonce, uint64_t idx) {
  constexpr int N = P::N;
  static_assert(N <= 8, "scalar fields have at most 8 limbs");
  constexpr int TOP = P::BITS - 32 * (N - 1);
  for (uint64_t attempt = 0;; attempt++) {
    uint32_t blk[16];
    // counter: low 40 bits... idx in the low 56 bits, the attempt number above (2^8 attempts never happen)
    chacha20_block(key, idx | (attempt << 56), nonce, blk);
    for (int half = 0; half < 2; half++) {
      Fp<P> r;
#pragma unroll
      for (int i = 0; i < N; i++) r.v[i] = blk[8 * half + i];
      if (TOP < 32) r.v[N - 1] &= ((1u << TOP) - 1);
      if (r.is_canonical()) return r.to_mont();
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

}  // namespace zk
