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

}  // namespace zk
