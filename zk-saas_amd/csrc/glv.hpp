// Host-side scalar recoding for the dealer's point-packing kernels (groth16.hpp pss_pack_points_jsf_kernel, pack_split.hpp):
// Solinas' joint sparse form of a pair of integers and the split of a scalar by the curve's endomorphism.  Plain host code over
// field.hpp (no HIP), so tests/native/field_host_test.cpp checks it on the CPU.
#pragma once
#include <vector>

#include "field.hpp"
#include "glv_params.hpp"

namespace zk {

// Solinas' joint sparse form of two N-limb integers (canonical, not Montgomery): digits in {-1, 0, 1}, least
// significant first, at most 32 N + 1 columns; u0[j] 2^j sums to a, u1[j] 2^j to b.
template <int N>
inline void jsf_digits(const uint32_t* a, const uint32_t* b, std::vector<int8_t>& u0, std::vector<int8_t>& u1) {
  uint32_t k[2][N];
  for (int i = 0; i < N; i++) k[0][i] = a[i], k[1][i] = b[i];
  int d[2] = {0, 0};
  auto nonzero = [&](int i) {
    uint32_t acc = 0;
    for (int q = 0; q < N; q++) acc |= k[i][q];
    return acc != 0 || d[i] != 0;
  };
  u0.clear();
  u1.clear();
  while (nonzero(0) || nonzero(1)) {
    int l[2], u[2] = {0, 0};
    for (int i = 0; i < 2; i++) l[i] = (d[i] + (int)(k[i][0] & 7u)) & 7;
    for (int i = 0; i < 2; i++)
      if (l[i] & 1) {
        u[i] = 2 - (l[i] & 3);
        if ((l[i] == 3 || l[i] == 5) && (l[1 - i] & 3) == 2) u[i] = -u[i];
      }
    for (int i = 0; i < 2; i++) {
      if (2 * d[i] == 1 + u[i]) d[i] = 1 - d[i];
      for (int q = 0; q < N - 1; q++) k[i][q] = (k[i][q] >> 1) | (k[i][q + 1] << 31);
      k[i][N - 1] >>= 1;
    }
    u0.push_back((int8_t)u[0]);
    u1.push_back((int8_t)u[1]);
  }
}

// k = k1 + lambda k2 (mod r) with short k1, k2 (glv_params.hpp; tools/gen_glv.py says where the constants come from).
// k_mont: Montgomery form.  mag1 / mag2: |k1|, |k2| as canonical limbs, neg1 / neg2 their signs.  Returns false when the
// curve has no constants, when the identity k1 + lambda k2 = k does not hold, or when a part is longer than Glv::BITS
// bits -- the caller then keeps the plain digits.
template <class FrP>
inline bool glv_split(const Fp<FrP>& k_mont, uint32_t* mag1, bool* neg1, uint32_t* mag2, bool* neg2) {
  if constexpr (!Glv<FrP>::OK || FrP::N != 8) {
    return false;
  } else {
    using G = Glv<FrP>;
    using Fr = Fp<FrP>;
    constexpr int N = 8, NG = G::NG;
    const Fr kc = k_mont.from_mont();
    auto mulhi = [&](const uint32_t* g, uint32_t* out) {      // (k * g) >> SHIFT as N limbs
      uint32_t t[N + NG + 1] = {0};
      for (int i = 0; i < N; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < NG; j++) {
          const uint64_t v = (uint64_t)kc.v[i] * g[j] + t[i + j] + carry;
          t[i + j] = (uint32_t)v;
          carry = v >> 32;
        }
        t[i + NG] = (uint32_t)carry;
      }
      for (int i = 0; i < N; i++) out[i] = (G::SHIFT / 32 + i) < N + NG ? t[G::SHIFT / 32 + i] : 0u;
    };
    uint32_t c1[N], c2[N];
    mulhi(G::G1C, c1);
    mulhi(G::G2C, c2);
    const Fr C1 = Fr::from_limbs(c1).to_mont(), C2 = Fr::from_limbs(c2).to_mont();
    const Fr K1 = k_mont + C1 * Fr::from_limbs(G::N11).to_mont() + C2 * Fr::from_limbs(G::N12).to_mont();
    const Fr K2 = C1 * Fr::from_limbs(G::N21).to_mont() + C2 * Fr::from_limbs(G::N22).to_mont();
    if (!(K1 + Fr::from_limbs(G::LAMBDA).to_mont() * K2 == k_mont)) return false;
    auto signed_mag = [&](const Fr& x, uint32_t* mag, bool* neg) {
      const Fr a = x.from_mont(), b = x.neg().from_mont();      // the value and r - value: the shorter one is the magnitude
      int top_a = -1, top_b = -1;
      for (int i = 0; i < N * 32; i++) {
        if ((a.v[i >> 5] >> (i & 31)) & 1u) top_a = i;
        if ((b.v[i >> 5] >> (i & 31)) & 1u) top_b = i;
      }
      *neg = top_b < top_a;
      const Fr& m = *neg ? b : a;
      for (int i = 0; i < N; i++) mag[i] = m.v[i];
      return (*neg ? top_b : top_a) < G::BITS;
    };
    return signed_mag(K1, mag1, neg1) && signed_mag(K2, mag2, neg2);
  }
}

}  // namespace zk
