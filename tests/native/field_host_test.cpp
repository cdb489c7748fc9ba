// Host-side check of csrc/field.hpp's inversions by the binary Euclid (Fp::inverse_gcd) and by batched divsteps
// (Fp::inverse_safegcd, the one inversion of d_pp's carry kernel) against
// the Fermat ladder (Fp::inverse) and against a * a^-1 = 1, on the six fields of the library: 3 000 values each, incl. small
// ones, 1 and p - 1.  Built and run by tests/test_native_field.py with the host compiler (the same source compiles for the
// device).
#include <cstdio>
#include <cstdlib>
#include "curves.hpp"
#include "field.hpp"
using namespace zk;
template <class P> int run(const char* name) {
  using F = Fp<P>;
  int bad = 0;
  srand(5);
  for (int it = 0; it < 3000; it++) {
    F a;
    for (int i = 0; i < F::N; i++) a.v[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    if (it < 40) { for (int i = 1; i < F::N; i++) a.v[i] = 0; a.v[0] = it + 1; }
    a.v[F::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;     // below 2^(BITS-1) < p
    if (it == 41) { a = F::zero() - F::one(); }
    if (it == 42) { a = F::one(); }
    if (a.is_zero()) continue;
    F g = a.inverse_gcd(), f = a.inverse(), h = a.inverse_safegcd();
    if (!(g == f) || !(h == f) || !(a * g == F::one())) bad++;
  }
  printf("%s: %d mismatches\n", name, bad);
  return bad;
}
int main() {
  int b = run<Bn254Fr>("bn254 fr") + run<Bls381Fr>("bls381 fr") + run<Bls377Fr>("bls377 fr") + run<Bn254Fq>("bn254 fq") +
          run<Bls381Fq>("bls381 fq") + run<Bls377Fq>("bls377 fq");
  return b != 0;
}
