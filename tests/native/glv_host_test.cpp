// Host-side dump of csrc/glv.hpp's scalar recoding for tests/test_native_field.py to check with big integers: for each scalar
// field with an endomorphism, 300 scalars k (Montgomery form inside, printed as canonical hex) -> the split k = k1 + lambda k2
// (signed magnitudes) and the joint sparse form digits of (|k1|, |k2|).  One line per scalar:
//   <curve> <k> <neg1> <|k1|> <neg2> <|k2|> <digits of |k1|, least significant first, as -,0,+> <digits of |k2|>
#include <cstdio>
#include <cstdlib>
#include "curves.hpp"
#include "glv.hpp"
using namespace zk;
template <class P> static void hex(const uint32_t* v) {
  for (int i = P::N - 1; i >= 0; i--) printf("%08x", v[i]);
}
template <class P> int run(const char* name) {
  using F = Fp<P>;
  srand(11);
  int fails = 0;
  for (int it = 0; it < 300; it++) {
    F a;
    for (int i = 0; i < F::N; i++) a.v[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    a.v[F::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;     // below 2^(BITS-1) < r
    if (it < 20) { for (int i = 1; i < F::N; i++) a.v[i] = 0; a.v[0] = it; }        // 0, 1, 2, ...
    if (it == 20) a = (F::zero() - F::one()).from_mont();                          // r - 1 (canonical)
    const F km = a.to_mont();
    uint32_t m1[F::N], m2[F::N];
    bool n1, n2;
    if (!glv_split<P>(km, m1, &n1, m2, &n2)) { fails++; continue; }
    std::vector<int8_t> u0, u1;
    jsf_digits<F::N>(m1, m2, u0, u1);
    printf("%s ", name); hex<P>(a.v); printf(" %d ", (int)n1); hex<P>(m1); printf(" %d ", (int)n2); hex<P>(m2); printf(" ");
    for (auto d : u0) putchar(d < 0 ? '-' : (d ? '+' : '0'));
    putchar(' ');
    for (auto d : u1) putchar(d < 0 ? '-' : (d ? '+' : '0'));
    if (u0.empty()) printf("_ _");
    putchar('\n');
  }
  printf("%s: %d refused\n", name, fails);
  return fails;
}
int main() {
  return (run<Bn254Fr>("bn254") + run<Bls381Fr>("bls12_381") + run<Bls377Fr>("bls12_377")) != 0;
}
