// Go / no-go experiment: a 256-bit Montgomery multiplication on the FP64 FMA pipe -- five limbs of
// 52 bits held as doubles, every 52x52-bit limb product split exactly into a high and a low half by two fused
// multiply-adds in round-toward-zero mode (p_hi = fma(a, b, 2^104), p_lo = fma(a, b, 2^104 + 2^52 - p_hi)), column sums
// accumulated as 64-bit integers over the bit patterns -- against the v_mad_u64_u32 multiplier of csrc/field.hpp, in the
// harness of tools/mulbench.hip: throughput of a dependent chain over 2^20 lanes, single-wave latency, and
// BIT-EXACTNESS on 2^20 random pairs (r52 * 2^260 == a * b mod p, checked with the production multiplier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mulbench52.hip -o gpurun_out/mulbench52 && gpurun_out/mulbench52
// Adoption rule: >= 1.2x on BN254 Fq AND bit-exact.  The result is recorded in profiles/ either way.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../zk-saas_amd/csrc/field.hpp"
using namespace zk;

constexpr uint64_t MASK52 = (1ull << 52) - 1;
constexpr long long B52 = 0x4330000000000000ll;    // bit pattern of 2^52
constexpr long long B104 = 0x4670000000000000ll;   // bit pattern of 2^104

struct F52 {
  double v[5];
};
struct Mod52 {
  double p[5];
  uint64_t pi[5];
  uint64_t n0;      // -p^-1 mod 2^52
};

__device__ __forceinline__ void set_round_toward_zero_f64() {
  // MODE register, FP_ROUND bits [3:2] = double/half rounding: 3 = toward zero.  Inline asm on purpose: the compiler's
  // mode-register pass restores the default rounding mode after __builtin_amdgcn_s_setreg before the first FP operation.
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}
__device__ __forceinline__ double u52_to_double(uint64_t x) { return __longlong_as_double((long long)(x | (uint64_t)B52)) - 0x1p52; }

// r = a * b * 2^-260 mod p, limbs < 2^52, result fully reduced
__device__ __forceinline__ F52 mul52(const F52& a, const F52& b, const Mod52& M) {
  const double C1 = 0x1p104, C12 = 0x1p104 + 0x1p52;
  long long acc[11];
#pragma unroll
  for (int k = 0; k < 11; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 5; j++) {
      double ph = __builtin_fma(a.v[i], b.v[j], C1);
      double pl = __builtin_fma(a.v[i], b.v[j], C12 - ph);
      acc[i + j + 1] += __double_as_longlong(ph) - B104;
      acc[i + j] += __double_as_longlong(pl) - B52;
    }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    uint64_t lo = (uint64_t)acc[i] & MASK52;
    uint64_t q = (lo * M.n0) & MASK52;
    double qd = u52_to_double(q);
#pragma unroll
    for (int j = 0; j < 5; j++) {
      double ph = __builtin_fma(qd, M.p[j], C1);
      double pl = __builtin_fma(qd, M.p[j], C12 - ph);
      acc[i + j + 1] += __double_as_longlong(ph) - B104;
      acc[i + j] += __double_as_longlong(pl) - B52;
    }
    acc[i + 1] += acc[i] >> 52;          // acc[i] is a multiple of 2^52 now
  }
  uint64_t t[5];
  long long carry = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    long long s = acc[5 + k] + carry;
    t[k] = (uint64_t)s & MASK52;
    carry = s >> 52;
  }
  // t < 2p: subtract p if t >= p
  uint64_t d[5];
  long long borrow = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    long long s = (long long)t[k] - (long long)M.pi[k] + borrow;
    d[k] = (uint64_t)s & MASK52;
    borrow = s >> 52;                    // 0 or -1
  }
  const bool ge = borrow == 0;           // (the carry out of the top limb is 0: t < 2p < 2^260)
  (void)carry;
  F52 r;
#pragma unroll
  for (int k = 0; k < 5; k++) r.v[k] = u52_to_double(ge ? d[k] : t[k]);
  return r;
}

template <class F>
__device__ __forceinline__ F52 to52(const F& x) {       // the 8 x 32-bit limbs as an integer -> 5 x 52-bit limbs
  static_assert(F::N == 8, "256-bit fields");
  const uint64_t l0 = x.v[0] | ((uint64_t)x.v[1] << 32), l1 = x.v[2] | ((uint64_t)x.v[3] << 32),
                 l2 = x.v[4] | ((uint64_t)x.v[5] << 32), l3 = x.v[6] | ((uint64_t)x.v[7] << 32);
  const uint64_t w[5] = {l0 & MASK52, ((l0 >> 52) | (l1 << 12)) & MASK52, ((l1 >> 40) | (l2 << 24)) & MASK52,
                         ((l2 >> 28) | (l3 << 36)) & MASK52, l3 >> 16};
  F52 r;
#pragma unroll
  for (int k = 0; k < 5; k++) r.v[k] = u52_to_double(w[k]);
  return r;
}
template <class F>
__device__ __forceinline__ F from52(const F52& x) {
  uint64_t w[5];
#pragma unroll
  for (int k = 0; k < 5; k++) w[k] = (uint64_t)__double_as_longlong(x.v[k] + 0x1p52) & MASK52;   // exact: integers below 2^52
  const uint64_t l0 = w[0] | (w[1] << 52), l1 = (w[1] >> 12) | (w[2] << 40), l2 = (w[2] >> 24) | (w[3] << 28),
                 l3 = (w[3] >> 36) | (w[4] << 16);
  F r;
  r.v[0] = (uint32_t)l0, r.v[1] = (uint32_t)(l0 >> 32), r.v[2] = (uint32_t)l1, r.v[3] = (uint32_t)(l1 >> 32);
  r.v[4] = (uint32_t)l2, r.v[5] = (uint32_t)(l2 >> 32), r.v[6] = (uint32_t)l3, r.v[7] = (uint32_t)(l3 >> 32);
  return r;
}

template <class F>
__global__ void chain52(F* x, const F* y, Mod52 M, int iters) {
  set_round_toward_zero_f64();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  F52 a = to52(x[i]), b = to52(y[i]);
  for (int k = 0; k < iters; k++) {
    a = mul52(a, b, M);
    b = mul52(b, a, M);
  }
  x[i] = from52<F>(a) + from52<F>(b);
}
template <class F>
__global__ void chain32(F* x, const F* y, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  F a = x[i], b = y[i];
  for (int k = 0; k < iters; k++) {
    a = F::mul_inline(a, b);
    b = F::mul_inline(b, a);
  }
  x[i] = a + b;
}
// r52 * 2^260 == a * b (mod p)  <=>  mont256(r52, 2^260 mod p) == mont256(a, b)
template <class F>
__global__ void check52(const F* x, const F* y, Mod52 M, F c260, int* bad, size_t n) {
  set_round_toward_zero_f64();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F r = from52<F>(mul52(to52(x[i]), to52(y[i]), M));
  if (F::mul_ref(r, c260) != F::mul_ref(x[i], y[i]) || !r.is_canonical()) atomicAdd(bad, 1);
}

template <class F>
void run(const char* name) {
  using P = typename F::Params;
  const size_t n = 1 << 20;
  std::vector<F> hx(n), hy(n);
  srand(1);
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < F::N; k++) {
      hx[i].v[k] = ((uint32_t)rand() << 16) ^ rand();
      hy[i].v[k] = ((uint32_t)rand() << 16) ^ rand();
    }
    hx[i].v[F::N - 1] &= 0x0fffffff;
    hy[i].v[F::N - 1] &= 0x0fffffff;
    if (i < 4) {   // edge values: 0, p-1, all ones below the top limb
      for (int k = 0; k < F::N; k++) hx[i].v[k] = i == 0 ? 0 : (i == 1 ? P::MOD[k] : 0xffffffffu);
      if (i == 1) hx[i].v[0] -= 1;
      if (i >= 2) hx[i].v[F::N - 1] = P::MOD[F::N - 1] - 1;
    }
  }
  // modulus as 52-bit limbs, -p^-1 mod 2^52, 2^260 mod p
  Mod52 M;
  for (int k = 0; k < 5; k++) {
    uint64_t w = 0;
    for (int b = 0; b < 52; b++) {
      int bit = 52 * k + b;
      if (bit < 32 * F::N && ((P::MOD[bit / 32] >> (bit % 32)) & 1u)) w |= 1ull << b;
    }
    M.pi[k] = w;
    M.p[k] = (double)w;
  }
  uint64_t p0 = (uint64_t)P::MOD[0] | ((uint64_t)P::MOD[1] << 32);
  uint64_t xinv = P::N0INV;               // -p^-1 mod 2^32, one Newton step -> mod 2^64
  xinv = xinv * (2 + p0 * xinv);
  M.n0 = xinv & MASK52;
  F c260 = F::zero();
  c260.v[0] = 1;
  for (int k = 0; k < 260; k++) c260 = c260 + c260;      // host: modular doubling of the integer 1
  F *dx, *dy;
  int* dbad;
  hipMalloc(&dx, n * sizeof(F));
  hipMalloc(&dy, n * sizeof(F));
  hipMalloc(&dbad, 4);
  hipMemcpy(dx, hx.data(), n * sizeof(F), hipMemcpyHostToDevice);
  hipMemcpy(dy, hy.data(), n * sizeof(F), hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 4);
  check52<F><<<n / 256, 256>>>(dx, dy, M, c260, dbad, n);
  int bad = -1;
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("%s: fp64/52-bit multiplier, mismatches vs the production multiplier = %d of %zu pairs\n", name, bad, n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200;
  double rate[2] = {0, 0};
  for (int var = 0; var < 2; var++) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemcpy(dx, hx.data(), n * sizeof(F), hipMemcpyHostToDevice);
      hipEventRecord(e0);
      if (var == 0) chain32<F><<<n / 256, 256>>>(dx, dy, iters);
      else chain52<F><<<n / 256, 256>>>(dx, dy, M, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) {
        rate[var] = n * 2.0 * iters / ms / 1e6;
        printf("%s %s: %.3f ms for %.1f M mul -> %.1f G mul/s\n", name, var ? "fp64 52-bit limbs " : "v_mad_u64_u32 (prod)",
               ms, n * 2.0 * iters / 1e6, rate[var]);
      }
    }
  }
  for (int var = 0; var < 2; var++) {
    hipEventRecord(e0);
    if (var == 0) chain32<F><<<1, 64>>>(dx, dy, 2000);
    else chain52<F><<<1, 64>>>(dx, dy, M, 2000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s %s: single wave %.3f us per dependent mul\n", name, var ? "fp64 52-bit limbs " : "v_mad_u64_u32 (prod)",
           ms * 1e3 / 4000);
  }
  printf("%s: speedup %.2fx, bit-exact: %s  => %s\n", name, rate[1] / rate[0], bad == 0 ? "yes" : "NO",
         (bad == 0 && rate[1] / rate[0] >= 1.2) ? "GO" : "NO-GO (adoption needs >= 1.2x and bit-exactness)");
}

int main() {
  run<Fp<Bn254Fq>>("bn254_fq");
  run<Fp<Bn254Fr>>("bn254_fr");
  return 0;
}
