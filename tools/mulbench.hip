// Micro-benchmark + cross-check of Montgomery multiply variants on the GPU (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mulbench.hip -o gpurun_out/mulbench && gpurun_out/mulbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zk-saas_amd/csrc/field.hpp"
using namespace zk;

template <class F, int VAR>
__device__ __forceinline__ F mulv(const F& a, const F& b) {
  if constexpr (VAR == 0) return F::mul_ref(a, b);
  else if constexpr (VAR == 2) return F::mul_fips(a, b);
  else return F::mul_pairs(a, b);
}

// squaring chains: variant 2 = the production multiplier called with equal operands, variant 3 = sqr_fips
template <class F, int VAR>
__global__ void sqr_chain(F* x, const F* y, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  F a = x[i], b = y[i];
  for (int k = 0; k < iters; k++) {
    a = VAR == 3 ? F::sqr_fips(a) : F::mul_fips(a, a);
    b = VAR == 3 ? F::sqr_fips(b) : F::mul_fips(b, b);
  }
  x[i] = a + b;
}
template <class F>
__global__ void check_sqr(const F* x, int* bad, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (F::mul_ref(x[i], x[i]) != F::sqr_fips(x[i])) atomicAdd(bad, 1);
}

template <class F, int VAR>
__global__ void chain(F* x, const F* y, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  F a = x[i], b = y[i];
  for (int k = 0; k < iters; k++) {
    a = mulv<F, VAR>(a, b);
    b = mulv<F, VAR>(b, a);
  }
  x[i] = a + b;
}
template <class F>
__global__ void check(const F* x, const F* y, int* bad, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F r0 = F::mul_ref(x[i], y[i]);
  F r1 = F::mul_pairs(x[i], y[i]);
  F r2 = F::mul_fips(x[i], y[i]);
  if (r0 != r1 || r0 != r2) atomicAdd(bad, 1);
}

template <class F>
void run(const char* name) {
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
    if (i < 4) {   // edge values: 0, p-1-ish, all ones below the top limb
      for (int k = 0; k < F::N; k++) hx[i].v[k] = i == 0 ? 0 : (i == 1 ? F::Params::MOD[k] : 0xffffffffu);
      if (i == 1) hx[i].v[0] -= 1;
      if (i >= 2) hx[i].v[F::N - 1] = F::Params::MOD[F::N - 1] - 1;
    }
  }
  F *dx, *dy;
  int* dbad;
  hipMalloc(&dx, n * sizeof(F));
  hipMalloc(&dy, n * sizeof(F));
  hipMalloc(&dbad, 4);
  hipMemcpy(dx, hx.data(), n * sizeof(F), hipMemcpyHostToDevice);
  hipMemcpy(dy, hy.data(), n * sizeof(F), hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 4);
  check<F><<<n / 256, 256>>>(dx, dy, dbad, n);
  int bad = -1;
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("%s: mismatches new-vs-ref = %d of %zu\n", name, bad, n);
  hipMemset(dbad, 0, 4);
  check_sqr<F><<<n / 256, 256>>>(dx, dbad, n);
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("%s: mismatches sqr_fips-vs-ref = %d of %zu\n", name, bad, n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200;
  for (int var = 0; var < 3; var++) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemcpy(dx, hx.data(), n * sizeof(F), hipMemcpyHostToDevice);
      hipEventRecord(e0);
      if (var == 0) chain<F, 0><<<n / 256, 256>>>(dx, dy, iters);
      else if (var == 2) chain<F, 2><<<n / 256, 256>>>(dx, dy, iters);
      else chain<F, 1><<<n / 256, 256>>>(dx, dy, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s variant %d: %.3f ms for %.1f M mul -> %.1f G mul/s\n", name, var, ms, n * 2.0 * iters / 1e6,
                      n * 2.0 * iters / ms / 1e6);
    }
  }
  for (int var = 2; var <= 3; var++) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemcpy(dx, hx.data(), n * sizeof(F), hipMemcpyHostToDevice);
      hipEventRecord(e0);
      if (var == 2) sqr_chain<F, 2><<<n / 256, 256>>>(dx, dy, iters);
      else sqr_chain<F, 3><<<n / 256, 256>>>(dx, dy, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s SQUARING %s: %.3f ms for %.1f M sqr -> %.1f G sqr/s\n", name, var == 2 ? "mul_fips(a, a)" : "sqr_fips(a)",
                      ms, n * 2.0 * iters / 1e6, n * 2.0 * iters / ms / 1e6);
    }
  }
  // single-wave latency: 64 lanes only
  for (int var = 0; var < 3; var++) {
    hipEventRecord(e0);
    if (var == 0) chain<F, 0><<<1, 64>>>(dx, dy, 2000);
    else if (var == 2) chain<F, 2><<<1, 64>>>(dx, dy, 2000);
    else chain<F, 1><<<1, 64>>>(dx, dy, 2000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s variant %d: single wave %.3f us per dependent mul\n", name, var, ms * 1e3 / 4000);
  }
}

int main() {
  run<Fp<Bn254Fr>>("bn254_fr");
  run<Fp<Bn254Fq>>("bn254_fq");
  run<Fp<Bls381Fq>>("bls381_fq(12 limbs)");
  return 0;
}
