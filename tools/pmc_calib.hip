// FETCH_SIZE / WRITE_SIZE calibration (MI355X_MICROARCH.md "HBM": "Other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern before trusting an absolute").  One kernel per ACCESS SHAPE the library's
// streaming kernels use, each moving an exactly known number of bytes between two buffers far larger than the Infinity
// Cache (so that every byte comes from / goes to HBM), run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two
// passes; tools/pmc_calibrate.py divides known by reported and records the factor per shape, which tools/pmc_summary.py
// then applies instead of a per-kernel yes/no doubling.
//
//   calib_lane16        : one 16-byte load + store per lane, consecutive lanes consecutive (the guide's calibrated case)
//   calib_elem32        : one 32-byte element per lane as two 16-byte accesses 16 bytes apart (load_elem / store_elem:
//                         king / deg_red / d_pp / vector kernels; a wave instruction covers 64 x 32 B with 16-byte holes)
//   calib_ntt_pass0     : ntt_pass_kernel's tile load of the FIRST pass: thread t loads elements t, t + T, t + 2T, t + 3T
//                         of a contiguous 4T-element tile (T = 512), stores the same way
//   calib_ntt_pass1     : ... of a LATER pass: a tile is R rows of C = 8 consecutive elements, rows 2^s0 elements apart
//                         (256-byte runs at a large stride)
//   calib_rows8         : the king kernels' gather: lane j reads element j of 8 party rows (row pitch = len), writes 8 rows
//   calib_gather64      : one 64-byte affine point per lane at a pseudo-random index (the MSM accumulate's base fetch)
//
// build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
// run  : rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dirF> -o p -- tools/pmc_calib   (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct E32 {
  uint4 a, b;
};

__global__ void calib_lane16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void calib_elem32(const E32* __restrict__ in, E32* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  E32 v;
  v.a = in[i].a;
  v.b = in[i].b;
  out[i].a = v.a;
  out[i].b = v.b;
}
__global__ void calib_ntt_pass0(const E32* __restrict__ in, E32* __restrict__ out) {
  const size_t base = (size_t)blockIdx.x * 2048;
  E32 v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    v[q].a = in[base + threadIdx.x + q * 512].a;
    v[q].b = in[base + threadIdx.x + q * 512].b;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    out[base + threadIdx.x + q * 512].a = v[q].a;
    out[base + threadIdx.x + q * 512].b = v[q].b;
  }
}
// tile = 256 rows x 8 columns; row r of tile (h, c0) is element (h << s1) + (r << s0) + c0 + c; s0 = 11, s1 = 19
__global__ void calib_ntt_pass1(const E32* __restrict__ in, E32* __restrict__ out, int s0, int s1) {
  const uint32_t tiles_per_h = 1u << (s0 - 3);
  const size_t h = blockIdx.x / tiles_per_h;
  const uint32_t c0 = (blockIdx.x % tiles_per_h) << 3;
  E32 v[4];
  size_t gi[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t x = threadIdx.x + q * 512, c = x & 7, r = x >> 3;
    gi[q] = (h << s1) + ((size_t)r << s0) + c0 + c;
    v[q].a = in[gi[q]].a;
    v[q].b = in[gi[q]].b;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    out[gi[q]].a = v[q].a;
    out[gi[q]].b = v[q].b;
  }
}
__global__ void calib_rows8(const E32* __restrict__ in, E32* __restrict__ out, size_t len) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= len) return;
  E32 v[8];
#pragma unroll
  for (int p = 0; p < 8; p++) {
    v[p].a = in[p * len + j].a;
    v[p].b = in[p * len + j].b;
  }
#pragma unroll
  for (int p = 0; p < 8; p++) {
    out[p * len + j].a = v[p].a;
    out[p * len + j].b = v[p].b;
  }
}
struct P64 {
  uint4 q[4];
};
__global__ void calib_gather64(const P64* __restrict__ in, P64* __restrict__ out, size_t n, size_t npts) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // a permutation-like scatter of indices (odd multiplier mod a power of two): every point fetched exactly once
  const size_t j = (i * 0x9E3779B97F4A7C15ull) & (npts - 1);
  P64 v = in[j];
  out[i] = v;
}

int main() {
  const size_t bytes = (size_t)2 << 30;                    // 2 GiB in, 2 GiB out: 8x the Infinity Cache each
  void *in = nullptr, *out = nullptr;
  CHECK(hipMalloc(&in, bytes));
  CHECK(hipMalloc(&out, bytes));
  CHECK(hipMemset(in, 1, bytes));
  CHECK(hipMemset(out, 0, bytes));
  CHECK(hipDeviceSynchronize());
  const size_t n16 = bytes / 16, n32 = bytes / 32, n64 = bytes / 64;
  for (int rep = 0; rep < 3; rep++) {
    calib_lane16<<<dim3((unsigned)(n16 / 256)), dim3(256)>>>((const uint4*)in, (uint4*)out, n16);
    calib_elem32<<<dim3((unsigned)(n32 / 256)), dim3(256)>>>((const E32*)in, (E32*)out, n32);
    calib_ntt_pass0<<<dim3((unsigned)(n32 / 2048)), dim3(512)>>>((const E32*)in, (E32*)out);
    calib_ntt_pass1<<<dim3((unsigned)(n32 / 2048)), dim3(512)>>>((const E32*)in, (E32*)out, 11, 19);
    calib_rows8<<<dim3((unsigned)(n32 / 8 / 256)), dim3(256)>>>((const E32*)in, (E32*)out, n32 / 8);
    calib_gather64<<<dim3((unsigned)(n64 / 256)), dim3(256)>>>((const P64*)in, (P64*)out, n64, n64);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
  }
  // every kernel reads `bytes` and writes `bytes`
  printf("{\"bytes_read_per_launch\": %zu, \"bytes_written_per_launch\": %zu}\n", bytes, bytes);
  CHECK(hipFree(in));
  CHECK(hipFree(out));
  return 0;
}
