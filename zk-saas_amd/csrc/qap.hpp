// Circom front end on the device: R1CS -> QAP evaluation vectors (groth16/src/qap.rs:42-89) and the canonical
// byte form of Fr vectors (ark-serialize's CanonicalSerialize for prime fields: little-endian canonical integer,
// which is what mpc-net frames carry, mpc-net/src/ser_net.rs:24-25).
#pragma once
#include "field.hpp"
#include "ntt.hpp"

namespace zk {
#if defined(__HIPCC__)

// One lane per domain element.  A and B are CSR (row pointers [nc+1], wire indices, Montgomery coefficients).
//   i < nc          : a_i = <A_i, w>, b_i = <B_i, w>, c_i = a_i * b_i      (qap.rs:58-70; c from a*b as the circom
//                     reduction does, so the C matrix is never read)
//   nc <= i < nc+ni : a_i = w[i - nc], b_i = c_i = 0                         (qap.rs:72-75)
//   otherwise       : zero padding up to the domain size
// HBM-bound gather: 36 bytes per nonzero (index + coefficient) plus the touched witness elements.
template <class F>
__global__ __launch_bounds__(256) void r1cs_qap_kernel(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ ca,
                                                       const F* __restrict__ va, const uint32_t* __restrict__ pb,
                                                       const uint32_t* __restrict__ cb, const F* __restrict__ vb,
                                                       const F* __restrict__ w, uint32_t nvars, uint32_t nc,
                                                       uint32_t ni, size_t m, F* __restrict__ a, F* __restrict__ b,
                                                       F* __restrict__ c, uint32_t* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  F ai = F::zero(), bi = F::zero(), ci = F::zero();
  if (i < nc) {
    for (uint32_t k = pa[i]; k < pa[i + 1]; k++) {
      uint32_t col = ca[k];
      if (col >= nvars) {
        atomicAdd(bad, 1u);
        continue;
      }
      ai = ai + load_elem(va + k) * load_elem(w + col);
    }
    for (uint32_t k = pb[i]; k < pb[i + 1]; k++) {
      uint32_t col = cb[k];
      if (col >= nvars) {
        atomicAdd(bad, 1u);
        continue;
      }
      bi = bi + load_elem(vb + k) * load_elem(w + col);
    }
    ci = ai * bi;
  } else if (i < (size_t)nc + ni) {
    ai = load_elem(w + (i - nc));
  }
  store_elem(a + i, ai);
  store_elem(b + i, bi);
  store_elem(c + i, ci);
}

// Montgomery -> canonical little-endian bytes (same limb layout, value taken out of Montgomery form)
template <class F>
__global__ void fr_to_bytes_kernel(const F* __restrict__ x, size_t len, F* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) store_elem(out + i, load_elem(x + i).from_mont());
}

// canonical bytes -> Montgomery; a value >= p is what ark-serialize rejects (SerializationError::InvalidData)
template <class F>
__global__ void fr_from_bytes_kernel(const F* __restrict__ in, size_t len, F* __restrict__ out,
                                     uint32_t* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  F v = load_elem(in + i);
  if (!v.is_canonical()) {
    atomicAdd(bad, 1u);
    v = F::zero();
  }
  store_elem(out + i, v.to_mont());
}

#endif  // __HIPCC__
}  // namespace zk
