// Pippenger kernels for G1 of one curve (own translation unit: see msm_impl.hpp).
#include "curves.hpp"
#include "msm_impl.hpp"
namespace zk {
ZK_INSTANTIATE_MSM(Bls381Fr, Fp<Bls381Fq>)
}  // namespace zk
