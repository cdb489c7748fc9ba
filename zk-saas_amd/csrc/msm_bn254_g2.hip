// Pippenger kernels for G2 of one curve (own translation unit: see msm_impl.hpp).
#include "curves.hpp"
#include "msm_impl.hpp"
namespace zk {
ZK_INSTANTIATE_MSM(Bn254Fr, Fp2<Bn254Fq>)
}  // namespace zk
