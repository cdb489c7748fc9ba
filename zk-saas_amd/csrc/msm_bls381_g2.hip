// Pippenger kernels for G2 of one curve (own translation unit: see msm_impl.hpp).
#define ZK_MUL_INLINE_LIMBS 12
#include "curves.hpp"
#include "msm_impl.hpp"
namespace zk {
ZK_INSTANTIATE_MSM(Bls381Fr, Fp2<Bls381Fq>)
}  // namespace zk
