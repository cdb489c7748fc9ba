// Pippenger kernels for G1 of one curve (own translation unit: see msm_impl.hpp).
// the mixed addition of a 12-limb base field: inline products (see field.hpp ZK_MUL_INLINE_LIMBS)
#define ZK_MUL_INLINE_LIMBS 12
#include "curves.hpp"
#include "msm_impl.hpp"
namespace zk {
ZK_INSTANTIATE_MSM(Bls381Fr, Fp<Bls381Fq>)
}  // namespace zk
