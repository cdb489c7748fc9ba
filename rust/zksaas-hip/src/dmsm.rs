//! `d_msm` with the reference's signature (`dist-primitives/src/dmsm/mod.rs:59-102`): the local `G::msm` (`:73`) is the
//! Pippenger bucket MSM of `csrc/msm.hpp`; the king's `unpack2` + sum (`:85-86`) is one linear form over the parties'
//! results, gathered and broadcast through `zk_dist_d_msm`.
use core::ffi::c_void;
use core::ptr;

use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::PrimeField;
use dist_primitives::dmsm::MsmMask;
use mpc_net::{MpcNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip_sys as sys;

use crate::net::HipBacked;
use crate::{check, group_of, pack_affine, pack_jacobian, unpack_jacobian, DeviceBuf};

/// `dist-primitives/src/dmsm/mod.rs:59-102` for `G = Projective<C>` (every curve of the reference is short
/// Weierstrass).  A length mismatch is the `Generic(min_len.to_string())` the reference gets from `G::msm` (`:73`).
pub async fn d_msm<C: SWCurveConfig, Net: HipBacked>(
    bases: &[Affine<C>],
    scalars: &[C::ScalarField],
    msm_mask: &MsmMask<Projective<C>>,
    _pp: &PackedSharingParams<C::ScalarField>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Projective<C>, MpcNetError>
where
    C::ScalarField: PrimeField + 'static,
{
    if bases.len() != scalars.len() {
        return Err(MpcNetError::Generic(bases.len().min(scalars.len()).to_string()));
    }
    let ctx = net.ctx();
    let k = net.parties_per_rank();
    if k != 1 {
        return Err(MpcNetError::BadInput { err: "d_msm: this signature carries one party's vectors (see d_msm_rows)" });
    }
    let b = DeviceBuf::from_slice(ctx, &pack_affine(bases))?;
    let s = DeviceBuf::from_slice(ctx, scalars)?;
    let im = pack_jacobian(&[msm_mask.in_mask]);
    let om = pack_jacobian(&[msm_mask.out_mask]);
    let mut out = vec![0u64; im.len()];
    check(ctx, unsafe {
        sys::zk_dist_d_msm(ctx.raw(), net.raw_net(), sid as i32, group_of::<C>(), b.ptr(), s.ptr(), bases.len(),
                           im.as_ptr() as *const c_void, om.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void,
                           ptr::null_mut())
    })?;
    Ok(unpack_jacobian::<C>(&out, 1)[0])
}

/// The same for a rank that drives `k` parties: rows `[k][len]`, `k` masks, `k` results.
pub async fn d_msm_rows<C: SWCurveConfig, Net: HipBacked>(
    bases: &[Affine<C>],
    scalars: &[C::ScalarField],
    len: usize,
    masks: &[MsmMask<Projective<C>>],
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<Projective<C>>, MpcNetError>
where
    C::ScalarField: PrimeField + 'static,
{
    let k = net.parties_per_rank();
    if bases.len() != k * len || scalars.len() != k * len || masks.len() != k {
        return Err(MpcNetError::BadInput { err: "d_msm_rows: [k][len] rows and k masks expected" });
    }
    let ctx = net.ctx();
    let b = DeviceBuf::from_slice(ctx, &pack_affine(bases))?;
    let s = DeviceBuf::from_slice(ctx, scalars)?;
    let im = pack_jacobian(&masks.iter().map(|m| m.in_mask).collect::<Vec<_>>());
    let om = pack_jacobian(&masks.iter().map(|m| m.out_mask).collect::<Vec<_>>());
    let mut out = vec![0u64; im.len()];
    check(ctx, unsafe {
        sys::zk_dist_d_msm(ctx.raw(), net.raw_net(), sid as i32, group_of::<C>(), b.ptr(), s.ptr(), len,
                           im.as_ptr() as *const c_void, om.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void,
                           ptr::null_mut())
    })?;
    Ok(unpack_jacobian::<C>(&out, k))
}

/// `MsmMask::sample` (`dmsm/mod.rs:21-47`) by the library's dealer: one mask per party.
pub fn sample_msm_masks<C: SWCurveConfig>(ctx: &crate::Context, gen: Affine<C>)
                                          -> Result<Vec<MsmMask<Projective<C>>>, MpcNetError> {
    let g = pack_affine(&[gen]);
    let per = 3 * g.len() / 2;
    let (mut im, mut om) = (vec![0u64; ctx.n * per], vec![0u64; ctx.n * per]);
    check(ctx, unsafe {
        sys::zk_msm_mask_sample(ctx.raw(), group_of::<C>(), g.as_ptr() as *const c_void, 0,
                                im.as_mut_ptr() as *mut c_void, om.as_mut_ptr() as *mut c_void)
    })?;
    let (a, b) = (unpack_jacobian::<C>(&im, ctx.n), unpack_jacobian::<C>(&om, ctx.n));
    Ok(a.into_iter().zip(b).map(|(x, y)| MsmMask::new(x, y)).collect())
}
