//! `deg_red` with the reference's signature (`dist-primitives/src/utils/deg_red.rs:80-126`), for `T = F` and for
//! `T` = a curve point (the reference is generic over `T: DomainCoeff<F>`).
use core::ffi::c_void;
use core::ptr;

use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ec::CurveGroup;
use ark_ff::{FftField, PrimeField};
use dist_primitives::utils::deg_red::DegRedMask;
use mpc_net::{MpcNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip_sys as sys;

use crate::net::HipBacked;
use crate::{check, group_of, pack_affine, unpack_affine, DeviceBuf};

/// `deg_red.rs:80-126` over field elements: mask add -> gather -> king `unpack2` then `pack` per chunk -> scatter ->
/// unmask, as `zk_dist_deg_red`.
pub async fn deg_red<F: FftField + PrimeField + 'static, Net: HipBacked>(
    x_share: Vec<F>,
    degred_mask: &DegRedMask<F, F>,
    _pp: &PackedSharingParams<F>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<F>, MpcNetError> {
    let len = x_share.len();
    if degred_mask.in_mask.len() != len || degred_mask.out_mask.len() != len {
        return Err(MpcNetError::BadInput { err: "DegRedMask length differs from the share vector" });   // :92-93
    }
    let ctx = net.ctx();
    let k = net.parties_per_rank();
    let x = DeviceBuf::from_slice(ctx, &x_share)?;
    let im = DeviceBuf::from_slice(ctx, &degred_mask.in_mask)?;
    let om = DeviceBuf::from_slice(ctx, &degred_mask.out_mask)?;
    check(ctx, unsafe {
        sys::zk_dist_deg_red(ctx.raw(), net.raw_net(), sid as i32, x.ptr(), im.ptr(), om.ptr(), len / k, 0,
                             ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(net.raw_net(), sid as i32) })?;
    x.to_vec(len)
}

/// `deg_red.rs:80-126` with `T = G`: shares of group elements (CRS shares, `proving_key.rs:47-123`).
pub async fn deg_red_points<C: SWCurveConfig, Net: HipBacked>(
    x_share: Vec<Projective<C>>,
    degred_mask: &DegRedMask<C::ScalarField, Projective<C>>,
    _pp: &PackedSharingParams<C::ScalarField>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<Projective<C>>, MpcNetError>
where
    C::ScalarField: FftField + PrimeField + 'static,
{
    let len = x_share.len();
    if degred_mask.in_mask.len() != len || degred_mask.out_mask.len() != len {
        return Err(MpcNetError::BadInput { err: "DegRedMask length differs from the share vector" });
    }
    let ctx = net.ctx();
    let k = net.parties_per_rank();
    let aff = |v: &[Projective<C>]| pack_affine(&Projective::<C>::normalize_batch(v));
    let x = DeviceBuf::from_slice(ctx, &aff(&x_share))?;
    let im = DeviceBuf::from_slice(ctx, &aff(&degred_mask.in_mask))?;
    let om = DeviceBuf::from_slice(ctx, &aff(&degred_mask.out_mask))?;
    let out = DeviceBuf::alloc(ctx, x.bytes)?;
    let gen = pack_affine(&[Affine::<C>::generator()]);
    check(ctx, unsafe {
        sys::zk_dist_deg_red_points(ctx.raw(), net.raw_net(), sid as i32, group_of::<C>(), x.ptr(), im.ptr(), om.ptr(),
                                    len / k, gen.as_ptr() as *const c_void, 0, out.ptr(), ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(net.raw_net(), sid as i32) })?;
    let limbs: Vec<u64> = out.to_vec(x.bytes / 8)?;
    Ok(unpack_affine::<C>(&limbs, len).into_iter().map(Into::into).collect())
}

/// `DegRedMask::sample` with `gen = 1` (`deg_red.rs:40-66`) by the library's dealer: one mask per party.
pub fn sample_degred_masks<F: FftField + PrimeField + 'static>(ctx: &crate::Context, len: usize)
                                                                -> Result<Vec<DegRedMask<F, F>>, MpcNetError> {
    let bytes = ctx.n * len * core::mem::size_of::<F>();
    let (im, om) = (DeviceBuf::alloc(ctx, bytes)?, DeviceBuf::alloc(ctx, bytes)?);
    check(ctx, unsafe { sys::zk_degred_mask_sample(ctx.raw(), len, 0, im.ptr(), om.ptr(), ptr::null_mut()) })?;
    let (a, b): (Vec<F>, Vec<F>) = (im.to_vec(ctx.n * len)?, om.to_vec(ctx.n * len)?);
    Ok(a.chunks(len).zip(b.chunks(len)).map(|(x, y)| DegRedMask::new(x.to_vec(), y.to_vec())).collect())
}
