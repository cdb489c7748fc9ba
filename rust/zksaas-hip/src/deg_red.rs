//! `deg_red` behind the reference's signature (`dist-primitives/src/utils/deg_red.rs:80-126`), generic over
//! `T: DomainCoeff<F>` like the reference: `T = F` (field elements) and `T` = a curve group (CRS shares,
//! `proving_key.rs:47-123`) are told apart by `TypeId`, which is why `T` carries `'static` here -- the one bound the
//! patch adds to `dist-primitives`' own `deg_red` (every type the reference instantiates it with is `'static`).
//! `degred_mask: &DegRedMask<F, T>` arrives as `in_mask, out_mask`.
use core::any::TypeId;
use core::ffi::c_void;
use core::ptr;

use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ec::{AffineRepr, CurveGroup};
use ark_ff::{FftField, PrimeField};
use ark_poly::domain::DomainCoeff;
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use ark_std::UniformRand;
use mpc_net::ser_net::MpcSerNet;
use mpc_net::{MpcNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip_sys as sys;

use crate::{check, group_of, pack_affine, same_slice, same_vec, unpack_affine, Context, DeviceBuf, HipNet};

/// `deg_red.rs:80-126`: mask add -> gather -> king `unpack2` then `pack` per chunk -> scatter -> unmask.
pub async fn deg_red<
    F: FftField,
    T: DomainCoeff<F> + CanonicalSerialize + CanonicalDeserialize + UniformRand + 'static,
    Net: MpcSerNet,
>(
    x_share: Vec<T>,
    in_mask: &[T],
    out_mask: &[T],
    pp: &PackedSharingParams<F>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<T>, MpcNetError> {
    let hip = HipNet::of(net)?;
    hip.ctx().expect_field::<F>(pp.l)?;
    let len = x_share.len();
    if in_mask.len() != len || out_mask.len() != len {
        return Err(MpcNetError::BadInput { err: "DegRedMask length differs from the share vector" });   // :92-93
    }
    if TypeId::of::<T>() == TypeId::of::<F>() {
        let x = same_vec::<T, F>(x_share).map_err(|_| MpcNetError::BadInput { err: "unreachable" })?;
        let (im, om) = (same_slice::<T, F>(in_mask).unwrap(), same_slice::<T, F>(out_mask).unwrap());
        let out = deg_red_field(hip, x, im, om, sid)?;
        return same_vec::<F, T>(out).map_err(|_| MpcNetError::BadInput { err: "unreachable" });
    }
    crate::sw_dispatch!(T, C => {
        let x = same_vec::<T, Projective<C>>(x_share).map_err(|_| MpcNetError::BadInput { err: "unreachable" })?;
        let im = same_slice::<T, Projective<C>>(in_mask).unwrap();
        let om = same_slice::<T, Projective<C>>(out_mask).unwrap();
        let out = deg_red_points::<C>(hip, x, im, om, sid)?;
        same_vec::<Projective<C>, T>(out).map_err(|_| MpcNetError::BadInput { err: "unreachable" })
    })
}

fn deg_red_field<F: FftField>(hip: &HipNet, x_share: Vec<F>, in_mask: &[F], out_mask: &[F], sid: MultiplexedStreamID)
                              -> Result<Vec<F>, MpcNetError> {
    let (ctx, k, len) = (hip.ctx(), hip.parties_per_rank(), x_share.len());
    let x = DeviceBuf::from_slice(ctx, &x_share)?;
    let im = DeviceBuf::from_slice(ctx, in_mask)?;
    let om = DeviceBuf::from_slice(ctx, out_mask)?;
    check(ctx, unsafe {
        sys::zk_dist_deg_red(ctx.raw(), hip.raw_net(), sid as i32, x.ptr(), im.ptr(), om.ptr(), len / k, 0,
                             ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid as i32) })?;
    x.to_vec(len)
}

/// `deg_red.rs:80-126` with `T = G`: shares of group elements.
pub fn deg_red_points<C: SWCurveConfig>(hip: &HipNet, x_share: Vec<Projective<C>>, in_mask: &[Projective<C>],
                                        out_mask: &[Projective<C>], sid: MultiplexedStreamID)
                                        -> Result<Vec<Projective<C>>, MpcNetError> {
    let (ctx, k, len) = (hip.ctx(), hip.parties_per_rank(), x_share.len());
    let aff = |v: &[Projective<C>]| pack_affine(&Projective::<C>::normalize_batch(v));
    let x = DeviceBuf::from_slice(ctx, &aff(&x_share))?;
    let im = DeviceBuf::from_slice(ctx, &aff(in_mask))?;
    let om = DeviceBuf::from_slice(ctx, &aff(out_mask))?;
    let out = DeviceBuf::alloc(ctx, x.bytes)?;
    let gen = pack_affine(&[Affine::<C>::generator()]);
    check(ctx, unsafe {
        sys::zk_dist_deg_red_points(ctx.raw(), hip.raw_net(), sid as i32, group_of::<C>(), x.ptr(), im.ptr(), om.ptr(),
                                    len / k, gen.as_ptr() as *const c_void, 0, out.ptr(), ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid as i32) })?;
    let limbs: Vec<u64> = out.to_vec(x.bytes / 8)?;
    Ok(unpack_affine::<C>(&limbs, len).into_iter().map(Into::into).collect())
}

/// `DegRedMask::sample` with `gen = 1` (`deg_red.rs:40-66`) by the library's dealer: one `(in_mask, out_mask)` per party.
pub fn sample_degred_masks<F: FftField + PrimeField>(ctx: &Context, len: usize)
                                                     -> Result<Vec<(Vec<F>, Vec<F>)>, MpcNetError> {
    let bytes = ctx.n * len * core::mem::size_of::<F>();
    let (im, om) = (DeviceBuf::alloc(ctx, bytes)?, DeviceBuf::alloc(ctx, bytes)?);
    check(ctx, unsafe { sys::zk_degred_mask_sample(ctx.raw(), len, 0, im.ptr(), om.ptr(), ptr::null_mut()) })?;
    let (a, b): (Vec<F>, Vec<F>) = (im.to_vec(ctx.n * len)?, om.to_vec(ctx.n * len)?);
    Ok(a.chunks(len).zip(b.chunks(len)).map(|(x, y)| (x.to_vec(), y.to_vec())).collect())
}
