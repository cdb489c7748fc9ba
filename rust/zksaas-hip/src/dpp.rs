//! `d_pp` behind the reference's signature (`dist-primitives/src/dpp/mod.rs:15-87`).
//!
//! The king's closure (`:40-73`: unpack `num || den`, divide, prefix product, `pack_vec`) runs as three launches of
//! `csrc/dpp.hpp` -- a prefix scan of the numerators, a suffix scan of the denominators and ONE inversion for the whole
//! vector give the same field elements as the reference's `inverse()` per element (`:54-57`) -- followed by the
//! `deg_red` round (`:86`).  A zero denominator is `MpcNetError::Generic` (the reference panics on `unwrap()`, `:55`).
//! Bounds as the reference (`F: FftField + PrimeField + Field`, `Net: MpcSerNet`); `degred_mask` as its two fields.
use core::ptr;

use ark_ff::{FftField, Field, PrimeField};
use mpc_net::ser_net::MpcSerNet;
use mpc_net::{MpcNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip_sys as sys;

use crate::{check, DeviceBuf, HipNet};

/// `dist-primitives/src/dpp/mod.rs:15-87`.
pub async fn d_pp<F: FftField + PrimeField + Field, Net: MpcSerNet>(
    num: Vec<F>,
    den: Vec<F>,
    in_mask: &[F],
    out_mask: &[F],
    pp: &PackedSharingParams<F>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<F>, MpcNetError> {
    let hip = HipNet::of(net)?;
    let len = num.len();
    if den.len() != len || in_mask.len() != len || out_mask.len() != len {
        return Err(MpcNetError::BadInput { err: "d_pp: num, den and the DegRedMask must have one length" });
    }
    let ctx = hip.ctx();
    ctx.expect_field::<F>(pp.l)?;
    let k = hip.parties_per_rank();
    let n_d = DeviceBuf::from_slice(ctx, &num)?;
    let d_d = DeviceBuf::from_slice(ctx, &den)?;
    let im = DeviceBuf::from_slice(ctx, in_mask)?;
    let om = DeviceBuf::from_slice(ctx, out_mask)?;
    let out = DeviceBuf::alloc(ctx, n_d.bytes)?;
    check(ctx, unsafe {
        sys::zk_dist_d_pp(ctx.raw(), hip.raw_net(), sid as i32, n_d.ptr(), d_d.ptr(), im.ptr(), om.ptr(), len / k, 0,
                          out.ptr(), ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid as i32) })?;
    out.to_vec(len)
}

/// All `n` parties in one process on one GPU (the reference's `simulate_network_round` tests, `dpp_test.rs:16-91`):
/// rows `[n][len]`; the `deg_red` round is fused into the last kernel (`zk_d_pp`).
pub fn d_pp_all_parties<F: FftField + PrimeField>(
    ctx: &crate::Context, num: &[F], den: &[F], masks: Option<(&[F], &[F])>, len: usize,
) -> Result<Vec<F>, MpcNetError> {
    if num.len() != ctx.n * len || den.len() != ctx.n * len {
        return Err(MpcNetError::BadInput { err: "d_pp: [n][len] rows expected" });
    }
    let n_d = DeviceBuf::from_slice(ctx, num)?;
    let d_d = DeviceBuf::from_slice(ctx, den)?;
    let (im, om) = match masks {
        Some((a, b)) => (Some(DeviceBuf::from_slice(ctx, a)?), Some(DeviceBuf::from_slice(ctx, b)?)),
        None => (None, None),
    };
    let out = DeviceBuf::alloc(ctx, n_d.bytes)?;
    let p = |b: &Option<DeviceBuf>| b.as_ref().map(|b| b.ptr() as *const core::ffi::c_void).unwrap_or(ptr::null());
    check(ctx, unsafe {
        sys::zk_d_pp(ctx.raw(), n_d.ptr(), d_d.ptr(), p(&im), p(&om), len, 0, out.ptr(), ptr::null_mut())
    })?;
    out.to_vec(ctx.n * len)
}
