//! `d_fft` / `d_ifft` behind the reference's signatures (`dist-primitives/src/dfft/mod.rs:99-175`).
//!
//! Called collectively by every rank, like the reference's by every party.  The local stages (`fft1_in_place`,
//! `:178-208`) and the king's closure (`fft2_with_rearrange`, `:240-320`: mask add, gather, unpack -> fft2 -> g^i ->
//! pack, scatter, unmask) run in `zk_dist_d_fft` / `zk_dist_d_ifft`; the exchange is RCCL gather / scatter over xGMI
//! (or the all-to-all king when the context option `king_alltoall` is set).
//!
//! Generic bounds and argument order are the reference's (`F: FftField + PrimeField`, `D: EvaluationDomain<F>`,
//! `Net: MpcSerNet`); `fft_mask: &FftMask<F>` arrives as `in_mask, out_mask` (module doc of `lib.rs`).
use core::ffi::c_void;
use core::ptr;

use ark_ff::{FftField, PrimeField};
use ark_poly::EvaluationDomain;
use mpc_net::ser_net::MpcSerNet;
use mpc_net::{MpcNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip_sys as sys;

use crate::{check, Context, DeviceBuf, HipNet};

fn masks_dev<F: PrimeField>(ctx: &Context, in_mask: &[F], out_mask: &[F], len: usize)
                            -> Result<(Option<DeviceBuf>, Option<DeviceBuf>), MpcNetError> {
    // FftMask::zero(mbyl) (`dfft/mod.rs:89-94`) is a pair of zero vectors: pass NULL instead of uploading zeros
    let up = |v: &[F]| -> Result<Option<DeviceBuf>, MpcNetError> {
        if v.iter().all(|x| x.is_zero()) {
            Ok(None)
        } else if v.len() != len {
            Err(MpcNetError::BadInput { err: "FftMask length differs from the share vector" })
        } else {
            Ok(Some(DeviceBuf::from_slice(ctx, v)?))
        }
    };
    Ok((up(in_mask)?, up(out_mask)?))
}
fn p(b: &Option<DeviceBuf>) -> *const c_void {
    b.as_ref().map(|b| b.ptr() as *const c_void).unwrap_or(ptr::null())
}

/// `dist-primitives/src/dfft/mod.rs:99-134`.  `pcoeff_share`: this rank's `k` parties' vectors, `[k][m/l]` flattened
/// (`k = 1` in the reference's one-process-per-party deployment, where it is exactly the reference's argument).
pub async fn d_fft<
    F: FftField + PrimeField,
    D: EvaluationDomain<F>,
    Net: MpcSerNet,
>(
    pcoeff_share: Vec<F>,
    in_mask: &[F],
    out_mask: &[F],
    rearrange: bool,
    dom: &D,
    pp: &PackedSharingParams<F>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<F>, MpcNetError> {
    let hip = HipNet::of(net)?;
    let k = hip.parties_per_rank();
    if pcoeff_share.len() * pp.l != dom.size() * k {
        // the reference debug_asserts (`:112-118`); here the mismatch is an error in every build
        return Err(MpcNetError::BadInput { err: "Mismatch of size in FFT" });
    }
    let ctx = hip.ctx();
    ctx.expect_field::<F>(pp.l)?;
    let sh = DeviceBuf::from_slice(ctx, &pcoeff_share)?;
    let (im, om) = masks_dev(ctx, in_mask, out_mask, pcoeff_share.len())?;
    check(ctx, unsafe {
        sys::zk_dist_d_fft(ctx.raw(), hip.raw_net(), sid as i32, sh.ptr(), p(&im), p(&om), rearrange as i32,
                           dom.log_size_of_group() as i32, 0, ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid as i32) })?;
    sh.to_vec(pcoeff_share.len())
}

/// `dist-primitives/src/dfft/mod.rs:137-175`: scales by `1/m` (`:159`), runs the stages with `group_gen_inv` and lets
/// the king multiply coefficient `i` by `g^i`.
pub async fn d_ifft<
    F: FftField + PrimeField,
    D: EvaluationDomain<F>,
    Net: MpcSerNet,
>(
    peval_share: Vec<F>,
    in_mask: &[F],
    out_mask: &[F],
    rearrange: bool,
    dom: &D,
    g: F,
    pp: &PackedSharingParams<F>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<F>, MpcNetError> {
    let hip = HipNet::of(net)?;
    let k = hip.parties_per_rank();
    if peval_share.len() * pp.l != dom.size() * k {
        return Err(MpcNetError::BadInput { err: "Mismatch of size in IFFT" });
    }
    let ctx = hip.ctx();
    ctx.expect_field::<F>(pp.l)?;
    let sh = DeviceBuf::from_slice(ctx, &peval_share)?;
    let (im, om) = masks_dev(ctx, in_mask, out_mask, peval_share.len())?;
    let gl = [g];
    check(ctx, unsafe {
        sys::zk_dist_d_ifft(ctx.raw(), hip.raw_net(), sid as i32, sh.ptr(), p(&im), p(&om), rearrange as i32,
                            dom.log_size_of_group() as i32, crate::fr_ptr(&gl), 0, ptr::null_mut())
    })?;
    check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid as i32) })?;
    sh.to_vec(peval_share.len())
}

/// `FftMask::sample` (`dfft/mod.rs:30-85`) by the library's dealer, for all `n` parties: one `(in_mask, out_mask)` per
/// party (`dist-primitives` wraps them: `FftMask::new(in_mask, out_mask)`).
pub fn sample_fft_masks<F: FftField + PrimeField>(
    ctx: &Context, rearrange: bool, g: F, inverse: bool, log2_m: u32,
) -> Result<Vec<(Vec<F>, Vec<F>)>, MpcNetError> {
    let len = (1usize << log2_m) / ctx.l;
    let bytes = ctx.n * len * core::mem::size_of::<F>();
    let (im, om) = (DeviceBuf::alloc(ctx, bytes)?, DeviceBuf::alloc(ctx, bytes)?);
    let gl = [g];
    check(ctx, unsafe {
        sys::zk_fft_mask_sample(ctx.raw(), rearrange as i32, crate::fr_ptr(&gl), inverse as i32, log2_m as i32, 0,
                                im.ptr(), om.ptr(), ptr::null_mut())
    })?;
    let (a, b): (Vec<F>, Vec<F>) = (im.to_vec(ctx.n * len)?, om.to_vec(ctx.n * len)?);
    Ok(a.chunks(len).zip(b.chunks(len)).map(|(x, y)| (x.to_vec(), y.to_vec())).collect())
}
