//! `d_msm` behind the reference's signature (`dist-primitives/src/dmsm/mod.rs:59-102`): the local `G::msm` (`:73`) is
//! the Pippenger bucket MSM of `csrc/msm.hpp`; the king's `unpack2` + sum (`:85-86`) is one linear form over the
//! parties' results, gathered and broadcast through `zk_dist_d_msm`.
//!
//! `G: CurveGroup, Net: MpcSerNet` as in the reference; `msm_mask: &MsmMask<G>` arrives as `in_mask, out_mask`.
use core::ffi::c_void;
use core::ptr;

use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ec::CurveGroup;
use mpc_net::ser_net::MpcSerNet;
use mpc_net::{MpcNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip_sys as sys;

use crate::{check, group_of, pack_affine, pack_jacobian, same_slice, same_value, unpack_jacobian, DeviceBuf, HipNet};

/// `dist-primitives/src/dmsm/mod.rs:59-102`.  A length mismatch is the `Generic(min_len.to_string())` the reference
/// gets from `G::msm` (`:73`).  One party per rank (`d_msm_rows` carries `k` parties).
pub async fn d_msm<G: CurveGroup, Net: MpcSerNet>(
    bases: &[G::Affine],
    scalars: &[G::ScalarField],
    in_mask: &G,
    out_mask: &G,
    pp: &PackedSharingParams<G::ScalarField>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<G, MpcNetError> {
    let hip = HipNet::of(net)?;
    hip.ctx().expect_field::<G::ScalarField>(pp.l)?;
    if hip.parties_per_rank() != 1 {
        return Err(MpcNetError::BadInput { err: "d_msm: this signature carries one party's vectors (see d_msm_rows)" });
    }
    crate::sw_dispatch!(G, C => {
        // inside this branch G == Projective<C>: the casts below are identities
        let b = same_slice::<G::Affine, Affine<C>>(bases).expect("G::Affine is Affine<C>");
        let s = same_slice::<G::ScalarField, <C as ark_ec::CurveConfig>::ScalarField>(scalars).expect("same scalar field");
        let im = same_value::<G, Projective<C>>(*in_mask).expect("G is Projective<C>");
        let om = same_value::<G, Projective<C>>(*out_mask).expect("G is Projective<C>");
        let out = d_msm_rows::<C>(b, s, b.len().min(s.len()), &[im], &[om], hip, sid, b.len() != s.len())?;
        Ok(same_value::<Projective<C>, G>(out[0]).expect("G is Projective<C>"))
    })
}

/// The same for a rank that drives `k` parties: rows `[k][len]`, `k` masks, `k` results.
#[allow(clippy::too_many_arguments)]
pub fn d_msm_rows<C: SWCurveConfig>(
    bases: &[Affine<C>],
    scalars: &[C::ScalarField],
    len: usize,
    in_masks: &[Projective<C>],
    out_masks: &[Projective<C>],
    hip: &HipNet,
    sid: MultiplexedStreamID,
    ragged: bool,
) -> Result<Vec<Projective<C>>, MpcNetError> {
    if ragged {
        return Err(MpcNetError::Generic(len.to_string()));       // `G::msm(bases, scalars)?` (`dmsm/mod.rs:73`)
    }
    let k = hip.parties_per_rank();
    if bases.len() != k * len || scalars.len() != k * len || in_masks.len() != k || out_masks.len() != k {
        return Err(MpcNetError::BadInput { err: "d_msm_rows: [k][len] rows and k masks expected" });
    }
    let ctx = hip.ctx();
    let b = DeviceBuf::from_slice(ctx, &pack_affine(bases))?;
    let s = DeviceBuf::from_slice(ctx, scalars)?;
    let im = pack_jacobian(in_masks);
    let om = pack_jacobian(out_masks);
    let mut out = vec![0u64; im.len()];
    check(ctx, unsafe {
        sys::zk_dist_d_msm(ctx.raw(), hip.raw_net(), sid as i32, group_of::<C>(), b.ptr(), s.ptr(), len,
                           im.as_ptr() as *const c_void, om.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void,
                           ptr::null_mut())
    })?;
    Ok(unpack_jacobian::<C>(&out, k))
}

/// `MsmMask::sample` (`dmsm/mod.rs:21-47`) by the library's dealer: one `(in_mask, out_mask)` per party.
pub fn sample_msm_masks<C: SWCurveConfig>(ctx: &crate::Context, gen: Affine<C>)
                                          -> Result<Vec<(Projective<C>, Projective<C>)>, MpcNetError> {
    let g = pack_affine(&[gen]);
    let per = 3 * g.len() / 2;
    let (mut im, mut om) = (vec![0u64; ctx.n * per], vec![0u64; ctx.n * per]);
    check(ctx, unsafe {
        sys::zk_msm_mask_sample(ctx.raw(), group_of::<C>(), g.as_ptr() as *const c_void, 0,
                                im.as_mut_ptr() as *mut c_void, om.as_mut_ptr() as *mut c_void)
    })?;
    let (a, b) = (unpack_jacobian::<C>(&im, ctx.n), unpack_jacobian::<C>(&om, ctx.n));
    Ok(a.into_iter().zip(b).collect())
}
