//! `PackedSharingParams` batch methods on the GPU (`secret-sharing/src/pss.rs:69-221`,
//! `dist-primitives/src/utils/pack.rs:8-35`).  The reference packs one chunk per call and transposes `Vec<Vec<F>>`;
//! here a whole vector is packed / unpacked per launch and the party-major layout `[n][m/l]` needs no transpose.
use core::ptr;

use ark_ff::PrimeField;
use mpc_net::MpcNetError;
use zksaas_hip_sys as sys;

use crate::{check, fr_ptr, Context, DeviceBuf};

/// `pack_vec` + `transpose` (`pack.rs:8-35`): secrets `[m]` -> `n` share vectors of length `m / l`.
pub fn pack_vec<F: PrimeField + 'static>(ctx: &Context, secrets: &[F]) -> Result<Vec<Vec<F>>, MpcNetError> {
    pack_impl(ctx, secrets, 0, false)
}
/// Stride packing (`dfft/mod.rs:286-299`, `qap.rs:103-112`): chunk `j` packs `secrets[j], secrets[j + m/l], ..`.
pub fn pack_stride<F: PrimeField + 'static>(ctx: &Context, secrets: &[F]) -> Result<Vec<Vec<F>>, MpcNetError> {
    pack_impl(ctx, secrets, 1, false)
}
/// `det_pack` per chunk (`pss.rs:69-87`).
pub fn det_pack_vec<F: PrimeField + 'static>(ctx: &Context, secrets: &[F]) -> Result<Vec<Vec<F>>, MpcNetError> {
    pack_impl(ctx, secrets, 0, true)
}

fn pack_impl<F: PrimeField + 'static>(ctx: &Context, secrets: &[F], order: i32, det: bool)
                                      -> Result<Vec<Vec<F>>, MpcNetError> {
    if secrets.len() % ctx.l != 0 {
        return Err(MpcNetError::BadInput { err: "pack: length is not a multiple of l" });
    }
    let nchunks = secrets.len() / ctx.l;
    let sec = DeviceBuf::from_slice(ctx, secrets)?;
    let sh = DeviceBuf::alloc(ctx, ctx.n * nchunks * core::mem::size_of::<F>())?;
    let rc = unsafe {
        if det {
            sys::zk_pss_det_pack(ctx.raw(), sec.ptr(), nchunks, order, sh.ptr(), ptr::null_mut())
        } else {
            sys::zk_pss_pack(ctx.raw(), sec.ptr(), nchunks, order, 0, sh.ptr(), ptr::null_mut())
        }
    };
    check(ctx, rc)?;
    let flat: Vec<F> = sh.to_vec(ctx.n * nchunks)?;
    Ok(flat.chunks(nchunks.max(1)).map(|c| c.to_vec()).collect())
}

/// `unpack` of every chunk (`pss.rs:125-138`): shares `[n][m/l]` -> secrets `[m]` in `pack_vec` order.
pub fn unpack_vec<F: PrimeField + 'static>(ctx: &Context, shares: &[Vec<F>]) -> Result<Vec<F>, MpcNetError> {
    let (flat, nchunks) = flatten(ctx, shares, ctx.n)?;
    let sh = DeviceBuf::from_slice(ctx, &flat)?;
    let out = DeviceBuf::alloc(ctx, nchunks * ctx.l * core::mem::size_of::<F>())?;
    check(ctx, unsafe { sys::zk_pss_unpack(ctx.raw(), sh.ptr(), nchunks, out.ptr(), ptr::null_mut()) })?;
    out.to_vec(nchunks * ctx.l)
}

/// `unpack_missing_shares` of every chunk (`pss.rs:210-221`): `unpack2` when all `n` parties are present, the
/// Lagrange form (`:170-205`) for the listed subset otherwise.
pub fn unpack_missing_shares_vec<F: PrimeField + 'static>(ctx: &Context, shares: &[Vec<F>], parties: &[u32])
                                                          -> Result<Vec<F>, MpcNetError> {
    let (flat, nchunks) = flatten(ctx, shares, parties.len())?;
    let sh = DeviceBuf::from_slice(ctx, &flat)?;
    let out = DeviceBuf::alloc(ctx, nchunks * ctx.l * core::mem::size_of::<F>())?;
    check(ctx, unsafe {
        sys::zk_pss_unpack2(ctx.raw(), sh.ptr(), parties.as_ptr(), parties.len() as i32, nchunks, out.ptr(),
                            ptr::null_mut())
    })?;
    out.to_vec(nchunks * ctx.l)
}

fn flatten<F: PrimeField>(_ctx: &Context, shares: &[Vec<F>], want: usize) -> Result<(Vec<F>, usize), MpcNetError> {
    if shares.len() != want || shares.is_empty() {
        return Err(MpcNetError::BadInput { err: "unpack: one share vector per listed party expected" });
    }
    let nchunks = shares[0].len();
    if shares.iter().any(|v| v.len() != nchunks) {
        return Err(MpcNetError::BadInput { err: "unpack: share vectors differ in length" });
    }
    let mut flat = Vec::with_capacity(want * nchunks);
    for v in shares {
        flat.extend_from_slice(v);
    }
    let _ = fr_ptr(&flat);
    Ok((flat, nchunks))
}
