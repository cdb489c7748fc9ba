//! `circom_h`, `libsnark_h` (`groth16/src/ext_wit.rs:14-181`) and the distributed prover `dsha256`
//! (`groth16/examples/sha256.rs:32-129`) as ONE device composition each, with the reference's signatures token for
//! token (`PackedQAPShare<F, D>`, `&[FftMask<F>; 6]`, `&DegRedMask<F, F>`, `pp`, `Net: MpcSerNet`;
//! `PackedProvingKeyShare<E>`, `Net: MpcNet`): this crate depends on `groth16` and `dist-primitives`, so it can name
//! their types (`zksaas-hip` cannot: `dist-primitives` depends on it).
//!
//! OPTIONAL.  `groth16/` needs no line changed to run on the GPU: its `circom_h` calls `d_ifft` / `d_fft` / `deg_red` of
//! `dist-primitives`, which hand their rounds to the device when the net is a `HipNet`
//! (`rust/patches/dist-primitives.diff`).  What this crate adds is the fused form -- `zk_dist_circom_h` keeps the three
//! channels in flight inside the library, forms `a b - c` at `deg_red`'s load and never returns the intermediate vectors
//! to the host; `zk_dist_groth16_prove` also overlaps the five MSMs with `circom_h` -- for an application that changes
//! its `use groth16::ext_wit` line to `use zksaas_hip_groth16 as ext_wit`.
//!
//! NOT compiled in the build image (no Rust toolchain): `tests/test_rust_ffi.py` checks the `sys::zk_*` calls against
//! the header and these signatures against the reference's.
use core::ffi::c_void;
use core::ptr;

use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::{FftField, PrimeField};
use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
use dist_primitives::dfft::FftMask;
use dist_primitives::dmsm::MsmMask;
use dist_primitives::utils::deg_red::DegRedMask;
use groth16::proving_key::PackedProvingKeyShare;
use groth16::qap::{self, PackedQAPShare};
use mpc_net::ser_net::MpcSerNet;
use mpc_net::{MpcNet, MpcNetError};
use secret_sharing::pss::PackedSharingParams;
use zksaas_hip::{check, pack_affine, pack_jacobian, unpack_jacobian, DeviceBuf, HipNet};
use zksaas_hip_sys as sys;

/// Device-resident masks of one proof in the layout of `zk_groth16_masks` (six `FftMask`, the `DegRedMask`; MSM masks
/// stay on the host).
struct MasksDev {
    keep: Vec<DeviceBuf>,
    host: Vec<Vec<u64>>,
    ct: sys::ZkGroth16Masks,
}

fn fr_masks<F: FftField + PrimeField>(hip: &HipNet, fft_mask: &[FftMask<F>], degred: Option<&DegRedMask<F, F>>)
                                      -> Result<MasksDev, MpcNetError> {
    let ctx = hip.ctx();
    let mut m = MasksDev {
        keep: Vec::new(),
        host: Vec::new(),
        ct: sys::ZkGroth16Masks {
            fft_in: [ptr::null(); 6],
            fft_out: [ptr::null(); 6],
            degred_in: ptr::null(),
            degred_out: ptr::null(),
            msm_in: [ptr::null(); 5],
            msm_out: [ptr::null(); 5],
        },
    };
    for i in 0..6 {
        let a = DeviceBuf::from_slice(ctx, &fft_mask[i].in_mask)?;
        let b = DeviceBuf::from_slice(ctx, &fft_mask[i].out_mask)?;
        m.ct.fft_in[i] = a.ptr();
        m.ct.fft_out[i] = b.ptr();
        m.keep.push(a);
        m.keep.push(b);
    }
    if let Some(degred) = degred {
        let a = DeviceBuf::from_slice(ctx, &degred.in_mask)?;
        let b = DeviceBuf::from_slice(ctx, &degred.out_mask)?;
        m.ct.degred_in = a.ptr();
        m.ct.degred_out = b.ptr();
        m.keep.push(a);
        m.keep.push(b);
    }
    Ok(m)
}

/// `groth16/src/ext_wit.rs:104-181`: three `d_ifft` with the coset shift `w_2m` and rearranged output joined on
/// channels 0..2, three `d_fft` likewise, `a b - c` share-wise, `deg_red` -- one `zk_dist_circom_h`.
pub async fn circom_h<
    F: FftField + PrimeField,
    D: EvaluationDomain<F>,
    Net: MpcSerNet,
>(
    qap_share: PackedQAPShare<F, D>,
    fft_mask: &[FftMask<F>; 6], // 3 ifft and 3 fft
    degred_mask: &DegRedMask<F, F>,
    pp: &PackedSharingParams<F>,
    net: &Net,
) -> Result<Vec<F>, MpcNetError> {
    let hip = HipNet::of(net)?;
    let ctx = hip.ctx();
    ctx.expect_field::<F>(pp.l)?;
    let len = qap_share.a.len();
    let (a, b, c) = (DeviceBuf::from_slice(ctx, &qap_share.a)?, DeviceBuf::from_slice(ctx, &qap_share.b)?,
                     DeviceBuf::from_slice(ctx, &qap_share.c)?);
    let masks = fr_masks(hip, fft_mask, Some(degred_mask))?;
    let h = DeviceBuf::alloc(ctx, a.bytes)?;
    check(ctx, unsafe {
        sys::zk_dist_circom_h(ctx.raw(), hip.raw_net(), a.ptr(), b.ptr(), c.ptr(), qap_share.domain.log_size_of_group() as i32,
                              &masks.ct, 0, h.ptr(), ptr::null_mut())
    })?;
    for sid in 0..3 {
        check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid) })?;
    }
    h.to_vec(len)
}

/// `groth16/src/ext_wit.rs:14-102`: three `d_ifft` with the coset shift `F::GENERATOR`, three `d_fft` on the coset,
/// `(a b - c) / Z(g)`, one coset `d_ifft` -- one `zk_dist_libsnark_h`.
pub async fn libsnark_h<
    F: FftField + PrimeField,
    D: EvaluationDomain<F>,
    Net: MpcSerNet,
>(
    qap_share: PackedQAPShare<F, D>,
    fft_mask: &[FftMask<F>; 7], // 3 ifft, 3 fft and 1 coset ifft
    pp: &PackedSharingParams<F>,
    net: &Net,
) -> Result<Vec<F>, MpcNetError> {
    let hip = HipNet::of(net)?;
    let ctx = hip.ctx();
    ctx.expect_field::<F>(pp.l)?;
    let len = qap_share.a.len();
    let (a, b, c) = (DeviceBuf::from_slice(ctx, &qap_share.a)?, DeviceBuf::from_slice(ctx, &qap_share.b)?,
                     DeviceBuf::from_slice(ctx, &qap_share.c)?);
    let mut keep = Vec::new();
    let (mut ins, mut outs) = ([ptr::null::<c_void>(); 7], [ptr::null::<c_void>(); 7]);
    for i in 0..7 {
        let x = DeviceBuf::from_slice(ctx, &fft_mask[i].in_mask)?;
        let y = DeviceBuf::from_slice(ctx, &fft_mask[i].out_mask)?;
        ins[i] = x.ptr();
        outs[i] = y.ptr();
        keep.push(x);
        keep.push(y);
    }
    let h = DeviceBuf::alloc(ctx, a.bytes)?;
    check(ctx, unsafe {
        sys::zk_dist_libsnark_h(ctx.raw(), hip.raw_net(), a.ptr(), b.ptr(), c.ptr(), qap_share.domain.log_size_of_group() as i32,
                                ins.as_ptr(), outs.as_ptr(), 0, h.ptr(), ptr::null_mut())
    })?;
    for sid in 0..3 {
        check(ctx, unsafe { sys::zk_net_sync(hip.raw_net(), sid) })?;
    }
    h.to_vec(len)
}

/// `groth16/examples/sha256.rs:32-129` (`dsha256`): `circom_h`, then `A`, `B` in G1 and G2, `C` (`prove.rs:11-238`) --
/// one `zk_dist_groth16_prove`; returns this party's `(pi_a, pi_b, pi_c)` shares.  The reference's function is private
/// to its example, unwraps every `Result` and returns the bare tuple; this one has its argument list and returns the
/// error instead of panicking.  The `where` clause adds what reaching coordinates needs: the pairing's groups are
/// short-Weierstrass (true of every `Pairing` in arkworks 0.4).
#[allow(clippy::too_many_arguments)]
pub async fn dsha256<E, Net, C1, C2>(
    pp: &PackedSharingParams<E::ScalarField>,
    crs_share: &PackedProvingKeyShare<E>,
    qap_share: qap::PackedQAPShare<
        E::ScalarField,
        Radix2EvaluationDomain<E::ScalarField>,
    >,
    a_share: &[E::ScalarField],
    ax_share: &[E::ScalarField],
    r_share: E::ScalarField,
    s_share: E::ScalarField,
    fft_mask: &[FftMask<E::ScalarField>; 6],
    f_degred_mask: &DegRedMask<E::ScalarField, E::ScalarField>,
    g1_msm_mask: &[MsmMask<E::G1>; 4],
    g2_msm_mask: &MsmMask<E::G2>,
    net: &Net,
) -> Result<(E::G1, E::G2, E::G1), MpcNetError>
where
    E: Pairing<G1Affine = Affine<C1>, G2Affine = Affine<C2>, G1 = Projective<C1>, G2 = Projective<C2>>,
    Net: MpcNet,
    C1: SWCurveConfig<ScalarField = E::ScalarField>,
    C2: SWCurveConfig<ScalarField = E::ScalarField>,
    E::ScalarField: FftField + PrimeField,
{
    let hip = HipNet::of(net)?;
    let ctx = hip.ctx();
    ctx.expect_field::<E::ScalarField>(pp.l)?;
    let up1 = |v: &[Affine<C1>]| DeviceBuf::from_slice(ctx, &pack_affine(v));
    let (s, u, w, hq) = (up1(&crs_share.s)?, up1(&crs_share.u)?, up1(&crs_share.w)?, up1(&crs_share.h)?);
    let v = DeviceBuf::from_slice(ctx, &pack_affine(&crs_share.v))?;
    let consts1: Vec<Vec<u64>> = [crs_share.a_query0, crs_share.b_g1_query0, crs_share.delta_g1, crs_share.alpha_g1,
                                  crs_share.beta_g1].iter().map(|p| pack_affine(&[*p])).collect();
    let consts2: Vec<Vec<u64>> = [crs_share.b_g2_query0, crs_share.delta_g2, crs_share.beta_g2].iter()
        .map(|p| pack_affine(&[*p])).collect();
    let cp = |v: &Vec<u64>| v.as_ptr() as *const c_void;
    let crs = sys::ZkCrsShare {
        s_d: s.ptr(),
        h_d: hq.ptr(),
        v_d: v.ptr(),
        w_d: w.ptr(),
        u_d: u.ptr(),
        len_a: crs_share.s.len(),
        len_w: crs_share.w.len(),
        len_u: crs_share.u.len(),
        a_query0: cp(&consts1[0]),
        b_g1_query0: cp(&consts1[1]),
        delta_g1: cp(&consts1[2]),
        alpha_g1: cp(&consts1[3]),
        beta_g1: cp(&consts1[4]),
        b_g2_query0: cp(&consts2[0]),
        delta_g2: cp(&consts2[1]),
        beta_g2: cp(&consts2[2]),
    };
    let (qa, qb, qc) = (DeviceBuf::from_slice(ctx, &qap_share.a)?, DeviceBuf::from_slice(ctx, &qap_share.b)?,
                        DeviceBuf::from_slice(ctx, &qap_share.c)?);
    let (asd, axd) = (DeviceBuf::from_slice(ctx, a_share)?, DeviceBuf::from_slice(ctx, ax_share)?);
    let mut masks = fr_masks(hip, fft_mask, Some(f_degred_mask))?;
    // MSM masks in the order of zk_groth16_masks: A, B-in-G1, B-in-G2, C.w, C.u (sha256.rs:226-291)
    let g1 = |m: &MsmMask<Projective<C1>>| (pack_jacobian(&[m.in_mask]), pack_jacobian(&[m.out_mask]));
    let order = [Some(0usize), Some(1), None, Some(2), Some(3)];
    for which in order.iter() {
        let (i, o) = match which {
            Some(k) => g1(&g1_msm_mask[*k]),
            None => (pack_jacobian(&[g2_msm_mask.in_mask]), pack_jacobian(&[g2_msm_mask.out_mask])),
        };
        masks.host.push(i);
        masks.host.push(o);
    }
    for slot in 0..5 {
        masks.ct.msm_in[slot] = masks.host[2 * slot].as_ptr() as *const c_void;
        masks.ct.msm_out[slot] = masks.host[2 * slot + 1].as_ptr() as *const c_void;
    }
    let (rr, ss) = ([r_share], [s_share]);
    let n1 = 3 * consts1[0].len() / 2;
    let n2 = 3 * consts2[0].len() / 2;
    let (mut pa, mut pb, mut pc) = (vec![0u64; n1], vec![0u64; n2], vec![0u64; n1]);
    check(ctx, unsafe {
        sys::zk_dist_groth16_prove(ctx.raw(), hip.raw_net(), &crs, qa.ptr(), qb.ptr(), qc.ptr(), asd.ptr(), axd.ptr(),
                                   rr.as_ptr() as *const c_void, ss.as_ptr() as *const c_void,
                                   qap_share.domain.log_size_of_group() as i32, &masks.ct, 0,
                                   pa.as_mut_ptr() as *mut c_void, pb.as_mut_ptr() as *mut c_void,
                                   pc.as_mut_ptr() as *mut c_void, ptr::null_mut())
    })?;
    Ok((unpack_jacobian::<C1>(&pa, 1)[0], unpack_jacobian::<C2>(&pb, 1)[0], unpack_jacobian::<C1>(&pc, 1)[0]))
}
