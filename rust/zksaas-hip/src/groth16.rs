//! `circom_h` (`groth16/src/ext_wit.rs:104-181`) and the distributed prover `dsha256`
//! (`groth16/examples/sha256.rs:32-129`) with the reference's argument lists.  `groth16/` is not a dependency of this
//! crate (it depends on `dist-primitives`, not the other way round), so the two functions take the FIELDS of
//! `PackedQAPShare` / `PackedProvingKeyShare` (`qap.rs:29-40`, `proving_key.rs:18-45`) through the small structs below;
//! the call sites in `groth16/` pass `&qap_share.a`, `&crs_share.s` … unchanged otherwise.
use core::ffi::c_void;
use core::ptr;

use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::{FftField, PrimeField};
use dist_primitives::dfft::FftMask;
use dist_primitives::dmsm::MsmMask;
use dist_primitives::utils::deg_red::DegRedMask;
use mpc_net::MpcNetError;
use zksaas_hip_sys as sys;

use crate::net::HipBacked;
use crate::{check, pack_affine, pack_jacobian, unpack_jacobian, DeviceBuf};

/// The vectors of `PackedQAPShare<F, D>` (`qap.rs:29-40`) of this rank's party.
pub struct QapShare<'a, F> {
    pub a: &'a [F],
    pub b: &'a [F],
    pub c: &'a [F],
    pub log2_m: u32,
}

/// Device-resident masks of one proof in the layout of `zk_groth16_masks` (six `FftMask`, the `DegRedMask`; MSM masks
/// stay on the host).
struct MasksDev {
    keep: Vec<DeviceBuf>,
    host: Vec<Vec<u64>>,
    ct: sys::ZkGroth16Masks,
}

fn fr_masks<F: PrimeField>(net: &impl HipBacked, fft_mask: &[FftMask<F>; 6], degred: &DegRedMask<F, F>)
                           -> Result<MasksDev, MpcNetError> {
    let ctx = net.ctx();
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
    let a = DeviceBuf::from_slice(ctx, &degred.in_mask)?;
    let b = DeviceBuf::from_slice(ctx, &degred.out_mask)?;
    m.ct.degred_in = a.ptr();
    m.ct.degred_out = b.ptr();
    m.keep.push(a);
    m.keep.push(b);
    Ok(m)
}

/// `groth16/src/ext_wit.rs:104-181`: three `d_ifft` with the coset shift `w_2m` and rearranged output joined on
/// channels 0..2, three `d_fft` likewise, `a b - c` share-wise, `deg_red` -- one `zk_dist_circom_h`.
pub async fn circom_h<F: FftField + PrimeField + 'static, Net: HipBacked>(
    qap_share: QapShare<'_, F>,
    fft_mask: &[FftMask<F>; 6],
    degred_mask: &DegRedMask<F, F>,
    net: &Net,
) -> Result<Vec<F>, MpcNetError> {
    let ctx = net.ctx();
    let len = qap_share.a.len();
    let (a, b, c) = (DeviceBuf::from_slice(ctx, qap_share.a)?, DeviceBuf::from_slice(ctx, qap_share.b)?,
                     DeviceBuf::from_slice(ctx, qap_share.c)?);
    let masks = fr_masks(net, fft_mask, degred_mask)?;
    let h = DeviceBuf::alloc(ctx, a.bytes)?;
    check(ctx, unsafe {
        sys::zk_dist_circom_h(ctx.raw(), net.raw_net(), a.ptr(), b.ptr(), c.ptr(), qap_share.log2_m as i32, &masks.ct, 0,
                              h.ptr(), ptr::null_mut())
    })?;
    for sid in 0..3 {
        check(ctx, unsafe { sys::zk_net_sync(net.raw_net(), sid) })?;
    }
    h.to_vec(len)
}

/// The query vectors and constants of `PackedProvingKeyShare<E>` (`proving_key.rs:18-45`) of this rank's party.
pub struct CrsShare<'a, E: Pairing> {
    pub s: &'a [E::G1Affine],
    pub u: &'a [E::G1Affine],
    pub w: &'a [E::G1Affine],
    pub h: &'a [E::G1Affine],
    pub v: &'a [E::G2Affine],
    pub a_query0: E::G1Affine,
    pub b_g1_query0: E::G1Affine,
    pub b_g2_query0: E::G2Affine,
    pub delta_g1: E::G1Affine,
    pub delta_g2: E::G2Affine,
    pub alpha_g1: E::G1Affine,
    pub beta_g1: E::G1Affine,
    pub beta_g2: E::G2Affine,
}

/// `groth16/examples/sha256.rs:32-129` (`dsha256`): `circom_h`, then `A`, `B` in G1 and G2, `C` (`prove.rs:11-238`) --
/// one `zk_dist_groth16_prove`; returns this party's `(pi_a, pi_b, pi_c)` shares.
#[allow(clippy::too_many_arguments)]
pub async fn dsha256<E, C1, C2, Net>(
    crs_share: &CrsShare<'_, E>,
    qap_share: QapShare<'_, E::ScalarField>,
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
    C1: SWCurveConfig<ScalarField = E::ScalarField>,
    C2: SWCurveConfig<ScalarField = E::ScalarField>,
    E::ScalarField: FftField + PrimeField + 'static,
    Net: HipBacked,
{
    let ctx = net.ctx();
    let up1 = |v: &[Affine<C1>]| DeviceBuf::from_slice(ctx, &pack_affine(v));
    let (s, u, w, hq) = (up1(crs_share.s)?, up1(crs_share.u)?, up1(crs_share.w)?, up1(crs_share.h)?);
    let v = DeviceBuf::from_slice(ctx, &pack_affine(crs_share.v))?;
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
    let (qa, qb, qc) = (DeviceBuf::from_slice(ctx, qap_share.a)?, DeviceBuf::from_slice(ctx, qap_share.b)?,
                        DeviceBuf::from_slice(ctx, qap_share.c)?);
    let (asd, axd) = (DeviceBuf::from_slice(ctx, a_share)?, DeviceBuf::from_slice(ctx, ax_share)?);
    let mut masks = fr_masks(net, fft_mask, f_degred_mask)?;
    // MSM masks in the order of zk_groth16_masks: A, B-in-G1, B-in-G2, C.w, C.u (sha256.rs:226-291)
    let g1 = |m: &MsmMask<Projective<C1>>| (pack_jacobian(&[m.in_mask]), pack_jacobian(&[m.out_mask]));
    let order = [Some(0usize), Some(1), None, Some(2), Some(3)];
    for (slot, which) in order.iter().enumerate() {
        let (i, o) = match which {
            Some(k) => g1(&g1_msm_mask[*k]),
            None => (pack_jacobian(&[g2_msm_mask.in_mask]), pack_jacobian(&[g2_msm_mask.out_mask])),
        };
        masks.host.push(i);
        masks.host.push(o);
        masks.ct.msm_in[slot] = masks.host[2 * slot].as_ptr() as *const c_void;
        masks.ct.msm_out[slot] = masks.host[2 * slot + 1].as_ptr() as *const c_void;
    }
    let (rr, ss) = ([r_share], [s_share]);
    let n1 = 3 * consts1[0].len() / 2;
    let n2 = 3 * consts2[0].len() / 2;
    let (mut pa, mut pb, mut pc) = (vec![0u64; n1], vec![0u64; n2], vec![0u64; n1]);
    check(ctx, unsafe {
        sys::zk_dist_groth16_prove(ctx.raw(), net.raw_net(), &crs, qa.ptr(), qb.ptr(), qc.ptr(), asd.ptr(), axd.ptr(),
                                   crate::fr_ptr(&rr), crate::fr_ptr(&ss), qap_share.log2_m as i32, &masks.ct, 0,
                                   pa.as_mut_ptr() as *mut c_void, pb.as_mut_ptr() as *mut c_void,
                                   pc.as_mut_ptr() as *mut c_void, ptr::null_mut())
    })?;
    Ok((unpack_jacobian::<C1>(&pa, 1)[0], unpack_jacobian::<C2>(&pb, 1)[0], unpack_jacobian::<C1>(&pc, 1)[0]))
}
