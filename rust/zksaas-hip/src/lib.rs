//! MI355X back end for the zkSaaS hot path, behind the reference's own generic signatures.
//!
//! Every public `async fn` here has the name, the generic parameters WITH THEIR BOUNDS (`Net: MpcSerNet`, `G: CurveGroup`,
//! `D: EvaluationDomain<F>`), the argument order and the error type of the reference function it replaces (file:line in
//! its doc line).  The one systematic difference: the mask structs (`FftMask`, `MsmMask`, `DegRedMask`) are defined in
//! `dist-primitives`, which depends on THIS crate (never the other way round: cargo rejects cycles), so a mask argument
//! arrives as its two fields (`in_mask`, `out_mask`).  `dist-primitives` keeps its functions and calls down
//! (`rust/patches/dist-primitives.diff`, at most ten lines per function); `groth16/` compiles unchanged with
//! `Net = HipNet`.  `tests/test_rust_ffi.py` compares every signature here with the reference's token for token.
//! The arithmetic runs in `libzksaas_hip.so` (hand-written HIP kernels for gfx950) through `zksaas_hip_sys`; nothing
//! here computes on the CPU and nothing falls back to arkworks when the library reports an error.
//!
//! Field / curve selection is by `TypeId` of the arkworks scalar field (`Curve::of::<F>()`): BN254, BLS12-381 and
//! BLS12-377 are the three the library is built for.  Scalars cross the boundary as their in-memory Montgomery limbs
//! (`Fp<MontBackend<_, N>, N>` is `[u64; N]` little endian, what `include/zksaas.h` specifies); points are repacked to
//! `x || y` because `ark_ec::short_weierstrass::Affine` is not `repr(C)`.
//!
//! Group values are dispatched the same way: `G: CurveGroup` is matched by `TypeId` against the six short-Weierstrass
//! groups the library is built for (`sw_dispatch!`), inside which `G` IS `Projective<C>` and `G::Affine` IS `Affine<C>`.
//!
//! NOT compiled in the build image (no Rust toolchain there): `tests/test_rust_ffi.py` checks every `sys::zk_*` call of
//! this crate against the header (exported name, argument count), the signatures against the reference, and the
//! dependency graph of the three manifests plus the patches.
pub mod deg_red;
pub mod dfft;
pub mod dmsm;
pub mod dpp;
pub mod error;
pub mod net;
pub mod pss;

use core::any::TypeId;
use core::ffi::c_void;
use core::ptr;
use std::sync::Arc;

use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::{Field, PrimeField, Zero};
use mpc_net::MpcNetError;
use zksaas_hip_sys as sys;

pub use error::check;
pub use net::{HipNet, Transport};

/// `Some(v)` seen as a slice of `B` iff `A` and `B` are the same type (`TypeId`): the monomorphised dispatch from a
/// generic parameter to the concrete type a branch is written for.  Not a transmute between different types.
pub(crate) fn same_slice<A: 'static, B: 'static>(v: &[A]) -> Option<&[B]> {
    if TypeId::of::<A>() == TypeId::of::<B>() {
        Some(unsafe { &*(v as *const [A] as *const [B]) })
    } else {
        None
    }
}
pub(crate) fn same_vec<A: 'static, B: 'static>(v: Vec<A>) -> Result<Vec<B>, Vec<A>> {
    if TypeId::of::<A>() == TypeId::of::<B>() {
        let mut v = core::mem::ManuallyDrop::new(v);
        Ok(unsafe { Vec::from_raw_parts(v.as_mut_ptr() as *mut B, v.len(), v.capacity()) })
    } else {
        Err(v)
    }
}
pub(crate) fn same_value<A: 'static + Copy, B: 'static + Copy>(v: A) -> Option<B> {
    same_slice::<A, B>(core::slice::from_ref(&v)).map(|s| s[0])
}

/// Runs `$body` with `$C` bound to the `SWCurveConfig` whose `Projective<$C>` IS the generic group `$G` (`TypeId`), for the
/// six groups of the three curves; `BadInput` for any other group.
#[macro_export]
macro_rules! sw_dispatch {
    ($G:ty, $C:ident => $body:expr) => {{
        use ark_ec::short_weierstrass::Projective;
        let t = core::any::TypeId::of::<$G>();
        if t == core::any::TypeId::of::<Projective<ark_bn254::g1::Config>>() {
            type $C = ark_bn254::g1::Config;
            $body
        } else if t == core::any::TypeId::of::<Projective<ark_bn254::g2::Config>>() {
            type $C = ark_bn254::g2::Config;
            $body
        } else if t == core::any::TypeId::of::<Projective<ark_bls12_381::g1::Config>>() {
            type $C = ark_bls12_381::g1::Config;
            $body
        } else if t == core::any::TypeId::of::<Projective<ark_bls12_381::g2::Config>>() {
            type $C = ark_bls12_381::g2::Config;
            $body
        } else if t == core::any::TypeId::of::<Projective<ark_bls12_377::g1::Config>>() {
            type $C = ark_bls12_377::g1::Config;
            $body
        } else if t == core::any::TypeId::of::<Projective<ark_bls12_377::g2::Config>>() {
            type $C = ark_bls12_377::g2::Config;
            $body
        } else {
            Err(mpc_net::MpcNetError::BadInput { err: "zksaas-hip: group is not G1 / G2 of BN254, BLS12-381 or BLS12-377" })
        }
    }};
}

/// The three curves `libzksaas_hip.so` is instantiated for (`enum zk_curve`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Curve {
    Bn254,
    Bls12_381,
    Bls12_377,
}

impl Curve {
    /// The curve whose scalar field is `F`, by `TypeId` (monomorphised away).
    pub fn of<F: 'static>() -> Result<Curve, MpcNetError> {
        let t = TypeId::of::<F>();
        if t == TypeId::of::<ark_bn254::Fr>() {
            Ok(Curve::Bn254)
        } else if t == TypeId::of::<ark_bls12_381::Fr>() {
            Ok(Curve::Bls12_381)
        } else if t == TypeId::of::<ark_bls12_377::Fr>() {
            Ok(Curve::Bls12_377)
        } else {
            Err(MpcNetError::BadInput { err: "zksaas-hip: scalar field is not BN254 / BLS12-381 / BLS12-377 Fr" })
        }
    }
    pub fn id(self) -> i32 {
        match self {
            Curve::Bn254 => sys::ZK_BN254,
            Curve::Bls12_381 => sys::ZK_BLS12_381,
            Curve::Bls12_377 => sys::ZK_BLS12_377,
        }
    }
}

/// `zk_group` of a short-Weierstrass config: G2 iff the base field is a quadratic extension.
pub fn group_of<C: SWCurveConfig>() -> i32 {
    if <C::BaseField as Field>::extension_degree() == 2 {
        sys::ZK_G2
    } else {
        sys::ZK_G1
    }
}

struct CtxInner {
    raw: *mut sys::ZkCtx,
}
unsafe impl Send for CtxInner {}
unsafe impl Sync for CtxInner {}
impl Drop for CtxInner {
    fn drop(&mut self) {
        unsafe { sys::zk_ctx_destroy(self.raw) }
    }
}

/// `PackedSharingParams::new(l)` on one GPU (`secret-sharing/src/pss.rs:39-66` -> `zk_ctx_create`): the domains, the
/// pack / unpack matrices, twiddle tables and stream set of one device.  Cheap to clone.
#[derive(Clone)]
pub struct Context {
    inner: Arc<CtxInner>,
    pub curve: Curve,
    pub l: usize,
    pub n: usize,
}

impl Context {
    pub fn new<F: 'static>(l: usize, device: i32) -> Result<Self, MpcNetError> {
        let curve = Curve::of::<F>()?;
        let mut raw: *mut sys::ZkCtx = ptr::null_mut();
        let rc = unsafe { sys::zk_ctx_create(curve.id(), l as i32, device, &mut raw) };
        if raw.is_null() {
            return Err(MpcNetError::Generic(format!("zk_ctx_create failed ({rc})")));
        }
        let ctx = Context { inner: Arc::new(CtxInner { raw }), curve, l, n: 4 * l };
        check(&ctx, rc)?;
        Ok(ctx)
    }
    pub fn raw(&self) -> *mut sys::ZkCtx {
        self.inner.raw
    }
    /// The context was created for `F`'s curve and this packing factor (`pp.l`): a caller mixing fields gets
    /// `BadInput`, not a computation over the wrong modulus.
    pub fn expect_field<F: 'static>(&self, l: usize) -> Result<(), MpcNetError> {
        if Curve::of::<F>()? != self.curve || l != self.l {
            return Err(MpcNetError::BadInput { err: "zksaas-hip: the net's context was created for another field or packing factor" });
        }
        Ok(())
    }
    /// `zk_ctx_set_option` (e.g. `"rng_replay"`, `"king_alltoall"`, `"h_first_log_m"`).
    pub fn set_option(&self, name: &str, value: i64) -> Result<(), MpcNetError> {
        let c = std::ffi::CString::new(name).map_err(|_| MpcNetError::BadInput { err: "option name" })?;
        check(self, unsafe { sys::zk_ctx_set_option(self.raw(), c.as_ptr(), value) })
    }
    pub fn sync(&self) -> Result<(), MpcNetError> {
        check(self, unsafe { sys::zk_stream_sync(self.raw(), ptr::null_mut()) })
    }
}

/// Device allocation owned by a context (`zk_malloc` / `zk_free`).
pub struct DeviceBuf {
    ctx: Context,
    ptr: *mut c_void,
    pub bytes: usize,
}
unsafe impl Send for DeviceBuf {}

impl DeviceBuf {
    pub fn alloc(ctx: &Context, bytes: usize) -> Result<Self, MpcNetError> {
        let mut p: *mut c_void = ptr::null_mut();
        check(ctx, unsafe { sys::zk_malloc(ctx.raw(), bytes.max(1), &mut p) })?;
        Ok(DeviceBuf { ctx: ctx.clone(), ptr: p, bytes })
    }
    /// Upload a slice of plain-old-data elements (field elements: their Montgomery limbs).
    pub fn from_slice<T: Copy>(ctx: &Context, v: &[T]) -> Result<Self, MpcNetError> {
        let bytes = core::mem::size_of_val(v);
        let b = DeviceBuf::alloc(ctx, bytes)?;
        check(ctx, unsafe {
            sys::zk_memcpy_h2d(ctx.raw(), b.ptr, v.as_ptr() as *const c_void, bytes, ptr::null_mut())
        })?;
        Ok(b)
    }
    pub fn to_vec<T: Copy + Default>(&self, len: usize) -> Result<Vec<T>, MpcNetError> {
        if len == 0 {
            return Ok(Vec::new());
        }
        let mut v = vec![T::default(); len];
        let bytes = core::mem::size_of_val(&v[..]);
        debug_assert!(bytes <= self.bytes);
        check(&self.ctx, unsafe {
            sys::zk_memcpy_d2h(self.ctx.raw(), v.as_mut_ptr() as *mut c_void, self.ptr, bytes, ptr::null_mut())
        })?;
        check(&self.ctx, unsafe { sys::zk_stream_sync(self.ctx.raw(), ptr::null_mut()) })?;
        Ok(v)
    }
    pub fn ptr(&self) -> *mut c_void {
        self.ptr
    }
}
impl Drop for DeviceBuf {
    fn drop(&mut self) {
        unsafe {
            sys::zk_free(self.ctx.raw(), self.ptr);
        }
    }
}

/// Field elements as the bytes the ABI takes: arkworks' `Fp<MontBackend<_, N>, N>` holds `BigInt<N>([u64; N])` in
/// Montgomery form and a zero-sized marker, i.e. exactly the limbs of `include/zksaas.h`.
pub(crate) fn fr_ptr<F: PrimeField>(v: &[F]) -> *const c_void {
    debug_assert_eq!(core::mem::size_of::<F>(), ((F::MODULUS_BIT_SIZE as usize + 63) / 64) * 8);
    v.as_ptr() as *const c_void
}
pub(crate) fn fr_ptr_mut<F: PrimeField>(v: &mut [F]) -> *mut c_void {
    debug_assert_eq!(core::mem::size_of::<F>(), ((F::MODULUS_BIT_SIZE as usize + 63) / 64) * 8);
    v.as_mut_ptr() as *mut c_void
}

/// Limbs of one base-field element (Fq: N limbs; Fq2: c0 || c1), Montgomery form as held in memory.
fn push_base<B: Field>(x: &B, out: &mut Vec<u64>) {
    for c in x.to_base_prime_field_elements() {
        // BasePrimeField: PrimeField with MontBackend: the in-memory BigInt is the Montgomery residue
        let limbs: &[u64] = unsafe {
            core::slice::from_raw_parts(&c as *const _ as *const u64, core::mem::size_of_val(&c) / 8)
        };
        out.extend_from_slice(limbs);
    }
}
fn read_base<B: Field>(limbs: &[u64]) -> B {
    let k = B::extension_degree() as usize;
    let per = limbs.len() / k;
    let comps = (0..k).map(|i| {
        let mut c = B::BasePrimeField::zero();
        unsafe {
            core::ptr::copy_nonoverlapping(limbs[i * per..].as_ptr(), &mut c as *mut _ as *mut u64, per);
        }
        c
    });
    B::from_base_prime_field_elems(&comps.collect::<Vec<_>>()).expect("component count")
}

/// Affine points as packed `x || y` limb arrays, `(0, 0)` = identity (`include/zksaas.h`, data layout).
pub fn pack_affine<C: SWCurveConfig>(pts: &[Affine<C>]) -> Vec<u64> {
    let mut out = Vec::with_capacity(pts.len() * 2 * core::mem::size_of::<C::BaseField>() / 8);
    let zero = C::BaseField::zero();
    for p in pts {
        if p.infinity {
            push_base(&zero, &mut out);
            push_base(&zero, &mut out);
        } else {
            push_base(&p.x, &mut out);
            push_base(&p.y, &mut out);
        }
    }
    out
}
pub fn unpack_affine<C: SWCurveConfig>(limbs: &[u64], count: usize) -> Vec<Affine<C>> {
    let per = limbs.len() / count.max(1) / 2;
    (0..count)
        .map(|i| {
            let x: C::BaseField = read_base(&limbs[2 * i * per..(2 * i + 1) * per]);
            let y: C::BaseField = read_base(&limbs[(2 * i + 1) * per..(2 * i + 2) * per]);
            if x.is_zero() && y.is_zero() {
                Affine::<C>::identity()
            } else {
                Affine::<C>::new_unchecked(x, y)
            }
        })
        .collect()
}
/// Group values as Jacobian `X || Y || Z`, `Z = 0` = identity.
pub fn pack_jacobian<C: SWCurveConfig>(pts: &[Projective<C>]) -> Vec<u64> {
    let mut out = Vec::new();
    for p in pts {
        push_base(&p.x, &mut out);
        push_base(&p.y, &mut out);
        push_base(&p.z, &mut out);
    }
    out
}
pub fn unpack_jacobian<C: SWCurveConfig>(limbs: &[u64], count: usize) -> Vec<Projective<C>> {
    let per = limbs.len() / count.max(1) / 3;
    (0..count)
        .map(|i| {
            let x: C::BaseField = read_base(&limbs[3 * i * per..(3 * i + 1) * per]);
            let y: C::BaseField = read_base(&limbs[(3 * i + 1) * per..(3 * i + 2) * per]);
            let z: C::BaseField = read_base(&limbs[(3 * i + 2) * per..(3 * i + 3) * per]);
            Projective::<C> { x, y, z }
        })
        .collect()
}
