//! dg16-shim -- the arkworks-typed side of the boundary.
//!
//! Every function here has the signature of the reference call it replaces and forwards to ONE entry point of
//! `libdg16.so` (`dg16-sys`).  The reference's call sites change by a `#[cfg(feature = "dg16")]` line each
//! (`bindings/patches/*.patch`); with the feature off its code is byte-identical to today.
//!
//! | module      | replaces                                                                     | libdg16 entry points                 |
//! |-------------|-------------------------------------------------------------------------------|--------------------------------------|
//! | [`msm`]     | `G::msm(bases, scalars)`  dist-primitives/src/dmsm/mod.rs:82                  | `dg16_msm`, `dg16_bases_upload`, `dg16_msm_resident` |
//! | [`ntt`]     | `domain.fft_in_place` / `ifft_in_place`, the NTT part of `witness_map_from_matrices`  ark-circom/src/circom/qap.rs:64-91 | `dg16_ntt`, `dg16_h_poly` |
//! | [`prove`]   | `create_proof_with_reduction_and_matrices`  groth16/examples/sha256.rs:159, `qap::qap`  groth16/src/qap.rs:44-91 | `dg16_pk_create`, `dg16_qap`, `dg16_groth16_prove` |
//! | [`net`]     | `MpcNet` / `MpcSerNet` (mpc-net/src/lib.rs:46-140, dist-primitives/src/channel/mod.rs:7-60) and the packed-secret-sharing primitives over it | `dg16_net` vtable, `dg16_rccl_*`, `dg16_d_*`, `dg16_prove_a/_b/_c` |
//!
//! Memory layout: `ark_ec::short_weierstrass::Affine<P>` is `{x, y, infinity: bool}` with Rust's default (unspecified)
//! layout, so bases are repacked once per call to `x || y` Montgomery limbs with the identity as zeros ([`pack`]);
//! scalars (`Fp<MontBackend<_, 4>>`, `repr(transparent)` over `[u64; 4]` in arkworks 0.4) are passed as they lie in
//! memory with `DG16_F_SCALARS_MONT`.  Results come back as Jacobian `(x, y, z)` limbs = `Projective::new_unchecked`.

pub mod msm;
pub mod net;
pub mod ntt;
pub mod pack;
pub mod prove;

use dg16_sys as sys;
use once_cell::sync::Lazy;
use std::ffi::CStr;
use std::os::raw::c_int;

/// Error type of the shim: the text `dg16_last_error` gives, or the `min_len` of a length mismatch (what
/// `VariableBaseMSM::msm` returns as `Err(usize)` and `?` turns into `MpcNetError::Generic`, mpc-net/src/lib.rs:22-26).
#[derive(Debug, Clone)]
pub enum Dg16Error {
    LengthMismatch(usize),
    Status(c_int, String),
}
impl core::fmt::Display for Dg16Error {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        match self {
            Dg16Error::LengthMismatch(n) => write!(f, "{}", n),
            Dg16Error::Status(rc, msg) => write!(f, "dg16 status {}: {}", rc, msg),
        }
    }
}
impl std::error::Error for Dg16Error {}
#[cfg(feature = "mpc")]
impl From<Dg16Error> for mpc_net::MpcNetError {
    fn from(e: Dg16Error) -> Self {
        mpc_net::MpcNetError::Generic(e.to_string())
    }
}

/// One process-wide context per GPU (`DG16_DEVICE`, default 0).  The library is thread-safe per channel, which matches
/// the reference's concurrency: 8 party tasks, three futures in flight per task (mpc-net/src/multi.rs:305-314,
/// groth16/src/prove.rs:119-125).
pub struct Ctx(pub *mut sys::Dg16Ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { sys::dg16_ctx_destroy(self.0) }
    }
}
pub static CTX: Lazy<Ctx> = Lazy::new(|| {
    let device: c_int = std::env::var("DG16_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
    let mut p = core::ptr::null_mut();
    let rc = unsafe { sys::dg16_ctx_create(device, &mut p) };
    assert_eq!(rc, sys::DG16_OK, "dg16_ctx_create({}) failed: no MI355X / libdg16 has no CPU path", device);
    Ctx(p)
});

pub(crate) fn check(rc: c_int) -> Result<(), Dg16Error> {
    if rc == sys::DG16_OK {
        return Ok(());
    }
    let msg = unsafe {
        let p = sys::dg16_last_error(CTX.0);
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    Err(Dg16Error::Status(rc, msg))
}

/// (curve, group) of `include/dg16.h` for a short-Weierstrass config: implemented for G1 / G2 of the three curves.
pub trait Dg16Config: ark_ec::short_weierstrass::SWCurveConfig {
    const CURVE: c_int;
    const GROUP: c_int;
}
/// Curve id for a scalar field (the NTT / h-polynomial entry points take only the curve).
pub trait Dg16Scalar: ark_ff::PrimeField {
    const CURVE: c_int;
}
pub fn curve_of<F: Dg16Scalar>() -> c_int {
    F::CURVE
}
/// The same for a field that is only known as `F: PrimeField` -- the reference's transforms and dist-primitives are
/// generic over the scalar field (`witness_map_from_matrices<F: PrimeField, ..>`, `d_fft<F: FftField + PrimeField, ..>`)
/// and a `#[cfg(feature = "dg16")]` block cannot add a bound to them -- by its modulus: the low limb and the bit length
/// tell the three scalar fields apart.  None: not a field of this library (the call site falls through to arkworks).
pub fn curve_id_of<F: ark_ff::PrimeField>() -> Option<c_int> {
    let m = F::MODULUS;
    let limbs: &[u64] = m.as_ref();
    match (F::MODULUS_BIT_SIZE, limbs.first().copied()) {
        (254, Some(0x43E1_F593_F000_0001)) => Some(sys::DG16_BN254),
        (255, Some(0xFFFF_FFFF_0000_0001)) => Some(sys::DG16_BLS12_381),
        (253, Some(0x0A11_8000_0000_0001)) => Some(sys::DG16_BLS12_377),
        _ => None,
    }
}
macro_rules! curve_impls {
    ($feat:literal, $krate:ident, $id:expr) => {
        #[cfg(feature = $feat)]
        impl Dg16Config for $krate::g1::Config {
            const CURVE: c_int = $id;
            const GROUP: c_int = 1;
        }
        #[cfg(feature = $feat)]
        impl Dg16Config for $krate::g2::Config {
            const CURVE: c_int = $id;
            const GROUP: c_int = 2;
        }
        #[cfg(feature = $feat)]
        impl Dg16Scalar for $krate::Fr {
            const CURVE: c_int = $id;
        }
    };
}
curve_impls!("bn254", ark_bn254, sys::DG16_BN254);
curve_impls!("bls12-381", ark_bls12_381, sys::DG16_BLS12_381);
curve_impls!("bls12-377", ark_bls12_377, sys::DG16_BLS12_377);
