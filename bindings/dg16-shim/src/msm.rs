//! `G::msm(bases, scalars)` -> `dg16_msm` (dist-primitives/src/dmsm/mod.rs:82; also examples/msm_bench.rs:20,
//! dmsm_test.rs:49-50, groth16/examples/local_groth_bench.rs:139-147) and, for a fixed CRS, `dg16_bases_upload` +
//! `dg16_msm_resident` (window tables in HBM: one bucket set, no Horner tail).
use crate::pack::{pack_affine, scalars_as_bytes, unpack_projective, FieldBytes};
use crate::{check, sys, Dg16Config, Dg16Error, CTX};
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use std::os::raw::{c_int, c_uint};

/// `sum_i scalars[i] * bases[i]`, the value `VariableBaseMSM::msm` returns.  `channel` is the caller's
/// `MultiplexedStreamID as i32` (each channel owns a HIP stream: the three `d_msm` of `prove::C` run concurrently).
///
/// `Err(LengthMismatch(min_len))` mirrors the `Err(usize)` of `VariableBaseMSM::msm`.
///
/// `bases_in_subgroup`: every base is in the order-r subgroup -- true of any `G1Affine` / `G2Affine` that came out of
/// `deserialize_with_mode(.., Validate::Yes)` or of CRS generation (`pk.a_query` ..).  It lets the library split the
/// scalars with the curve's endomorphism (GLV); for a curve point OUTSIDE the subgroup the endomorphism is not a
/// scalar multiplication, so without the flag only cofactor-one groups (BN254 G1) take that path.
pub fn msm<P: Dg16Config>(
    bases: &[Affine<P>],
    scalars: &[P::ScalarField],
    channel: c_int,
    bases_in_subgroup: bool,
) -> Result<Projective<P>, Dg16Error>
where
    P::BaseField: FieldBytes,
{
    if bases.len() != scalars.len() {
        return Err(Dg16Error::LengthMismatch(bases.len().min(scalars.len())));
    }
    let packed = pack_affine(bases);
    let fe = <P::BaseField as FieldBytes>::BYTES;
    let mut out = vec![0u8; 3 * fe];
    let mut flags: c_uint = sys::DG16_F_SCALARS_MONT;
    if bases_in_subgroup {
        flags |= sys::DG16_F_BASES_IN_SUBGROUP;
    }
    check(unsafe {
        sys::dg16_msm(
            CTX.0,
            P::CURVE,
            P::GROUP,
            packed.as_ptr().cast(),
            scalars_as_bytes(scalars).as_ptr().cast(),
            bases.len(),
            scalars.len(),
            flags,
            channel,
            out.as_mut_ptr().cast(),
        )
    })?;
    Ok(unpack_projective::<P>(&out))
}

/// A fixed base vector made resident once (`PackedProvingKeyShare`'s five vectors, an SRS for `dpoly_commit`):
/// `dg16_bases_upload` builds the table of window multiples T[w][i] = 2^(c w) P_i in HBM.
pub struct ResidentBases<P: Dg16Config> {
    h: *mut sys::Dg16Bases,
    n: usize,
    _p: core::marker::PhantomData<P>,
}
unsafe impl<P: Dg16Config> Send for ResidentBases<P> {}
unsafe impl<P: Dg16Config> Sync for ResidentBases<P> {}
impl<P: Dg16Config> Drop for ResidentBases<P> {
    fn drop(&mut self) {
        unsafe { sys::dg16_bases_free(self.h) }
    }
}
impl<P: Dg16Config> ResidentBases<P>
where
    P::BaseField: FieldBytes,
{
    pub fn upload(bases: &[Affine<P>]) -> Result<Self, Dg16Error> {
        let packed = pack_affine(bases);
        let mut h = core::ptr::null_mut();
        check(unsafe {
            sys::dg16_bases_upload(CTX.0, P::CURVE, P::GROUP, packed.as_ptr().cast(), bases.len(), 0, &mut h)
        })?;
        Ok(Self { h, n: bases.len(), _p: core::marker::PhantomData })
    }
    pub fn raw(&self) -> *const sys::Dg16Bases {
        self.h
    }
    /// `G::msm(self.bases, scalars)` over the resident tables.
    pub fn msm(&self, scalars: &[P::ScalarField], channel: c_int) -> Result<Projective<P>, Dg16Error> {
        if scalars.len() != self.n {
            return Err(Dg16Error::LengthMismatch(self.n.min(scalars.len())));
        }
        let fe = <P::BaseField as FieldBytes>::BYTES;
        let mut out = vec![0u8; 3 * fe];
        check(unsafe {
            sys::dg16_msm_resident(
                CTX.0,
                self.h,
                scalars_as_bytes(scalars).as_ptr().cast(),
                scalars.len(),
                sys::DG16_F_SCALARS_MONT,
                channel,
                out.as_mut_ptr().cast(),
            )
        })?;
        Ok(unpack_projective::<P>(&out))
    }
}

/// The call-site form: `G::msm(bases, scalars)` for `G = Projective<P>` (what `d_msm::<G, _>` is instantiated with:
/// `E::G1`, `E::G2` of the three curves).  dist-primitives/src/dmsm/mod.rs:82 under `feature = "dg16"`.
pub trait Dg16Group: ark_ec::CurveGroup {
    fn dg16_msm(bases: &[Self::Affine], scalars: &[Self::ScalarField], channel: c_int, bases_in_subgroup: bool)
        -> Result<Self, Dg16Error>;
}
impl<P: Dg16Config> Dg16Group for Projective<P>
where
    P::BaseField: FieldBytes,
{
    fn dg16_msm(bases: &[Affine<P>], scalars: &[P::ScalarField], channel: c_int, bases_in_subgroup: bool)
        -> Result<Self, Dg16Error> {
        msm::<P>(bases, scalars, channel, bases_in_subgroup)
    }
}
pub fn msm_group<G: Dg16Group>(bases: &[G::Affine], scalars: &[G::ScalarField], channel: c_int, bases_in_subgroup: bool)
    -> Result<G, Dg16Error> {
    G::dg16_msm(bases, scalars, channel, bases_in_subgroup)
}
