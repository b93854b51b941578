//! Transport and the packed-secret-sharing primitives over it.
//!
//! * [`RcclNet`]: `MpcNet` for the GPUs of one node -- the native RCCL backend of libdg16 (`dg16_rccl_*`: one
//!   communicator per `MultiplexedStreamID`, ncclSend / ncclRecv on device buffers over xGMI, stream-ordered).  Replaces
//!   `ProdNet` (TCP/TLS star around the king, mpc-net/src/prod.rs) between the processes of one MI355X node.
//! * [`d_msm`], [`d_fft`], [`d_pp`], [`deg_red`], [`ext_wit_h`], [`prove_a`] / [`prove_b`] / [`prove_c`]: the bodies of
//!   dist-primitives/src/{dmsm,dfft,dpp}/mod.rs, utils/deg_red.rs, groth16/src/{ext_wit,prove}.rs as ONE library call
//!   each; the king-side work the reference does on the CPU (`fft2_with_rearrange_pad`, `unpackexp`, the prefix
//!   products of `d_pp`) runs on the king's GPU inside the call.
//!
//! Payloads are DEVICE buffers: between GPUs of one node the wire form is the HBM form (the reference serialises every
//! `Vec<F>` with ark-serialize for TCP, dist-primitives/src/channel/mod.rs:14,49); `dg16_wire_fr_encode` /
//! `dg16_points_compress` convert at the edge to a party that speaks ark-serialize.
#![cfg(feature = "mpc")]
use crate::pack::{pack_affine, scalars_as_bytes, scalars_as_bytes_mut, unpack_projective, FieldBytes};
use crate::{check, sys, Dg16Config, Dg16Error, Dg16Scalar, CTX};
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::PrimeField;
use mpc_net::MultiplexedStreamID;
use std::os::raw::{c_int, c_uint};

/// Packed sharing parameters on the device (`PackedSharingParams::new(l)`, secret-sharing/src/pss.rs:34-62).
pub struct Pss(pub *mut sys::Dg16Pss);
unsafe impl Send for Pss {}
unsafe impl Sync for Pss {}
impl Pss {
    pub fn new<F: Dg16Scalar>(l: usize) -> Result<Self, Dg16Error> {
        let mut p = core::ptr::null_mut();
        check(unsafe { sys::dg16_pss_create(CTX.0, F::CURVE, l as c_uint, &mut p) })?;
        Ok(Self(p))
    }
}
impl Drop for Pss {
    fn drop(&mut self) {
        unsafe { sys::dg16_pss_destroy(self.0) }
    }
}

/// One party = one process = one GPU.  Rank 0 is the king.  `id128` is made by rank 0 (`RcclNet::unique_id`) and
/// handed to the other ranks by the launcher's rendezvous.
pub struct RcclNet {
    h: *mut sys::Dg16Rccl,
}
unsafe impl Send for RcclNet {}
unsafe impl Sync for RcclNet {}
impl RcclNet {
    pub fn unique_id() -> Result<[u8; 128], Dg16Error> {
        let mut id = [0u8; 128];
        check(unsafe { sys::dg16_rccl_unique_id(id.as_mut_ptr().cast()) })?;
        Ok(id)
    }
    /// ncclCommInitRank (+ the two per-channel communicators): blocks until all ranks have joined.
    pub fn create(id128: &[u8; 128], n_ranks: u32, rank: u32) -> Result<Self, Dg16Error> {
        let mut h = core::ptr::null_mut();
        check(unsafe { sys::dg16_rccl_create(CTX.0, id128.as_ptr().cast(), n_ranks, rank, &mut h) })?;
        Ok(Self { h })
    }
    /// (rank count, this rank) as the communicator itself reports them.
    pub fn ranks(&self) -> Result<(u32, u32), Dg16Error> {
        let (mut n, mut r) = (0, 0);
        check(unsafe { sys::dg16_rccl_ranks(self.h, &mut n, &mut r) })?;
        Ok((n, r))
    }
    pub fn net(&self) -> *const sys::Dg16Net {
        unsafe { sys::dg16_rccl_net(self.h) }
    }
    pub fn comm(&self) -> *const sys::Dg16Comm {
        unsafe { sys::dg16_rccl_comm(self.h) }
    }
}
impl Drop for RcclNet {
    fn drop(&mut self) {
        unsafe { sys::dg16_rccl_destroy(self.h) }
    }
}

/// `d_msm::<G, _>(bases, scalars, pp, net, sid)` (dist-primitives/src/dmsm/mod.rs:70-98): local MSM of the share
/// vectors, gather to the king, `unpackexp` + sum there, the same point back to every party.
pub fn d_msm<P: Dg16Config>(
    bases: &[Affine<P>],
    scalars: &[P::ScalarField],
    pp: &Pss,
    net: *const sys::Dg16Net,
    sid: MultiplexedStreamID,
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
    // share vectors of CRS elements are linear combinations of subgroup points
    let flags = sys::DG16_F_SCALARS_MONT | sys::DG16_F_BASES_IN_SUBGROUP;
    check(unsafe {
        sys::dg16_d_msm(
            CTX.0, pp.0, net, P::GROUP, packed.as_ptr().cast(), scalars_as_bytes(scalars).as_ptr().cast(),
            bases.len(), scalars.len(), flags, sid as c_int, out.as_mut_ptr().cast(),
        )
    })?;
    Ok(unpack_projective::<P>(&out))
}

/// `d_fft` (inverse = false) / `d_ifft` (inverse = true), dist-primitives/src/dfft/mod.rs:17-95: `fft1_in_place` on every
/// party, `fft2_with_rearrange_pad` on the king, shares of the evaluations back.  `pcoeff_share.len() * l` must be the
/// domain size (the reference's debug assertion, :31-37).
pub fn d_fft<F: PrimeField>(
    pcoeff_share: &[F],
    rearrange: bool,
    pad: usize,
    degree2: bool,
    log_m: u32,
    inverse: bool,
    pp: &Pss,
    net: *const sys::Dg16Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<F>, Dg16Error> {
    let mut out = vec![F::zero(); pad * pcoeff_share.len()];
    check(unsafe {
        sys::dg16_d_fft(
            CTX.0, pp.0, net, scalars_as_bytes(pcoeff_share).as_ptr().cast(), pcoeff_share.len(), log_m, rearrange as c_int,
            pad as c_uint, degree2 as c_int, inverse as c_int, scalars_as_bytes_mut(&mut out).as_mut_ptr().cast(), 0,
            sid as c_int,
        )
    })?;
    Ok(out)
}

/// `d_pp` (dist-primitives/src/dpp/mod.rs:17-88): shares of the prefix products of num / den.
pub fn d_pp<F: PrimeField>(num: &[F], den: &[F], pp: &Pss, net: *const sys::Dg16Net, sid: MultiplexedStreamID)
    -> Result<Vec<F>, Dg16Error> {
    if num.len() != den.len() {
        return Err(Dg16Error::LengthMismatch(num.len().min(den.len())));
    }
    let mut out = vec![F::zero(); num.len()];
    check(unsafe {
        sys::dg16_d_pp(CTX.0, pp.0, net, scalars_as_bytes(num).as_ptr().cast(), scalars_as_bytes(den).as_ptr().cast(),
                       num.len(), scalars_as_bytes_mut(&mut out).as_mut_ptr().cast(), 0, sid as c_int)
    })?;
    Ok(out)
}

/// `deg_red` (dist-primitives/src/utils/deg_red.rs:10-28).
pub fn deg_red<F: PrimeField>(px: &[F], pp: &Pss, net: *const sys::Dg16Net, sid: MultiplexedStreamID) -> Result<Vec<F>, Dg16Error> {
    let mut out = vec![F::zero(); px.len()];
    check(unsafe {
        sys::dg16_deg_red(CTX.0, pp.0, net, scalars_as_bytes(px).as_ptr().cast(), px.len(),
                          scalars_as_bytes_mut(&mut out).as_mut_ptr().cast(), 0, sid as c_int)
    })?;
    Ok(out)
}

/// `ext_wit::h` (groth16/src/ext_wit.rs:16-101): packed shares of the h-polynomial from the PackedQAPShare vectors.
pub fn ext_wit_h<F: PrimeField>(a: &[F], b: &[F], c: &[F], log_m: u32, pp: &Pss, net: *const sys::Dg16Net) -> Result<Vec<F>, Dg16Error> {
    let mut out = vec![F::zero(); a.len()];
    check(unsafe {
        sys::dg16_ext_wit_h(CTX.0, pp.0, net, scalars_as_bytes(a).as_ptr().cast(), scalars_as_bytes(b).as_ptr().cast(),
                            scalars_as_bytes(c).as_ptr().cast(), log_m, scalars_as_bytes_mut(&mut out).as_mut_ptr().cast(), 0)
    })?;
    Ok(out)
}

/// `prove::A::compute` (groth16/src/prove.rs:21-46): `L + N * r + d_msm(S, a)`.
pub fn prove_a<P: Dg16Config>(
    l: Affine<P>, n: Affine<P>, r: P::ScalarField, s_bases: &[Affine<P>], a: &[P::ScalarField], pp: &Pss,
    net: *const sys::Dg16Net, sid: MultiplexedStreamID,
) -> Result<Projective<P>, Dg16Error>
where
    P::BaseField: FieldBytes,
{
    let (pl, pn, ps) = (pack_affine(&[l]), pack_affine(&[n]), pack_affine(s_bases));
    let fe = <P::BaseField as FieldBytes>::BYTES;
    let mut out = vec![0u8; 3 * fe];
    let rr = [r];
    check(unsafe {
        sys::dg16_prove_a(CTX.0, pp.0, net, pl.as_ptr().cast(), pn.as_ptr().cast(), scalars_as_bytes(&rr).as_ptr().cast(),
                          ps.as_ptr().cast(), scalars_as_bytes(a).as_ptr().cast(), s_bases.len(), a.len(),
                          sys::DG16_F_SCALARS_MONT | sys::DG16_F_BASES_IN_SUBGROUP, sid as c_int, out.as_mut_ptr().cast())
    })?;
    Ok(unpack_projective::<P>(&out))
}

/// `prove::B::compute` (prove.rs:62-85): `Z + K * s + d_msm(V, a)` in G2 -- the same entry point shape as A.
pub fn prove_b<P: Dg16Config>(
    z: Affine<P>, k: Affine<P>, s: P::ScalarField, v_bases: &[Affine<P>], a: &[P::ScalarField], pp: &Pss,
    net: *const sys::Dg16Net, sid: MultiplexedStreamID,
) -> Result<Projective<P>, Dg16Error>
where
    P::BaseField: FieldBytes,
{
    let (pz, pk, pv) = (pack_affine(&[z]), pack_affine(&[k]), pack_affine(v_bases));
    let fe = <P::BaseField as FieldBytes>::BYTES;
    let mut out = vec![0u8; 3 * fe];
    let ss = [s];
    check(unsafe {
        sys::dg16_prove_b(CTX.0, pp.0, net, pz.as_ptr().cast(), pk.as_ptr().cast(), scalars_as_bytes(&ss).as_ptr().cast(),
                          pv.as_ptr().cast(), scalars_as_bytes(a).as_ptr().cast(), v_bases.len(), a.len(),
                          sys::DG16_F_SCALARS_MONT | sys::DG16_F_BASES_IN_SUBGROUP, sid as c_int, out.as_mut_ptr().cast())
    })?;
    Ok(unpack_projective::<P>(&out))
}

/// `prove::C::compute` (prove.rs:106-136): `w + u + A s + M r + h r` with the three `d_msm` JOINED on channels 0 / 1 / 2
/// like `tokio::try_join!` (:113-125) -- three host threads inside the call.  `serial_channels`: for a transport whose
/// channels are one ordered pipe, run them one after another instead (`DG16_F_SERIAL_CHANNELS`; same result).
#[allow(clippy::too_many_arguments)]
pub fn prove_c<P: Dg16Config>(
    a_point: Projective<P>, m: Affine<P>, s: P::ScalarField, r: P::ScalarField, w_bases: &[Affine<P>], ax: &[P::ScalarField],
    u_bases: &[Affine<P>], h: &[P::ScalarField], h_bases: &[Affine<P>], a: &[P::ScalarField], pp: &Pss,
    net: *const sys::Dg16Net, serial_channels: bool,
) -> Result<Projective<P>, Dg16Error>
where
    P::BaseField: FieldBytes,
{
    let fe = <P::BaseField as FieldBytes>::BYTES;
    let mut pa = vec![0u8; 3 * fe];
    a_point.x.write_mont(&mut pa[..fe]);
    a_point.y.write_mont(&mut pa[fe..2 * fe]);
    a_point.z.write_mont(&mut pa[2 * fe..]);
    let (pm, pw, pu, ph) = (pack_affine(&[m]), pack_affine(w_bases), pack_affine(u_bases), pack_affine(h_bases));
    let (ss, rr) = ([s], [r]);
    let mut out = vec![0u8; 3 * fe];
    let mut flags = sys::DG16_F_SCALARS_MONT | sys::DG16_F_BASES_IN_SUBGROUP;
    if serial_channels {
        flags |= sys::DG16_F_SERIAL_CHANNELS;
    }
    check(unsafe {
        sys::dg16_prove_c(CTX.0, pp.0, net, pa.as_ptr().cast(), pm.as_ptr().cast(), scalars_as_bytes(&ss).as_ptr().cast(),
                          scalars_as_bytes(&rr).as_ptr().cast(), pw.as_ptr().cast(), scalars_as_bytes(ax).as_ptr().cast(),
                          w_bases.len(), ax.len(), pu.as_ptr().cast(), scalars_as_bytes(h).as_ptr().cast(), u_bases.len(),
                          h.len(), ph.as_ptr().cast(), scalars_as_bytes(a).as_ptr().cast(), h_bases.len(), a.len(), flags,
                          out.as_mut_ptr().cast())
    })?;
    Ok(unpack_projective::<P>(&out))
}

/// A transport the library can drive: anything that hands out the `dg16_net` vtable (MpcNet's required methods and its
/// two provided collectives on device buffers).  `RcclNet` is the native one; a caller that keeps its own `MpcNet`
/// (e.g. `ProdNet` to a remote party) implements the eight callbacks over it, staging through pinned host memory, and
/// sets `serial_channels` if its channels share one pipe.
pub trait AsDg16Net {
    fn vtable(&self) -> *const sys::Dg16Net;
}
impl AsDg16Net for RcclNet {
    fn vtable(&self) -> *const sys::Dg16Net {
        self.net()
    }
}
pub fn vtable_of<N: AsDg16Net>(net: &N) -> *const sys::Dg16Net {
    net.vtable()
}

/// The device-side twin of a `PackedSharingParams<F>` (one per packing factor l, made on first use).
pub fn pss_of<F: Dg16Scalar>(pp: &secret_sharing::pss::PackedSharingParams<F>) -> &'static Pss {
    use once_cell::sync::Lazy;
    use std::collections::HashMap;
    use std::sync::Mutex;
    static CACHE: Lazy<Mutex<HashMap<(c_int, usize), &'static Pss>>> = Lazy::new(|| Mutex::new(HashMap::new()));
    let mut m = CACHE.lock().unwrap();
    *m.entry((F::CURVE, pp.l)).or_insert_with(|| Box::leak(Box::new(Pss::new::<F>(pp.l).expect("dg16_pss_create"))))
}

/// `prove::C::compute` with the reference's own types (`E::G1`, `E::G1Affine`): groth16/src/prove.rs:104-136.
#[allow(clippy::too_many_arguments)]
pub fn prove_c_pairing<E, P>(
    a_point: Projective<P>, m: Affine<P>, s: P::ScalarField, r: P::ScalarField, w_bases: &[Affine<P>], ax: &[P::ScalarField],
    u_bases: &[Affine<P>], h: &[P::ScalarField], h_bases: &[Affine<P>], a: &[P::ScalarField], pp: &Pss,
    net: *const sys::Dg16Net, serial_channels: bool,
) -> Result<Projective<P>, Dg16Error>
where
    E: ark_ec::pairing::Pairing<G1 = Projective<P>, G1Affine = Affine<P>>,
    P: Dg16Config,
    P::BaseField: FieldBytes,
{
    prove_c::<P>(a_point, m, s, r, w_bases, ax, u_bases, h, h_bases, a, pp, net, serial_channels)
}
