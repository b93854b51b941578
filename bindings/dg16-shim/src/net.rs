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
use mpc_net::MpcNet as _;
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
        Self::new_for_curve(F::CURVE, l)
    }
    pub fn new_for_curve(curve: c_int, l: usize) -> Result<Self, Dg16Error> {
        let mut p = core::ptr::null_mut();
        check(unsafe { sys::dg16_pss_create(CTX.0, curve, l as c_uint, &mut p) })?;
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
/// two provided collectives on device buffers).  `RcclNet` is the native one; [`MpcNetAdapter`] wraps ANY `MpcNet` of
/// the reference (`LocalTestNet`'s connections, `ProdNet`), which is what the generic call sites of the patches hold.
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

// the HIP runtime libdg16.so is linked against (payloads of the vtable are device buffers)
extern "C" {
    fn hipMemcpyAsync(dst: *mut core::ffi::c_void, src: *const core::ffi::c_void, bytes: usize, kind: c_int,
                      stream: *mut core::ffi::c_void) -> c_int;
    fn hipStreamSynchronize(stream: *mut core::ffi::c_void) -> c_int;
}
const HIP_MEMCPY_HOST_TO_DEVICE: c_int = 1;
const HIP_MEMCPY_DEVICE_TO_HOST: c_int = 2;

/// `dg16_net` over any `MpcNet` of the reference: the eight callbacks stage the device payload through host memory
/// (`hipMemcpyAsync` on the stream the library hands over + a stream synchronisation) and drive the trait's async methods
/// on the tokio runtime the adapter was made on.  Made INSIDE an async fn of the reference (`MpcNetAdapter::new(net)`
/// captures `Handle::current()`); the library's calls then run under `tokio::task::block_in_place` (see [`blocking`]), so the
/// callbacks may `block_on` without starving the worker threads that drive the other parties' tasks.
/// A transport whose three `MultiplexedStreamID`s share one ordered pipe passes `serial_channels = true` to `prove_c`.
pub struct MpcNetAdapter<'a, N: mpc_net::MpcNet> {
    net: &'a N,
    rt: tokio::runtime::Handle,
    table: Box<sys::Dg16Net>,
}
impl<'a, N: mpc_net::MpcNet> MpcNetAdapter<'a, N> {
    pub fn new(net: &'a N) -> Box<Self> {
        let mut me = Box::new(Self {
            net,
            rt: tokio::runtime::Handle::current(),
            table: Box::new(sys::Dg16Net {
                self_: core::ptr::null_mut(),
                n_parties: Self::cb_n_parties,
                party_id: Self::cb_party_id,
                gather_to_king: Self::cb_gather,
                scatter_from_king: Self::cb_scatter,
                is_init: Self::cb_is_init,
                send_to: Self::cb_send_to,
                recv_from: Self::cb_recv_from,
            }),
        });
        let p: *mut Self = &mut *me;
        me.table.self_ = p.cast();
        me
    }
    unsafe fn this<'b>(p: *mut core::ffi::c_void) -> &'b Self {
        &*(p as *const Self)
    }
    fn sid(channel: c_int) -> Option<MultiplexedStreamID> {
        match channel {
            0 => Some(MultiplexedStreamID::Zero),
            1 => Some(MultiplexedStreamID::One),
            2 => Some(MultiplexedStreamID::Two),
            _ => None,
        }
    }
    unsafe fn d2h(src: *const core::ffi::c_void, bytes: usize, stream: *mut core::ffi::c_void) -> Option<Vec<u8>> {
        let mut v = vec![0u8; bytes];
        if hipMemcpyAsync(v.as_mut_ptr().cast(), src, bytes, HIP_MEMCPY_DEVICE_TO_HOST, stream) != 0 { return None; }
        if hipStreamSynchronize(stream) != 0 { return None; }
        Some(v)
    }
    unsafe fn h2d(dst: *mut core::ffi::c_void, src: &[u8], stream: *mut core::ffi::c_void) -> bool {
        hipMemcpyAsync(dst, src.as_ptr().cast(), src.len(), HIP_MEMCPY_HOST_TO_DEVICE, stream) == 0
            && hipStreamSynchronize(stream) == 0        // `src` is dropped by the caller right after
    }
    extern "C" fn cb_n_parties(p: *mut core::ffi::c_void) -> c_uint {
        unsafe { Self::this(p) }.net.n_parties() as c_uint
    }
    extern "C" fn cb_party_id(p: *mut core::ffi::c_void) -> c_uint {
        unsafe { Self::this(p) }.net.party_id()
    }
    extern "C" fn cb_is_init(p: *mut core::ffi::c_void) -> c_int {
        unsafe { Self::this(p) }.net.is_init() as c_int
    }
    // client_send_or_king_receive (mpc-net/src/lib.rs:61-99): king's recv = n_parties x bytes, party-major
    extern "C" fn cb_gather(p: *mut core::ffi::c_void, channel: c_int, send: *const core::ffi::c_void, bytes: usize,
                            recv: *mut core::ffi::c_void, stream: *mut core::ffi::c_void) -> c_int {
        let me = unsafe { Self::this(p) };
        let (Some(sid), Some(out)) = (Self::sid(channel), unsafe { Self::d2h(send, bytes, stream) }) else {
            return sys::DG16_ERR_BAD_ARG;
        };
        match me.rt.block_on(me.net.client_send_or_king_receive(&out, sid)) {
            Ok(Some(parts)) => {
                let mut all = Vec::with_capacity(bytes * parts.len());
                for b in &parts {
                    if b.len() != bytes { return sys::DG16_ERR_NET; }
                    all.extend_from_slice(b);
                }
                if unsafe { Self::h2d(recv, &all, stream) } { sys::DG16_OK } else { sys::DG16_ERR_HIP }
            }
            Ok(None) => sys::DG16_OK,
            Err(_) => sys::DG16_ERR_NET,
        }
    }
    // client_receive_or_king_send (mpc-net/src/lib.rs:102-140): the king passes n_parties x bytes
    extern "C" fn cb_scatter(p: *mut core::ffi::c_void, channel: c_int, send: *const core::ffi::c_void, bytes: usize,
                             recv: *mut core::ffi::c_void, stream: *mut core::ffi::c_void) -> c_int {
        let me = unsafe { Self::this(p) };
        let Some(sid) = Self::sid(channel) else { return sys::DG16_ERR_BAD_ARG };
        let out = if me.net.is_king() {
            let n = me.net.n_parties();
            let Some(all) = (unsafe { Self::d2h(send, bytes * n, stream) }) else { return sys::DG16_ERR_HIP };
            Some(all.chunks(bytes).map(bytes::Bytes::copy_from_slice).collect::<Vec<_>>())
        } else {
            None
        };
        match me.rt.block_on(me.net.client_receive_or_king_send(out, sid)) {
            Ok(mine) if mine.len() == bytes => {
                if unsafe { Self::h2d(recv, &mine, stream) } { sys::DG16_OK } else { sys::DG16_ERR_HIP }
            }
            _ => sys::DG16_ERR_NET,
        }
    }
    extern "C" fn cb_send_to(p: *mut core::ffi::c_void, peer: c_uint, channel: c_int, send: *const core::ffi::c_void,
                             bytes: usize, stream: *mut core::ffi::c_void) -> c_int {
        let me = unsafe { Self::this(p) };
        let (Some(sid), Some(out)) = (Self::sid(channel), unsafe { Self::d2h(send, bytes, stream) }) else {
            return sys::DG16_ERR_BAD_ARG;
        };
        match me.rt.block_on(me.net.send_to(peer, bytes::Bytes::from(out), sid)) {
            Ok(()) => sys::DG16_OK,
            Err(_) => sys::DG16_ERR_NET,
        }
    }
    extern "C" fn cb_recv_from(p: *mut core::ffi::c_void, peer: c_uint, channel: c_int, recv: *mut core::ffi::c_void,
                               bytes: usize, stream: *mut core::ffi::c_void) -> c_int {
        let me = unsafe { Self::this(p) };
        let Some(sid) = Self::sid(channel) else { return sys::DG16_ERR_BAD_ARG };
        match me.rt.block_on(me.net.recv_from(peer, sid)) {
            Ok(b) if b.len() == bytes => {
                if unsafe { Self::h2d(recv, &b, stream) } { sys::DG16_OK } else { sys::DG16_ERR_HIP }
            }
            _ => sys::DG16_ERR_NET,
        }
    }
}
impl<'a, N: mpc_net::MpcNet> AsDg16Net for MpcNetAdapter<'a, N> {
    fn vtable(&self) -> *const sys::Dg16Net {
        &*self.table
    }
}
/// The vtable for whatever `MpcNet` a generic call site of the reference holds (the patches' `net: &Net`).
pub fn adapter_of<N: mpc_net::MpcNet>(net: &N) -> Box<MpcNetAdapter<'_, N>> {
    MpcNetAdapter::new(net)
}
/// Runs a blocking library call from inside an async fn of the reference without stalling the tokio worker it is on
/// (the 8 party tasks of `simulate_network_round` share those workers; the library's collectives wait for the peers).
pub fn blocking<T>(f: impl FnOnce() -> T) -> T {
    tokio::task::block_in_place(f)
}

/// The device-side twin of a `PackedSharingParams<F>` (one per curve and packing factor l, made on first use); `F` is
/// only known as a `PrimeField` at the reference's generic call sites: identified by its modulus (`crate::curve_id_of`).
pub fn pss_of<F: PrimeField>(pp: &secret_sharing::pss::PackedSharingParams<F>) -> Result<&'static Pss, Dg16Error> {
    use once_cell::sync::Lazy;
    use std::collections::HashMap;
    use std::sync::Mutex;
    static CACHE: Lazy<Mutex<HashMap<(c_int, usize), &'static Pss>>> = Lazy::new(|| Mutex::new(HashMap::new()));
    let curve = crate::curve_id_of::<F>()
        .ok_or_else(|| Dg16Error::Status(sys::DG16_ERR_BAD_CURVE, "scalar field of no curve of libdg16".into()))?;
    let mut m = CACHE.lock().unwrap();
    if let Some(p) = m.get(&(curve, pp.l)) {
        return Ok(*p);
    }
    let made: &'static Pss = Box::leak(Box::new(Pss::new_for_curve(curve, pp.l)?));
    m.insert((curve, pp.l), made);
    Ok(made)
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
