//! dg16-sys -- raw bindings of `libdg16.so`, the MI355X (gfx950) Groth16 hot path behind the `dist-primitives` surface.
//!
//! One `extern "C"` item per function `include/dg16.h` declares (the header is the contract: `tests/test_abi.py` of the
//! library's repository diffs this file against it and against the symbols the shared object exports), the
//! `#[repr(C)]` mirrors of its structs, and its flag / status constants.  No logic lives here; `dg16-shim` holds the
//! arkworks-typed wrappers the reference's call sites use.
#![allow(non_camel_case_types, non_snake_case, clippy::too_many_arguments)]

use std::os::raw::{c_char, c_int, c_uint, c_void};

macro_rules! opaque {
    ($($name:ident),*) => { $(#[repr(C)] pub struct $name { _p: [u8; 0], _m: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)> })* };
}
opaque!(Dg16Ctx, Dg16Pk, Dg16Bases, Dg16Pss, Dg16LocalNet, Dg16Rccl, Dg16R1cs, Dg16Zkey);

// enum dg16_curve
pub const DG16_BN254: c_int = 0;
pub const DG16_BLS12_381: c_int = 1;
pub const DG16_BLS12_377: c_int = 2;
// enum dg16_status
pub const DG16_OK: c_int = 0;
pub const DG16_ERR_LENGTH_MISMATCH: c_int = 1; // mirrors Err(usize) of VariableBaseMSM::msm
pub const DG16_ERR_BAD_CURVE: c_int = 2;
pub const DG16_ERR_BAD_ARG: c_int = 3;
pub const DG16_ERR_OOM: c_int = 4;
pub const DG16_ERR_HIP: c_int = 5;
pub const DG16_ERR_NET: c_int = 6;
pub const DG16_ERR_UNSUPPORTED: c_int = 7;
// enum dg16_flags
pub const DG16_F_SCALARS_MONT: c_uint = 1; // arkworks memory is Montgomery form
pub const DG16_F_DEVICE_PTRS: c_uint = 2;
pub const DG16_F_OUT_AFFINE: c_uint = 4;
pub const DG16_F_H_CYCLIC: c_uint = 8;
pub const DG16_F_SERIAL_CHANNELS: c_uint = 16;
pub const DG16_F_OVERLAP_TAIL: c_uint = 32;
pub const DG16_F_BASES_IN_SUBGROUP: c_uint = 64;
// enum dg16_field_opcode
pub const DG16_OP_ADD: c_int = 0;
pub const DG16_OP_SUB: c_int = 1;
pub const DG16_OP_MUL: c_int = 2;
pub const DG16_OP_SQR: c_int = 3;
pub const DG16_OP_INV: c_int = 4;
pub const DG16_OP_TO_MONT: c_int = 5;
pub const DG16_OP_FROM_MONT: c_int = 6;
pub const DG16_OP_NEG: c_int = 7;
// dg16_zkey_points: which
pub const DG16_ZKEY_ALPHA_G1: c_int = 0;
pub const DG16_ZKEY_BETA_G1: c_int = 1;
pub const DG16_ZKEY_BETA_G2: c_int = 2;
pub const DG16_ZKEY_GAMMA_G2: c_int = 3;
pub const DG16_ZKEY_DELTA_G1: c_int = 4;
pub const DG16_ZKEY_DELTA_G2: c_int = 5;
pub const DG16_ZKEY_IC: c_int = 6;
pub const DG16_ZKEY_A: c_int = 7;
pub const DG16_ZKEY_B1: c_int = 8;
pub const DG16_ZKEY_B2: c_int = 9;
pub const DG16_ZKEY_L: c_int = 10;
pub const DG16_ZKEY_H: c_int = 11;

/// `dg16_pk_info`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Dg16PkInfo {
    pub n_ab: u64,
    pub n_l: u64,
    pub n_h: u64,
    pub c_ab: u32,
    pub c_l: u32,
    pub c_h: u32,
    pub shard: u32,
    pub n_shards: u32,
    pub table_bytes: u64,
    pub table_stride: u32,
}

/// `dg16_comm`: what the single-statement prover needs from a transport (all-gather + all-to-all on device buffers,
/// ordered on the HIP stream the payload was produced on).
#[repr(C)]
pub struct Dg16Comm {
    pub self_: *mut c_void,
    pub n_ranks: extern "C" fn(*mut c_void) -> c_uint,
    pub rank: extern "C" fn(*mut c_void) -> c_uint,
    pub all_gather: extern "C" fn(*mut c_void, *const c_void, usize, *mut c_void, *mut c_void) -> c_int,
    pub all_to_all: extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize, *mut c_void) -> c_int,
}

/// `dg16_net`: MpcNet (mpc-net/src/lib.rs:46-140) on device buffers.
#[repr(C)]
pub struct Dg16Net {
    pub self_: *mut c_void,
    pub n_parties: extern "C" fn(*mut c_void) -> c_uint,
    pub party_id: extern "C" fn(*mut c_void) -> c_uint,
    /// king: recv = n_parties * bytes (party-major); others: recv ignored.  (self, channel, send, bytes, recv, hip_stream)
    pub gather_to_king: extern "C" fn(*mut c_void, c_int, *const c_void, usize, *mut c_void, *mut c_void) -> c_int,
    pub scatter_from_king: extern "C" fn(*mut c_void, c_int, *const c_void, usize, *mut c_void, *mut c_void) -> c_int,
    pub is_init: extern "C" fn(*mut c_void) -> c_int,
    /// (self, peer, channel, send, bytes, hip_stream)
    pub send_to: extern "C" fn(*mut c_void, c_uint, c_int, *const c_void, usize, *mut c_void) -> c_int,
    pub recv_from: extern "C" fn(*mut c_void, c_uint, c_int, *mut c_void, usize, *mut c_void) -> c_int,
}

/// `dg16_r1cs_header`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Dg16R1csHeader {
    pub n_wires: u32,
    pub n_pub_out: u32,
    pub n_pub_in: u32,
    pub n_prv_in: u32,
    pub n_constraints: u32,
    pub has_wire_map: u32,
    pub n_labels: u64,
}
/// `dg16_zkey_header`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Dg16ZkeyHeader {
    pub n_vars: u32,
    pub n_public: u32,
    pub domain_size: u32,
    pub num_constraints: u32,
}
/// `dg16_csr` (owned by the handle; coeff: nnz x 32-byte little-endian field elements)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Dg16Csr {
    pub n_rows: u64,
    pub nnz: u64,
    pub row_ptr: *const u32,
    pub col: *const u32,
    pub coeff: *const c_void,
}
/// `dg16_arkkey_layout_t`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Dg16ArkKeyLayout {
    pub n_ic: u64, pub n_a: u64, pub n_b1: u64, pub n_b2: u64, pub n_h: u64, pub n_l: u64,
    pub off_alpha_g1: u64, pub off_beta_g2: u64, pub off_gamma_g2: u64, pub off_delta_g2: u64, pub off_ic: u64,
    pub off_beta_g1: u64, pub off_delta_g1: u64, pub off_a: u64, pub off_b1: u64, pub off_b2: u64, pub off_h: u64,
    pub off_l: u64,
    pub bytes: u64,
}

#[link(name = "dg16")]
extern "C" {
    pub fn dg16_ctx_create(device: c_int, out: *mut *mut Dg16Ctx) -> c_int;
    pub fn dg16_ctx_destroy(ctx: *mut Dg16Ctx);
    pub fn dg16_last_error(ctx: *mut Dg16Ctx) -> *const c_char;
    pub fn dg16_msm(ctx: *mut Dg16Ctx, curve: c_int, group: c_int, bases: *const c_void,
                    scalars: *const c_void, n_bases: usize, n_scalars: usize, flags: c_uint,
                    channel: c_int, out: *mut c_void) -> c_int;
    pub fn dg16_ntt(ctx: *mut Dg16Ctx, curve: c_int, data: *mut c_void, log_n: c_uint,
                    inverse: c_int, coset_offset: *const c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_h_poly(ctx: *mut Dg16Ctx, curve: c_int, a: *const c_void, b: *const c_void,
                       c: *const c_void, log_m: c_uint, out: *mut c_void, flags: c_uint,
                       channel: c_int) -> c_int;
    pub fn dg16_pk_create(ctx: *mut Dg16Ctx, curve: c_int, num_vars: usize, num_inputs: usize,
                          domain_size: usize, a_query: *const c_void, b_g1_query: *const c_void,
                          b_g2_query: *const c_void, h_query: *const c_void, l_query: *const c_void,
                          fixed_points: *const c_void, flags: c_uint, out: *mut *mut Dg16Pk) -> c_int;
    pub fn dg16_pk_destroy(pk: *mut Dg16Pk);
    pub fn dg16_groth16_prove(ctx: *mut Dg16Ctx, pk: *const Dg16Pk, a: *const c_void,
                              b: *const c_void, c: *const c_void, full_assignment: *const c_void,
                              r_s: *const c_void, flags: c_uint, proof_out: *mut c_void) -> c_int;

    // ---- the rest of include/dg16.h (every entry point; sections 3.4-3.7 show the call sites) ----
    pub fn dg16_set_stream(ctx: *mut Dg16Ctx, channel: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn dg16_sync(ctx: *mut Dg16Ctx, channel: c_int) -> c_int;
    pub fn dg16_device_info(ctx: *mut Dg16Ctx, name: *mut c_char, name_len: usize, cus: *mut c_int) -> c_int;
    pub fn dg16_last_kernel_ms(ctx: *mut Dg16Ctx, channel: c_int, which: c_int, ms: *mut f32) -> c_int;
    pub fn dg16_field_op(ctx: *mut Dg16Ctx, field_id: c_int, op: c_int, a: *const c_void, b: *const c_void,
                         out: *mut c_void, n: usize, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_qap(ctx: *mut Dg16Ctx, curve: c_int, num_constraints: usize, num_inputs: usize, num_vars: usize,
                    log_m: c_uint, a_row_ptr: *const u32, a_col: *const u32, a_coeff: *const c_void,
                    b_row_ptr: *const u32, b_col: *const u32, b_coeff: *const c_void,
                    full_assignment: *const c_void, a_out: *mut c_void, b_out: *mut c_void, c_out: *mut c_void,
                    flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_qap_rows(ctx: *mut Dg16Ctx, curve: c_int, num_constraints: usize, num_inputs: usize,
                    num_vars: usize, log_m: c_uint, a_row_ptr: *const u32, a_col: *const u32,
                    a_coeff: *const c_void, b_row_ptr: *const u32, b_col: *const u32, b_coeff: *const c_void,
                    full_assignment: *const c_void, row_start: usize, row_stride: usize, a_out: *mut c_void,
                    b_out: *mut c_void, c_out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_gen_bases(ctx: *mut Dg16Ctx, curve: c_int, group: c_int, seed: u64, n: usize, out: *mut c_void,
                          flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_to_affine(ctx: *mut Dg16Ctx, curve: c_int, group: c_int, jac: *const c_void, out: *mut c_void,
                          n: usize, flags: c_uint, channel: c_int) -> c_int;
    // resident bases (a CRS is uploaded once; msm over its window tables)
    pub fn dg16_bases_upload(ctx: *mut Dg16Ctx, curve: c_int, group: c_int, bases: *const c_void, n: usize,
                             flags: c_uint, out: *mut *mut Dg16Bases) -> c_int;
    pub fn dg16_ctx_set_table_budget(ctx: *mut Dg16Ctx, bytes: u64) -> c_int;   // HBM budget of one key's window tables
    pub fn dg16_bases_free(h: *mut Dg16Bases);
    pub fn dg16_bases_info(h: *const Dg16Bases, n: *mut usize, window_bits: *mut c_uint, table_bytes: *mut u64) -> c_int;
    pub fn dg16_msm_resident(ctx: *mut Dg16Ctx, h: *const Dg16Bases, scalars: *const c_void, n_scalars: usize,
                             flags: c_uint, channel: c_int, out: *mut c_void) -> c_int;
    pub fn dg16_d_msm_resident(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, bases: *const Dg16Bases,
                               scalars: *const c_void, n_scalars: usize, flags: c_uint, channel: c_int,
                               out: *mut c_void) -> c_int;
    // keys, shards, the halves of a distributed proof
    pub fn dg16_pk_info_get(pk: *const Dg16Pk, out: *mut Dg16PkInfo) -> c_int;
    pub fn dg16_pk_create_shard(ctx: *mut Dg16Ctx, curve: c_int, num_vars: usize, num_inputs: usize,
                                domain_size: usize, a_query: *const c_void, b_g1_query: *const c_void,
                                b_g2_query: *const c_void, h_query: *const c_void, l_query: *const c_void,
                                fixed_points: *const c_void, shard: c_uint, n_shards: c_uint, flags: c_uint,
                                out: *mut *mut Dg16Pk) -> c_int;                 // flags | DG16_F_H_CYCLIC
    pub fn dg16_groth16_results_bytes(curve: c_int) -> usize;
    pub fn dg16_groth16_msms(ctx: *mut Dg16Ctx, pk: *const Dg16Pk, a: *const c_void, b: *const c_void,
                             c: *const c_void, w: *const c_void, r_s: *const c_void, flags: c_uint,
                             results_out: *mut c_void) -> c_int;
    pub fn dg16_groth16_msms_h(ctx: *mut Dg16Ctx, pk: *const Dg16Pk, h_shard: *const c_void, w: *const c_void,
                               r_s: *const c_void, flags: c_uint, results_out: *mut c_void) -> c_int;
    pub fn dg16_groth16_assemble(ctx: *mut Dg16Ctx, pk: *const Dg16Pk, gathered: *const c_void, n_shards: usize,
                                 r_s: *const c_void, flags: c_uint, proof_out: *mut c_void) -> c_int;
    // one process per GPU: collectives, sharded h-polynomial, the whole distributed proof (section 4)
    pub fn dg16_ntt_dist(ctx: *mut Dg16Ctx, curve: c_int, comm: *const Dg16Comm, input: *const c_void, out: *mut c_void,
                         log_n: c_uint, inverse: c_int, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_ntt_dist_stage(ctx: *mut Dg16Ctx, curve: c_int, log_n: c_uint, rank: c_uint, n_ranks: c_uint,
                               inverse: c_int, stage: c_int, input: *const c_void, out: *mut c_void, flags: c_uint,
                               channel: c_int) -> c_int;
    pub fn dg16_h_poly_dist(ctx: *mut Dg16Ctx, curve: c_int, comm: *const Dg16Comm, a_rows: *const c_void,
                            b_rows: *const c_void, c_rows: *const c_void, log_m: c_uint, out: *mut c_void,
                            flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_h_poly_dist_stage(ctx: *mut Dg16Ctx, curve: c_int, log_m: c_uint, rank: c_uint, n_ranks: c_uint,
                                  stage: c_int, inputs: *const *const c_void, out: *mut c_void, flags: c_uint,
                                  channel: c_int) -> c_int;
    pub fn dg16_groth16_prove_dist(ctx: *mut Dg16Ctx, pk: *const Dg16Pk, comm: *const Dg16Comm,
                                   a_rows: *const c_void, b_rows: *const c_void, c_rows: *const c_void,
                                   w: *const c_void, r_s: *const c_void, flags: c_uint, proof_out: *mut c_void) -> c_int;
    pub fn dg16_rccl_unique_id(out128: *mut c_void) -> c_int;
    pub fn dg16_rccl_create(ctx: *mut Dg16Ctx, unique_id128: *const c_void, n_ranks: c_uint, rank: c_uint,
                            out: *mut *mut Dg16Rccl) -> c_int;
    pub fn dg16_rccl_ranks(h: *mut Dg16Rccl, n_ranks: *mut c_uint, rank: *mut c_uint) -> c_int;  // ncclCommCount / UserRank
    pub fn dg16_rccl_channels_split(h: *mut Dg16Rccl) -> c_int;   // 1: channels 1, 2 are ncclCommSplit duplicates
    pub fn dg16_rccl_comm(h: *mut Dg16Rccl) -> *const Dg16Comm;
    pub fn dg16_rccl_net(h: *mut Dg16Rccl) -> *const Dg16Net;
    pub fn dg16_rccl_destroy(h: *mut Dg16Rccl);
    pub fn dg16_rccl_error() -> *const c_char;
    // in-process net (LocalTestNet), packed secret sharing
    pub fn dg16_localnet_create(n_parties: c_uint, out: *mut *mut Dg16LocalNet) -> c_int;
    pub fn dg16_localnet_party(net: *mut Dg16LocalNet, id: c_uint) -> *const Dg16Net;
    pub fn dg16_localnet_destroy(net: *mut Dg16LocalNet);
    pub fn dg16_localnet_abort(net: *mut Dg16LocalNet);
    pub fn dg16_localnet_reset(net: *mut Dg16LocalNet, timeout_s: c_uint);
    pub fn dg16_pss_destroy(pp: *mut Dg16Pss);
    pub fn dg16_pss_apply(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, which: c_int, input: *const c_void, count: usize,
                          out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_pss_apply_exp(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, group: c_int, which: c_int,
                              input: *const c_void, count: usize, out: *mut c_void, flags: c_uint,
                              channel: c_int) -> c_int;
    // file formats, proof codec, verifier (host code of the library)
    pub fn dg16_io_error() -> *const c_char;
    pub fn dg16_r1cs_parse(data: *const c_void, bytes: usize, out: *mut *mut Dg16R1cs) -> c_int;
    pub fn dg16_r1cs_header_get(f: *const Dg16R1cs, out: *mut Dg16R1csHeader) -> c_int;
    pub fn dg16_r1cs_matrix(f: *const Dg16R1cs, which: c_int, out: *mut Dg16Csr) -> c_int;
    pub fn dg16_r1cs_wire_map(f: *const Dg16R1cs, map: *mut *const u64) -> c_int;
    pub fn dg16_r1cs_free(f: *mut Dg16R1cs);
    pub fn dg16_zkey_parse(data: *const c_void, bytes: usize, out: *mut *mut Dg16Zkey) -> c_int;
    pub fn dg16_zkey_header_get(z: *const Dg16Zkey, out: *mut Dg16ZkeyHeader) -> c_int;
    pub fn dg16_zkey_points(z: *const Dg16Zkey, which: c_int, ptr: *mut *const c_void, count: *mut usize) -> c_int;
    pub fn dg16_zkey_matrix(z: *const Dg16Zkey, which: c_int, out: *mut Dg16Csr) -> c_int;
    pub fn dg16_zkey_free(z: *mut Dg16Zkey);
    pub fn dg16_serialize_error() -> *const c_char;
    pub fn dg16_proof_compress(curve: c_int, proof_jacobian: *const c_void, out128: *mut c_void) -> c_int;
    pub fn dg16_proof_decompress(curve: c_int, in128: *const c_void, validate: c_int, proof_affine: *mut c_void) -> c_int;
    pub fn dg16_codec_error() -> *const c_char;
    pub fn dg16_arkkey_layout(data: *const c_void, bytes: usize, verifying_key_only: c_int, out: *mut Dg16ArkKeyLayout) -> c_int;
    pub fn dg16_points_compress(ctx: *mut Dg16Ctx, curve: c_int, group: c_int, affine: *const c_void, n: usize,
                                out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_points_decompress(ctx: *mut Dg16Ctx, curve: c_int, group: c_int, input: *const c_void, n: usize,
                                  validate: c_int, affine_out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_wire_fr_bytes(n: usize) -> usize;                                     // 8 + 32 n
    pub fn dg16_wire_fr_encode(ctx: *mut Dg16Ctx, curve: c_int, mont: *const c_void, n: usize, out: *mut c_void,
                               flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_wire_fr_decode(ctx: *mut Dg16Ctx, curve: c_int, input: *const c_void, bytes: usize,
                               out_mont: *mut c_void, n_out: *mut usize, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_verify_error() -> *const c_char;
    pub fn dg16_groth16_verify(curve: c_int, alpha_g1: *const c_void, beta_g2: *const c_void, gamma_g2: *const c_void,
                               delta_g2: *const c_void, ic: *const c_void, n_ic: usize, public_inputs: *const c_void,
                               n_public: usize, proof_affine: *const c_void, flags: c_uint, accepted: *mut c_int) -> c_int;
    // ---- dist-primitives over an MpcNet (packed secret sharing): d_fft, d_msm, d_pp, deg_red, ext_wit::h, prove::A/B/C ----
    pub fn dg16_pss_create(ctx: *mut Dg16Ctx, curve: c_int, l: c_uint, out: *mut *mut Dg16Pss) -> c_int;
    pub fn dg16_d_fft(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, share: *const c_void,
                      share_len: usize, log_m: c_uint, rearrange: c_int, pad: c_uint, degree2: c_int,
                      inverse: c_int, out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_d_msm(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, group: c_int,
                      bases: *const c_void, scalars: *const c_void, n_bases: usize, n_scalars: usize,
                      flags: c_uint, channel: c_int, out: *mut c_void) -> c_int;
    pub fn dg16_d_pp(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, num: *const c_void,
                     den: *const c_void, count: usize, out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_deg_red(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, px: *const c_void,
                        count: usize, out: *mut c_void, flags: c_uint, channel: c_int) -> c_int;
    pub fn dg16_ext_wit_h(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, a: *const c_void,
                          b: *const c_void, c: *const c_void, log_m: c_uint, out: *mut c_void, flags: c_uint) -> c_int;
    // prove::A / B / C::compute (groth16/src/prove.rs:21-46, 62-85, 106-136)
    pub fn dg16_prove_a(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, l: *const c_void, n: *const c_void,
                        r: *const c_void, s_bases: *const c_void, a: *const c_void, n_s: usize, n_a: usize,
                        flags: c_uint, channel: c_int, out: *mut c_void) -> c_int;
    pub fn dg16_prove_b(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, z: *const c_void, k: *const c_void,
                        s: *const c_void, v_bases: *const c_void, a: *const c_void, n_v: usize, n_a: usize,
                        flags: c_uint, channel: c_int, out: *mut c_void) -> c_int;
    pub fn dg16_prove_c(ctx: *mut Dg16Ctx, pp: *const Dg16Pss, net: *const Dg16Net, a_point: *const c_void,
                        m: *const c_void, s: *const c_void, r: *const c_void, w_bases: *const c_void,
                        ax: *const c_void, n_w: usize, n_ax: usize, u_bases: *const c_void, h: *const c_void,
                        n_u: usize, n_h: usize, h_bases: *const c_void, a: *const c_void, n_hb: usize, n_a: usize,
                        flags: c_uint, out: *mut c_void) -> c_int;
}
