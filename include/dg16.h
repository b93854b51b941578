/* dg16 -- C ABI of the MI355X (gfx950) Groth16 hot path: MSM, NTT, h-polynomial, prover glue.
 *
 * This is the drop-in boundary for zkHubHQ/distributed-groth16.  The reference has no FFI; the
 * seams a maintainer re-points at this library are Rust call sites (INTEGRATION.md shows the
 * `extern "C"` block and the feature-gated replacements):
 *
 *   dg16_msm            <- `G::msm(bases, scalars)`            dist-primitives/src/dmsm/mod.rs:82
 *                          (also examples/msm_bench.rs:20, groth16/examples/local_groth_bench.rs:139-147)
 *   dg16_ntt            <- `EvaluationDomain::{fft,ifft}_in_place` ark-circom/src/circom/qap.rs:64-85
 *                          and the butterfly loops fft1/fft2_in_place  dist-primitives/src/dfft/mod.rs:98-182
 *   dg16_h_poly         <- `CircomReduction::witness_map_from_matrices` (NTT part)
 *                                                              ark-circom/src/circom/qap.rs:64-91
 *   dg16_qap            <- `qap::qap` (R1CS x witness)          groth16/src/qap.rs:44-91
 *   dg16_field_op       <- element-wise ark-ff ops (parity probe for the Montgomery kernels)
 *   dg16_gen_bases      <- `PackedProvingKeyShare::rand`       groth16/src/proving_key.rs:112-155
 *   dg16_groth16_prove  <- `create_proof_with_reduction_and_matrices`  groth16/examples/sha256.rs:159
 *                          and the A/B/C assembly of groth16/src/prove.rs:21-136
 *
 * Conventions
 *   - Field elements: little-endian limbs, 32 bytes (Fr of all curves, BN254 Fq) or 48 bytes
 *     (BLS12-381 / BLS12-377 Fq), Montgomery form with R = 2^256 / 2^384 -- the arkworks in-memory
 *     representation (pinned by ark-circom/src/zkey.rs:417-427).
 *   - G1 affine point: x || y (64 / 96 bytes).  G2 affine: x.c0 || x.c1 || y.c0 || y.c1 (128 / 192
 *     bytes).  Identity = all-zero bytes (ark-circom/src/zkey.rs:353-361).  No `infinity` flag
 *     byte: the Rust shim repacks `Affine<P>`.
 *   - Results of group operations are Jacobian (x, y, z) in Montgomery form (== ark-ec
 *     `Projective` for short-Weierstrass curves); z = 0 encodes the identity.
 *   - Scalars: 32 bytes each, canonical integers unless DG16_F_SCALARS_MONT is set.
 *   - Pointers are host pointers unless DG16_F_DEVICE_PTRS is set (then ALL data pointers of the
 *     call, inputs and outputs, are device pointers on the context's GPU).
 *   - `channel` (0..2) mirrors `MultiplexedStreamID::{Zero,One,Two}` (mpc-net/src/lib.rs:29-33):
 *     each channel owns a HIP stream and a workspace; calls on different channels may run
 *     concurrently from different host threads, calls on one channel are serialised.
 *   - Every function returns a dg16_status; nothing throws or aborts across this boundary.  The
 *     Rust side maps non-zero to `MpcNetError::Generic` (mpc-net/src/lib.rs:15-27).
 *   - The library owns device memory; the caller owns every buffer it passes and may free it on
 *     return (host-pointer calls are synchronous; device-pointer calls are stream-ordered on the
 *     channel's stream -- use dg16_sync or the stream you installed with dg16_set_stream).
 */
#ifndef DG16_H
#define DG16_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dg16_ctx dg16_ctx;

enum dg16_curve { DG16_BN254 = 0, DG16_BLS12_381 = 1, DG16_BLS12_377 = 2 };

enum dg16_status {
  DG16_OK = 0,
  DG16_ERR_LENGTH_MISMATCH = 1, /* mirrors Err(usize) of VariableBaseMSM::msm */
  DG16_ERR_BAD_CURVE = 2,
  DG16_ERR_BAD_ARG = 3,
  DG16_ERR_OOM = 4,
  DG16_ERR_HIP = 5,
  DG16_ERR_NET = 6,
  DG16_ERR_UNSUPPORTED = 7
};

enum dg16_flags {
  DG16_F_SCALARS_MONT = 1u, /* scalars are in Montgomery form (arkworks memory) */
  DG16_F_DEVICE_PTRS = 2u,  /* all data pointers are device pointers */
  DG16_F_OUT_AFFINE = 4u,   /* group result as affine x || y (one inversion on the device) */
  DG16_F_H_CYCLIC = 8u,     /* dg16_pk_create_shard: this shard's h_query bases are h_query[shard + n_shards * j]
                               (the output layout of the sharded h-polynomial) instead of a contiguous slice */
  DG16_F_SERIAL_CHANNELS = 16u, /* dg16_prove_c: run the three d_msm one after another (channel 0, 1, 2 in that order
                               on every party) instead of joined from three host threads -- for a dg16_net whose
                               channels are not independent (one ordered pipe); the result is the same */
  DG16_F_OVERLAP_TAIL = 32u, /* dg16_groth16_prove and dg16_groth16_prove_dist with DG16_F_DEVICE_PTRS, for a queue of proofs on
                               one context (prove_dist: also the all-gather of the records, on channel 2's stream): the last
                               MSM's bucket reduction, the assembly and the copy to proof_out are ordered on CHANNEL 2's
                               stream instead of channel 0's, so the work enqueued next on channel 0 (the next proof's
                               dg16_qap and h-polynomial) starts under that latency-bound tail.  proof_out is complete
                               after dg16_sync(ctx, 2) (or stream-ordered work on channel 2); every other call on the
                               context orders itself behind the tail.  Ignored with host pointers. */
  DG16_F_BASES_IN_SUBGROUP = 64u /* dg16_msm, dg16_d_msm, dg16_prove_a / _b / _c: the caller guarantees that every base is
                               in the order-r subgroup (true of any arkworks G1Affine / G2Affine obtained through
                               Validate::Yes or from CRS generation).  The library may then split the scalars with the
                               curve's endomorphism (GLV: phi(P) = lambda P holds only in that subgroup), which is what
                               its headline MSM rates are measured with.  Without the flag only cofactor-one groups
                               (BN254 G1) take that path and every other group runs plain Pippenger, which -- like
                               VariableBaseMSM::msm -- is correct for ANY point of the curve.  Resident keys / tables
                               (dg16_pk_create*, dg16_bases_upload) never split and ignore the flag. */
};

/* field ids for dg16_field_op: curve for the base field Fq, 16 + curve for the scalar field Fr */
enum dg16_field_opcode {
  DG16_OP_ADD = 0, DG16_OP_SUB = 1, DG16_OP_MUL = 2, DG16_OP_SQR = 3, DG16_OP_INV = 4,
  DG16_OP_TO_MONT = 5, DG16_OP_FROM_MONT = 6, DG16_OP_NEG = 7
};

/* ---- context ------------------------------------------------------------------------------- */
int dg16_ctx_create(int device, dg16_ctx **out);
void dg16_ctx_destroy(dg16_ctx *ctx);
const char *dg16_last_error(dg16_ctx *ctx);
/* Run `channel` on a caller-owned hipStream_t (e.g. torch's current stream). NULL restores the
 * context's own stream. */
int dg16_set_stream(dg16_ctx *ctx, int channel, void *hip_stream);
int dg16_sync(dg16_ctx *ctx, int channel);
/* Device name, CU count (for reports). */
int dg16_device_info(dg16_ctx *ctx, char *name, size_t name_len, int *compute_units);

/* ---- element-wise field arithmetic (parity probe) -------------------------------------------- */
int dg16_field_op(dg16_ctx *ctx, int field_id, int op, const void *a, const void *b, void *out,
                  size_t n, unsigned flags, int channel);

/* ---- NTT ---------------------------------------------------------------------------------------
 * In-place Radix2EvaluationDomain transform of 2^log_n Montgomery-form Fr elements, natural order
 * in and out.  inverse != 0 scales by n^-1.  coset_offset: NULL, or a HOST pointer to the domain
 * offset g (Montgomery form): forward multiplies coefficient i by g^i first, inverse multiplies
 * output i by g^-i. */
int dg16_ntt(dg16_ctx *ctx, int curve, void *data, unsigned log_n, int inverse,
             const void *coset_offset, unsigned flags, int channel);

/* h = NTT(shift(iNTT a)) * NTT(shift(iNTT b)) - NTT(shift(iNTT c)) on the size-2^log_m domain,
 * shift = multiply coefficient i by w_{2m}^i.  a, b, c are not modified; out may alias a. */
int dg16_h_poly(dg16_ctx *ctx, int curve, const void *a, const void *b, const void *c,
                unsigned log_m, void *out, unsigned flags, int channel);

/* ---- QAP evaluation vectors (sparse R1CS x witness) ------------------------------------------------
 * Replaces `qap::qap` (groth16/src/qap.rs:44-91; the same loops open ark-circom/src/circom/qap.rs:34-62):
 * a[i] = <A_i, w>, b[i] = <B_i, w> for the num_constraints CSR rows (coefficients in Montgomery form),
 * a[num_constraints + j] = w[j] for j < num_inputs, c = a o b on the constraint rows, zero padding up to
 * 2^log_m = D::new(num_constraints + num_inputs).size().  full_assignment: num_vars Fr elements
 * (Montgomery iff DG16_F_SCALARS_MONT); outputs: 2^log_m Montgomery elements each. */
int dg16_qap(dg16_ctx *ctx, int curve, size_t num_constraints, size_t num_inputs, size_t num_vars,
             unsigned log_m, const uint32_t *a_row_ptr, const uint32_t *a_col, const void *a_coeff,
             const uint32_t *b_row_ptr, const uint32_t *b_col, const void *b_coeff,
             const void *full_assignment, void *a_out, void *b_out, void *c_out, unsigned flags,
             int channel);

/* The same for the rows i = row_start + row_stride * j, j < 2^log_m / row_stride, written densely (a_out[j] is row
 * i): rank `row_start` of `row_stride` ranks computes exactly the cyclic rows the sharded h-polynomial consumes. */
int dg16_qap_rows(dg16_ctx *ctx, int curve, size_t num_constraints, size_t num_inputs, size_t num_vars,
                  unsigned log_m, const uint32_t *a_row_ptr, const uint32_t *a_col, const void *a_coeff,
                  const uint32_t *b_row_ptr, const uint32_t *b_col, const void *b_coeff,
                  const void *full_assignment, size_t row_start, size_t row_stride, void *a_out, void *b_out,
                  void *c_out, unsigned flags, int channel);

/* ---- MSM ----------------------------------------------------------------------------------------
 * out = sum_i scalars[i] * bases[i] in G1 (group = 1) or G2 (group = 2).
 * n_bases != n_scalars returns DG16_ERR_LENGTH_MISMATCH.  out: Jacobian (3 field elements of the
 * group's coordinate field) or affine with DG16_F_OUT_AFFINE.  Scalars are integers below 2^255 (canonical field
 * elements are; a set bit 255 is not part of the contract).  DG16_F_BASES_IN_SUBGROUP: see enum dg16_flags. */
int dg16_msm(dg16_ctx *ctx, int curve, int group, const void *bases, const void *scalars,
             size_t n_bases, size_t n_scalars, unsigned flags, int channel, void *out);

/* Resident bases (bases_upload -> handle, msm_resident): a CRS is fixed, so its bases go to HBM once,
 * as the table of window multiples T[w][i] = 2^(c w) P_i (288 GB of HBM: 15 rows x 64 B x 2^20 = 1 GB per G1
 * vector) -- an MSM over them then has ONE bucket set and no Horner tail (2^20 points: ~2 ms G1 / ~6 ms G2 against
 * ~4.7 / ~14 for dg16_msm).  This is what `PackedProvingKeyShare` (groth16/src/proving_key.rs:26-46) holds per party:
 * dg16_d_msm_resident is d_msm (dist-primitives/src/dmsm/mod.rs:70-98) over such a handle.
 * n_scalars != n returns DG16_ERR_LENGTH_MISMATCH like dg16_msm. */
typedef struct dg16_bases dg16_bases;
int dg16_bases_upload(dg16_ctx *ctx, int curve, int group, const void *bases, size_t n, unsigned flags,
                      dg16_bases **out);
void dg16_bases_free(dg16_bases *h);
int dg16_bases_info(const dg16_bases *h, size_t *n, unsigned *window_bits, uint64_t *table_bytes);
int dg16_msm_resident(dg16_ctx *ctx, const dg16_bases *h, const void *scalars, size_t n_scalars, unsigned flags,
                      int channel, void *out);

/* Synthetic bases P_i = (k0 + i*k1) * G (distinct, prime-order subgroup), written as affine points
 * to `out` (device or host per flags). */
int dg16_gen_bases(dg16_ctx *ctx, int curve, int group, uint64_t seed, size_t n, void *out,
                   unsigned flags, int channel);

/* Jacobian -> affine for n points (n inversions on the device, one thread each). */
int dg16_to_affine(dg16_ctx *ctx, int curve, int group, const void *jac, void *out, size_t n,
                   unsigned flags, int channel);

/* ---- Groth16 prover (single prover; the value the n-party run must equal) ------------------------
 * Replaces `Groth16::<E, CircomReduction>::create_proof_with_reduction_and_matrices` (third-party
 * fork; call sites groth16/examples/sha256.rs:159, mpc-api/src/main.rs:393) from the point where
 * the QAP evaluation vectors exist (groth16/src/qap.rs:44-91 produces a, b, c).
 *
 * dg16_pk_create makes the proving key resident in HBM.  Queries are the arkworks `ProvingKey`
 * vectors, element 0 included: a_query, b_g1_query, b_g2_query have num_vars entries, l_query has
 * num_vars - num_inputs, h_query has domain_size entries (CircomReduction::h_query_scalars,
 * ark-circom/src/circom/qap.rs:94-110).  fixed_points = alpha_g1 | beta_g1 | delta_g1 (G1 affine)
 * | beta_g2 | delta_g2 (G2 affine), contiguous.  The base-vector mapping follows
 * groth16/src/proving_key.rs:48-65.  flags: DG16_F_DEVICE_PTRS if every pointer is a device pointer. */
/* HBM budget, in bytes, of the window tables of ONE resident key (dg16_pk_create*: the five tables together) or ONE
 * base set (dg16_bases_upload) built on this context from now on; 0 (the default) = unlimited.  A full table has one
 * row T[w] = 2^(c w) P per c-bit window (BN254, 2^20 wires, c = 17: 6.0 GB per key; BLS12-381 at 2^24: 126 GB).  Under a
 * budget the tables keep every k-th row, the smallest k that fits: the MSMs then run k bucket sets joined by a
 * Horner tail of (k - 1) c doublings -- same results, more bucket reductions.  k = W degenerates to the plain bases. */
int dg16_ctx_set_table_budget(dg16_ctx *ctx, uint64_t bytes);
typedef struct dg16_pk dg16_pk;
int dg16_pk_create(dg16_ctx *ctx, int curve, size_t num_vars, size_t num_inputs, size_t domain_size,
                   const void *a_query, const void *b_g1_query, const void *b_g2_query,
                   const void *h_query, const void *l_query, const void *fixed_points, unsigned flags,
                   dg16_pk **out);
void dg16_pk_destroy(dg16_pk *pk);

/* What a resident key holds (for reports: bench.py prints it next to every timing, because the proof time depends
 * on the window tables built here, once per key).  n_*: points per MSM launch of this shard (slice + delta slots);
 * c_*: window bits of the tables; table_bytes: HBM held by the five tables.  (Facts of the key only: measured
 * rates and instruction counts of the kernels live with the benchmarks -- profiles/, bench.py -- not in the ABI.) */
typedef struct dg16_pk_info {
  uint64_t n_ab, n_l, n_h;
  uint32_t c_ab, c_l, c_h;
  uint32_t shard, n_shards;
  uint64_t table_bytes;
  uint32_t table_stride; /* 1: one table row per window; k > 1: every k-th row kept (dg16_ctx_set_table_budget) */
} dg16_pk_info;
int dg16_pk_info_get(const dg16_pk *pk, dg16_pk_info *out);

/* Multi-GPU form: the key holds slice `shard` of `n_shards` of every MSM range (contiguous slices of
 * a_query[1..], b_g1_query[1..], b_g2_query[1..]; the l_query elements of the same wires -- L shares the digit sort
 * of A / B1 / B; a contiguous slice of h_query, or with DG16_F_H_CYCLIC the elements shard + n_shards * j); the
 * delta pairs ride on the last shard.  Pass the FULL queries; only the slice is copied to the device. */
int dg16_pk_create_shard(dg16_ctx *ctx, int curve, size_t num_vars, size_t num_inputs,
                         size_t domain_size, const void *a_query, const void *b_g1_query,
                         const void *b_g2_query, const void *h_query, const void *l_query,
                         const void *fixed_points, unsigned shard, unsigned n_shards, unsigned flags,
                         dg16_pk **out);

/* a, b, c: QAP evaluation vectors (domain_size Montgomery Fr elements each); full_assignment:
 * num_vars Fr elements [1, public.., witness..] (Montgomery iff DG16_F_SCALARS_MONT); r_s: HOST
 * pointer to r || s (2 x 32 bytes, same form as the assignment).  proof_out: A (G1 Jacobian) |
 * B (G2 Jacobian) | C (G1 Jacobian).  Uses all three channels. */
int dg16_groth16_prove(dg16_ctx *ctx, const dg16_pk *pk, const void *a, const void *b, const void *c,
                       const void *full_assignment, const void *r_s, unsigned flags, void *proof_out);

/* The two halves of dg16_groth16_prove, for one-process-per-GPU runs:
 *   dg16_groth16_msms      h-polynomial + this shard's five MSMs -> results record
 *                          (A, B1, L, H as G1 Jacobian, then B as G2 Jacobian; dg16_groth16_results_bytes)
 *   <all-gather of the records over RCCL, done by the caller>
 *   dg16_groth16_assemble  per-MSM sum of the n_shards records + the A/B/C assembly of prove.rs:21-136
 * This is d_msm's "gather to king, sum, broadcast" (dist-primitives/src/dmsm/mod.rs:88-97) with the
 * sum done on every rank. */
size_t dg16_groth16_results_bytes(int curve);
int dg16_groth16_msms(dg16_ctx *ctx, const dg16_pk *pk, const void *a, const void *b, const void *c,
                      const void *full_assignment, const void *r_s, unsigned flags, void *results_out);
int dg16_groth16_assemble(dg16_ctx *ctx, const dg16_pk *pk, const void *gathered_results,
                          size_t n_shards, const void *r_s, unsigned flags, void *proof_out);

/* ---- one process per GPU: collectives, sharded h-polynomial, distributed prove ----------------------------------
 * The king / client exchange of the reference (mpc-net/src/lib.rs:61-140 under dist-primitives/src/channel/mod.rs:8-57)
 * re-mapped to the GPUs of one node: `dg16_comm` is what the prover needs from a transport, dg16_rccl_* is the
 * native implementation (RCCL grouped send / recv on device buffers over xGMI, stream-ordered: no host
 * synchronisation on the data path).  A comm may also be implemented by the caller (tests drive these entry points
 * with a torch.distributed / gloo-backed comm).
 *
 *   all_gather   every rank contributes `bytes`; recv_dev gets n_ranks * bytes ordered by rank
 *                (d_msm's "gather to king, sum, send back" with the sum on every rank, dmsm/mod.rs:88-97)
 *   all_to_all   bytes_per_peer bytes at send_dev + p * bytes_per_peer go to rank p and land at
 *                recv_dev + me * bytes_per_peer there (the butterfly exchange of the sharded NTT; the reference
 *                does gather -> fft2_in_place -> scatter through the king, dfft/mod.rs:185-256)
 * Both are ordered on `hip_stream`: the payload was produced on it, the result may be consumed on it. */
typedef struct dg16_comm {
  void *self;
  unsigned (*n_ranks)(void *self);
  unsigned (*rank)(void *self);
  int (*all_gather)(void *self, const void *send_dev, size_t bytes, void *recv_dev, void *hip_stream);
  int (*all_to_all)(void *self, const void *send_dev, void *recv_dev, size_t bytes_per_peer, void *hip_stream);
} dg16_comm;

/* ONE 2^log_n-point transform over n_ranks GPUs (2, 4 or 8; 2^log_n >= n_ranks^2) with ONE all-to-all -- the "all-to-all
 * of NTT butterflies": what d_fft / d_ifft (dist-primitives/src/dfft/mod.rs:17-95) do through the king (local levels,
 * gather, remaining levels, scatter), as a four-step transform without a king.  in: this rank's CYCLIC elements
 * x[n_ranks * j + rank], j < M = 2^log_n / n_ranks.  out (M elements, may not alias in): the transposed layout of a
 * four-step FFT, out[k1 * S + j] = X[M * k1 + rank * S + j], S = M / n_ranks, k1 < n_ranks (rank sigma ends with slice
 * sigma of every length-M block of the natural-order output).  inverse != 0: inverse root and the 1 / 2^log_n scale.
 * dg16_ntt_dist_stage: the two local stages for callers that run the exchange themselves (stage 0: in -> send buffer,
 * natural order = n_ranks pieces of S; stage 1: receive buffer -> out). */
int dg16_ntt_dist(dg16_ctx *ctx, int curve, const dg16_comm *comm, const void *in, void *out, unsigned log_n,
                  int inverse, unsigned flags, int channel);
int dg16_ntt_dist_stage(dg16_ctx *ctx, int curve, unsigned log_n, unsigned rank, unsigned n_ranks, int inverse,
                        int stage, const void *in, void *out, unsigned flags, int channel);

/* h-polynomial over n_ranks GPUs (2, 4 or 8; 2^log_m >= n_ranks^2).  a, b, c: this rank's CYCLIC rows of the QAP
 * evaluation vectors, a[n_ranks * j + rank], j < 2^log_m / n_ranks (dg16_qap_rows); out: h[rank + n_ranks * j] --
 * the scalars of a DG16_F_H_CYCLIC key shard.  Two all-to-alls of 3 * 32 * 2^log_m / n_ranks bytes per rank.
 * Device pointers only (DG16_F_DEVICE_PTRS must be set). */
int dg16_h_poly_dist(dg16_ctx *ctx, int curve, const dg16_comm *comm, const void *a_rows, const void *b_rows,
                     const void *c_rows, unsigned log_m, void *out, unsigned flags, int channel);
/* The three local stages of dg16_h_poly_dist, for callers that run the exchanges themselves:
 *   stage 0  in = {a_rows, b_rows, c_rows}         out = send buffer 1  [peer][vector][S], S = 2^log_m / n_ranks^2
 *   stage 1  in = {receive buffer 1}               out = send buffer 2  [peer][vector][S]
 *   stage 2  in = {receive buffer 2}               out = h[rank + n_ranks * j]
 * Buffers hold 3 * 2^log_m / n_ranks elements. */
int dg16_h_poly_dist_stage(dg16_ctx *ctx, int curve, unsigned log_m, unsigned rank, unsigned n_ranks, int stage,
                           const void *const *in, void *out, unsigned flags, int channel);

/* dg16_groth16_msms with the h-polynomial already made (h_shard: the n_h scalars of this key's h slice, Montgomery
 * form, device pointer or host per flags) -- the half of a distributed proof after the sharded h-polynomial. */
int dg16_groth16_msms_h(dg16_ctx *ctx, const dg16_pk *pk, const void *h_shard, const void *full_assignment,
                        const void *r_s, unsigned flags, void *results_out);

/* The whole distributed proof on this rank, stream-ordered: sharded h-polynomial (two all-to-alls), this shard's
 * five MSMs, one all-gather of the results records, assembly.  pk: a DG16_F_H_CYCLIC shard (rank = shard,
 * n_ranks = n_shards); a_rows, b_rows, c_rows as for dg16_h_poly_dist.  With comm == NULL (or one rank) and an
 * unsharded key this is dg16_groth16_prove.  Every rank receives the same proof. */
int dg16_groth16_prove_dist(dg16_ctx *ctx, const dg16_pk *pk, const dg16_comm *comm, const void *a_rows,
                            const void *b_rows, const void *c_rows, const void *full_assignment, const void *r_s,
                            unsigned flags, void *proof_out);

/* Native RCCL transport.  Rank 0 makes the 128-byte id (dg16_rccl_unique_id) and hands it to the other ranks out
 * of band (the launcher's rendezvous); every rank then calls dg16_rccl_create on its own context (blocking until all
 * ranks have joined).  dg16_rccl_comm serves dg16_h_poly_dist / dg16_groth16_prove_dist; dg16_rccl_net is the MpcNet
 * vtable below (king = rank 0), so that d_fft / d_msm / d_pp / ext_wit::h / prove::A,B,C run one party per GPU.
 * The handle holds THREE communicators, one per MultiplexedStreamID (mpc-net/src/lib.rs:29-33): channel c of the
 * vtable only ever touches communicator c, under that communicator's mutex, so the three d_msm that prove::C joins
 * (groth16/src/prove.rs:113-125; dg16_prove_c drives them from three host threads) cannot meet each other's
 * payloads whatever order the threads run in on each party.  Communicator 0 also carries the dg16_comm collectives.
 * Channels 1 and 2 are ncclCommSplit duplicates of the first communicator (dg16_rccl_channels_split() == 1) or, on a
 * librccl without that entry point, joined through two fresh ids broadcast over it.  librccl is bound at run time
 * (dlopen): without it these return DG16_ERR_UNSUPPORTED and the rest of the library is unaffected. */
typedef struct dg16_rccl dg16_rccl;
struct dg16_net;
int dg16_rccl_unique_id(void *out128);
int dg16_rccl_create(dg16_ctx *ctx, const void *unique_id128, unsigned n_ranks, unsigned rank, dg16_rccl **out);
/* rank count and this rank's index as the communicator itself reports them (ncclCommCount / ncclCommUserRank) */
int dg16_rccl_ranks(dg16_rccl *h, unsigned *n_ranks, unsigned *rank);
int dg16_rccl_channels_split(dg16_rccl *h);
const dg16_comm *dg16_rccl_comm(dg16_rccl *h);
const struct dg16_net *dg16_rccl_net(dg16_rccl *h);
void dg16_rccl_destroy(dg16_rccl *h);
const char *dg16_rccl_error(void);

/* ---- dist-primitives, literally (packed secret sharing over an MpcNet) -----------------------------
 * These mirror the reference's functions one to one; "party" = one caller (a host thread with its own
 * dg16_ctx for the in-process LocalNet, or one process per GPU with RCCL-backed callbacks).  All
 * payloads are device buffers.
 *
 *   dg16_pss_create        PackedSharingParams::new(l)                 secret-sharing/src/pss.rs:34-62
 *   dg16_pss_apply         pack_from_public / unpack / unpack2 (batched) pss.rs:86-148
 *   dg16_pss_apply_exp     packexp_from_public / unpackexp             dist-primitives/src/dmsm/mod.rs:7-68
 *   dg16_d_fft             d_fft (inverse = 0) / d_ifft (inverse = 1)  dist-primitives/src/dfft/mod.rs:17-95
 *                          incl. fft1_in_place :98-140, fft2_in_place :142-182,
 *                          fft2_with_rearrange_pad :185-256, fft_in_place_rearrange :258-271,
 *                          pack_vec / transpose utils/pack.rs:4-33
 *   dg16_d_msm             d_msm                                        dist-primitives/src/dmsm/mod.rs:70-98
 *   dg16_deg_red           deg_red                                      dist-primitives/src/utils/deg_red.rs:10-28
 *   dg16_d_pp              d_pp                                         dist-primitives/src/dpp/mod.rs:17-88
 *   dg16_ext_wit_h         ext_wit::h                                   groth16/src/ext_wit.rs:16-101
 *   dg16_prove_a / _b / _c prove::A / B / C::compute                    groth16/src/prove.rs:21-46, 62-85, 106-136
 *   dg16_net               MpcNet's provided gather / scatter           mpc-net/src/lib.rs:61-140
 *   dg16_localnet_*        LocalTestNet                                 mpc-net/src/multi.rs:227-329
 */
typedef struct dg16_pss dg16_pss;
typedef struct dg16_localnet dg16_localnet;

/* Transport vtable (MpcNet): both collectives are blocking and ordered per channel.  send/recv are
 * device pointers; `hip_stream` is the stream the payload was produced on (the callee synchronises or
 * orders on it).  gather: every party contributes `bytes`; on the king (party 0) `recv_dev` receives
 * n_parties * bytes ordered by party id (client_send_or_king_receive, lib.rs:61-99).  scatter: the king
 * provides n_parties * bytes in `send_dev`, every party receives its `bytes` (lib.rs:102-140; equal
 * lengths are enforced by the single `bytes`, as lib.rs:116-124 checks). */
typedef struct dg16_net {
  void *self;
  unsigned (*n_parties)(void *self);
  unsigned (*party_id)(void *self);
  int (*gather_to_king)(void *self, int channel, const void *send_dev, size_t bytes, void *recv_dev,
                        void *hip_stream);
  int (*scatter_from_king)(void *self, int channel, const void *send_dev, size_t bytes, void *recv_dev,
                           void *hip_stream);
  /* MpcNet's required methods (mpc-net/src/lib.rs:46-58): is_init, and the point-to-point pair the two provided
   * collectives above are written in terms of.  A send_to(peer) completes against the peer's recv_from(me) with
   * the same channel and the same `bytes` (a length mismatch is DG16_ERR_NET on both sides, like the reference's
   * framing error); both are ordered on `hip_stream`. */
  int (*is_init)(void *self);
  int (*send_to)(void *self, unsigned peer, int channel, const void *send_dev, size_t bytes, void *hip_stream);
  int (*recv_from)(void *self, unsigned peer, int channel, void *recv_dev, size_t bytes, void *hip_stream);
} dg16_net;

int dg16_localnet_create(unsigned n_parties, dg16_localnet **out);
const dg16_net *dg16_localnet_party(dg16_localnet *net, unsigned id);
void dg16_localnet_destroy(dg16_localnet *net);
/* A party that fails outside a collective aborts the net: pending and future collectives return
 * DG16_ERR_NET on every party ("Stream died", mpc-net/src/multi.rs:393) instead of waiting; a
 * collective that is not joined by all parties within the timeout aborts by itself. */
void dg16_localnet_abort(dg16_localnet *net);
void dg16_localnet_reset(dg16_localnet *net, unsigned timeout_s);

int dg16_pss_create(dg16_ctx *ctx, int curve, unsigned l, dg16_pss **out);
void dg16_pss_destroy(dg16_pss *pp);
/* which: 0 pack ([count][l] -> [count][n]), 1 unpack ([count][n] -> [count][l]), 2 unpack2 */
int dg16_pss_apply(dg16_ctx *ctx, const dg16_pss *pp, int which, const void *in, size_t count,
                   void *out, unsigned flags, int channel);
/* same on affine group elements ("in the exponent"); group 1 or 2 */
int dg16_pss_apply_exp(dg16_ctx *ctx, const dg16_pss *pp, int group, int which, const void *in,
                       size_t count, void *out, unsigned flags, int channel);

/* share: this party's share_len packed shares; share_len * l must equal 2^log_m (the reference's
 * debug assertion, dfft/mod.rs:31-37) else DG16_ERR_BAD_ARG; out: pad * share_len elements. */
int dg16_d_fft(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *share,
               size_t share_len, unsigned log_m, int rearrange, unsigned pad, int degree2, int inverse,
               void *out, unsigned flags, int channel);
/* bases / scalars: this party's share vectors; out: Jacobian point, identical on all parties. */
int dg16_d_msm(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, int group, const void *bases,
               const void *scalars, size_t n_bases, size_t n_scalars, unsigned flags, int channel,
               void *out);
/* d_msm with this party's base shares resident (dg16_bases_upload of the share vector). */
int dg16_d_msm_resident(dg16_ctx *ctx, const dg16_pss *pp, const struct dg16_net *net, const dg16_bases *bases,
                        const void *scalars, size_t n_scalars, unsigned flags, int channel, void *out);
int dg16_deg_red(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *px, size_t count,
                 void *out, unsigned flags, int channel);
int dg16_d_pp(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *num, const void *den,
              size_t count, void *out, unsigned flags, int channel);
/* a/b/c_share: this party's PackedQAPShare vectors (m/l each); out: m/l packed shares of h.
 * Uses channel 0 (the reference multiplexes channels 0..2 for the three transforms). */
int dg16_ext_wit_h(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *a_share,
                   const void *b_share, const void *c_share, unsigned log_m, void *out, unsigned flags);
/* prove::A::compute (groth16/src/prove.rs:21-46): out = L + N * r + d_msm(S, a) on `channel` (the reference's `sid`).
 * L, N: G1 affine, in the clear (the identity (0, 0) is arkworks' default, as groth16/examples/sha256.rs:46-48 passes
 * it); r: one scalar (Montgomery iff DG16_F_SCALARS_MONT, like a); S, a: this party's packed shares; out: G1
 * Jacobian (E::G1), identical on all parties.  n_S != n_a is DG16_ERR_LENGTH_MISMATCH (the Err of G::msm). */
int dg16_prove_a(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *L, const void *N,
                 const void *r, const void *S, const void *a, size_t n_S, size_t n_a, unsigned flags, int channel,
                 void *out);
/* prove::B::compute (prove.rs:62-85): out = Z + K * s + d_msm(V, a) in G2 (Z, K: G2 affine; out: G2 Jacobian). */
int dg16_prove_b(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *Z, const void *K,
                 const void *s, const void *V, const void *a, size_t n_V, size_t n_a, unsigned flags, int channel,
                 void *out);
/* prove::C::compute (prove.rs:106-136): out = w + u + A * s + M * r + h * r with w = d_msm(W, ax), u = d_msm(U, h),
 * h = d_msm(H, a) JOINED on channels 0 / 1 / 2 like the reference's tokio::try_join! (:113-125): the call drives the
 * three channels of `ctx` from three host threads at once (every party must do the same, so the three collectives
 * of a channel meet their peers).  A: G1 Jacobian (E::G1, the value A::compute returned); M: G1 affine. */
int dg16_prove_c(dg16_ctx *ctx, const dg16_pss *pp, const dg16_net *net, const void *A, const void *M,
                 const void *s, const void *r, const void *W, const void *ax, size_t n_W, size_t n_ax,
                 const void *U, const void *h, size_t n_U, size_t n_h, const void *H, const void *a, size_t n_H,
                 size_t n_a, unsigned flags, void *out);

/* ---- circom `.r1cs` / snarkjs `.zkey` readers (host side of the library; no GPU involved) ----------------
 *   dg16_r1cs_parse   <- R1CSFile::new              ark-circom/src/circom/r1cs_reader.rs:54-249
 *   dg16_zkey_parse   <- read_zkey / BinFile         ark-circom/src/zkey.rs:53-388
 * Same acceptance rules as the reference (magic, version 1, 32-byte fields, BN254 moduli, section sizes, wire 0).
 * On failure the status is non-zero and dg16_io_error() (thread-local) gives the reference's error text.
 * dg16_r1cs copies what it keeps.  dg16_zkey is ZERO-COPY for points -- a zkey stores x || y Montgomery limbs with
 * the identity as (0, 0), i.e. this library's base layout -- so `data` must outlive the handle and the pointers
 * of dg16_zkey_points go straight into dg16_pk_create.  Matrix coefficients of a zkey are value * R^2 as stored
 * (zkey.rs:332-337): one dg16_field_op(from_mont) gives the Montgomery values dg16_qap takes; the rows snarkjs
 * appends for the public inputs are dropped and num_constraints = max row - n_public (zkey.rs:176-198). */
typedef struct dg16_r1cs dg16_r1cs;
typedef struct dg16_zkey dg16_zkey;
typedef struct dg16_r1cs_header {
  uint32_t n_wires, n_pub_out, n_pub_in, n_prv_in, n_constraints, has_wire_map;
  uint64_t n_labels;
} dg16_r1cs_header;
typedef struct dg16_zkey_header {
  uint32_t n_vars, n_public, domain_size, num_constraints;
} dg16_zkey_header;
typedef struct dg16_csr {   /* owned by the handle; coeff: nnz x 32-byte little-endian field elements */
  uint64_t n_rows, nnz;
  const uint32_t *row_ptr, *col;
  const void *coeff;
} dg16_csr;
enum {
  DG16_ZKEY_ALPHA_G1 = 0, DG16_ZKEY_BETA_G1 = 1, DG16_ZKEY_BETA_G2 = 2, DG16_ZKEY_GAMMA_G2 = 3,
  DG16_ZKEY_DELTA_G1 = 4, DG16_ZKEY_DELTA_G2 = 5, DG16_ZKEY_IC = 6, DG16_ZKEY_A = 7, DG16_ZKEY_B1 = 8,
  DG16_ZKEY_B2 = 9, DG16_ZKEY_L = 10, DG16_ZKEY_H = 11
};
const char *dg16_io_error(void);
int dg16_r1cs_parse(const void *data, size_t bytes, dg16_r1cs **out);
int dg16_r1cs_header_get(const dg16_r1cs *f, dg16_r1cs_header *out);
int dg16_r1cs_matrix(const dg16_r1cs *f, int which /* 0 A, 1 B, 2 C; canonical coefficients */, dg16_csr *out);
int dg16_r1cs_wire_map(const dg16_r1cs *f, const uint64_t **map /* NULL when the file has none */);
void dg16_r1cs_free(dg16_r1cs *f);
int dg16_zkey_parse(const void *data, size_t bytes, dg16_zkey **out);
int dg16_zkey_header_get(const dg16_zkey *z, dg16_zkey_header *out);
int dg16_zkey_points(const dg16_zkey *z, int which /* DG16_ZKEY_* */, const void **ptr, size_t *count);
int dg16_zkey_matrix(const dg16_zkey *z, int which /* 0 A, 1 B; value * R^2 */, dg16_csr *out);
void dg16_zkey_free(dg16_zkey *z);


/* ---- arkworks compressed Proof<Bn254> (host side; no GPU involved) ---------------------------------------
 *   dg16_proof_compress    <- proof.serialize_compressed       mpc-api/src/main.rs:154-171 (proof.bin)
 *   dg16_proof_decompress  <- Proof::deserialize_compressed     zk-cli verify path, test-circuits/sha256/proof.bin
 * 128 bytes: A (G1, 32) || B (G2, 64) || C (G1, 32); little-endian x, flags in the two top bits of the last byte
 * (bit 7: y > -y, Fq2 compared on (c1, c0); bit 6: infinity).  compress takes the 12 field elements that
 * dg16_groth16_prove writes (A, B, C Jacobian, Montgomery limbs); decompress writes A.x A.y | B.x B.y | C.x C.y
 * (8 field elements, Montgomery limbs, identity = zeros) and, with validate != 0, checks like Validate::Yes:
 * reduced coordinates, on the curve, B in the order-r subgroup.  BN254 only (the reference's proof curve). */
const char *dg16_serialize_error(void);
int dg16_proof_compress(int curve, const void *proof_jacobian, void *out128);
int dg16_proof_decompress(int curve, const void *in128, int validate, void *proof_affine);

/* ---- arkworks compressed key files (BN254) and compressed points (BN254, BLS12-377, BLS12-381) ----------------------
 *   <- pk.serialize_with_mode(.., Compress::Yes) / ProvingKey::deserialize_with_mode(.., Compress::Yes, Validate::No)
 *      and the same for VerifyingKey                               mpc-api/src/main.rs:154-171, :459-512
 * dg16_arkkey_layout (host code) walks the container: struct field order of ark-groth16's derive, a u64 little-endian
 * length in front of every Vec; offsets are byte offsets into the file, counts are points.  The point sections are
 * then (de)compressed in batches ON THE GPU by dg16_points_compress / _decompress -- one lane per point running the
 * same routines as the proof.bin codec above (pinned by the reference's own proof.bin): affine x || y Montgomery limbs
 * with the identity as zeros on one side (the layout dg16_pk_create and dg16_bases_upload take), 32 (G1) / 64 (G2)
 * bytes per point on the other; for BLS12-377 (ark-bls12-377 uses the same default SWFlags encoding -- the group
 * elements the reference's d_msm tests put on the wire, dist-primitives/examples/dmsm_test.rs) 48 / 96 bytes, and
 * validate != 0 checks the subgroup for G1 as well (cofactor != 1).  BLS12-381 (BASELINE config 5's curve; not a
 * dependency of the reference): ark-bls12-381 0.4 overrides the format with the zcash / IETF encoding -- 48 / 96 bytes,
 * BIG-endian x (G2: x.c1 || x.c0), flags in the three top bits of the FIRST byte: 0x80 compressed (always written, and
 * required when reading), 0x40 infinity, 0x20 y is the larger of (y, -y).  In every form an infinity encoding must
 * carry x = 0 and no sign flag (stricter than arkworks 0.4, which ignores the rest of an infinity encoding).
 * dg16_points_decompress is synchronous: a coordinate that is not reduced, bad flags
 * or an x off the curve return DG16_ERR_BAD_ARG (dg16_codec_error names the first failing point), like the Err of
 * deserialize_with_mode; validate != 0 adds the order-r subgroup check for G2 (Validate::Yes). */
typedef struct dg16_arkkey_layout_t {
  uint64_t n_ic, n_a, n_b1, n_b2, n_h, n_l;
  uint64_t off_alpha_g1, off_beta_g2, off_gamma_g2, off_delta_g2, off_ic;              /* VerifyingKey */
  uint64_t off_beta_g1, off_delta_g1, off_a, off_b1, off_b2, off_h, off_l;             /* rest of ProvingKey */
  uint64_t bytes;
} dg16_arkkey_layout_t;
const char *dg16_codec_error(void);
int dg16_arkkey_layout(const void *data, size_t bytes, int verifying_key_only, dg16_arkkey_layout_t *out);
int dg16_points_compress(dg16_ctx *ctx, int curve, int group, const void *affine, size_t n, void *out,
                         unsigned flags, int channel);
int dg16_points_decompress(dg16_ctx *ctx, int curve, int group, const void *in, size_t n, int validate,
                           void *affine_out, unsigned flags, int channel);

/* Vec<F> as MpcSerNet puts it on the wire (`out.serialize_compressed`, dist-primitives/src/channel/mod.rs:14,49):
 * u64 little-endian length || canonical little-endian 32-byte elements.  Between GPUs of one node the payloads of
 * dg16_net stay in the HBM form; these two convert at the edge to a party that speaks ark-serialize (group elements:
 * dg16_points_compress above).  Decode checks the prefix against `bytes` and every element < r. */
size_t dg16_wire_fr_bytes(size_t n);
int dg16_wire_fr_encode(dg16_ctx *ctx, int curve, const void *mont, size_t n, void *out, unsigned flags, int channel);
int dg16_wire_fr_decode(dg16_ctx *ctx, int curve, const void *in, size_t bytes, void *out_mont, size_t *n_out,
                        unsigned flags, int channel);

/* ---- Groth16 verification, BN254 (host side; no GPU involved: four pairings) --------------------------------
 *   dg16_groth16_verify  <- Groth16::<Bn254>::verify_proof   groth16/examples/sha256.rs:228-254, mpc-api verify
 * e(A, B) = e(alpha, beta) e(IC_0 + sum x_i IC_i, gamma) e(C, delta).  Points are affine x || y Montgomery limbs
 * with the identity as zeros -- the layout of a zkey's header / IC section (dg16_zkey_points) and of
 * dg16_proof_decompress.  public_inputs: n_public scalars of 32 bytes, canonical or (DG16_F_SCALARS_MONT)
 * Montgomery.  n_ic != n_public + 1 returns DG16_ERR_LENGTH_MISMATCH (ark-groth16's MalformedVerifyingKey).
 * Validation (what arkworks' Validate::Yes deserialisation guarantees before verify_proof runs): a verifying-key
 * point with a non-reduced coordinate, off its curve or (G2) outside the order-r subgroup, or a public input >= r,
 * returns DG16_ERR_BAD_ARG (no input aliasing: x and x + r are not the same input); a PROOF point with a
 * non-reduced coordinate, off its curve, or B outside the subgroup is a rejection (*accepted = 0), not an error.
 * *accepted = 1 iff all checks pass and the equation holds. */
const char *dg16_verify_error(void);
int dg16_groth16_verify(int curve, const void *alpha_g1, const void *beta_g2, const void *gamma_g2,
                        const void *delta_g2, const void *ic, size_t n_ic, const void *public_inputs,
                        size_t n_public, const void *proof_affine, unsigned flags, int *accepted);

/* Duration in milliseconds of the dominant kernel(s) of the most recent call on `channel`
 * (HIP events recorded on the channel's stream); 0 if none.  which: 0 = whole call,
 * 1 = bucket accumulation (MSM) / butterfly passes (NTT); 2 = NOT a duration: the shader clock in MHz the chip held
 * under the bucket accumulation of `which = 1` (the kernel measures it: s_memtime against the 100-MHz s_memrealtime,
 * summed over its workgroups), 0 if that call had none. */
int dg16_last_kernel_ms(dg16_ctx *ctx, int channel, int which, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* DG16_H */
