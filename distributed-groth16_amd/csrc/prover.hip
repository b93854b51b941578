// C ABI of the Groth16 prover (include/dg16.h); the per-curve instantiations live in prover_<curve>.hip.
#include <stdio.h>
#include <string.h>

#include "pk.h"

namespace dg16 {
#define DECL_P(name)                                                                                                   \
  void pk_build_##name(dg16_ctx*, PkDev&, const void*, const void*, const void*, const void*, const void*, const void*, bool); \
  void prove_##name(dg16_ctx*, const PkDev&, const void*, const void*, const void*, const void*, const void*, bool, bool, void*, \
                    bool); \
  void msms_##name(dg16_ctx*, Call&, Call&, Call&, const PkDev&, const void*, const void*, const void*, const void*,    \
                   const void*, bool, bool, uint8_t*, const dg16_comm*, const void*);                                   \
  void prove_dist_##name(dg16_ctx*, const PkDev&, const dg16_comm*, const void*, const void*, const void*, const void*, \
                         const void*, bool, bool, void*, bool);                                                         \
  void assemble_##name(Call&, const uint8_t*, size_t, uint8_t*);                                                        \
  size_t results_bytes_##name();                                                                                        \
  size_t proof_bytes_##name();
DECL_P(bn254) DECL_P(bls12_381)
}  // namespace dg16


using namespace dg16;

extern "C" {

int dg16_pk_create(dg16_ctx* ctx, int curve, size_t num_vars, size_t num_inputs, size_t domain_size,
                   const void* a_query, const void* b_g1_query, const void* b_g2_query, const void* h_query,
                   const void* l_query, const void* fixed_points, unsigned flags, dg16_pk** out) {
  return dg16_pk_create_shard(ctx, curve, num_vars, num_inputs, domain_size, a_query, b_g1_query, b_g2_query,
                              h_query, l_query, fixed_points, 0, 1, flags, out);
}

int dg16_pk_create_shard(dg16_ctx* ctx, int curve, size_t num_vars, size_t num_inputs, size_t domain_size,
                         const void* a_query, const void* b_g1_query, const void* b_g2_query,
                         const void* h_query, const void* l_query, const void* fixed_points, unsigned shard,
                         unsigned n_shards, unsigned flags, dg16_pk** out) {
  if (!ctx || !out) return DG16_ERR_BAD_ARG;
  *out = nullptr;
  dg16_pk* pk = new dg16_pk{ctx, {}};
  int rc = guarded(ctx, [&] {
    DG_REQUIRE(curve == DG16_BN254 || curve == DG16_BLS12_381, DG16_ERR_BAD_CURVE,
               "Groth16 needs G2: BN254 or BLS12-381");
    DG_REQUIRE(num_vars >= 2 && num_inputs >= 1 && num_inputs <= num_vars, DG16_ERR_BAD_ARG, "bad variable counts");
    DG_REQUIRE(domain_size && !(domain_size & (domain_size - 1)), DG16_ERR_BAD_ARG, "domain must be a power of two");
    DG_REQUIRE(a_query && b_g1_query && b_g2_query && h_query && l_query && fixed_points, DG16_ERR_BAD_ARG,
               "null query");
    pk->d.curve = curve;
    pk->d.num_vars = num_vars;
    pk->d.num_inputs = num_inputs;
    pk->d.m = domain_size;
    DG_REQUIRE(n_shards >= 1 && shard < n_shards, DG16_ERR_BAD_ARG, "shard index out of range");
    pk->d.shard = shard;
    pk->d.nshards = n_shards;
    pk->d.h_cyclic = (flags & DG16_F_H_CYCLIC) != 0;
    bool dev = flags & DG16_F_DEVICE_PTRS;
    if (curve == DG16_BN254)
      pk_build_bn254(ctx, pk->d, a_query, b_g1_query, b_g2_query, h_query, l_query, fixed_points, dev);
    else
      pk_build_bls12_381(ctx, pk->d, a_query, b_g1_query, b_g2_query, h_query, l_query, fixed_points, dev);
  });
  if (rc != DG16_OK) {
    pk_free(pk->d);
    delete pk;
    return rc;
  }
  *out = pk;
  return DG16_OK;
}

int dg16_pk_info_get(const dg16_pk* pk, dg16_pk_info* out) {
  if (!pk || !out) return DG16_ERR_BAD_ARG;
  memset(out, 0, sizeof(*out));
  const PkDev& d = pk->d;
  out->n_ab = d.ab_hi - d.ab_lo + 3;
  out->n_l = out->n_ab;
  out->n_h = d.h_hi - d.h_lo;
  out->c_ab = d.c_ab;
  out->c_l = d.c_l;
  out->c_h = d.c_h;
  out->shard = d.shard;
  out->n_shards = d.nshards;
  out->table_bytes = d.table_bytes;
  out->table_stride = d.stride;
  return DG16_OK;
}

void dg16_pk_destroy(dg16_pk* pk) {
  if (!pk) return;
  hipSetDevice(pk->ctx->device);
  hipDeviceSynchronize();
  pk_free(pk->d);
  delete pk;
}

int dg16_groth16_prove(dg16_ctx* ctx, const dg16_pk* pk, const void* a, const void* b, const void* c,
                       const void* full_assignment, const void* r_s, unsigned flags, void* proof_out) {
  if (!ctx || !pk) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pk->ctx == ctx, DG16_ERR_BAD_ARG, "proving key belongs to another context");
    DG_REQUIRE(a && b && c && full_assignment && r_s && proof_out, DG16_ERR_BAD_ARG, "null operand");
    bool mont = flags & DG16_F_SCALARS_MONT, dev = flags & DG16_F_DEVICE_PTRS;
    const bool overlap = (flags & DG16_F_OVERLAP_TAIL) && dev;
    if (pk->d.curve == DG16_BN254)
      prove_bn254(ctx, pk->d, a, b, c, full_assignment, r_s, mont, dev, proof_out, overlap);
    else
      prove_bls12_381(ctx, pk->d, a, b, c, full_assignment, r_s, mont, dev, proof_out, overlap);
  });
}

size_t dg16_groth16_results_bytes(int curve) {
  return curve == DG16_BN254 ? results_bytes_bn254() : curve == DG16_BLS12_381 ? results_bytes_bls12_381() : 0;
}

int dg16_groth16_msms(dg16_ctx* ctx, const dg16_pk* pk, const void* a, const void* b, const void* c,
                      const void* full_assignment, const void* r_s, unsigned flags, void* results_out) {
  if (!ctx || !pk) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pk->ctx == ctx, DG16_ERR_BAD_ARG, "proving key belongs to another context");
    DG_REQUIRE(a && b && c && full_assignment && r_s && results_out, DG16_ERR_BAD_ARG, "null operand");
    bool mont = flags & DG16_F_SCALARS_MONT, dev = flags & DG16_F_DEVICE_PTRS;
    Call k0(ctx, 0), k1(ctx, 1), k2(ctx, 2);
    size_t rec = dg16_groth16_results_bytes(pk->d.curve);
    uint8_t* res_dev = dev ? (uint8_t*)results_out : (uint8_t*)ws(k0.c, 16, 8192);
    k0.begin_dominant();
    if (pk->d.curve == DG16_BN254)
      msms_bn254(ctx, k0, k1, k2, pk->d, a, b, c, full_assignment, r_s, mont, dev, res_dev, nullptr, nullptr);
    else
      msms_bls12_381(ctx, k0, k1, k2, pk->d, a, b, c, full_assignment, r_s, mont, dev, res_dev, nullptr, nullptr);
    k0.end_dominant();
    if (!dev) stage_out(k0, results_out, res_dev, rec, false);
    k0.finish();
    k1.finish();
    k2.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k0.s()));
  });
}

int dg16_groth16_msms_h(dg16_ctx* ctx, const dg16_pk* pk, const void* h_shard, const void* full_assignment,
                        const void* r_s, unsigned flags, void* results_out) {
  if (!ctx || !pk) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pk->ctx == ctx, DG16_ERR_BAD_ARG, "proving key belongs to another context");
    DG_REQUIRE(h_shard && full_assignment && r_s && results_out, DG16_ERR_BAD_ARG, "null operand");
    bool mont = flags & DG16_F_SCALARS_MONT, dev = flags & DG16_F_DEVICE_PTRS;
    Call k0(ctx, 0), k1(ctx, 1), k2(ctx, 2);
    size_t rec = dg16_groth16_results_bytes(pk->d.curve);
    uint8_t* res_dev = dev ? (uint8_t*)results_out : (uint8_t*)ws(k0.c, 16, 8192);
    k0.begin_dominant();
    if (pk->d.curve == DG16_BN254)
      msms_bn254(ctx, k0, k1, k2, pk->d, nullptr, nullptr, nullptr, full_assignment, r_s, mont, dev, res_dev, nullptr, h_shard);
    else
      msms_bls12_381(ctx, k0, k1, k2, pk->d, nullptr, nullptr, nullptr, full_assignment, r_s, mont, dev, res_dev, nullptr,
                     h_shard);
    k0.end_dominant();
    if (!dev) stage_out(k0, results_out, res_dev, rec, false);
    k0.finish();
    k1.finish();
    k2.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k0.s()));
  });
}

int dg16_groth16_prove_dist(dg16_ctx* ctx, const dg16_pk* pk, const dg16_comm* comm, const void* a_rows,
                            const void* b_rows, const void* c_rows, const void* full_assignment, const void* r_s,
                            unsigned flags, void* proof_out) {
  if (!ctx || !pk) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pk->ctx == ctx, DG16_ERR_BAD_ARG, "proving key belongs to another context");
    DG_REQUIRE(a_rows && b_rows && c_rows && full_assignment && r_s && proof_out, DG16_ERR_BAD_ARG, "null operand");
    bool mont = flags & DG16_F_SCALARS_MONT, dev = flags & DG16_F_DEVICE_PTRS;
    const bool overlap = (flags & DG16_F_OVERLAP_TAIL) && dev;
    if (pk->d.curve == DG16_BN254)
      prove_dist_bn254(ctx, pk->d, comm, a_rows, b_rows, c_rows, full_assignment, r_s, mont, dev, proof_out, overlap);
    else
      prove_dist_bls12_381(ctx, pk->d, comm, a_rows, b_rows, c_rows, full_assignment, r_s, mont, dev, proof_out, overlap);
  });
}

int dg16_groth16_assemble(dg16_ctx* ctx, const dg16_pk* pk, const void* gathered_results, size_t n_shards,
                          const void* r_s, unsigned flags, void* proof_out) {
  if (!ctx || !pk) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pk->ctx == ctx, DG16_ERR_BAD_ARG, "proving key belongs to another context");
    DG_REQUIRE(gathered_results && r_s && proof_out && n_shards >= 1, DG16_ERR_BAD_ARG, "null operand");
    bool mont = flags & DG16_F_SCALARS_MONT, dev = flags & DG16_F_DEVICE_PTRS;
    Call k0(ctx, 0);
    size_t rec = dg16_groth16_results_bytes(pk->d.curve);
    const uint8_t* g = (const uint8_t*)stage_in(k0, 19, gathered_results, n_shards * rec, dev);
    uint8_t* proof_dev = (uint8_t*)ws(k0.c, 16, 8192) + 4096;
    size_t proof_bytes;
    if (pk->d.curve == DG16_BN254) {
      assemble_bn254(k0, g, n_shards, proof_dev);
      proof_bytes = proof_bytes_bn254();
    } else {
      assemble_bls12_381(k0, g, n_shards, proof_dev);
      proof_bytes = proof_bytes_bls12_381();
    }
    stage_out(k0, proof_out, proof_dev, proof_bytes, dev);
    k0.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k0.s()));
  });
}

}  // extern "C"
