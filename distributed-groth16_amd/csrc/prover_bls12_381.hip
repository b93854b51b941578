// Groth16 prover instantiated for curve id 1 (bls12_381).
#include "msm_impl.h"

namespace dg16 {
DG16_MSM_EXTERN(CurveTypes<1>)   // compiled in msm_group.hip / msm_reduce.hip
}  // namespace dg16

#include "prover_impl.h"

namespace dg16 {
void pk_build_bls12_381(dg16_ctx* ctx, PkDev& d, const void* a, const void* b1, const void* b2, const void* h, const void* l,
                 const void* fx, bool dev) { pk_build<1>(ctx, d, a, b1, b2, h, l, fx, dev); }
void prove_bls12_381(dg16_ctx* ctx, const PkDev& pk, const void* a, const void* b, const void* c, const void* w,
              const void* rs, bool mont, bool dev, void* out, bool overlap) {
  prove_typed<1>(ctx, pk, a, b, c, w, rs, mont, dev, out, overlap);
}
void msms_bls12_381(dg16_ctx* ctx, Call& k0, Call& k1, Call& k2, const PkDev& pk, const void* a, const void* b, const void* c,
             const void* w, const void* rs, bool mont, bool dev, uint8_t* res, const dg16_comm* comm, const void* h_given) {
  msms_typed<1>(ctx, k0, k1, k2, pk, a, b, c, w, rs, mont, dev, res, comm, h_given);
}
void prove_dist_bls12_381(dg16_ctx* ctx, const PkDev& pk, const dg16_comm* comm, const void* a, const void* b, const void* c,
              const void* w, const void* rs, bool mont, bool dev, void* out, bool overlap) {
  prove_dist_typed<1>(ctx, pk, comm, a, b, c, w, rs, mont, dev, out, overlap);
}
void assemble_bls12_381(Call& k0, const uint8_t* gathered, size_t n_shards, uint8_t* proof) {
  assemble_typed<1>(k0, gathered, n_shards, proof);
}
size_t results_bytes_bls12_381() { return msm_results_bytes<1>(); }
size_t proof_bytes_bls12_381() {
  using CT = CurveTypes<1>;
  return 2 * sizeof(Jacobian<CT::Fq>) + sizeof(Jacobian<CT::Fq2>);
}
}  // namespace dg16
