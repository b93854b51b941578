// Groth16 prover instantiated for curve id 0 (bn254).
#include "msm_impl.h"

namespace dg16 {
DG16_MSM_EXTERN(CurveTypes<0>)   // compiled in msm_group.hip / msm_reduce.hip
}  // namespace dg16

#include "prover_impl.h"

namespace dg16 {
void pk_build_bn254(dg16_ctx* ctx, PkDev& d, const void* a, const void* b1, const void* b2, const void* h, const void* l,
                 const void* fx, bool dev) { pk_build<0>(ctx, d, a, b1, b2, h, l, fx, dev); }
void prove_bn254(dg16_ctx* ctx, const PkDev& pk, const void* a, const void* b, const void* c, const void* w,
              const void* rs, bool mont, bool dev, void* out, bool overlap) {
  prove_typed<0>(ctx, pk, a, b, c, w, rs, mont, dev, out, overlap);
}
void msms_bn254(dg16_ctx* ctx, Call& k0, Call& k1, Call& k2, const PkDev& pk, const void* a, const void* b, const void* c,
             const void* w, const void* rs, bool mont, bool dev, uint8_t* res, const dg16_comm* comm, const void* h_given) {
  msms_typed<0>(ctx, k0, k1, k2, pk, a, b, c, w, rs, mont, dev, res, comm, h_given);
}
void prove_dist_bn254(dg16_ctx* ctx, const PkDev& pk, const dg16_comm* comm, const void* a, const void* b, const void* c,
              const void* w, const void* rs, bool mont, bool dev, void* out, bool overlap) {
  prove_dist_typed<0>(ctx, pk, comm, a, b, c, w, rs, mont, dev, out, overlap);
}
void assemble_bn254(Call& k0, const uint8_t* gathered, size_t n_shards, uint8_t* proof) {
  assemble_typed<0>(k0, gathered, n_shards, proof);
}
size_t results_bytes_bn254() { return msm_results_bytes<0>(); }
size_t proof_bytes_bn254() {
  using CT = CurveTypes<0>;
  return 2 * sizeof(Jacobian<CT::Fq>) + sizeof(Jacobian<CT::Fq2>);
}
}  // namespace dg16
