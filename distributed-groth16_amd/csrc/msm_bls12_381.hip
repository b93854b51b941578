// MSM / base generation / affine conversion instantiated for curve id 1 (bls12_381).
#include "msm_impl.h"

namespace dg16 {
using CT = CurveTypes<1>;

void msm_bls12_381(Call& k, int group, const void* bases, const void* scalars, size_t n, bool mont, bool affine,
                 void* out) {
  if (group == 1) msm_run<CT::Fq, CT::Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mont, affine, out);
  else msm_run<CT::Fq2, CT::Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mont, affine, out);
}
void gen_bases_bls12_381(Call& k, int group, uint64_t seed, size_t n, void* out) {
  if (group == 1) gen_bases_run<CT::Fq, CT::G1c>(k, seed, n, out);
  else gen_bases_run<CT::Fq2, CT::G2c>(k, seed, n, out);
}
void to_affine_bls12_381(Call& k, int group, const void* jac, void* out, size_t n) {
  if (group == 1) to_affine_run<CT::Fq>(k, jac, out, n);
  else to_affine_run<CT::Fq2>(k, jac, out, n);
}
}  // namespace dg16
