// MSM / base generation / affine conversion instantiated for curve id 2 (bls12_377).
#include "msm_impl.h"

namespace dg16 {
using CT = CurveTypes<2>;

void msm_bls12_377(Call& k, int group, const void* bases, const void* scalars, size_t n, bool mont, bool affine,
                 void* out) {
  if (group == 1) msm_run<CT::Fq, CT::Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mont, affine, out);
  else throw StatusError{DG16_ERR_UNSUPPORTED, "BLS12-377 G2 is not on the reference's path"};
}
void gen_bases_bls12_377(Call& k, int group, uint64_t seed, size_t n, void* out) {
  if (group == 1) gen_bases_run<CT::Fq, CT::G1c>(k, seed, n, out);
  else throw StatusError{DG16_ERR_UNSUPPORTED, "BLS12-377 G2 is not on the reference's path"};
}
void to_affine_bls12_377(Call& k, int group, const void* jac, void* out, size_t n) {
  if (group == 1) to_affine_run<CT::Fq>(k, jac, out, n);
  else throw StatusError{DG16_ERR_UNSUPPORTED, "BLS12-377 G2 is not on the reference's path"};
}
}  // namespace dg16
