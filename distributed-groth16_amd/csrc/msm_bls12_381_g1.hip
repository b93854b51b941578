// MSM / base generation / affine conversion instantiated for curve id 1 (bls12_381), G1.
#include "msm_impl.h"

namespace dg16 {
using CT = CurveTypes<1>;
void msm_bls12_381_g1(Call& k, const void* bases, const void* scalars, size_t n, bool mont, bool affine, void* out) {
  msm_run<CT::Fq, CT::Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mont, affine, out);
}
void gen_bases_bls12_381_g1(Call& k, uint64_t seed, size_t n, void* out) { gen_bases_run<CT::Fq, CT::G1c>(k, seed, n, out); }
void to_affine_bls12_381_g1(Call& k, const void* jac, void* out, size_t n) { to_affine_run<CT::Fq>(k, jac, out, n); }
}  // namespace dg16
