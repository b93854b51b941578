// Curve dispatch for the group-valued entry points (per-curve instantiations live in msm_<curve>.hip
// so they compile in parallel).
#include "ctx.h"

namespace dg16 {
#define DECL(name)                                                                                          \
  void msm_##name(Call&, int, const void*, const void*, size_t, bool, bool, void*);                         \
  void gen_bases_##name(Call&, int, uint64_t, size_t, void*);                                               \
  void to_affine_##name(Call&, int, const void*, void*, size_t);
DECL(bn254)
DECL(bls12_381)
DECL(bls12_377)

void msm_launch(Call& k, int curve, int group, const void* bases, const void* scalars, size_t n, bool mont,
                bool affine, void* out) {
  switch (curve) {
    case 0: msm_bn254(k, group, bases, scalars, n, mont, affine, out); break;
    case 1: msm_bls12_381(k, group, bases, scalars, n, mont, affine, out); break;
    default: msm_bls12_377(k, group, bases, scalars, n, mont, affine, out); break;
  }
}
void gen_bases_launch(Call& k, int curve, int group, uint64_t seed, size_t n, void* out) {
  switch (curve) {
    case 0: gen_bases_bn254(k, group, seed, n, out); break;
    case 1: gen_bases_bls12_381(k, group, seed, n, out); break;
    default: gen_bases_bls12_377(k, group, seed, n, out); break;
  }
}
void to_affine_launch(Call& k, int curve, int group, const void* jac, void* out, size_t n) {
  switch (curve) {
    case 0: to_affine_bn254(k, group, jac, out, n); break;
    case 1: to_affine_bls12_381(k, group, jac, out, n); break;
    default: to_affine_bls12_377(k, group, jac, out, n); break;
  }
}
}  // namespace dg16
