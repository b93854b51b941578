// Curve / group dispatch for the group-valued entry points (one translation unit per (curve, group)
// so the heavy template instantiations compile in parallel).
#include "ctx.h"

namespace dg16 {
#define DECL(name)                                                                              \
  void msm_##name(Call&, const void*, const void*, size_t, unsigned, bool, void*);                  \
  void gen_bases_##name(Call&, uint64_t, size_t, void*);                                        \
  void to_affine_##name(Call&, const void*, void*, size_t);                                     \
  void* bases_table_##name(Call&, const void*, size_t, size_t, unsigned*, unsigned*, unsigned*); \
  void msm_resident_##name(Call&, const void*, size_t, unsigned, unsigned, const void*, bool, bool, void*);
DECL(bn254_g1) DECL(bn254_g2) DECL(bls12_381_g1) DECL(bls12_381_g2) DECL(bls12_377_g1) DECL(bls12_377_g2)

#define DISPATCH(fn, ...)                                                                       \
  switch (curve * 2 + group - 1) {                                                              \
    case 0: fn##_bn254_g1(__VA_ARGS__); break;                                                  \
    case 1: fn##_bn254_g2(__VA_ARGS__); break;                                                  \
    case 2: fn##_bls12_381_g1(__VA_ARGS__); break;                                              \
    case 3: fn##_bls12_381_g2(__VA_ARGS__); break;                                              \
    case 4: fn##_bls12_377_g1(__VA_ARGS__); break;                                              \
    case 5: fn##_bls12_377_g2(__VA_ARGS__); break;                                              \
    default: throw StatusError{DG16_ERR_BAD_ARG, "unknown (curve, group)"};                     \
  }

void msm_launch(Call& k, int curve, int group, const void* bases, const void* scalars, size_t n, unsigned mode,
                bool affine, void* out) {
  DISPATCH(msm, k, bases, scalars, n, mode, affine, out)
}
void gen_bases_launch(Call& k, int curve, int group, uint64_t seed, size_t n, void* out) {
  DISPATCH(gen_bases, k, seed, n, out)
}
void to_affine_launch(Call& k, int curve, int group, const void* jac, void* out, size_t n) {
  DISPATCH(to_affine, k, jac, out, n)
}
void* bases_table_launch(Call& k, int curve, int group, const void* bases, size_t n, size_t budget, unsigned* c,
                         unsigned* nwin, unsigned* stride) {
  void* t = nullptr;
  switch (curve * 2 + group - 1) {
    case 0: t = bases_table_bn254_g1(k, bases, n, budget, c, nwin, stride); break;
    case 1: t = bases_table_bn254_g2(k, bases, n, budget, c, nwin, stride); break;
    case 2: t = bases_table_bls12_381_g1(k, bases, n, budget, c, nwin, stride); break;
    case 3: t = bases_table_bls12_381_g2(k, bases, n, budget, c, nwin, stride); break;
    case 4: t = bases_table_bls12_377_g1(k, bases, n, budget, c, nwin, stride); break;
    case 5: t = bases_table_bls12_377_g2(k, bases, n, budget, c, nwin, stride); break;
    default: throw StatusError{DG16_ERR_BAD_ARG, "unknown (curve, group)"};
  }
  return t;
}
void msm_resident_launch(Call& k, int curve, int group, const void* table, size_t n, unsigned c, unsigned stride,
                         const void* scalars, bool mont, bool affine, void* out) {
  DISPATCH(msm_resident, k, table, n, c, stride, scalars, mont, affine, out)
}
}  // namespace dg16
