// Concrete field / group types per curve id (include/dg16.h enum dg16_curve).
#pragma once
#include "consts_gen.h"
#include "ec.h"

namespace dg16 {

using bn254_fq = Fp<bn254_fq_params>;
using bn254_fr = Fp<bn254_fr_params>;
using bls12_381_fq = Fp<bls12_381_fq_params>;
using bls12_381_fr = Fp<bls12_381_fr_params>;
using bls12_377_fq = Fp<bls12_377_fq_params>;
using bls12_377_fr = Fp<bls12_377_fr_params>;

template <int CURVE> struct CurveTypes;
template <> struct CurveTypes<0> {
  using Fq = bn254_fq; using Fr = bn254_fr; using Fq2 = Fp2<bn254_fq>;
  using G1c = bn254_g1_consts; using G2c = bn254_g2_consts;
  static constexpr int SCALAR_BITS = 254;
};
template <> struct CurveTypes<1> {
  using Fq = bls12_381_fq; using Fr = bls12_381_fr; using Fq2 = Fp2<bls12_381_fq>;
  using G1c = bls12_381_g1_consts; using G2c = bls12_381_g2_consts;
  static constexpr int SCALAR_BITS = 255;
};
template <> struct CurveTypes<2> {
  using Fq = bls12_377_fq; using Fr = bls12_377_fr; using Fq2 = Fp2<bls12_377_fq>;
  using G1c = bls12_377_g1_consts; using G2c = bls12_377_g2_consts;
  static constexpr int SCALAR_BITS = 253;
};

// group generator as an affine point of coordinate field F
template <class F, class C> struct GenLoader;
template <class P, class C> struct GenLoader<Fp<P>, C> {
  DG_HD static Affine<Fp<P>> get() {
    Affine<Fp<P>> g;
#pragma unroll
    for (int i = 0; i < P::NL; i++) { g.x.l[i] = C::GX[i]; g.y.l[i] = C::GY[i]; }
    return g;
  }
};
template <class P, class C> struct GenLoader<Fp2<Fp<P>>, C> {
  DG_HD static Affine<Fp2<Fp<P>>> get() {
    Affine<Fp2<Fp<P>>> g;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
      g.x.c0.l[i] = C::GX_C0[i]; g.x.c1.l[i] = C::GX_C1[i];
      g.y.c0.l[i] = C::GY_C0[i]; g.y.c1.l[i] = C::GY_C1[i];
    }
    return g;
  }
};

}  // namespace dg16
