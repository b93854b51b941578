// Quadratic extension Fq[u]/(u^2 + BETA): BETA = 1 for BN254 and BLS12-381 (ark-bn254 / ark-bls12-381 Fq2Config
// NONRESIDUE = -1), BETA = 5 for BLS12-377 (ark-bls12-377 Fq2Config NONRESIDUE = -5; the G2 of
// groth16/examples/local_groth_bench.rs:141).  Memory layout c0 || c1, each an Fp in Montgomery form,
// i.e. the arkworks `QuadExtField { c0, c1 }` the reference passes for E::G2Affine at
// groth16/src/prove.rs:62-85.
#pragma once
#include "fp.h"

namespace dg16 {

// u^2 = -Fq2Beta<P>::value over the base field with parameters P (a small positive integer)
struct bls12_377_fq_params;
template <class P> struct Fq2Beta { static constexpr int value = 1; };
template <> struct Fq2Beta<bls12_377_fq_params> { static constexpr int value = 5; };
// BETA * a by additions (BETA is 1 or 5)
template <class F>
DG_HD F fq2_beta_mul(const F& a) {
  constexpr int BETA = Fq2Beta<typename F::Params>::value;
  static_assert(BETA == 1 || BETA == 5, "non-residue not wired");
  if constexpr (BETA == 1) return a;
  else return a.dbl().dbl() + a;
}

template <class F>
struct Fp2 {
  using Base = F;
  F c0, c1;

  DG_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
  DG_HD static Fp2 one() { return {F::one(), F::zero()}; }
  DG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  DG_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
  DG_HD bool operator!=(const Fp2& o) const { return !(*this == o); }
  DG_HD friend Fp2 operator+(const Fp2& a, const Fp2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  DG_HD friend Fp2 operator-(const Fp2& a, const Fp2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  DG_HD Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  DG_HD Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // Karatsuba: 3 base multiplications
  DG_HD friend Fp2 operator*(const Fp2& a, const Fp2& b) {
    F v0 = F::mul_call(a.c0, b.c0);
    F v1 = F::mul_call(a.c1, b.c1);
    F s = F::mul_call(a.c0 + a.c1, b.c0 + b.c1);
    return {v0 - fq2_beta_mul(v1), s - v0 - v1};
  }
  // complex squaring: 2 base multiplications.  (c0 + c1)(c0 - BETA c1) = c0^2 - BETA c1^2 - (BETA - 1) c0 c1
  DG_HD Fp2 sqr() const {
    F t = F::mul_call(c0, c1);
    F r0 = F::mul_call(c0 + c1, c0 - fq2_beta_mul(c1));
    if constexpr (Fq2Beta<typename F::Params>::value != 1) r0 = r0 + fq2_beta_mul(t) - t;
    return {r0, t.dbl()};
  }
  DG_HD Fp2 inv() const {
    F n = (c0.sqr() + fq2_beta_mul(c1.sqr())).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  DG_HD static Fp2 select(bool c, const Fp2& a, const Fp2& b) {
    return {F::select(c, a.c0, b.c0), F::select(c, a.c1, b.c1)};
  }
};

}  // namespace dg16
