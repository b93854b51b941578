// Short-Weierstrass (a = 0) group arithmetic for G1 (F = Fp) and G2 (F = Fp2), gfx950.
//
// Point formats:
//   Affine<F>   x || y, identity = (0, 0)           -- what the C ABI takes (include/dg16.h); the
//               reference's G::Affine at dist-primitives/src/dmsm/mod.rs:82 repacked without the
//               `infinity` flag, as ark-circom/src/zkey.rs:353-361 encodes it.
//   XYZZ<F>     (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, identity = ZZ == 0.  Bucket accumulator:
//               mixed add 8M + 2S (madd-2008-s), general add 12M + 2S (add-2008-s), doubling
//               6M + 3S (dbl-2008-s-1) -- one representation for every bucket phase.
//   Jacobian<F> (X, Y, Z), what the C ABI returns (== ark-ec `Projective` for sw curves);
//               from XYZZ without an inversion: (X*ZZ, Y*ZZZ, ZZ).
// Every operation is complete (identity / doubling / inverse operands handled), because inputs the
// reference tests use hit them (dmsm/mod.rs:155-159: M copies of one point, scalars all 1).
#pragma once
#include "fp.h"
#include "fp2.h"

namespace dg16 {

#if defined(__HIPCC__)
#define DG_COLD __host__ __device__ __attribute__((noinline))
#else
#define DG_COLD __attribute__((noinline))
#endif

template <class F>
struct Affine {
  F x, y;
  DG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  DG_HD static Affine inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct Jacobian {
  F x, y, z;
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;

  DG_HD bool is_inf() const { return zz.is_zero(); }
  DG_HD static XYZZ inf() { return {F::one(), F::one(), F::zero(), F::zero()}; }
  DG_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
  DG_HD XYZZ neg() const { return {x, y.neg(), zz, zzz}; }

  // 2 * (affine p), p != identity                                   (mdbl-2008-s-1, a = 0)
  // (forceinline: a call here makes hipcc keep the loaded point in scratch memory -- the G2 bucket
  // kernel then spent 7x its arithmetic time on scratch round trips at one wave per SIMD)
  static DG_HD XYZZ dbl_affine(const F& px, const F& py) {
    F u = py.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = px * v;
    F xx = px.sqr();
    F m = xx.dbl() + xx;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * py;
    return {x3, y3, v, w};
  }
  // 2 * p                                                             (dbl-2008-s-1, a = 0)
  DG_COLD XYZZ dbl() const {
    if (is_inf()) return *this;
    F u = y.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = x * v;
    F xx = x.sqr();
    F m = xx.dbl() + xx;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * y;
    return {x3, y3, v * zz, w * zzz};
  }
  // this + (negate ? -q : q), q affine                                (madd-2008-s)
  DG_HD XYZZ madd(const Affine<F>& q, bool negate) const {
    if (q.is_inf()) return *this;
    F qy = negate ? q.y.neg() : q.y;
    if (is_inf()) return {q.x, qy, F::one(), F::one()};
    F u2 = q.x * zz;
    F s2 = qy * zzz;
    F p = u2 - x;
    F r = s2 - y;
    if (p.is_zero()) {
      if (r.is_zero()) return dbl_affine(q.x, qy);
      return inf();
    }
    F pp = p.sqr();
    F ppp = p * pp;
    F q_ = x * pp;
    F x3 = r.sqr() - ppp - q_.dbl();
    F y3 = r * (q_ - x3) - y * ppp;
    return {x3, y3, zz * pp, zzz * ppp};
  }
  // this + o                                                           (add-2008-s)
  // noinline: only the bucket-reduction / tail kernels use it, and inlining 14 multiplications at
  // every call site multiplies compile time and code size for nothing
  DG_COLD XYZZ add(const XYZZ& o) const {
    if (o.is_inf()) return *this;
    if (is_inf()) return o;
    F u1 = x * o.zz;
    F u2 = o.x * zz;
    F s1 = y * o.zzz;
    F s2 = o.y * zzz;
    F p = u2 - u1;
    F r = s2 - s1;
    if (p.is_zero()) {
      if (r.is_zero()) return dbl();
      return inf();
    }
    F pp = p.sqr();
    F ppp = p * pp;
    F q_ = u1 * pp;
    F x3 = r.sqr() - ppp - q_.dbl();
    F y3 = r * (q_ - x3) - s1 * ppp;
    return {x3, y3, zz * o.zz * pp, zzz * o.zzz * ppp};
  }
  // (X*ZZ, Y*ZZZ, ZZ) is the same point in Jacobian coordinates with Z = ZZ
  DG_HD Jacobian<F> to_jacobian() const {
    if (is_inf()) return {F::one(), F::one(), F::zero()};
    return {x * zz, y * zzz, zz};
  }
  // serial (one inversion); only for O(1)-sized tails
  DG_HD Affine<F> to_affine() const {
    if (is_inf()) return Affine<F>::inf();
    F zi = zzz.inv();          // 1/z^3
    F zi2 = (zi * zz).sqr();   // (z^2/z^3)^2 = 1/z^2
    return {x * zi2, y * zi};
  }
  DG_HD static XYZZ from_jacobian(const Jacobian<F>& j) {
    if (j.z.is_zero()) return inf();
    F zz = j.z.sqr();
    return {j.x, j.y, zz, zz * j.z};
  }
};

// k * p by double-and-add, k = NW little-endian 32-bit words (plain integer)
template <class F, int NW>
DG_HD XYZZ<F> scalar_mul(const XYZZ<F>& p, const uint32_t* k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = NW * 32 - 1; i >= 0; i--) {
    acc = acc.dbl();
    if ((k[i / 32] >> (i % 32)) & 1) acc = acc.add(p);
  }
  return acc;
}

}  // namespace dg16
