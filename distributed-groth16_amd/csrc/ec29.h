// Short-Weierstrass (a = 0) group arithmetic on the reduced-radix field types of fp29.h -- the bucket phases of the
// MSM (segment accumulation, bucket reduction).  Same formulas and completeness rules as ec.h (XYZZ coordinates:
// mixed add 8M + 2S, add 12M + 2S, doubling 6M + 3S; over a prime field Y3 = R (Q - X3) - PPP Y1 is ONE dual product,
// mul_sub: nine Montgomery reductions per mixed addition, not ten; identity / doubling / inverse operands handled because the
// reference's own tests feed them: dist-primitives/src/dmsm/mod.rs:155-159), but every temporary carries its bounds
// in its type (`auto`), products normalise an operand only where a column could overflow, and coordinates are
// stored normalised below a fixed storage bound BS p: nothing is compared or conditionally subtracted between
// the products.  A formula that compiles cannot overflow (static_asserts of fp29.h).
//
// Coordinates are x R mod p with R = 2^(W N) (NOT the arkworks R = 2^(32 NL) of the C ABI): bases enter through
// msm_to_internal_kernel / the table builder, results leave through to_xyzz32().
// Identity: ZZ with all limbs zero (only ever created by inf(); never the result of arithmetic).
#pragma once
#include "ec.h"
#include "fp29.h"

namespace dg16 {

// storage bound (in p / 64) of a coordinate.  7 p closes every formula below without a reduction when the field
// has >= 7 spare bits and the coordinates are base-field elements; quadratic-extension coordinates over a field
// with little slack (BN254: 7 bits) are reduced to ~p when stored (Fq2 products double the bounds twice).
template <class P, bool EXT>
struct StoreBound {
  static constexpr int value = (EXT && RR<P>::SLACK < 9) ? 130 : 448;
};

template <class F> struct FieldOf;   // maps a 32-bit-limb field type (fp.h / fp2.h) to its reduced-radix storage type
template <class P> struct FieldOf<Fp<P>> {
  using Params = P;
  static constexpr bool EXT = false;
  static constexpr int BS = StoreBound<P, false>::value;
  using Store = Fe<P, BS, 1>;
  using Canon = Fe<P, 64, 1>;
  static constexpr int WORDS = RR<P>::NL;           // packed words per element
  DG_HD static Canon load_packed(const uint32_t* w) { return fe_from_words<P>(w); }
  DG_HD static void store_packed(const Canon& a, uint32_t* w) { fe_to_words<P>(a, w); }
  DG_HD static Canon canon_of(const Store& a) { return canon(a); }
  DG_HD static Store one() { return fe_one<P>().template as<BS, 1>(); }
  DG_HD static Store zero() { return Store::zero(); }
  DG_HD static Fp<P> to32(const Store& a) { return fe_to_fp(a); }
  DG_HD static Store from32(const Fp<P>& a) { return fit<BS>(fe_from_fp(a)); }
};
template <class P> struct FieldOf<Fp2<Fp<P>>> {
  using Params = P;
  static constexpr bool EXT = true;
  static constexpr int BS = StoreBound<P, true>::value;
  using Store = Fe2<P, BS, 1>;
  using Canon = Fe2<P, 64, 1>;
  static constexpr int WORDS = 2 * RR<P>::NL;
  DG_HD static Canon load_packed(const uint32_t* w) { return {fe_from_words<P>(w), fe_from_words<P>(w + RR<P>::NL)}; }
  DG_HD static void store_packed(const Canon& a, uint32_t* w) {
    fe_to_words<P>(a.c0, w);
    fe_to_words<P>(a.c1, w + RR<P>::NL);
  }
  DG_HD static Canon canon_of(const Store& a) { return {canon(a.c0), canon(a.c1)}; }
  DG_HD static Store one() { return {fe_one<P>().template as<BS, 1>(), Fe<P, BS, 1>::zero()}; }
  DG_HD static Store zero() { return {Fe<P, BS, 1>::zero(), Fe<P, BS, 1>::zero()}; }
  DG_HD static Fp2<Fp<P>> to32(const Store& a) { return {fe_to_fp(a.c0), fe_to_fp(a.c1)}; }
  DG_HD static Store from32(const Fp2<Fp<P>>& a) { return {fit<BS>(fe_from_fp(a.c0)), fit<BS>(fe_from_fp(a.c1))}; }
};

// affine point in internal form, canonical coordinates; identity = (0, 0)
template <class F>
struct Affine29 {
  using FO = FieldOf<F>;
  typename FO::Canon x, y;
  DG_HD bool is_inf() const { return limbs_all_zero(x) && limbs_all_zero(y); }
  // from / to the packed words the library's tables hold (x || y, FO::WORDS words each)
  DG_HD static Affine29 load(const uint32_t* w) { return {FO::load_packed(w), FO::load_packed(w + FO::WORDS)}; }
};

template <class F>
struct XYZZ29 {
  using FO = FieldOf<F>;
  using S = typename FO::Store;
  static constexpr int BS = FO::BS;
  S x, y, zz, zzz;

  DG_HD bool is_inf() const { return limbs_all_zero(zz); }
  DG_HD static XYZZ29 inf() { return {FO::one(), FO::one(), FO::zero(), FO::zero()}; }
  DG_HD XYZZ29 neg_pt() const { return {x, fit<BS>(neg(y)), zz, zzz}; }

  // 2 (qx, qy), the affine point not the identity                         (mdbl-2008-s-1, a = 0)
  template <class FX, class FY>
  DG_HD static XYZZ29 dbl_affine(const FX& qx, const FY& qy) {
    const auto u = dbl(qy);
    const auto v = sqr(u);
    const auto w = u * v;
    const auto s = qx * v;
    const auto xx = sqr(qx);
    const auto m = dbl(xx) + xx;
    const auto x3 = fit<BS>(sqr(m) - dbl(s));
    const auto y3 = mul_sub(m, s - x3, w, qy);
    return {x3, fit<BS>(y3), fit<BS>(v), fit<BS>(w)};
  }
  // 2 this                                                                 (dbl-2008-s-1, a = 0)
  DG_HD XYZZ29 dbl_pt() const {
    if (is_inf()) return *this;
    const auto u = dbl(y);
    const auto v = sqr(u);
    const auto w = u * v;
    const auto s = x * v;
    const auto xx = sqr(x);
    const auto m = dbl(xx) + xx;
    const auto x3 = fit<BS>(sqr(m) - dbl(s));
    const auto y3 = mul_sub(m, s - x3, w, y);
    return {x3, fit<BS>(y3), fit<BS>(v * zz), fit<BS>(w * zzz)};
  }
  // this + (negate ? -q : q), q affine                                      (madd-2008-s)
  DG_HD XYZZ29 madd(const Affine29<F>& q, bool negate) const {
    if (q.is_inf()) return *this;
    const auto nqy = neg(q.y);
    const auto qy = select(negate, nqy, q.y.template as<decltype(nqy)::Bound, decltype(nqy)::Limb>());
    if (is_inf()) return {q.x.template as<BS, 1>(), fit<BS>(qy), FO::one(), FO::one()};
    const auto u2 = q.x * zz;
    const auto s2 = qy * zzz;
    const auto p_ = norm(u2 - x);
    const auto r_ = norm(s2 - y);
    if (is_zero(p_)) {
      if (is_zero(r_)) return dbl_affine(q.x, qy);
      return inf();
    }
    const auto pp = sqr(p_);
    const auto ppp = p_ * pp;
    const auto q_ = x * pp;
    const auto x3 = fit<BS>(sqr(r_) - (ppp + dbl(q_)));
    const auto y3 = mul_sub(r_, q_ - x3, ppp, y);
    return {x3, fit<BS>(y3), fit<BS>(zz * pp), fit<BS>(zzz * ppp)};
  }
  // this + o                                                                 (add-2008-s)
  DG_HD XYZZ29 add(const XYZZ29& o) const {
    if (o.is_inf()) return *this;
    if (is_inf()) return o;
    const auto u1 = x * o.zz;
    const auto u2 = o.x * zz;
    const auto s1 = y * o.zzz;
    const auto s2 = o.y * zzz;
    const auto p_ = norm(u2 - u1);
    const auto r_ = norm(s2 - s1);
    if (is_zero(p_)) {
      if (is_zero(r_)) return dbl_pt();
      return inf();
    }
    const auto pp = sqr(p_);
    const auto ppp = p_ * pp;
    const auto q_ = u1 * pp;
    const auto x3 = fit<BS>(sqr(r_) - (ppp + dbl(q_)));
    const auto y3 = mul_sub(r_, q_ - x3, ppp, s1);
    return {x3, fit<BS>(y3), fit<BS>((zz * o.zz) * pp), fit<BS>((zzz * o.zzz) * ppp)};
  }
  // ---- the same operations on operands that stay in memory (LDS / global / private) --------------------------
  // Coordinates are loaded where a product consumes them and results are stored as soon as every input has been
  // read, so the live set is ~6 field elements instead of both points + the temporaries (an Fq2 addition of
  // BLS12-381 would otherwise need > 512 registers).  dst may alias a or b.
#if defined(__HIP_DEVICE_COMPILE__)
#define DG29_STAGE() asm volatile("" ::: "memory")   /* keep the loads where they are written */
#else
#define DG29_STAGE()
#endif
  DG_HD static void add_mem(XYZZ29* dst, const XYZZ29* a, const XYZZ29* b) {
    if (b->is_inf()) {
      if (dst != a) *dst = *a;
      return;
    }
    if (a->is_inf()) {
      if (dst != b) *dst = *b;
      return;
    }
    const auto u1 = a->x * b->zz;
    DG29_STAGE();
    const auto p_ = norm(b->x * a->zz - u1);
    DG29_STAGE();
    const auto s1 = a->y * b->zzz;
    DG29_STAGE();
    const auto r_ = norm(b->y * a->zzz - s1);
    DG29_STAGE();
    if (is_zero(p_)) {
      if (is_zero(r_)) {
        const XYZZ29 d = a->dbl_pt();
        *dst = d;
      } else {
        *dst = inf();
      }
      return;
    }
    const auto pp = sqr(p_);
    const auto ppp = p_ * pp;
    const auto zz3 = fit<BS>((a->zz * b->zz) * pp);
    DG29_STAGE();
    const auto zzz3 = fit<BS>((a->zzz * b->zzz) * ppp);
    DG29_STAGE();
    dst->zz = zz3;          // every coordinate of a and b has been read by now
    dst->zzz = zzz3;
    DG29_STAGE();
    const auto q_ = u1 * pp;
    const auto x3 = fit<BS>(sqr(r_) - (ppp + dbl(q_)));
    dst->x = x3;
    DG29_STAGE();
    dst->y = fit<BS>(mul_sub(r_, q_ - x3, ppp, s1));
  }
  // The same addition on operands behind ACCESSORS (get(coord) loads a coordinate where a product consumes it,
  // put(coord, v) stores one): the in-workgroup bucket tree of the accumulation kernels keeps its operands in LDS
  // columns ([coordinate word][lane]).  coord: 0 x, 1 y, 2 zz, 3 zzz.  dst may be the accessor of a.
  template <class D, class A, class B>
  DG_HD static void add_acc(const D& d, const A& a, const B& b) {
    const auto bzz = b.get(2);
    if (limbs_all_zero(bzz)) return;                          // (dst == a in every use: nothing to copy)
    const auto azz = a.get(2);
    if (limbs_all_zero(azz)) {
      d.put(0, b.get(0)); d.put(1, b.get(1)); d.put(2, bzz); d.put(3, b.get(3));
      return;
    }
    const auto u1 = a.get(0) * bzz;
    DG29_STAGE();
    const auto p_ = norm(b.get(0) * azz - u1);
    DG29_STAGE();
    const auto s1 = a.get(1) * b.get(3);
    DG29_STAGE();
    const auto r_ = norm(b.get(1) * a.get(3) - s1);
    DG29_STAGE();
    if (is_zero(p_)) {
      if (is_zero(r_)) {
        const XYZZ29 t = XYZZ29{a.get(0), a.get(1), a.get(2), a.get(3)}.dbl_cold();
        d.put(0, t.x); d.put(1, t.y); d.put(2, t.zz); d.put(3, t.zzz);
      } else {
        d.put(2, FO::zero());                                  // the identity: zz = 0
      }
      return;
    }
    const auto pp = sqr(p_);
    const auto ppp = p_ * pp;
    const auto zz3 = fit<BS>((azz * bzz) * pp);
    DG29_STAGE();
    const auto zzz3 = fit<BS>((a.get(3) * b.get(3)) * ppp);
    DG29_STAGE();
    d.put(2, zz3);          // every coordinate of a and b that is still needed has been read by now ...
    d.put(3, zzz3);
    DG29_STAGE();
    const auto q_ = u1 * pp;
    const auto x3 = fit<BS>(sqr(r_) - (ppp + dbl(q_)));
    d.put(0, x3);
    DG29_STAGE();
    d.put(1, fit<BS>(mul_sub(r_, q_ - x3, ppp, s1)));           // ... (s1 holds a's y)
  }
  // d += b with d in LDS columns and the SMALLEST live set: U1 = X1 ZZ2 and S1 = Y1 ZZZ2 overwrite X1 / Y1 in place as
  // soon as they exist, so that -- like the mixed addition of the accumulation loop -- only P, R, PP, PPP live across
  // the products (add_acc keeps U1 and S1 in registers next to them: 256 VGPRs + 0.8 KB of scratch per lane for Fq2).
  // b is read-only and intact throughout, so the rare a == b case doubles b.
  template <class D, class B>
  DG_HD static void add_into(const D& d, const B& b) {
    if (limbs_all_zero(b.get(2))) return;
    if (limbs_all_zero(d.get(2))) {
      d.put(0, b.get(0)); d.put(1, b.get(1)); d.put(2, b.get(2)); d.put(3, b.get(3));
      return;
    }
    d.put(0, fit<BS>(d.get(0) * b.get(2)));                   // U1
    DG29_STAGE();
    d.put(1, fit<BS>(d.get(1) * b.get(3)));                   // S1
    DG29_STAGE();
    const auto p_ = norm(b.get(0) * d.get(2) - d.get(0));
    DG29_STAGE();
    const auto r_ = norm(b.get(1) * d.get(3) - d.get(1));
    DG29_STAGE();
    if (is_zero(p_)) {
      if (is_zero(r_)) {
        const XYZZ29 t = XYZZ29{b.get(0), b.get(1), b.get(2), b.get(3)}.dbl_pt()   /* inline: a call makes the kernel take the callee's 256 VGPRs + its spills */;
        d.put(0, t.x); d.put(1, t.y); d.put(2, t.zz); d.put(3, t.zzz);
      } else {
        d.put(2, FO::zero());                                  // the identity: zz = 0
      }
      return;
    }
    const auto pp = sqr(p_);
    const auto ppp = p_ * pp;
    DG29_STAGE();
    d.put(2, fit<BS>((d.get(2) * b.get(2)) * pp));
    DG29_STAGE();
    d.put(3, fit<BS>((d.get(3) * b.get(3)) * ppp));
    DG29_STAGE();
    const auto q_ = d.get(0) * pp;
    DG29_STAGE();
    const auto x3 = fit<BS>(sqr(r_) - (ppp + dbl(q_)));
    DG29_STAGE();
    const auto y3 = fit<BS>(mul_sub(r_, q_ - x3, ppp, d.get(1)));
    d.put(0, x3);
    d.put(1, y3);
  }
  // doubling behind a call: the equal-operands branch of add_acc is rare, its code must not sit in the caller's loop
#if defined(__HIPCC__)
  __host__ __device__ __attribute__((noinline)) XYZZ29 dbl_cold() const { return dbl_pt(); }
#else
  __attribute__((noinline)) XYZZ29 dbl_cold() const { return dbl_pt(); }
#endif
  DG_HD static void dbl_mem(XYZZ29* dst, const XYZZ29* a) {
    if (a->is_inf()) {
      if (dst != a) *dst = *a;
      return;
    }
    const auto u = dbl(a->y);
    const auto v = sqr(u);
    const auto w = u * v;
    DG29_STAGE();
    const auto s = a->x * v;
    const auto xx = sqr(a->x);
    DG29_STAGE();
    const auto m = dbl(xx) + xx;
    const auto x3 = fit<BS>(sqr(m) - dbl(s));
    const auto y3 = fit<BS>(mul_sub(m, s - x3, w, a->y));
    DG29_STAGE();
    const auto zz3 = fit<BS>(v * a->zz);
    const auto zzz3 = fit<BS>(w * a->zzz);
    DG29_STAGE();
    dst->x = x3;
    dst->y = y3;
    dst->zz = zz3;
    dst->zzz = zzz3;
  }
#undef DG29_STAGE

  // -> the 32-bit-limb XYZZ of ec.h (arkworks Montgomery form, canonical): what the older kernels and the C ABI read
  DG_HD XYZZ<F> to_xyzz32() const {
    if (is_inf()) return XYZZ<F>::inf();
    return {FO::to32(x), FO::to32(y), FO::to32(zz), FO::to32(zzz)};
  }
  DG_HD static XYZZ29 from_xyzz32(const XYZZ<F>& p) {
    if (p.is_inf()) return inf();
    return {FO::from32(p.x), FO::from32(p.y), FO::from32(p.zz), FO::from32(p.zzz)};
  }
};

// arkworks-form affine point (C ABI layout) -> packed internal form, same byte size: x R32 -> x R, canonical
template <class F>
DG_HD void affine_to_internal(const Affine<F>& p, uint32_t* out_words) {
  using FO = FieldOf<F>;
  if (p.is_inf()) {
#pragma unroll
    for (int i = 0; i < 2 * FO::WORDS; i++) out_words[i] = 0;
    return;
  }
  FO::store_packed(FO::canon_of(FO::from32(p.x)), out_words);
  FO::store_packed(FO::canon_of(FO::from32(p.y)), out_words + FO::WORDS);
}

}  // namespace dg16
