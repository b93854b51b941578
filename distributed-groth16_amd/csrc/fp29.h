// Reduced-radix prime-field arithmetic for the hot kernels of gfx950 (bucket accumulation, bucket reduction, NTT).
//
// Why a second representation next to fp.h.  With 32-bit limbs every partial product of a Montgomery
// multiplication needs a carry instruction next to its v_mad_u64_u32 (128 mad + 128 addc + ~47 mov for 8 limbs;
// addc issues at 31 T lane-op/s, mad at 34 T: profiles/r1_ubench_instr_rate.txt), and every addition ends in a
// compare-and-subtract.  Here a field element is N limbs of W bits (W = 29 for the 254/255-bit fields, 28 for the
// 377/381-bit ones) in 32-bit words:
//   * a column of <= 2N products of < 2^58 fits a 64-bit accumulator, so a product is a plain chain of
//     v_mad_u64_u32 (2 N^2 = 162 for N = 9, no carries) + one mask/shift per column; a square with a doubled
//     operand needs N(N+1)/2 + N^2 = 126.  Measured (tools/ubench/montmul29_rate, profiles/r2_ubench_montmul29_rate.txt):
//     155-166 G products/s chip-wide against 104-114 for the 32-bit product, 190 G squares/s, and almost no
//     sensitivity to occupancy (164 G/s at ONE wave per SIMD against 88);
//   * R = 2^(W N) leaves SLACK = W N - bits(p) spare bits (7 for BN254 Fq, 11 for BLS12-381 Fq), so values live
//     in [0, B p) for small B and additions / subtractions are N independent v_add_u32 (subtraction adds a
//     multiple of p whose limbs dominate the subtrahend's); nothing is compared or conditionally subtracted.
//
// Bounds are part of the type: Fe<P, B, LU> holds a value < (B / 64) p whose limbs below the top one are
// < LU 2^W.  Every operation computes the bounds of its result at compile time and static_asserts that (a) no
// column of a product can overflow 64 bits, (b) no limb can overflow 32 bits -- a formula that compiles cannot
// overflow.  Products normalise an operand first when (a) would fail, so formulas are written once for every
// field.  Values are only congruent mod p until canon(); equality with zero is tested against every multiple
// of p below the bound (is_zero).
//
// Memory stays the arkworks layout at the C ABI (fp.h: 32-bit limbs, R = 2^(32 NL)); tables the library builds
// for itself hold x R mod p of THIS representation packed into the same NL words (to_internal / from_words).
// Plain C++: the same header runs on the host in tests/host_arith.  On the device the products alone are explicit
// instruction sequences (fp29_asm_gen.h, see there), checked instruction by instruction on the CPU
// (tests/test_fp29_asm_isa.py).
#pragma once
#include "fp.h"
#include "fp2.h"

namespace dg16 {

template <int N>
struct LimbArr {
  uint32_t v[N];
};

namespace rr {

template <class P>
constexpr int modulus_bits() {
  int top = P::NL - 1;
  while (top > 0 && P::P[top] == 0) top--;
  int b = 32;
  while (b > 0 && !((P::P[top] >> (b - 1)) & 1)) b--;
  return top * 32 + b;
}
template <class P>
constexpr int limb_bits() { return modulus_bits<P>() <= 256 ? 29 : 28; }
template <class P>
constexpr int limb_count() { return (modulus_bits<P>() + 5 + limb_bits<P>() - 1) / limb_bits<P>(); }

// NW 32-bit words -> N limbs of W bits (the value must fit)
template <int N, int W, int NW>
constexpr LimbArr<N> split(const uint32_t (&w)[NW]) {
  LimbArr<N> r{};
  for (int i = 0; i < N; i++) {
    const int bit = W * i, k = bit / 32, o = bit % 32;
    uint64_t v = 0;
    if (k < NW) v = w[k];
    if (k + 1 < NW) v |= (uint64_t)w[k + 1] << 32;
    r.v[i] = (uint32_t)(v >> o) & ((1u << W) - 1);
  }
  return r;
}
// 2^e mod p as NL + 1 words (repeated doubling; compile time only)
template <class P>
constexpr LimbArr<P::NL + 1> pow2_words(int e) {
  constexpr int NL = P::NL;
  LimbArr<NL + 1> x{};
  x.v[0] = 1;
  for (int s = 0; s < e; s++) {
    uint32_t carry = 0;
    for (int i = 0; i <= NL; i++) {
      const uint32_t nc = x.v[i] >> 31;
      x.v[i] = (x.v[i] << 1) | carry;
      carry = nc;
    }
    bool ge = x.v[NL] != 0;
    if (!ge) {
      ge = true;
      for (int i = NL - 1; i >= 0; i--)
        if (x.v[i] != P::P[i]) { ge = x.v[i] > P::P[i]; break; }
    }
    if (ge) {
      uint64_t borrow = 0;
      for (int i = 0; i <= NL; i++) {
        const uint64_t d = (uint64_t)x.v[i] - (i < NL ? P::P[i] : 0u) - borrow;
        x.v[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
      }
    }
  }
  return x;
}
template <class P>
constexpr LimbArr<limb_count<P>()> pow2_limbs(int e) {
  const LimbArr<P::NL + 1> w = pow2_words<P>(e);
  return split<limb_count<P>(), limb_bits<P>(), P::NL + 1>(w.v);
}
template <class P>
constexpr LimbArr<limb_count<P>()> p_limbs() { return split<limb_count<P>(), limb_bits<P>(), P::NL>(P::P); }

// k p as W-bit limbs with s 2^W moved from every limb into the one below (same value; every limb below the top is
// >= s 2^W - s, so subtracting limbs < s 2^W from it cannot go negative... see Fe::sub)
template <class P>
constexpr LimbArr<limb_count<P>()> kp_limbs(int k, int s) {
  constexpr int N = limb_count<P>(), W = limb_bits<P>();
  const LimbArr<N> p = p_limbs<P>();
  LimbArr<N> r{};
  uint64_t carry = 0;
  for (int i = 0; i < N; i++) {
    const uint64_t v = (uint64_t)p.v[i] * (uint32_t)k + carry;
    r.v[i] = (uint32_t)(v & ((1u << W) - 1));
    carry = v >> W;
  }
  r.v[N - 1] += (uint32_t)(carry << W);   // (never: k p fits N limbs for every k in use)
  for (int i = 0; i < N - 1; i++) {
    r.v[i] += (uint32_t)s << W;
    r.v[i + 1] -= (uint32_t)s;
  }
  return r;
}

// does k p in the borrowed form (s) dominate every limb bound lu 2^W - 1 below the top?
template <class P>
constexpr bool kp_dominates(int k, int s, int lu) {
  constexpr int N = limb_count<P>(), W = limb_bits<P>();
  const LimbArr<N> r = kp_limbs<P>(k, s);
  for (int i = 0; i < N - 1; i++)
    if ((uint64_t)r.v[i] + 1 < ((uint64_t)lu << W)) return false;
  return true;
}
template <class P>
constexpr int kp_borrow(int k, int lu) { return kp_dominates<P>(k, lu, lu) ? lu : lu + 1; }

}  // namespace rr

// per-field constants of the reduced-radix form
template <class P>
struct RR {
  static constexpr int NL = P::NL;
  static constexpr int BITS = rr::modulus_bits<P>();
  static constexpr int W = rr::limb_bits<P>();
  static constexpr int N = rr::limb_count<P>();
  static constexpr int RBITS = W * N;
  static constexpr int SLACK = RBITS - BITS;            // R / p > 2^SLACK
  static constexpr uint32_t MASK = (1u << W) - 1;
  static constexpr int COLCAP = 1 << (64 - 2 * W);      // N (sum of LU products + 1) must stay below this
  static constexpr int LIMBCAP = 1 << (32 - W);         // LU must stay <= this
  static constexpr uint32_t INV = P::INV & MASK;        // -p^-1 mod 2^W
  static constexpr LimbArr<N> PL = rr::p_limbs<P>();
  static constexpr LimbArr<N> ONE = rr::pow2_limbs<P>(RBITS);                // Montgomery one
  static constexpr LimbArr<N> R2 = rr::pow2_limbs<P>(2 * RBITS);             // x -> x R
  static constexpr LimbArr<N> FROM32 = rr::pow2_limbs<P>(2 * RBITS - 32 * NL);   // x R32 -> x R
  static constexpr LimbArr<N> TO32 = rr::pow2_limbs<P>(32 * NL);             // x R -> x R32
  static constexpr uint32_t PTOP = PL.v[N - 1];         // floor(p / 2^(W (N - 1)))
  // words of R mod p: the 32-bit Montgomery product (fp.h) of x R32 with this is x R (for table builders)
  static constexpr LimbArr<NL + 1> R_WORDS = rr::pow2_words<P>(RBITS);
  // ... and of R32^2 / R mod p: the 32-bit Montgomery product of x R with this is x R32
  static constexpr LimbArr<NL + 1> R32SQ_OVER_R_WORDS = rr::pow2_words<P>(64 * NL - RBITS);
};

template <class P, int K, int S>
struct KPConst {
  static constexpr LimbArr<RR<P>::N> v = rr::kp_limbs<P>(K, S);
};

constexpr int rr_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// value < (B / 64) p; limbs below the top < LU 2^W
template <class P, int B, int LU = 1>
struct Fe {
  using T = RR<P>;
  static constexpr int N = T::N;
  static constexpr int Bound = B, Limb = LU;
  static_assert(LU >= 1 && LU <= T::LIMBCAP, "limb bound exceeds 32 bits");
  static_assert(B >= 1 && B <= (64 << T::SLACK), "value bound exceeds R");
  uint32_t l[N];

  DG_HD static Fe zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  // same limbs under a TIGHTER static bound that the caller has proved by other means (a loop whose bound grows by
  // a known amount per iteration cannot carry it in a loop-invariant type): unchecked
  template <int B2, int LU2 = LU>
  DG_HD Fe<P, B2, LU2> unsafe_assume() const {
    Fe<P, B2, LU2> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = l[i];
    return r;
  }
  // same limbs under a looser static bound
  template <int B2, int LU2 = LU>
  DG_HD Fe<P, B2, LU2> as() const {
    static_assert(B2 >= B && LU2 >= LU, "as<>() can only loosen a bound");
    Fe<P, B2, LU2> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = l[i];
    return r;
  }
};

template <class P>
using FeOne = Fe<P, 64, 1>;   // canonical (< p)

// all limbs zero (the encoding of "no value": identity coordinates), NOT zero mod p -- see is_zero
template <class P, int B, int LU>
DG_HD bool limbs_all_zero(const Fe<P, B, LU>& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) acc |= a.l[i];
  return acc == 0;
}
template <class P, int B, int LU>
DG_HD Fe<P, B, LU> select(bool c, const Fe<P, B, LU>& a, const Fe<P, B, LU>& b) {   // c ? a : b
  Fe<P, B, LU> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

template <class P>
DG_HD Fe<P, 64, 1> fe_const(const LimbArr<RR<P>::N>& c) {
  Fe<P, 64, 1> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = c.v[i];
  return r;
}
template <class P>
DG_HD Fe<P, 64, 1> fe_one() { return fe_const<P>(RR<P>::ONE); }

// carry propagation: same value, limbs below the top < 2^W
template <class P, int B, int LU>
DG_HD Fe<P, B, 1> norm(const Fe<P, B, LU>& a) {
  using T = RR<P>;
  Fe<P, B, 1> r;
  if constexpr (LU == 1) {
#pragma unroll
    for (int i = 0; i < T::N; i++) r.l[i] = a.l[i];
  } else {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < T::N - 1; i++) {
      const uint32_t v = a.l[i] + c;
      r.l[i] = v & T::MASK;
      c = v >> T::W;
    }
    r.l[T::N - 1] = a.l[T::N - 1] + c;
  }
  return r;
}

template <class P, int B1, int L1, int B2, int L2>
DG_HD Fe<P, B1 + B2, L1 + L2> operator+(const Fe<P, B1, L1>& a, const Fe<P, B2, L2>& b) {
  Fe<P, B1 + B2, L1 + L2> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <class P, int B, int LU>
DG_HD Fe<P, 2 * B, 2 * LU> dbl(const Fe<P, B, LU>& a) {
  Fe<P, 2 * B, 2 * LU> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = a.l[i] << 1;
  return r;
}
// K a for a small constant K (the non-residue of a quadratic extension): N independent multiplications by K
template <int K, class P, int B, int LU>
DG_HD Fe<P, K * B, K * LU> mul_small(const Fe<P, B, LU>& a) {
  Fe<P, K * B, K * LU> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = a.l[i] * (uint32_t)K;
  return r;
}
// a - b + K with K = k p >= b whose limbs dominate b's: k = ceil(B2 / 64) + 1, s 2^W borrowed into every limb
// below the top (s = L2, or L2 + 1 when a limb of k p is too small for L2 to dominate)
template <class P, int B2, int L2>
struct SubK {
  static constexpr int K = rr_ceil_div(B2, 64) + 1;
  static constexpr int S = rr::kp_borrow<P>(K, L2);
  // top limb of K (>= K PTOP - S) must cover b's (< (B2 / 64) (PTOP + 1))
  static_assert((long long)RR<P>::PTOP * (K * 64 - B2) / 64 > S + B2 / 64 + 2, "subtraction constant too small");
};
template <class P, int B1, int L1, int B2, int L2>
DG_HD Fe<P, B1 + 64 * SubK<P, B2, L2>::K, L1 + SubK<P, B2, L2>::S + 1> operator-(const Fe<P, B1, L1>& a,
                                                                                const Fe<P, B2, L2>& b) {
  using K = SubK<P, B2, L2>;
  Fe<P, B1 + 64 * K::K, L1 + K::S + 1> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = a.l[i] + (KPConst<P, K::K, K::S>::v.v[i] - b.l[i]);
  return r;
}
template <class P, int B, int LU>
DG_HD Fe<P, 64 * SubK<P, B, LU>::K, SubK<P, B, LU>::S + 1> neg(const Fe<P, B, LU>& a) {
  using K = SubK<P, B, LU>;
  Fe<P, 64 * K::K, K::S + 1> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = KPConst<P, K::K, K::S>::v.v[i] - a.l[i];
  return r;
}

// bound of a Montgomery product of values < (B1/64) p, (B2/64) p (and optionally a second pair): out < p + T / R
template <class P>
constexpr int rr_mul_bound(long long b1b2) {
  return 64 + rr_ceil_div(b1b2, 64LL << RR<P>::SLACK) + 1;
}
template <class P>
constexpr bool rr_cols_fit(int lu_products) { return RR<P>::N * (lu_products + 1) < RR<P>::COLCAP; }

}  // namespace dg16
// Device form of the products below, as explicit instruction sequences (generated: tools/gen_fp29_asm.py).  The host --
// tests/host_arith -- and -DDG29_NO_ASM_MAD builds use the plain loops.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DG29_NO_ASM_MAD)
#define DG29_ASM_MAD 1
#include "fp29_asm_gen.h"
#define DG29_SEQ(kind, n) mont_asm_##kind##_##n
#endif
namespace dg16 {

namespace rr {
// (a b + c d) / R mod p on raw limbs (c, d may be null); the caller has checked the column bound
template <class P, bool DUAL>
DG_HD void mont_inl(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
  using T = RR<P>;
  constexpr int N = T::N;
#ifdef DG29_ASM_MAD
  if constexpr (N == 9 && T::W == 29) {
    if constexpr (DUAL) DG29_SEQ(dual, 9)<P>(r, a, b, c, d);
    else DG29_SEQ(mul, 9)<P>(r, a, b);
    return;
  } else if constexpr (N == 14 && T::W == 28) {
    if constexpr (DUAL) DG29_SEQ(dual, 14)<P>(r, a, b, c, d);
    else DG29_SEQ(mul, 14)<P>(r, a, b);
    return;
  }
#endif
  uint64_t acc = 0;
  uint32_t m[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      acc += (uint64_t)a[i] * b[k - i];
      if constexpr (DUAL) acc += (uint64_t)c[i] * d[k - i];
    }
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * T::PL.v[k - i];
    m[k] = ((uint32_t)acc * T::INV) & T::MASK;
    acc += (uint64_t)m[k] * T::PL.v[0];
    acc >>= T::W;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) {
      acc += (uint64_t)a[i] * b[k - i];
      if constexpr (DUAL) acc += (uint64_t)c[i] * d[k - i];
    }
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * T::PL.v[k - i];
    r[k - N] = (uint32_t)acc & T::MASK;
    acc >>= T::W;
  }
  r[N - 1] = (uint32_t)acc;
}
// a^2 / R mod p with the doubled operand: N (N + 1) / 2 + N^2 products
template <class P>
DG_HD void mont_sqr_inl(uint32_t* __restrict__ r, const uint32_t* a) {
  using T = RR<P>;
  constexpr int N = T::N;
#ifdef DG29_ASM_MAD
  if constexpr (N == 9 && T::W == 29) {
    DG29_SEQ(sqr, 9)<P>(r, a);
    return;
  } else if constexpr (N == 14 && T::W == 28) {
    DG29_SEQ(sqr, 14)<P>(r, a);
    return;
  }
#endif
  uint32_t a2[N];
#pragma unroll
  for (int i = 0; i < N; i++) a2[i] = a[i] << 1;
  uint64_t acc = 0;
  uint32_t m[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a[k - i];
    if (k % 2 == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * T::PL.v[k - i];
    m[k] = ((uint32_t)acc * T::INV) & T::MASK;
    acc += (uint64_t)m[k] * T::PL.v[0];
    acc >>= T::W;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; 2 * i < k; i++) acc += (uint64_t)a2[i] * a[k - i];
    if (k % 2 == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * T::PL.v[k - i];
    r[k - N] = (uint32_t)acc & T::MASK;
    acc >>= T::W;
  }
  r[N - 1] = (uint32_t)acc;
}
// (a b + c d + e f + g h) / R mod p with one reduction (the caller has checked the column bound)
template <class P>
DG_HD void mont4_inl(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d,
                     const uint32_t* e, const uint32_t* f, const uint32_t* g, const uint32_t* h) {
  using T = RR<P>;
  constexpr int N = T::N;
#ifdef DG29_ASM_MAD
  if constexpr (N == 9 && T::W == 29) {
    DG29_SEQ(quad, 9)<P>(r, a, b, c, d, e, f, g, h);
    return;
  } else if constexpr (N == 14 && T::W == 28) {
    DG29_SEQ(quad, 14)<P>(r, a, b, c, d, e, f, g, h);
    return;
  }
#endif
  uint64_t acc = 0;
  uint32_t m[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++)
      acc += (uint64_t)a[i] * b[k - i] + (uint64_t)c[i] * d[k - i] + (uint64_t)e[i] * f[k - i] + (uint64_t)g[i] * h[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * T::PL.v[k - i];
    m[k] = ((uint32_t)acc * T::INV) & T::MASK;
    acc += (uint64_t)m[k] * T::PL.v[0];
    acc >>= T::W;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++)
      acc += (uint64_t)a[i] * b[k - i] + (uint64_t)c[i] * d[k - i] + (uint64_t)e[i] * f[k - i] + (uint64_t)g[i] * h[k - i];
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * T::PL.v[k - i];
    r[k - N] = (uint32_t)acc & T::MASK;
    acc >>= T::W;
  }
  r[N - 1] = (uint32_t)acc;
}
// Translation units of LATENCY-bound kernels (the bucket reduction: chains of a few dozen dependent group operations
// at ~1 wave per SIMD) define DG29_OUTLINE_MUL: the products become calls (operands by value in VGPRs) so that a
// group addition is ~4 KB of code instead of ~45 KB (G1) / ~120 KB (BN254 G2) / ~250 KB (BLS12-381 G2) -- the
// instruction cache is 64 KB per two CUs, and the inlined kernels were instruction-fetch-bound (measured: 55 us per
// dependent G1 addition against ~8 us of issue time).  Throughput kernels (bucket accumulation) keep them inline.
#if defined(__HIP_DEVICE_COMPILE__) && defined(DG29_OUTLINE_MUL)
template <class P>
struct RawFe {
  uint32_t l[RR<P>::N];
};
template <class P>
__device__ __attribute__((noinline)) RawFe<P> mont_call(RawFe<P> a, RawFe<P> b) {
  RawFe<P> r;
  mont_inl<P, false>(r.l, a.l, b.l, nullptr, nullptr);
  return r;
}
template <class P>
__device__ __attribute__((noinline)) RawFe<P> mont_dual_call(RawFe<P> a, RawFe<P> b, RawFe<P> c, RawFe<P> d) {
  RawFe<P> r;
  mont_inl<P, true>(r.l, a.l, b.l, c.l, d.l);
  return r;
}
template <class P>
__device__ __attribute__((noinline)) RawFe<P> mont_sqr_call(RawFe<P> a) {
  RawFe<P> r;
  mont_sqr_inl<P>(r.l, a.l);
  return r;
}
template <class P, bool DUAL>
__device__ __forceinline__ void mont(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* c,
                                     const uint32_t* d) {
  RawFe<P> x, y, o;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) { x.l[i] = a[i]; y.l[i] = b[i]; }
  if constexpr (DUAL) {
    RawFe<P> z, w;
#pragma unroll
    for (int i = 0; i < RR<P>::N; i++) { z.l[i] = c[i]; w.l[i] = d[i]; }
    o = mont_dual_call<P>(x, y, z, w);
  } else {
    o = mont_call<P>(x, y);
  }
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r[i] = o.l[i];
}
template <class P>
__device__ __forceinline__ void mont_sqr(uint32_t* __restrict__ r, const uint32_t* a) {
  RawFe<P> x;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) x.l[i] = a[i];
  const RawFe<P> o = mont_sqr_call<P>(x);
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r[i] = o.l[i];
}
#else
template <class P, bool DUAL>
DG_HD void mont(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
  mont_inl<P, DUAL>(r, a, b, c, d);
}
template <class P>
DG_HD void mont_sqr(uint32_t* __restrict__ r, const uint32_t* a) { mont_sqr_inl<P>(r, a); }
#endif
}  // namespace rr

// a b / R.  An operand is normalised first when a column could overflow otherwise.
template <class P, int B1, int L1, int B2, int L2>
DG_HD Fe<P, rr_mul_bound<P>((long long)B1 * B2), 1> operator*(const Fe<P, B1, L1>& a, const Fe<P, B2, L2>& b) {
  if constexpr (rr_cols_fit<P>(L1 * L2)) {
    Fe<P, rr_mul_bound<P>((long long)B1 * B2), 1> r;
    rr::mont<P, false>(r.l, a.l, b.l, nullptr, nullptr);
    return r;
  } else if constexpr (L1 >= L2) {
    return norm(a) * b;
  } else {
    return a * norm(b);
  }
}
template <class P, int B, int LU>
DG_HD Fe<P, rr_mul_bound<P>((long long)B * B), 1> sqr(const Fe<P, B, LU>& a) {
  if constexpr (rr_cols_fit<P>(LU * LU) && 2 * LU <= RR<P>::LIMBCAP) {
    Fe<P, rr_mul_bound<P>((long long)B * B), 1> r;
    rr::mont_sqr<P>(r.l, a.l);
    return r;
  } else {
    return sqr(norm(a));
  }
}
// (a b + c d) / R with one reduction
template <class P, int B1, int L1, int B2, int L2, int B3, int L3, int B4, int L4>
DG_HD Fe<P, rr_mul_bound<P>((long long)B1 * B2 + (long long)B3 * B4), 1> mul_add(const Fe<P, B1, L1>& a, const Fe<P, B2, L2>& b,
                                                                                 const Fe<P, B3, L3>& c, const Fe<P, B4, L4>& d) {
  if constexpr (rr_cols_fit<P>(L1 * L2 + L3 * L4)) {
    Fe<P, rr_mul_bound<P>((long long)B1 * B2 + (long long)B3 * B4), 1> r;
    rr::mont<P, true>(r.l, a.l, b.l, c.l, d.l);
    return r;
  } else if constexpr (L1 > 1) {
    return mul_add(norm(a), b, c, d);
  } else if constexpr (L3 > 1) {
    return mul_add(a, b, norm(c), d);
  } else if constexpr (L2 > 1) {
    return mul_add(a, norm(b), c, d);
  } else {
    return mul_add(a, b, c, norm(d));
  }
}

// (a b + c d + e f + g h) / R with one reduction; the operand with the loosest limbs is normalised while a column
// could overflow
template <class P, int B1, int L1, int B2, int L2, int B3, int L3, int B4, int L4, int B5, int L5, int B6, int L6, int B7,
          int L7, int B8, int L8>
DG_HD auto mul_add4(const Fe<P, B1, L1>& a, const Fe<P, B2, L2>& b, const Fe<P, B3, L3>& c, const Fe<P, B4, L4>& d,
                    const Fe<P, B5, L5>& e, const Fe<P, B6, L6>& f, const Fe<P, B7, L7>& g, const Fe<P, B8, L8>& h) {
  if constexpr (rr_cols_fit<P>(L1 * L2 + L3 * L4 + L5 * L6 + L7 * L8)) {
    Fe<P, rr_mul_bound<P>((long long)B1 * B2 + (long long)B3 * B4 + (long long)B5 * B6 + (long long)B7 * B8), 1> r;
    rr::mont4_inl<P>(r.l, a.l, b.l, c.l, d.l, e.l, f.l, g.l, h.l);
    return r;
  } else {
    constexpr int m12 = L1 > L2 ? L1 : L2, m34 = L3 > L4 ? L3 : L4, m56 = L5 > L6 ? L5 : L6, m78 = L7 > L8 ? L7 : L8;
    constexpr int ma = m12 > m34 ? m12 : m34, mb = m56 > m78 ? m56 : m78, mx = ma > mb ? ma : mb;
    static_assert(mx > 1, "mul_add4: normalised operands must fit");
    if constexpr (L1 == mx) return mul_add4(norm(a), b, c, d, e, f, g, h);
    else if constexpr (L2 == mx) return mul_add4(a, norm(b), c, d, e, f, g, h);
    else if constexpr (L3 == mx) return mul_add4(a, b, norm(c), d, e, f, g, h);
    else if constexpr (L4 == mx) return mul_add4(a, b, c, norm(d), e, f, g, h);
    else if constexpr (L5 == mx) return mul_add4(a, b, c, d, norm(e), f, g, h);
    else if constexpr (L6 == mx) return mul_add4(a, b, c, d, e, norm(f), g, h);
    else if constexpr (L7 == mx) return mul_add4(a, b, c, d, e, f, norm(g), h);
    else return mul_add4(a, b, c, d, e, f, g, norm(h));
  }
}
// (a b - c d) / R with ONE reduction: c is negated limb by limb (N subtractions from a multiple of p that dominates
// it) and rides as the second product of a dual product -- the Y3 of every group addition / doubling is of this form
// (R (Q - X3) - PPP Y1): nine reductions per mixed addition instead of ten.  Round 4: for the 14-limb fields too -- their
// G1 accumulation then runs at TWO waves per SIMD (178 VGPRs, no scratch; unfused at three it spilled 44-56 B per
// lane when fused): measured on BLS12-381 2^20, same box, same call, A's accumulation 2.97 -> 2.83 ms, proof 39.1 ->
// 37.9 ms (profiles/r4b_ab_variants.md).
template <class P>
constexpr bool rr_fuse_mul_sub() { return true; }
template <class P, int B1, int L1, int B2, int L2, int B3, int L3, int B4, int L4>
DG_HD auto mul_sub(const Fe<P, B1, L1>& a, const Fe<P, B2, L2>& b, const Fe<P, B3, L3>& c, const Fe<P, B4, L4>& d) {
  if constexpr (rr_fuse_mul_sub<P>()) return mul_add(a, b, neg(c), d);
  else return a * b - d * c;
}

// value < (B / 64) p  ->  the same residue below (66 / 64) p: one quotient estimate from the top limb
// (q = floor(top / (PTOP + 1)) <= floor(v / p), and v - q p < p (1 + (B / 64 + 1) / PTOP)), one multiply-subtract pass
template <class P, int B, int LU>
DG_HD Fe<P, 66, 1> reduce(const Fe<P, B, LU>& a_) {
  using T = RR<P>;
  const Fe<P, B, 1> a = norm(a_);
  static_assert((long long)(B / 64 + 2) * 64 < (long long)T::PTOP, "reduce(): top limb too coarse for this bound");
  const uint32_t q = a.l[T::N - 1] / (T::PTOP + 1);
  Fe<P, 66, 1> r;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < T::N - 1; i++) {
    const int64_t v = (int64_t)a.l[i] - (int64_t)((uint64_t)q * T::PL.v[i]) + carry;
    r.l[i] = (uint32_t)v & T::MASK;
    carry = v >> T::W;
  }
  r.l[T::N - 1] = (uint32_t)((int64_t)a.l[T::N - 1] - (int64_t)((uint64_t)q * T::PL.v[T::N - 1]) + carry);
  return r;
}
// storage helper: normalised and below (BS / 64) p, reducing only when the static bound requires it
template <int BS, class P, int B, int LU>
DG_HD Fe<P, BS, 1> fit(const Fe<P, B, LU>& a) {
  static_assert(BS >= 66, "storage bound below what reduce() guarantees");
  if constexpr (B <= BS) return norm(a).template as<BS, 1>();
  else return reduce(a).template as<BS, 1>();
}

// a == 0 (mod p) for a normalised value below J p: a = k p for some k < J, and the low limb rejects almost always; the
// full comparison runs only for lanes whose low limb matches a candidate.
// 9-limb fields: the low limb NAMES the only candidate -- a.l[0] = k p[0] mod 2^W  <=>  k = -(a.l[0] INV) mod 2^W
// (INV = -p^-1) -- one multiplication, a mask and a compare (the accumulation loop asks this once per addition with
// J = 10: the chain of ten compares and scalar branches it replaces was 680 SALU + 120 VALU instructions of the G1 loop).
// 14-limb fields keep the compare chain: with the one-compare filter hipcc's schedule of their accumulation loop needs
// more than the 168 registers of three waves (a dozen scratch accesses per iteration; none with the chain).
template <class P, int B>
DG_HD bool is_zero(const Fe<P, B, 1>& a) {
  using T = RR<P>;
  constexpr int J = rr_ceil_div(B, 64);   // candidates 0, p, .., (J - 1) p  (value < (B/64) p <= J p)
  if constexpr (T::N <= 9) {
    const uint32_t k = (0u - a.l[0] * T::INV) & T::MASK;
    if (k >= (uint32_t)J) return false;
  } else {
    bool hit = false;
#pragma unroll
    for (int j = 0; j < J; j++) {
      const uint32_t low = (uint32_t)(((uint64_t)T::PL.v[0] * (uint32_t)j) & T::MASK);
      hit = hit || (a.l[0] == low);
    }
    if (!hit) return false;
  }
  for (int j = 0; j < J; j++) {
    uint64_t carry = 0;
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < T::N; i++) {
      const uint64_t v = (uint64_t)T::PL.v[i] * (uint32_t)j + carry;
      const uint32_t limb = i < T::N - 1 ? (uint32_t)(v & T::MASK) : (uint32_t)v;
      carry = v >> T::W;
      diff |= limb ^ a.l[i];
    }
    if (diff == 0) return true;
  }
  return false;
}
template <class P, int B, int LU>
DG_HD bool is_zero(const Fe<P, B, LU>& a) { return is_zero(norm(a)); }
// The same test with the full comparison as a ROLLED loop over the candidates (cold code; unrolled it is ~420
// instructions per call for a 14-limb field: 1 700 inside the step loop of msm_accumulate_steps_kernel, whose code
// must fit the instruction cache).
template <class P, int B>
DG_HD bool is_zero_compact(const Fe<P, B, 1>& a) {
  using T = RR<P>;
  constexpr int J = rr_ceil_div(B, 64);
  bool hit = false;
#pragma unroll
  for (int j = 0; j < J; j++) {
    const uint32_t low = (uint32_t)(((uint64_t)T::PL.v[0] * (uint32_t)j) & T::MASK);
    hit = hit || (a.l[0] == low);
  }
  if (!hit) return false;
#pragma unroll 1
  for (int j = 0; j < J; j++) {
    uint64_t carry = 0;
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < T::N; i++) {
      const uint64_t v = (uint64_t)T::PL.v[i] * (uint32_t)j + carry;
      const uint32_t limb = i < T::N - 1 ? (uint32_t)(v & T::MASK) : (uint32_t)v;
      carry = v >> T::W;
      diff |= limb ^ a.l[i];
    }
    if (diff == 0) return true;
  }
  return false;
}

// fully reduced representative in [0, p)
template <class P, int B, int LU>
DG_HD Fe<P, 64, 1> canon(const Fe<P, B, LU>& a_) {
  using T = RR<P>;
  Fe<P, 66, 1> a = reduce(a_);          // < 2 p: at most one subtraction left
  uint32_t s[T::N];
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < T::N - 1; i++) {
    const int64_t v = (int64_t)a.l[i] - (int64_t)T::PL.v[i] + carry;
    s[i] = (uint32_t)v & T::MASK;
    carry = v >> T::W;
  }
  const int64_t top = (int64_t)a.l[T::N - 1] - (int64_t)T::PL.v[T::N - 1] + carry;
  s[T::N - 1] = (uint32_t)top;
  const bool ge = top >= 0;
  Fe<P, 64, 1> r;
#pragma unroll
  for (int i = 0; i < T::N; i++) r.l[i] = ge ? s[i] : a.l[i];
  return r;
}

// ---- packed words <-> limbs ---------------------------------------------------------------------------------
// NL words holding a value < p (a table entry in internal Montgomery form, or an R32-form element about to be
// converted) -> limbs
template <class P>
DG_HD Fe<P, 64, 1> fe_from_words(const uint32_t* w) {
  using T = RR<P>;
  Fe<P, 64, 1> r;
#pragma unroll
  for (int i = 0; i < T::N; i++) {
    const int bit = T::W * i, k = bit / 32, o = bit % 32;
    uint32_t v = 0;
    if (k < T::NL) {
      v = w[k] >> o;
      if (o + T::W > 32 && k + 1 < T::NL) v |= w[k + 1] << (32 - o);
    }
    r.l[i] = v & T::MASK;
  }
  return r;
}
// canonical limbs -> NL words
template <class P>
DG_HD void fe_to_words(const Fe<P, 64, 1>& a, uint32_t* w) {
  using T = RR<P>;
#pragma unroll
  for (int k = 0; k < T::NL; k++) {
    // word k covers bits [32 k, 32 k + 32)
    const int lo = (32 * k) / T::W;
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i = lo + j;
      if (i < T::N) {
        const int sh = T::W * i - 32 * k;   // may be negative for j = 0
        if (sh >= 0) {
          if (sh < 32) v |= (uint64_t)a.l[i] << sh;
        } else {
          v |= (uint64_t)a.l[i] >> (-sh);
        }
      }
    }
    w[k] = (uint32_t)v;
  }
}
// arkworks memory (x R32 mod p, fp.h) -> internal x R, and back (canonical words)
template <class P>
DG_HD Fe<P, rr_mul_bound<P>(64 * 64), 1> fe_from_fp(const Fp<P>& x) {
  return fe_from_words<P>(x.l) * fe_const<P>(RR<P>::FROM32);
}
template <class P, int B, int LU>
DG_HD Fp<P> fe_to_fp(const Fe<P, B, LU>& a) {
  Fp<P> r;
  fe_to_words<P>(canon(a * fe_const<P>(RR<P>::TO32)), r.l);
  return r;
}
// internal form, packed and canonical (what the library's own tables hold)
template <class P, int B, int LU>
DG_HD void fe_store_packed(const Fe<P, B, LU>& a, uint32_t* w) { fe_to_words<P>(canon(a), w); }

// ---- quadratic extension u^2 = -BETA (fp2.h: BETA = 1 for BN254 / BLS12-381, 5 for BLS12-377) -------------------
template <class P, int B, int LU = 1>
struct Fe2 {
  Fe<P, B, LU> c0, c1;
  static constexpr int Bound = B, Limb = LU;
  template <int B2, int LU2 = LU>
  DG_HD Fe2<P, B2, LU2> as() const { return {c0.template as<B2, LU2>(), c1.template as<B2, LU2>()}; }
};
template <class P, int B, int LU>
DG_HD bool limbs_all_zero(const Fe2<P, B, LU>& a) { return limbs_all_zero(a.c0) && limbs_all_zero(a.c1); }
template <class P, int B, int LU>
DG_HD Fe2<P, B, LU> select(bool c, const Fe2<P, B, LU>& a, const Fe2<P, B, LU>& b) {
  return {select(c, a.c0, b.c0), select(c, a.c1, b.c1)};
}
template <class P, int B1, int L1, int B2, int L2>
DG_HD auto operator+(const Fe2<P, B1, L1>& a, const Fe2<P, B2, L2>& b) -> Fe2<P, B1 + B2, L1 + L2> {
  return {a.c0 + b.c0, a.c1 + b.c1};
}
template <class P, int B1, int L1, int B2, int L2>
DG_HD auto operator-(const Fe2<P, B1, L1>& a, const Fe2<P, B2, L2>& b)
    -> Fe2<P, B1 + 64 * SubK<P, B2, L2>::K, L1 + SubK<P, B2, L2>::S + 1> {
  return {a.c0 - b.c0, a.c1 - b.c1};
}
template <class P, int B, int LU>
DG_HD auto dbl(const Fe2<P, B, LU>& a) -> Fe2<P, 2 * B, 2 * LU> { return {dbl(a.c0), dbl(a.c1)}; }
template <class P, int B, int LU>
DG_HD auto neg(const Fe2<P, B, LU>& a) -> Fe2<P, 64 * SubK<P, B, LU>::K, SubK<P, B, LU>::S + 1> {
  return {neg(a.c0), neg(a.c1)};
}
template <class P, int B, int LU>
DG_HD Fe2<P, B, 1> norm(const Fe2<P, B, LU>& a) { return {norm(a.c0), norm(a.c1)}; }
template <int BS, class P, int B, int LU>
DG_HD Fe2<P, BS, 1> fit(const Fe2<P, B, LU>& a) { return {fit<BS>(a.c0), fit<BS>(a.c1)}; }
template <class P, int B, int LU>
DG_HD bool is_zero(const Fe2<P, B, LU>& a) { return is_zero(a.c0) && is_zero(a.c1); }
template <class P, int B>
DG_HD bool is_zero_compact(const Fe2<P, B, 1>& a) { return is_zero_compact(a.c0) && is_zero_compact(a.c1); }
// (a0 b0 - BETA a1 b1) + (a0 b1 + a1 b0) u: two dual products, one reduction each; b.c1 is the component negated
// (pass the tighter operand second), BETA rides on a.c1 (N small multiplications)
template <class P, int B1, int L1, int B2, int L2>
DG_HD auto operator*(const Fe2<P, B1, L1>& a, const Fe2<P, B2, L2>& b) {
  constexpr int BETA = Fq2Beta<P>::value;
  const auto nb1 = neg(b.c1);
  const auto c1 = mul_add(a.c0, b.c1, a.c1, b.c0);
  if constexpr (BETA == 1) {
    const auto c0 = mul_add(a.c0, b.c0, a.c1, nb1);
    constexpr int BO = decltype(c0)::Bound > decltype(c1)::Bound ? decltype(c0)::Bound : decltype(c1)::Bound;
    return Fe2<P, BO, 1>{c0.template as<BO, 1>(), c1.template as<BO, 1>()};
  } else {
    const auto c0 = mul_add(a.c0, b.c0, mul_small<BETA>(norm(a.c1)), nb1);
    constexpr int BO = decltype(c0)::Bound > decltype(c1)::Bound ? decltype(c0)::Bound : decltype(c1)::Bound;
    return Fe2<P, BO, 1>{c0.template as<BO, 1>(), c1.template as<BO, 1>()};
  }
}
// (a0 + a1)(a0 - BETA a1) + (BETA - 1) a0 a1  +  2 a0 a1 u   [= a0^2 - BETA a1^2 + 2 a0 a1 u]
template <class P, int B, int LU>
DG_HD auto sqr(const Fe2<P, B, LU>& a_) {
  constexpr int BETA = Fq2Beta<P>::value;
  const auto a = norm(a_);
  if constexpr (BETA == 1) {
    const auto c0 = (a.c0 + a.c1) * (a.c0 + neg(a.c1));
    const auto c1 = dbl(a.c0) * a.c1;
    constexpr int BO = decltype(c0)::Bound > decltype(c1)::Bound ? decltype(c0)::Bound : decltype(c1)::Bound;
    return Fe2<P, BO, 1>{c0.template as<BO, 1>(), c1.template as<BO, 1>()};
  } else {
    const auto v = a.c0 * a.c1;
    const auto c0 = norm((a.c0 + a.c1) * (a.c0 + neg(mul_small<BETA>(a.c1))) + mul_small<BETA - 1>(v));
    const auto c1 = norm(dbl(v));
    constexpr int BO = decltype(c0)::Bound > decltype(c1)::Bound ? decltype(c0)::Bound : decltype(c1)::Bound;
    return Fe2<P, BO, 1>{c0.template as<BO, 1>(), c1.template as<BO, 1>()};
  }
}

// a b - c d in the quadratic extension.  9-limb fields, inline products: each component is ONE four-product sum
//   (a0 b0 + (BETA a1)(-b1) + (-c0) d0 + (BETA c1) d1)  +  (a0 b1 + a1 b0 + (-c0) d1 + (-c1) d0) u
// (two reductions instead of four; operands normalised first so that the four products of a column fit 64 bits).
// Otherwise two products and a subtraction: the 14-limb kernels sit at their register limits (eight operands live at
// once), and with out-of-line products (DG29_OUTLINE_MUL) eight operands by value do not fit the argument registers.
template <class P, int B1, int L1, int B2, int L2, int B3, int L3, int B4, int L4>
DG_HD auto mul_sub(const Fe2<P, B1, L1>& a_, const Fe2<P, B2, L2>& b_, const Fe2<P, B3, L3>& c_, const Fe2<P, B4, L4>& d_) {
#if !defined(DG29_OUTLINE_MUL) && !defined(DG29_NO_QUAD)
  if constexpr (rr_fuse_mul_sub<P>()) {
    constexpr int BETA = Fq2Beta<P>::value;
    const auto a = norm(a_);
    const auto b = norm(b_);
    const auto c = norm(c_);
    const auto d = norm(d_);
    const auto nb1 = neg(b.c1);
    const auto nc0 = neg(c.c0);
    const auto nc1 = neg(c.c1);
    const auto r1 = mul_add4(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0);
    if constexpr (BETA == 1) {
      const auto r0 = mul_add4(a.c0, b.c0, a.c1, nb1, nc0, d.c0, c.c1, d.c1);
      constexpr int BO = decltype(r0)::Bound > decltype(r1)::Bound ? decltype(r0)::Bound : decltype(r1)::Bound;
      return Fe2<P, BO, 1>{r0.template as<BO, 1>(), r1.template as<BO, 1>()};
    } else {
      const auto r0 = mul_add4(a.c0, b.c0, mul_small<BETA>(a.c1), nb1, nc0, d.c0, mul_small<BETA>(c.c1), d.c1);
      constexpr int BO = decltype(r0)::Bound > decltype(r1)::Bound ? decltype(r0)::Bound : decltype(r1)::Bound;
      return Fe2<P, BO, 1>{r0.template as<BO, 1>(), r1.template as<BO, 1>()};
    }
  } else {
    return a_ * b_ - d_ * c_;
  }
#else
  return a_ * b_ - d_ * c_;      // (d c: the operand order the Fq2 kernels were measured with -- call sites pass c = PPP, d = Y1)
#endif
}

}  // namespace dg16
