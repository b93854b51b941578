// Prime-field arithmetic for gfx950: Montgomery form, 32-bit limbs (8 for 254/255-bit fields, 12 for
// the 381/377-bit base fields), R = 2^(32*NL) -- byte-identical to the arkworks in-memory BigInt
// (4/6 x u64 LE) the reference hands over at dist-primitives/src/dmsm/mod.rs:82 and
// dist-primitives/src/dfft/mod.rs:98-182.
//
// gfx950 notes (measured, tools/ubench): v_mad_u64_u32 issues at ~0.6x the rate of a plain 32-bit
// VALU op (NOT quarter rate), so the multiply is written as product scanning with one
// v_mad_u64_u32 + one v_addc per partial product (128+128+8 for NL = 8), which is 1.33x faster
// than what hipcc makes of the C++ CIOS loop (286 v_mov + 130 v_lshl_add_u64 of glue).
// No MFMA: this is carry-chained integer arithmetic, not a dense contraction.
//
// The same header compiles for the host (plain C++ path) so the arithmetic can be unit-tested
// without a GPU; the product never runs it there.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DG_HD __host__ __device__ __forceinline__
#else
#define DG_HD inline
#endif

namespace dg16 {

#if defined(__HIP_DEVICE_COMPILE__)
// acc(64) += x*y ; carry-out accumulated in c2
#define DG_MADC(acc, c2, x, y)                                                            \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, vcc"        \
      : "+v"(acc), "+v"(c2) : "v"(x), "v"(y) : "vcc")
#define DG_MADC_S(acc, c2, x, y)                                                          \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, vcc"        \
      : "+v"(acc), "+v"(c2) : "v"(x), "s"(y) : "vcc")
// first product of a column: c2 is SET to the carry (no zeroing move at every column hand-over)
#define DG_MADC0(acc, c2, x, y)                                                           \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc"         \
      : "+v"(acc), "=v"(c2) : "v"(x), "v"(y) : "vcc")
#endif

template <class P>
struct alignas(16) Fp {
  static constexpr int NL = P::NL;
  using Params = P;
  uint32_t l[NL];

  DG_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    return r;
  }
  DG_HD static Fp one() {  // R mod p
    Fp r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = P::R[i];
    return r;
  }
  DG_HD static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = P::R2[i];
    return r;
  }
  DG_HD bool is_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) acc |= l[i];
    return acc == 0;
  }
  DG_HD bool operator==(const Fp& o) const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) acc |= l[i] ^ o.l[i];
    return acc == 0;
  }
  DG_HD bool operator!=(const Fp& o) const { return !(*this == o); }

  // r = t - p if t >= p (t < 2p given as NL limbs + `top` overflow word)
  DG_HD static Fp reduce_once(const uint32_t* t, uint32_t top) {
    uint32_t s[NL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      uint64_t d = (uint64_t)t[i] - P::P[i] - borrow;
      s[i] = (uint32_t)d;
      borrow = (uint32_t)(d >> 32) & 1;
    }
    bool ge = (top != 0) || !borrow;
    Fp r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = ge ? s[i] : t[i];
    return r;
  }

  DG_HD friend Fp operator+(const Fp& a, const Fp& b) {
    uint32_t t[NL];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      uint64_t s = (uint64_t)a.l[i] + b.l[i] + carry;
      t[i] = (uint32_t)s;
      carry = (uint32_t)(s >> 32);
    }
    return reduce_once(t, carry);  // p < 2^(32NL-1): carry is always 0, kept for generality
  }
  DG_HD friend Fp operator-(const Fp& a, const Fp& b) {
    uint32_t t[NL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
      t[i] = (uint32_t)d;
      borrow = (uint32_t)(d >> 32) & 1;
    }
    uint32_t mask = 0u - borrow;  // add p back if we went negative
    Fp r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      uint64_t s = (uint64_t)t[i] + (P::P[i] & mask) + carry;
      r.l[i] = (uint32_t)s;
      carry = (uint32_t)(s >> 32);
    }
    return r;
  }
  DG_HD Fp neg() const {
    if (is_zero()) return *this;
    Fp r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      uint64_t d = (uint64_t)P::P[i] - l[i] - borrow;
      r.l[i] = (uint32_t)d;
      borrow = (uint32_t)(d >> 32) & 1;
    }
    return r;
  }
  DG_HD Fp dbl() const { return *this + *this; }

  // Montgomery product a*b*R^-1 mod p
  DG_HD friend Fp operator*(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // product scanning; (c2:acc) is a 96-bit column accumulator
    uint64_t acc = 0;
    uint32_t c2;
    uint32_t m[NL];
    uint32_t t[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      // column k: a_0 b_k first (sets the carry word), then the remaining a_i b_(k-i) and m_i p_(k-i)
      DG_MADC0(acc, c2, a.l[0], b.l[k]);
#pragma unroll
      for (int i = 1; i <= k; i++) DG_MADC(acc, c2, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = 0; i < k; i++) DG_MADC_S(acc, c2, m[i], P::P[k - i]);
      m[k] = (uint32_t)acc * P::INV;
      DG_MADC_S(acc, c2, m[k], P::P[0]);
      acc = (acc >> 32) | ((uint64_t)c2 << 32);
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
      DG_MADC0(acc, c2, a.l[k - NL + 1], b.l[NL - 1]);
#pragma unroll
      for (int i = k - NL + 2; i < NL; i++) DG_MADC(acc, c2, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++) DG_MADC_S(acc, c2, m[i], P::P[k - i]);
      t[k - NL] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)c2 << 32);
    }
    t[NL - 1] = (uint32_t)acc;          // column 2NL-1 holds no products
    acc >>= 32;
    return reduce_once(t, (uint32_t)acc);
#else
    // portable CIOS (host-side unit tests only)
    uint32_t t[NL + 2];
    for (int i = 0; i < NL + 2; i++) t[i] = 0;
    for (int i = 0; i < NL; i++) {
      uint32_t carry = 0;
      for (int j = 0; j < NL; j++) {
        uint64_t acc = (uint64_t)a.l[j] * b.l[i] + t[j] + carry;
        t[j] = (uint32_t)acc;
        carry = (uint32_t)(acc >> 32);
      }
      uint64_t acc = (uint64_t)t[NL] + carry;
      t[NL] = (uint32_t)acc;
      t[NL + 1] = (uint32_t)(acc >> 32);
      uint32_t m = t[0] * P::INV;
      acc = (uint64_t)m * P::P[0] + t[0];
      carry = (uint32_t)(acc >> 32);
      for (int j = 1; j < NL; j++) {
        acc = (uint64_t)m * P::P[j] + t[j] + carry;
        t[j - 1] = (uint32_t)acc;
        carry = (uint32_t)(acc >> 32);
      }
      acc = (uint64_t)t[NL] + carry;
      t[NL - 1] = (uint32_t)acc;
      t[NL] = t[NL + 1] + (uint32_t)(acc >> 32);
    }
    return reduce_once(t, t[NL]);
#endif
  }
  DG_HD Fp sqr() const { return *this * *this; }
  // Out-of-line product for code that is register-bound rather than call-bound (the G2 bucket kernel:
  // with every base-field multiplication inlined hipcc needs 449 registers = ONE wave per SIMD).
#if defined(__HIPCC__)
  static __host__ __device__ __attribute__((noinline)) Fp mul_call(Fp a, Fp b) { return a * b; }
#else
  static __attribute__((noinline)) Fp mul_call(Fp a, Fp b) { return a * b; }
#endif

  DG_HD Fp to_mont() const { return *this * r2(); }  // canonical integer (< p) -> Montgomery
  DG_HD Fp from_mont() const {                        // Montgomery -> canonical integer
    Fp o = zero();
    o.l[0] = 1;
    return *this * o;
  }
  DG_HD static Fp from_u32(uint32_t v) {
    Fp o = zero();
    o.l[0] = v;
    return o.to_mont();
  }
  DG_HD static Fp select(bool c, const Fp& a, const Fp& b) {  // c ? a : b
    Fp r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
  }
  // a^e for a small public exponent
  DG_HD Fp pow_u64(uint64_t e) const {
    Fp acc = one(), base = *this;
    while (e) {
      if (e & 1) acc = acc * base;
      base = base.sqr();
      e >>= 1;
    }
    return acc;
  }
  // limb `i` of p - 2 (borrow-propagated at compile time; BLS12-377 Fq has low word 1)
  static constexpr uint32_t pm2_limb(int i) {
    uint32_t borrow = 2, out = 0;
    for (int k = 0; k <= i; k++) {
      uint64_t d = (uint64_t)P::P[k] - borrow;
      out = (uint32_t)d;
      borrow = (uint32_t)(d >> 32) & 1;
    }
    return out;
  }
  // Fermat inversion a^(p-2) (inv(0) = 0).  Serial: only used in O(1)-sized tails.
  DG_HD Fp inv() const {
    Fp acc = one();
    bool started = false;
#pragma unroll
    for (int li = NL - 1; li >= 0; li--) {
      const uint32_t w = pm2_limb(li);
      for (int b = 31; b >= 0; b--) {
        if (started) acc = acc.sqr();
        if ((w >> b) & 1) {
          acc = acc * *this;
          started = true;
        }
      }
    }
    return acc;
  }
};

}  // namespace dg16
