// ark-ec short-Weierstrass point (de)compression (arkworks `CanonicalSerialize` with `Compress::Yes`; default SWFlags --
// BLS12-381's zcash form further down):
// little-endian x with the flags in the two top bits of the last byte (bit 7: y is the "negative" root, y > -y, for
// Fq2 compared on (c1, c0); bit 6: point at infinity).  ONE implementation for the host (proof.bin: serialize.hip,
// pinned by the reference's own proof.bin and the coordinates its CLI prints) and for the device (batched key
// files: ark_codec.hip) -- the device path is the pinned host code, compiled for gfx950.
#pragma once
#include <stdint.h>
#include <string.h>

#include "types.h"

#if defined(__HIPCC__)
#define DG_CODEC __host__ __device__ inline
#else
#define DG_CODEC inline
#endif

namespace dg16 {

// The same codec for every curve whose arkworks crate uses the DEFAULT short-Weierstrass serialisation (ark-ec 0.4
// `SWFlags`): BN254 (ark-bn254) and BLS12-377 (ark-bls12-377 -- the curve of the reference's d_msm / d_fft tests, whose
// MpcSerNet sends one compressed G per d_msm: dist-primitives/src/channel/mod.rs:14,49).
// BLS12-381 (BASELINE config 5's curve; not a dependency of the reference): ark-bls12-381 0.4 overrides the format with
// the zcash / IETF encoding (its curves/util.rs): BIG-endian x, three flag bits in the FIRST byte -- bit 7 "compressed"
// (always set here), bit 6 infinity, bit 5 "y is the lexicographically larger of (y, -y)" --, G2 as x.c1 || x.c0.  Same
// square roots, sign rule (Fq2 ordered on (c1, c0)) and subgroup checks as the default form; only the bytes differ.
template <int CURVE>
struct CodecT {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  using QP = typename Fq::Params;
  static constexpr int NL = Fq::NL;
  static constexpr int FB = NL * 4;                        // bytes of one base-field element
  static constexpr int BETA = Fq2Beta<QP>::value;          // Fq2 = Fq[u] / (u^2 + BETA)
  static constexpr bool Q3MOD4 = (QP::P[0] & 3u) == 3u;
  static constexpr bool ZCASH = CURVE == 1;               // BLS12-381: big-endian, flags in the first byte

  DG_CODEC static void canon(const Fq& a, uint32_t out[NL]) {
    Fq c = a.from_mont();
    for (int i = 0; i < NL; i++) out[i] = c.l[i];
  }
  DG_CODEC static int cmp(const uint32_t* a, const uint32_t* b) {
    for (int i = NL - 1; i >= 0; i--)
      if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
  }
  // ark: YIsNegative <=> y > -y
  DG_CODEC static bool is_neg(const Fq& y) {
    uint32_t a[NL], b[NL];
    canon(y, a);
    canon(y.neg(), b);
    return cmp(a, b) > 0;
  }
  DG_CODEC static bool is_neg(const Fq2& y) {   // QuadExtField ordering: c1 first, then c0
    uint32_t a[NL], b[NL];
    canon(y.c1, a);
    canon(y.c1.neg(), b);
    int c = cmp(a, b);
    if (c != 0) return c > 0;
    return is_neg(y.c0);
  }
  DG_CODEC static Fq pow_limbs(const Fq& base, const uint32_t e[NL]) {
    Fq acc = Fq::one();
    for (int i = NL * 32 - 1; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i / 32] >> (i % 32)) & 1) acc = acc * base;
    }
    return acc;
  }
  DG_CODEC static Fq small(int k) {                          // k >= 0 as a field element
    return Fq::from_u32((uint32_t)k);
  }
  // q = 3 mod 4 (BN254): a^((q + 1) / 4).  Otherwise (BLS12-377: q - 1 = 2^46 t) Tonelli-Shanks with the non-residue
  // -BETA (u^2 + BETA is irreducible, so -BETA is a non-square): z = (-BETA)^t generates the 2-Sylow subgroup.
  DG_CODEC static bool sqrt_fq(const Fq& a, Fq& r) {
    if constexpr (Q3MOD4) {
      uint32_t e[NL];
      uint64_t carry = 1;
      for (int i = 0; i < NL; i++) {
        uint64_t v = (uint64_t)QP::P[i] + carry;
        e[i] = (uint32_t)v;
        carry = v >> 32;
      }
      for (int i = 0; i < NL; i++) e[i] = (e[i] >> 2) | (i + 1 < NL ? e[i + 1] << 30 : 0);
      r = pow_limbs(a, e);
      return r.sqr() == a;
    } else {
      if (a.is_zero()) { r = a; return true; }
      // p - 1 = 2^S t, t odd
      uint32_t t[NL];
      for (int i = 0; i < NL; i++) t[i] = QP::P[i];
      t[0] -= 1;                                             // p is odd: no borrow
      int S = 0;
      while (!(t[0] & 1)) {
        for (int i = 0; i < NL; i++) t[i] = (t[i] >> 1) | (i + 1 < NL ? t[i + 1] << 31 : 0);
        S++;
      }
      uint32_t th[NL];                                       // (t - 1) / 2
      for (int i = 0; i < NL; i++) th[i] = (t[i] >> 1) | (i + 1 < NL ? t[i + 1] << 31 : 0);
      Fq z = pow_limbs(small(BETA).neg(), t);
      Fq w = pow_limbs(a, th);
      Fq x = a * w;                                          // a^((t + 1) / 2)
      Fq b = x * w;                                          // a^t
      int v = S;
      while (!(b == Fq::one())) {
        int k = 0;
        Fq b2 = b;
        while (!(b2 == Fq::one())) {
          b2 = b2.sqr();
          k++;
          if (k >= v) return false;                          // a is not a square
        }
        Fq ww = z;
        for (int i = 0; i < v - k - 1; i++) ww = ww.sqr();
        z = ww.sqr();
        b = b * z;
        x = x * ww;
        v = k;
      }
      r = x;
      return r.sqr() == a;
    }
  }
  // (x0 + x1 u)^2 = a0 + a1 u with u^2 = -BETA: x0^2 = (a0 +- sqrt(a0^2 + BETA a1^2)) / 2, x1 = a1 / (2 x0)
  DG_CODEC static bool sqrt_fq2(const Fq2& a, Fq2& r) {
    if (a.c1.is_zero()) {
      Fq t;
      if (sqrt_fq(a.c0, t)) { r = {t, Fq::zero()}; return true; }
      if (sqrt_fq(a.c0.neg() * small(BETA).inv(), t)) { r = {Fq::zero(), t}; return true; }   // (t u)^2 = -BETA t^2
      return false;
    }
    Fq n;
    if (!sqrt_fq(a.c0.sqr() + fq2_beta_mul(a.c1.sqr()), n)) return false;
    const Fq inv2 = small(2).inv();
    const Fq cand[2] = {(a.c0 + n) * inv2, (a.c0 - n) * inv2};
    for (const Fq& delta : cand) {
      Fq x0;
      if (!sqrt_fq(delta, x0) || x0.is_zero()) continue;
      Fq x1 = a.c1 * (x0 + x0).inv();
      Fq2 cnd = {x0, x1};
      if (cnd.sqr() == a) { r = cnd; return true; }
    }
    return false;
  }
  DG_CODEC static Fq b_g1() {
    Fq b;
    for (int i = 0; i < NL; i++) b.l[i] = CT::G1c::B[i];
    return b;
  }
  DG_CODEC static Fq2 b_g2() {
    Fq2 b;
    for (int i = 0; i < NL; i++) { b.c0.l[i] = CT::G2c::B_C0[i]; b.c1.l[i] = CT::G2c::B_C1[i]; }
    return b;
  }
  DG_CODEC static void put_fq(const Fq& a, uint8_t* out) {
    uint32_t c[NL];
    canon(a, c);
    memcpy(out, c, FB);
  }
  DG_CODEC static bool get_fq(const uint8_t* in, Fq& out) {   // canonical little-endian, must be < q
    uint32_t c[NL];
    memcpy(c, in, FB);
    if (cmp(c, QP::P) >= 0) return false;
    Fq t = Fq::zero();
    for (int i = 0; i < NL; i++) t.l[i] = c[i];
    out = t.to_mont();
    return true;
  }
  DG_CODEC static void put_fq_be(const Fq& a, uint8_t* out) {
    uint32_t c[NL];
    canon(a, c);
    for (int i = 0; i < FB; i++) out[i] = (uint8_t)(c[(FB - 1 - i) / 4] >> (8 * ((FB - 1 - i) % 4)));
  }
  DG_CODEC static bool get_fq_be(const uint8_t* in, Fq& out) {   // canonical big-endian, must be < q
    uint32_t c[NL];
    for (int i = 0; i < NL; i++) c[i] = 0;
    for (int i = 0; i < FB; i++) c[(FB - 1 - i) / 4] |= (uint32_t)in[i] << (8 * ((FB - 1 - i) % 4));
    if (cmp(c, QP::P) >= 0) return false;
    Fq t = Fq::zero();
    for (int i = 0; i < NL; i++) t.l[i] = c[i];
    out = t.to_mont();
    return true;
  }
  DG_CODEC static void encode(const Affine<Fq>& p, uint8_t* out) {
    memset(out, 0, FB);
    if constexpr (ZCASH) {
      if (p.is_inf()) { out[0] = 0xC0; return; }
      put_fq_be(p.x, out);
      out[0] |= is_neg(p.y) ? 0xA0 : 0x80;
      return;
    }
    if (p.is_inf()) { out[FB - 1] |= 0x40; return; }
    put_fq(p.x, out);
    if (is_neg(p.y)) out[FB - 1] |= 0x80;
  }
  DG_CODEC static void encode(const Affine<Fq2>& p, uint8_t* out) {
    memset(out, 0, 2 * FB);
    if constexpr (ZCASH) {
      if (p.is_inf()) { out[0] = 0xC0; return; }
      put_fq_be(p.x.c1, out);
      put_fq_be(p.x.c0, out + FB);
      out[0] |= is_neg(p.y) ? 0xA0 : 0x80;
      return;
    }
    if (p.is_inf()) { out[2 * FB - 1] |= 0x40; return; }
    put_fq(p.x.c0, out);
    put_fq(p.x.c1, out + FB);
    if (is_neg(p.y)) out[2 * FB - 1] |= 0x80;
  }
  template <class F>
  DG_CODEC static bool in_subgroup(const Affine<F>& p) {
    using Fr = typename CT::Fr;
    XYZZ<F> q = scalar_mul<F, Fr::NL>(XYZZ<F>::from_affine(p), Fr::Params::P);
    return q.is_inf();
  }
  // validate: order-r subgroup check (Validate::Yes).  BN254 G1 has cofactor 1: nothing to check there.
  DG_CODEC static int decode(const uint8_t* in, Affine<Fq>& p, bool validate) {
    uint8_t b[FB];
    memcpy(b, in, FB);
    bool neg, inf;
    Fq x;
    if constexpr (ZCASH) {
      if (!(b[0] & 0x80)) return 1;                          // an uncompressed encoding where a compressed one is read
      inf = b[0] & 0x40;
      neg = b[0] & 0x20;
      b[0] &= 0x1F;
      if (neg && inf) return 1;
      if (!get_fq_be(b, x)) return 2;
    } else {
      neg = b[FB - 1] & 0x80;
      inf = b[FB - 1] & 0x40;
      b[FB - 1] &= 0x3F;
      if (neg && inf) return 1;
      if (!get_fq(b, x)) return 2;
    }
    if (inf) {
      if (!x.is_zero()) return 1;
      p = Affine<Fq>::inf();
      return 0;
    }
    Fq y;
    if (!sqrt_fq(x.sqr() * x + b_g1(), y)) return 3;
    if (is_neg(y) != neg) y = y.neg();
    p = {x, y};
    if (validate && CURVE != 0 && !in_subgroup(p)) return 4;
    return 0;
  }
  DG_CODEC static int decode(const uint8_t* in, Affine<Fq2>& p, bool validate) {
    uint8_t b[2 * FB];
    memcpy(b, in, 2 * FB);
    bool neg, inf;
    Fq2 x;
    if constexpr (ZCASH) {
      if (!(b[0] & 0x80)) return 1;
      inf = b[0] & 0x40;
      neg = b[0] & 0x20;
      b[0] &= 0x1F;
      if (neg && inf) return 1;
      if (!get_fq_be(b, x.c1) || !get_fq_be(b + FB, x.c0)) return 2;
    } else {
      neg = b[2 * FB - 1] & 0x80;
      inf = b[2 * FB - 1] & 0x40;
      b[2 * FB - 1] &= 0x3F;
      if (neg && inf) return 1;
      if (!get_fq(b, x.c0) || !get_fq(b + FB, x.c1)) return 2;
    }
    if (inf) {
      if (!x.is_zero()) return 1;
      p = Affine<Fq2>::inf();
      return 0;
    }
    Fq2 y;
    if (!sqrt_fq2(x.sqr() * x + b_g2(), y)) return 3;
    if (is_neg(y) != neg) y = y.neg();
    p = {x, y};
    if (validate && !in_subgroup(p)) return 4;      // the twist has a cofactor
    return 0;
  }
};

// BN254 under the names the proof.bin codec (serialize.hip) and the verifier use
namespace codec {
using C = CodecT<0>;
using Fq = C::Fq;
using Fq2 = C::Fq2;
constexpr int NL = C::NL;
DG_CODEC bool sqrt_fq(const Fq& a, Fq& r) { return C::sqrt_fq(a, r); }
DG_CODEC bool sqrt_fq2(const Fq2& a, Fq2& r) { return C::sqrt_fq2(a, r); }
DG_CODEC bool is_neg(const Fq& y) { return C::is_neg(y); }
DG_CODEC bool is_neg(const Fq2& y) { return C::is_neg(y); }
DG_CODEC Fq b_g1() { return C::b_g1(); }
DG_CODEC Fq2 b_g2() { return C::b_g2(); }
DG_CODEC void put_fq(const Fq& a, uint8_t* out) { C::put_fq(a, out); }
DG_CODEC bool get_fq(const uint8_t* in, Fq& out) { return C::get_fq(in, out); }
DG_CODEC void encode(const Affine<Fq>& p, uint8_t* out) { C::encode(p, out); }
DG_CODEC void encode(const Affine<Fq2>& p, uint8_t* out) { C::encode(p, out); }
DG_CODEC int decode(const uint8_t* in, Affine<Fq>& p) { return C::decode(in, p, false); }
DG_CODEC int decode(const uint8_t* in, Affine<Fq2>& p, bool validate) { return C::decode(in, p, validate); }
}  // namespace codec
}  // namespace dg16
