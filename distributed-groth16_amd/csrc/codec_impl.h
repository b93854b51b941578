// ark-ec short-Weierstrass point (de)compression for BN254 (arkworks `CanonicalSerialize` with `Compress::Yes`):
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
namespace codec {

using Fq = Fp<bn254_fq_params>;
using Fq2 = Fp2<Fq>;
constexpr int NL = Fq::NL;

DG_CODEC void canon(const Fq& a, uint32_t out[NL]) {
  Fq c = a.from_mont();
  for (int i = 0; i < NL; i++) out[i] = c.l[i];
}
DG_CODEC int cmp(const uint32_t* a, const uint32_t* b) {
  for (int i = NL - 1; i >= 0; i--)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
// ark: YIsNegative <=> y > -y
DG_CODEC bool is_neg(const Fq& y) {
  uint32_t a[NL], b[NL];
  canon(y, a);
  canon(y.neg(), b);
  return cmp(a, b) > 0;
}
DG_CODEC bool is_neg(const Fq2& y) {   // QuadExtField ordering: c1 first, then c0
  uint32_t a[NL], b[NL];
  canon(y.c1, a);
  canon(y.c1.neg(), b);
  int c = cmp(a, b);
  if (c != 0) return c > 0;
  return is_neg(y.c0);
}
DG_CODEC Fq pow_limbs(const Fq& base, const uint32_t e[NL]) {
  Fq acc = Fq::one();
  for (int i = NL * 32 - 1; i >= 0; i--) {
    acc = acc.sqr();
    if ((e[i / 32] >> (i % 32)) & 1) acc = acc * base;
  }
  return acc;
}
DG_CODEC bool sqrt_fq(const Fq& a, Fq& r) {   // q = 3 mod 4: a^((q+1)/4)
  uint32_t e[NL];
  uint64_t carry = 1;
  for (int i = 0; i < NL; i++) {
    uint64_t v = (uint64_t)bn254_fq_params::P[i] + carry;
    e[i] = (uint32_t)v;
    carry = v >> 32;
  }
  for (int i = 0; i < NL; i++) e[i] = (e[i] >> 2) | (i + 1 < NL ? e[i + 1] << 30 : 0);
  r = pow_limbs(a, e);
  return r.sqr() == a;
}
DG_CODEC bool sqrt_fq2(const Fq2& a, Fq2& r) {
  if (a.c1.is_zero()) {
    Fq t;
    if (sqrt_fq(a.c0, t)) { r = {t, Fq::zero()}; return true; }
    if (sqrt_fq(a.c0.neg(), t)) { r = {Fq::zero(), t}; return true; }
    return false;
  }
  Fq n;
  if (!sqrt_fq(a.c0.sqr() + a.c1.sqr(), n)) return false;
  const Fq inv2 = Fq::from_u32(2).inv();
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
DG_CODEC Fq b_g1() { return Fq::from_u32(3); }
DG_CODEC Fq2 b_g2() {   // twist: y^2 = x^3 + 3 / (9 + u)
  Fq2 xi = {Fq::from_u32(9), Fq::from_u32(1)};
  return Fq2{Fq::from_u32(3), Fq::zero()} * xi.inv();
}

DG_CODEC void put_fq(const Fq& a, uint8_t* out) {
  uint32_t c[NL];
  canon(a, c);
  memcpy(out, c, 32);
}
DG_CODEC bool get_fq(const uint8_t* in, Fq& out) {   // canonical little-endian, must be < q
  uint32_t c[NL];
  memcpy(c, in, 32);
  if (cmp(c, bn254_fq_params::P) >= 0) return false;
  Fq t = Fq::zero();
  for (int i = 0; i < NL; i++) t.l[i] = c[i];
  out = t.to_mont();
  return true;
}

DG_CODEC void encode(const Affine<Fq>& p, uint8_t* out) {
  memset(out, 0, 32);
  if (p.is_inf()) { out[31] |= 0x40; return; }
  put_fq(p.x, out);
  if (is_neg(p.y)) out[31] |= 0x80;
}
DG_CODEC void encode(const Affine<Fq2>& p, uint8_t* out) {
  memset(out, 0, 64);
  if (p.is_inf()) { out[63] |= 0x40; return; }
  put_fq(p.x.c0, out);
  put_fq(p.x.c1, out + 32);
  if (is_neg(p.y)) out[63] |= 0x80;
}

DG_CODEC int decode(const uint8_t* in, Affine<Fq>& p) {
  uint8_t b[32];
  memcpy(b, in, 32);
  const bool neg = b[31] & 0x80, inf = b[31] & 0x40;
  b[31] &= 0x3F;
  if (neg && inf) return 1;
  Fq x;
  if (!get_fq(b, x)) return 2;
  if (inf) {
    if (!x.is_zero()) return 1;
    p = Affine<Fq>::inf();
    return 0;
  }
  Fq y;
  if (!sqrt_fq(x.sqr() * x + b_g1(), y)) return 3;
  if (is_neg(y) != neg) y = y.neg();
  p = {x, y};
  return 0;   // G1 has cofactor 1
}
DG_CODEC int decode(const uint8_t* in, Affine<Fq2>& p, bool validate) {
  uint8_t b[64];
  memcpy(b, in, 64);
  const bool neg = b[63] & 0x80, inf = b[63] & 0x40;
  b[63] &= 0x3F;
  if (neg && inf) return 1;
  Fq2 x;
  if (!get_fq(b, x.c0) || !get_fq(b + 32, x.c1)) return 2;
  if (inf) {
    if (!x.is_zero()) return 1;
    p = Affine<Fq2>::inf();
    return 0;
  }
  Fq2 y;
  if (!sqrt_fq2(x.sqr() * x + b_g2(), y)) return 3;
  if (is_neg(y) != neg) y = y.neg();
  p = {x, y};
  if (validate) {   // order-r subgroup (the twist has a cofactor)
    XYZZ<Fq2> q = scalar_mul<Fq2, bn254_fr_params::NL>(XYZZ<Fq2>::from_affine(p), bn254_fr_params::P);
    if (!q.is_inf()) return 4;
  }
  return 0;
}

}  // namespace codec
}  // namespace dg16
