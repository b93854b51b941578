// arkworks compressed Proof<Bn254> <-> the prover's output -- host side of libdg16 (no GPU work here).  The
// reference writes / reads this 128-byte encoding as `proof.bin` (`proof.serialize_compressed`,
// `Proof::deserialize_compressed`: mpc-api/src/main.rs:154-171, zk-cli/test-circuits/sha256/proof.bin):
// A (G1, 32 B) || B (G2, 64 B) || C (G1, 32 B), little-endian x with the ark-ec short-Weierstrass flags in the
// two top bits of the last byte (bit 7: y is the "negative" root, y > -y, for Fq2 compared on (c1, c0);
// bit 6: point at infinity).  Pinned by the reference's own proof.bin and the coordinates its CLI prints
// (tests/test_serialize.py).  Uses the portable host path of fp.h.
#include <stdint.h>
#include <string.h>

#include "../../include/dg16.h"
#include "types.h"

namespace {

using namespace dg16;
using Fq = Fp<bn254_fq_params>;
using Fq2 = Fp2<Fq>;
constexpr int NL = Fq::NL;

void canon(const Fq& a, uint32_t out[NL]) {
  Fq c = a.from_mont();
  for (int i = 0; i < NL; i++) out[i] = c.l[i];
}
int cmp(const uint32_t* a, const uint32_t* b) {
  for (int i = NL - 1; i >= 0; i--)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
// ark: YIsNegative <=> y > -y
bool is_neg(const Fq& y) {
  uint32_t a[NL], b[NL];
  canon(y, a);
  canon(y.neg(), b);
  return cmp(a, b) > 0;
}
bool is_neg(const Fq2& y) {   // QuadExtField ordering: c1 first, then c0
  uint32_t a[NL], b[NL];
  canon(y.c1, a);
  canon(y.c1.neg(), b);
  int c = cmp(a, b);
  if (c != 0) return c > 0;
  return is_neg(y.c0);
}
Fq pow_limbs(const Fq& base, const uint32_t e[NL]) {
  Fq acc = Fq::one();
  for (int i = NL * 32 - 1; i >= 0; i--) {
    acc = acc.sqr();
    if ((e[i / 32] >> (i % 32)) & 1) acc = acc * base;
  }
  return acc;
}
bool sqrt_fq(const Fq& a, Fq& r) {   // q = 3 mod 4: a^((q+1)/4)
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
bool sqrt_fq2(const Fq2& a, Fq2& r) {
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
Fq b_g1() { return Fq::from_u32(3); }
Fq2 b_g2() {   // twist: y^2 = x^3 + 3 / (9 + u)
  Fq2 xi = {Fq::from_u32(9), Fq::from_u32(1)};
  return Fq2{Fq::from_u32(3), Fq::zero()} * xi.inv();
}

void put_fq(const Fq& a, uint8_t* out) {
  uint32_t c[NL];
  canon(a, c);
  memcpy(out, c, 32);
}
bool get_fq(const uint8_t* in, Fq& out) {   // canonical little-endian, must be < q
  uint32_t c[NL];
  memcpy(c, in, 32);
  if (cmp(c, bn254_fq_params::P) >= 0) return false;
  Fq t = Fq::zero();
  for (int i = 0; i < NL; i++) t.l[i] = c[i];
  out = t.to_mont();
  return true;
}

void encode(const Affine<Fq>& p, uint8_t* out) {
  memset(out, 0, 32);
  if (p.is_inf()) { out[31] |= 0x40; return; }
  put_fq(p.x, out);
  if (is_neg(p.y)) out[31] |= 0x80;
}
void encode(const Affine<Fq2>& p, uint8_t* out) {
  memset(out, 0, 64);
  if (p.is_inf()) { out[63] |= 0x40; return; }
  put_fq(p.x.c0, out);
  put_fq(p.x.c1, out + 32);
  if (is_neg(p.y)) out[63] |= 0x80;
}

int decode(const uint8_t* in, Affine<Fq>& p) {
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
int decode(const uint8_t* in, Affine<Fq2>& p, bool validate) {
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

thread_local const char* g_err = "";
const char* kDecodeErr[5] = {"", "invalid flags", "coordinate not reduced", "x is not on the curve",
                             "point is not in the prime-order subgroup"};

}  // namespace

extern "C" {

const char* dg16_serialize_error(void) { return g_err; }

int dg16_proof_compress(int curve, const void* proof_jacobian, void* out) {
  if (curve != DG16_BN254) { g_err = "compressed proofs: BN254 only"; return DG16_ERR_UNSUPPORTED; }
  if (!proof_jacobian || !out) { g_err = "null argument"; return DG16_ERR_BAD_ARG; }
  const uint8_t* in = (const uint8_t*)proof_jacobian;
  Jacobian<Fq> a, c;
  Jacobian<Fq2> b;
  memcpy(&a, in, sizeof(a));
  memcpy(&b, in + sizeof(a), sizeof(b));
  memcpy(&c, in + sizeof(a) + sizeof(b), sizeof(c));
  uint8_t* o = (uint8_t*)out;
  encode(XYZZ<Fq>::from_jacobian(a).to_affine(), o);
  encode(XYZZ<Fq2>::from_jacobian(b).to_affine(), o + 32);
  encode(XYZZ<Fq>::from_jacobian(c).to_affine(), o + 96);
  return DG16_OK;
}

int dg16_proof_decompress(int curve, const void* in128, int validate, void* proof_affine) {
  if (curve != DG16_BN254) { g_err = "compressed proofs: BN254 only"; return DG16_ERR_UNSUPPORTED; }
  if (!in128 || !proof_affine) { g_err = "null argument"; return DG16_ERR_BAD_ARG; }
  const uint8_t* in = (const uint8_t*)in128;
  Affine<Fq> a, c;
  Affine<Fq2> b;
  int rc = decode(in, a);
  if (!rc) rc = decode(in + 32, b, validate != 0);
  if (!rc) rc = decode(in + 96, c);
  if (rc) { g_err = kDecodeErr[rc]; return DG16_ERR_BAD_ARG; }
  uint8_t* o = (uint8_t*)proof_affine;
  memcpy(o, &a, sizeof(a));
  memcpy(o + sizeof(a), &b, sizeof(b));
  memcpy(o + sizeof(a) + sizeof(b), &c, sizeof(c));
  return DG16_OK;
}

}  // extern "C"
