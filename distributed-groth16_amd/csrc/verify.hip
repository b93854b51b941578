// Groth16 verification on BN254 -- host side of libdg16 (no GPU work here: four pairings per proof).  Counterpart
// of `Groth16::<Bn254>::verify_proof` as the reference calls it (groth16/examples/sha256.rs:228-254, the verify
// endpoint of mpc-api/src/main.rs and zk-cli): e(A, B) = e(alpha, beta) * e(sum_i x_i IC_i, gamma) * e(C, delta).
//
// Any non-degenerate bilinear map decides that equation, so this is the textbook reduced Tate pairing
// t(P, Q) = f_{r,P}(psi(Q))^((q^12 - 1)/r): Miller loop over the bits of r with P in E(Fq) (vertical lines lie in a
// proper subfield and die in the final exponentiation), Q untwisted into E(Fq12) by psi(x, y) = (x w^2, y w^3)
// (D-twist, Fq12 = Fq2[w] / (w^6 - (9 + u))), one shared square-and-multiply final exponentiation for the product
// of the four Miller values.  No Frobenius constants or curve-specific shortcuts: ~0.1 s per proof on one core,
// irrelevant next to proving.  Pinned by the reference's snarkjs proof triple (tests/test_verify.py).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/dg16.h"
#include "types.h"

namespace {

using namespace dg16;
using Fq = Fp<bn254_fq_params>;
using Fq2 = Fp2<Fq>;
using Fr = Fp<bn254_fr_params>;
constexpr int NL = Fq::NL;

struct Fq12 {
  Fq2 c[6];
  static Fq12 one() {
    Fq12 r;
    for (auto& x : r.c) x = Fq2::zero();
    r.c[0] = Fq2::one();
    return r;
  }
  bool is_one() const {
    if (!(c[0] == Fq2::one())) return false;
    for (int i = 1; i < 6; i++)
      if (!c[i].is_zero()) return false;
    return true;
  }
};

Fq2 mul_xi(const Fq2& a) {   // (a0 + a1 u)(9 + u) = (9 a0 - a1) + (a0 + 9 a1) u
  Fq n0 = a.c0.dbl().dbl().dbl() + a.c0, n1 = a.c1.dbl().dbl().dbl() + a.c1;
  return {n0 - a.c1, a.c0 + n1};
}

Fq12 mul(const Fq12& a, const Fq12& b) {
  Fq2 t[11];
  for (auto& x : t) x = Fq2::zero();
  for (int i = 0; i < 6; i++) {
    if (a.c[i].is_zero()) continue;
    for (int j = 0; j < 6; j++) {
      if (b.c[j].is_zero()) continue;
      t[i + j] = t[i + j] + a.c[i] * b.c[j];
    }
  }
  Fq12 r;
  for (int i = 0; i < 6; i++) r.c[i] = i < 5 ? t[i] + mul_xi(t[i + 6]) : t[i];
  return r;
}

// (q^12 - 1) / r as little-endian 32-bit limbs, computed once with schoolbook big-number arithmetic
const std::vector<uint32_t>& final_exponent() {
  static const std::vector<uint32_t> e = [] {
    std::vector<uint32_t> p(1, 1);
    for (int k = 0; k < 12; k++) {   // p *= q
      std::vector<uint32_t> n(p.size() + NL, 0);
      for (size_t i = 0; i < p.size(); i++) {
        uint64_t carry = 0;
        for (int j = 0; j < NL; j++) {
          uint64_t v = (uint64_t)p[i] * bn254_fq_params::P[j] + n[i + j] + carry;
          n[i + j] = (uint32_t)v;
          carry = v >> 32;
        }
        n[i + NL] += (uint32_t)carry;
      }
      p.swap(n);
    }
    p[0] -= 1;   // q^12 is odd
    // long division by r, bit by bit (remainder < 2r fits NL + 1 limbs)
    std::vector<uint32_t> quot(p.size(), 0);
    uint32_t rem[NL + 1] = {};
    for (int bit = (int)p.size() * 32 - 1; bit >= 0; bit--) {
      for (int i = NL; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
      rem[0] = (rem[0] << 1) | ((p[bit / 32] >> (bit % 32)) & 1);
      bool ge = rem[NL] != 0;
      if (!ge) {
        ge = true;
        for (int i = NL - 1; i >= 0; i--)
          if (rem[i] != bn254_fr_params::P[i]) { ge = rem[i] > bn254_fr_params::P[i]; break; }
      }
      if (ge) {
        uint64_t borrow = 0;
        for (int i = 0; i < NL; i++) {
          uint64_t d = (uint64_t)rem[i] - bn254_fr_params::P[i] - borrow;
          rem[i] = (uint32_t)d;
          borrow = (d >> 32) & 1;
        }
        rem[NL] -= (uint32_t)borrow;
        quot[bit / 32] |= 1u << (bit % 32);
      }
    }
    while (quot.size() > 1 && quot.back() == 0) quot.pop_back();
    return quot;
  }();
  return e;
}

Fq12 final_exp(const Fq12& f) {
  const std::vector<uint32_t>& e = final_exponent();
  Fq12 acc = Fq12::one();
  bool started = false;
  for (int i = (int)e.size() * 32 - 1; i >= 0; i--) {
    if (started) acc = mul(acc, acc);
    if ((e[i / 32] >> (i % 32)) & 1) {
      acc = started ? mul(acc, f) : f;
      started = true;
    }
  }
  return acc;
}

// l(psi(Q)) = Y - y0 - lam (X - x0)  with  X = xq w^2, Y = yq w^3
Fq12 line(const Fq& lam, const Fq& x0, const Fq& y0, const Affine<Fq2>& q) {
  Fq12 l;
  for (auto& x : l.c) x = Fq2::zero();
  l.c[0] = {lam * x0 - y0, Fq::zero()};
  l.c[2] = Fq2{q.x.c0 * lam, q.x.c1 * lam}.neg();
  l.c[3] = q.y;
  return l;
}

// f_{r,P}(psi(Q)) without the final exponentiation
Fq12 miller(const Affine<Fq>& p, const Affine<Fq2>& q) {
  if (p.is_inf() || q.is_inf()) return Fq12::one();
  Fq xt = p.x, yt = p.y;
  Fq12 f = Fq12::one();
  const Fq three = Fq::from_u32(3);
  int top = NL * 32 - 1;
  while (!((bn254_fr_params::P[top / 32] >> (top % 32)) & 1)) top--;
  for (int i = top - 1; i >= 0; i--) {
    Fq lam = three * xt.sqr() * (yt + yt).inv();
    f = mul(mul(f, f), line(lam, xt, yt, q));
    Fq x3 = lam.sqr() - xt - xt;
    yt = lam * (xt - x3) - yt;
    xt = x3;
    if ((bn254_fr_params::P[i / 32] >> (i % 32)) & 1) {
      if (xt == p.x) continue;   // T = -P: only at the very last addition ((r-1)P + P); the line is vertical
      lam = (yt - p.y) * (xt - p.x).inv();
      f = mul(f, line(lam, xt, yt, q));
      x3 = lam.sqr() - xt - p.x;
      yt = lam * (xt - x3) - yt;
      xt = x3;
    }
  }
  return f;
}

bool on_curve(const Affine<Fq>& p) { return p.is_inf() || p.y.sqr() == p.x.sqr() * p.x + Fq::from_u32(3); }
bool on_curve(const Affine<Fq2>& p) {
  if (p.is_inf()) return true;
  Fq2 xi = {Fq::from_u32(9), Fq::from_u32(1)};
  Fq2 b = Fq2{Fq::from_u32(3), Fq::zero()} * xi.inv();
  return p.y.sqr() == p.x.sqr() * p.x + b;
}
Affine<Fq> neg(const Affine<Fq>& p) { return p.is_inf() ? p : Affine<Fq>{p.x, p.y.neg()}; }

// limb vector < modulus (arkworks rejects non-canonical field elements at deserialisation; without this check
// x, x + r, x + 2r -- all below 2^256 -- would be the same public input: input aliasing)
template <class P>
bool canonical(const Fp<P>& v) {
  for (int i = P::NL - 1; i >= 0; i--)
    if (v.l[i] != P::P[i]) return v.l[i] < P::P[i];
  return false;
}
bool canonical(const Fq2& v) { return canonical(v.c0) && canonical(v.c1); }
template <class F>
bool canonical(const Affine<F>& p) { return canonical(p.x) && canonical(p.y); }
// r * Q == identity (G2 has cofactor > 1: on-curve is not enough; G1 of BN254 has cofactor 1)
bool in_subgroup(const Affine<Fq2>& q) {
  if (q.is_inf()) return true;
  return scalar_mul<Fq2, Fr::NL>(XYZZ<Fq2>::from_affine(q), bn254_fr_params::P).is_inf();
}

thread_local const char* g_err = "";

}  // namespace

extern "C" {

const char* dg16_verify_error(void) { return g_err; }

int dg16_groth16_verify(int curve, const void* alpha_g1, const void* beta_g2, const void* gamma_g2,
                        const void* delta_g2, const void* ic, size_t n_ic, const void* public_inputs,
                        size_t n_public, const void* proof_affine, unsigned flags, int* accepted) {
  if (curve != DG16_BN254) { g_err = "verification: BN254 only"; return DG16_ERR_UNSUPPORTED; }
  if (!alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !ic || !proof_affine || !accepted || (n_public && !public_inputs)) {
    g_err = "null argument";
    return DG16_ERR_BAD_ARG;
  }
  *accepted = 0;
  if (n_ic != n_public + 1) {   // ark-groth16: SynthesisError::MalformedVerifyingKey
    g_err = "public input count does not match the verification key";
    return DG16_ERR_LENGTH_MISMATCH;
  }
  Affine<Fq> alpha, a, c;
  Affine<Fq2> beta, gamma, delta, b;
  memcpy(&alpha, alpha_g1, sizeof(alpha));
  memcpy(&beta, beta_g2, sizeof(beta));
  memcpy(&gamma, gamma_g2, sizeof(gamma));
  memcpy(&delta, delta_g2, sizeof(delta));
  const uint8_t* pr = (const uint8_t*)proof_affine;
  memcpy(&a, pr, sizeof(a));
  memcpy(&b, pr + sizeof(a), sizeof(b));
  memcpy(&c, pr + sizeof(a) + sizeof(b), sizeof(c));
  // verification key: every coordinate reduced, every point on its curve, G2 points in the order-r subgroup
  // (what `VerifyingKey::deserialize_compressed` with Validate::Yes guarantees on the reference's side)
  std::vector<Affine<Fq>> icv;
  try {
    icv.resize(n_ic);
  } catch (...) {
    g_err = "out of memory";
    return DG16_ERR_OOM;
  }
  memcpy(icv.data(), ic, n_ic * sizeof(Affine<Fq>));
  bool vk_ok = canonical(alpha) && on_curve(alpha) && canonical(beta) && on_curve(beta) && in_subgroup(beta) &&
               canonical(gamma) && on_curve(gamma) && in_subgroup(gamma) && canonical(delta) && on_curve(delta) &&
               in_subgroup(delta);
  for (size_t i = 0; vk_ok && i < n_ic; i++) vk_ok = canonical(icv[i]) && on_curve(icv[i]);
  if (!vk_ok) {
    g_err = "malformed verifying key (non-reduced coordinate, point off its curve or outside the subgroup)";
    return DG16_ERR_BAD_ARG;
  }
  for (size_t i = 0; i < n_public; i++) {
    Fr x;
    memcpy(&x, (const uint8_t*)public_inputs + i * sizeof(Fr), sizeof(Fr));
    if (!canonical(x)) {
      g_err = "public input is not a reduced field element (>= r)";
      return DG16_ERR_BAD_ARG;
    }
  }
  // proof: a non-reduced coordinate, a point off its curve or B outside the subgroup is a rejection
  if (!canonical(a) || !canonical(b) || !canonical(c) || !on_curve(a) || !on_curve(c) || !on_curve(b) ||
      !in_subgroup(b)) {
    g_err = "";
    return DG16_OK;
  }
  // prepared inputs: IC_0 + sum_i x_i IC_{i+1}
  XYZZ<Fq> acc = XYZZ<Fq>::from_affine(icv[0]);
  for (size_t i = 0; i < n_public; i++) {
    Fr x;
    memcpy(&x, (const uint8_t*)public_inputs + i * sizeof(Fr), sizeof(Fr));
    if (flags & DG16_F_SCALARS_MONT) x = x.from_mont();
    acc = acc.add(scalar_mul<Fq, Fr::NL>(XYZZ<Fq>::from_affine(icv[i + 1]), x.l));
  }
  Fq12 f = miller(a, b);
  f = mul(f, miller(neg(alpha), beta));
  f = mul(f, miller(neg(acc.to_affine()), gamma));
  f = mul(f, miller(neg(c), delta));
  *accepted = final_exp(f).is_one() ? 1 : 0;
  g_err = "";
  return DG16_OK;
}

}  // extern "C"
