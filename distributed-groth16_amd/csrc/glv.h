// GLV decomposition of a G1 scalar for the j = 0 curves of this library (BN254, BLS12-381, BLS12-377):
//     k = k1 + k2 LAMBDA (mod r),   |k1|, |k2| < 2^128,   phi(x, y) = (BETA x, y) = LAMBDA (x, y)
// so that sum_i k_i P_i over n points and 254-bit scalars becomes a sum over 2n points (P_i, phi(P_i)) and 128-bit
// scalars: half the windows, i.e. half the dependent doublings of Pippenger's Horner tail -- the one part of a plain
// MSM that nothing can hide.
//
// With a reduced basis (a1, b1), (a2, b2) of the lattice {(a, b): a + b LAMBDA = 0 mod r}, det = +r, b1 < 0 < b2
// (consts_gen.h: <curve>_glv_consts, generated and checked by oracle/gen_consts.py):
//     c1 = round(b2 k / r), c2 = round(-b1 k / r)            (both >= 0)
//     k1 = k - c1 a1 - c2 a2,   k2 = -c1 b1 - c2 b2
// and k1 + k2 LAMBDA = k - c1 (a1 + b1 LAMBDA) - c2 (a2 + b2 LAMBDA) = k (mod r) for ANY integers c1, c2: the
// division-free rounding below (c = (G k + 2^255) >> 256 with G = round(2^256 b / r)) can be off by one without
// touching correctness, only the size of the halves -- which tests/test_host_arith.py bounds on a few hundred
// thousand scalars per curve against the same integer arithmetic done with Python integers.
// Plain integer arithmetic on 32-bit words, host and device.
#pragma once
#include <stdint.h>
#include "fp.h"

namespace dg16 {
namespace glv {

// r[0 .. NA + NB) = a[0 .. NA) * b[0 .. NB)
template <int NA, int NB>
DG_HD void mul_words(const uint32_t* a, const uint32_t* b, uint32_t* r) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const uint64_t v = (uint64_t)a[i] * b[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)v;
      carry = v >> 32;
    }
    r[i + NB] = (uint32_t)carry;
  }
}
// acc[0 .. N) +-= v[0 .. N)   (two's complement, wraps)
template <int N>
DG_HD void add_words(uint32_t* acc, const uint32_t* v, bool subtract) {
  uint64_t carry = subtract ? 1 : 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const uint64_t s = (uint64_t)acc[i] + (subtract ? ~v[i] : v[i]) + carry;
    acc[i] = (uint32_t)s;
    carry = s >> 32;
  }
}
// two's complement acc[0 .. N) -> magnitude in out[0 .. 8) (the value fits 128 bits; the upper words are cleared),
// returns the sign
template <int N>
DG_HD bool magnitude(const uint32_t* acc, uint32_t* out) {
  const bool neg = (acc[N - 1] >> 31) != 0;
  uint64_t carry = neg ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t s = (uint64_t)(neg ? ~acc[i] : acc[i]) + carry;
    out[i] = (uint32_t)s;
    carry = s >> 32;
  }
  return neg;
}

// k (8 words, < 2^256) -> |k1|, |k2| as 8-word integers with the sign in bit 255 of each: the form the digit kernels of
// msm_impl.h take (a set bit 255 negates every digit of the scalar; a canonical field element never has it set)
template <class GC>
DG_HD void split(const uint32_t* k, uint32_t* h1, uint32_t* h2) {
  uint32_t t[13], c1[5], c2[5];
  const uint32_t half[13] = {0, 0, 0, 0, 0, 0, 0, 0x80000000u, 0, 0, 0, 0, 0};
  mul_words<5, 8>(GC::G1, k, t);
  add_words<13>(t, half, false);
#pragma unroll
  for (int i = 0; i < 5; i++) c1[i] = t[8 + i];
  mul_words<5, 8>(GC::G2, k, t);
  add_words<13>(t, half, false);
#pragma unroll
  for (int i = 0; i < 5; i++) c2[i] = t[8 + i];
  uint32_t p1[10], p2[10], acc[10];
  // k1 = k - c1 a1 - c2 a2
  mul_words<5, 5>(c1, GC::A1, p1);
  mul_words<5, 5>(c2, GC::A2, p2);
#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = i < 8 ? k[i] : 0u;
  add_words<10>(acc, p1, !GC::A1_NEG);
  add_words<10>(acc, p2, !GC::A2_NEG);
  const bool n1 = magnitude<10>(acc, h1);
  // k2 = -c1 b1 - c2 b2
  mul_words<5, 5>(c1, GC::B1, p1);
  mul_words<5, 5>(c2, GC::B2, p2);
#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = 0u;
  add_words<10>(acc, p1, !GC::B1_NEG);
  add_words<10>(acc, p2, !GC::B2_NEG);
  const bool n2 = magnitude<10>(acc, h2);
  if (n1) h1[7] |= 0x80000000u;
  if (n2) h2[7] |= 0x80000000u;
}

// The four-dimensional form for G2 (psi, psi^2, psi^3: <curve>_g2_glv4_consts): k -> four quarters of at most 64 bits,
//     k = k0 + k1 LAMBDA + k2 LAMBDA^2 + k3 LAMBDA^3 (mod r)
// Babai rounding on an LLL-reduced basis B: c_i = round(k A_i / det) through the multipliers G_i (the sign of c_i is a
// constant of the curve), k_j = [j = 0] k - sum_i c_i B[i][j].  Outputs as in split(): magnitude, sign in bit 255.
template <class GC>
DG_HD void split4(const uint32_t* k, uint32_t* h0, uint32_t* h1, uint32_t* h2, uint32_t* h3) {
  uint32_t c[4][7];
  {
    uint32_t t[15];
    uint32_t half[15];
#pragma unroll
    for (int i = 0; i < 15; i++) half[i] = i == 7 ? 0x80000000u : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      mul_words<7, 8>(GC::G[i], k, t);
      add_words<15>(t, half, false);
#pragma unroll
      for (int w = 0; w < 7; w++) c[i][w] = t[8 + w];
    }
  }
  uint32_t* out[4] = {h0, h1, h2, h3};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t acc[11];
#pragma unroll
    for (int w = 0; w < 11; w++) acc[w] = (j == 0 && w < 8) ? k[w] : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t p[11];
      mul_words<7, 3>(c[i], GC::B[4 * i + j], p);
      p[10] = 0;
      // the term c_i B[i][j] is negative iff exactly one of its factors is; k_j -= term
      add_words<11>(acc, p, GC::C_NEG[i] == GC::B_NEG[4 * i + j]);
    }
    if (magnitude<11>(acc, out[j])) out[j][7] |= 0x80000000u;
  }
}

}  // namespace glv
}  // namespace dg16
