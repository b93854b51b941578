// Probe translation unit of tests/test_fp29_asm_isa.py (CPU test, no GPU needed): the device products of fp29.h
// (fp29_asm_gen.h) behind non-inlined functions with the operands in VGPRs, compiled to gfx950 assembly that the test
// reads back and executes on Python integers.  Not part of libdg16.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../distributed-groth16_amd/csrc/consts_gen.h"
#include "../../distributed-groth16_amd/csrc/fp29.h"

using namespace dg16;
// Operands and result behind pointers (flat loads / stores): the only part of the calling convention the test relies on
// is "pointer arguments arrive in v[0:1], v[2:3], ...".
#define PROBE(name, P)                                                                                              \
  extern "C" __device__ __attribute__((noinline, used)) void probe_mul_##name(uint32_t* r, const uint32_t* a,       \
                                                                              const uint32_t* b) {                  \
    constexpr int N = RR<P>::N;                                                                                     \
    uint32_t x[N], y[N], o[N];                                                                                      \
    for (int i = 0; i < N; i++) { x[i] = a[i]; y[i] = b[i]; }                                                       \
    rr::mont_inl<P, false>(o, x, y, nullptr, nullptr);                                                              \
    for (int i = 0; i < N; i++) r[i] = o[i];                                                                        \
  }                                                                                                                 \
  extern "C" __device__ __attribute__((noinline, used)) void probe_dual_##name(uint32_t* r, const uint32_t* a,      \
                                                                               const uint32_t* b, const uint32_t* c, \
                                                                               const uint32_t* d) {                 \
    constexpr int N = RR<P>::N;                                                                                     \
    uint32_t x[N], y[N], z[N], w[N], o[N];                                                                          \
    for (int i = 0; i < N; i++) { x[i] = a[i]; y[i] = b[i]; z[i] = c[i]; w[i] = d[i]; }                             \
    rr::mont_inl<P, true>(o, x, y, z, w);                                                                           \
    for (int i = 0; i < N; i++) r[i] = o[i];                                                                        \
  }                                                                                                                 \
  extern "C" __device__ __attribute__((noinline, used)) void probe_quad_##name(                                     \
      uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, const uint32_t* e,   \
      const uint32_t* f, const uint32_t* g, const uint32_t* h) {                                                    \
    constexpr int N = RR<P>::N;                                                                                     \
    uint32_t x[8][N], o[N];                                                                                         \
    const uint32_t* src[8] = {a, b, c, d, e, f, g, h};                                                              \
    for (int j = 0; j < 8; j++)                                                                                     \
      for (int i = 0; i < N; i++) x[j][i] = src[j][i];                                                              \
    rr::mont4_inl<P>(o, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);                                            \
    for (int i = 0; i < N; i++) r[i] = o[i];                                                                        \
  }                                                                                                                 \
  extern "C" __device__ __attribute__((noinline, used)) void probe_sqr_##name(uint32_t* r, const uint32_t* a) {     \
    constexpr int N = RR<P>::N;                                                                                     \
    uint32_t x[N], o[N];                                                                                            \
    for (int i = 0; i < N; i++) x[i] = a[i];                                                                        \
    rr::mont_sqr_inl<P>(o, x);                                                                                      \
    for (int i = 0; i < N; i++) r[i] = o[i];                                                                        \
  }
PROBE(bn254_fq, bn254_fq_params)
PROBE(bn254_fr, bn254_fr_params)
PROBE(bls12_381_fq, bls12_381_fq_params)
PROBE(bls12_381_fr, bls12_381_fr_params)
PROBE(bls12_377_fq, bls12_377_fq_params)
PROBE(bls12_377_fr, bls12_377_fr_params)
