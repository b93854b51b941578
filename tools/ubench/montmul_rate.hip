// Montgomery-multiplication throughput probe (BN254 Fq, 8 x 32-bit limbs) on gfx950.
// Variant A: plain C++ CIOS (what hipcc makes of it).  Variant B: product scanning with
// v_mad_u64_u32 + v_addc carry word (inline asm).  Both are checked against a host evaluation
// of variant A before timing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define NL 8
static constexpr uint32_t Pk[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#define INV 0xe4866389u

__host__ __device__ inline void final_sub(uint32_t* r, const uint32_t* t, uint32_t top) {
  uint32_t s[NL]; uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { uint64_t d = (uint64_t)t[i] - Pk[i] - borrow; s[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1; }
  bool ge = (top != 0) || !borrow;
#pragma unroll
  for (int i = 0; i < NL; i++) r[i] = ge ? s[i] : t[i];
}

__host__ __device__ inline void mont_mul_A(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint32_t t[NL + 2];
#pragma unroll
  for (int i = 0; i < NL + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) { uint64_t acc = (uint64_t)a[j] * b[i] + t[j] + carry; t[j] = (uint32_t)acc; carry = (uint32_t)(acc >> 32); }
    uint64_t acc = (uint64_t)t[NL] + carry; t[NL] = (uint32_t)acc; t[NL + 1] = (uint32_t)(acc >> 32);
    uint32_t m = t[0] * INV;
    acc = (uint64_t)m * Pk[0] + t[0]; carry = (uint32_t)(acc >> 32);
#pragma unroll
    for (int j = 1; j < NL; j++) { acc = (uint64_t)m * Pk[j] + t[j] + carry; t[j - 1] = (uint32_t)acc; carry = (uint32_t)(acc >> 32); }
    acc = (uint64_t)t[NL] + carry; t[NL - 1] = (uint32_t)acc; t[NL] = t[NL + 1] + (uint32_t)(acc >> 32);
  }
  final_sub(r, t, t[NL]);
}

// acc (64-bit pair) += x*y, carry-out accumulated into c2
#define MADC(acc, c2, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(x), "v"(y) : "vcc")
#define MADC_S(acc, c2, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(x), "s"(y) : "vcc")

__device__ __forceinline__ void mont_mul_B(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t acc = 0; uint32_t c2 = 0; uint32_t m[NL]; uint32_t t[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i < k; i++) { MADC(acc, c2, a[i], b[k - i]); MADC_S(acc, c2, m[i], Pk[k - i]); }
    MADC(acc, c2, a[k], b[0]);
    m[k] = (uint32_t)acc * INV;
    MADC_S(acc, c2, m[k], Pk[0]);
    acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) { MADC(acc, c2, a[i], b[k - i]); MADC_S(acc, c2, m[i], Pk[k - i]); }
    t[k - NL] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
  }
  final_sub(r, t, (uint32_t)acc);
}


// Variant D = variant B as shipped at the end of round 1 (csrc/fp.h): the first product of a column SETS the carry
// word, so no zeroing move at the column hand-over and hipcc coalesces the carry word into the next accumulator pair.
#define MADC0(acc, c2, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc" : "+v"(acc), "=v"(c2) : "v"(x), "v"(y) : "vcc")
__device__ __forceinline__ void mont_mul_D(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t acc = 0; uint32_t c2; uint32_t m[NL]; uint32_t t[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) {
    MADC0(acc, c2, a[0], b[k]);
#pragma unroll
    for (int i = 1; i <= k; i++) MADC(acc, c2, a[i], b[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) MADC_S(acc, c2, m[i], Pk[k - i]);
    m[k] = (uint32_t)acc * INV;
    MADC_S(acc, c2, m[k], Pk[0]);
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
    MADC0(acc, c2, a[k - NL + 1], b[NL - 1]);
#pragma unroll
    for (int i = k - NL + 2; i < NL; i++) MADC(acc, c2, a[i], b[k - i]);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) MADC_S(acc, c2, m[i], Pk[k - i]);
    t[k - NL] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
  }
  t[NL - 1] = (uint32_t)acc;
  final_sub(r, t, (uint32_t)(acc >> 32));
}

// Variant C: column-parallel CIOS with lazy carries.  NL+1 live columns, each a 64-bit accumulator
// plus a 32-bit overflow counter; the 8 MADCs of a row hit 8 different columns (ILP 8), so a single
// wave is not bound by the mad->addc->mad dependency chain of variant B.
__device__ __forceinline__ void mont_mul_C(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t acc[NL + 1]; uint32_t cc[NL + 1];
#pragma unroll
  for (int j = 0; j <= NL; j++) { acc[j] = 0; cc[j] = 0; }
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int j = 0; j < NL; j++) MADC(acc[j], cc[j], a[j], b[i]);
    uint32_t m = (uint32_t)acc[0] * INV;
#pragma unroll
    for (int j = 0; j < NL; j++) MADC_S(acc[j], cc[j], m, Pk[j]);
    // column 0 now has zero low word: fold its upper part into column 1 and shift the window
    uint64_t up = (acc[0] >> 32) | ((uint64_t)cc[0] << 32);
    uint64_t s1 = acc[1] + up;
    cc[1] += (s1 < up) ? 1u : 0u;
    acc[1] = s1;
#pragma unroll
    for (int j = 0; j < NL; j++) { acc[j] = acc[j + 1]; cc[j] = cc[j + 1]; }
    acc[NL] = 0; cc[NL] = 0;
  }
  // resolve: value = sum_j (acc[j] + cc[j]*2^64) * 2^(32 j)
  uint32_t t[NL]; uint64_t carry = 0; uint32_t top = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    uint64_t lo = (acc[j] & 0xffffffffu) + (carry & 0xffffffffu);
    t[j] = (uint32_t)lo;
    // next carry = (acc[j] >> 32) + (cc[j] << 32) + (carry >> 32) + (lo >> 32)
    carry = (acc[j] >> 32) + ((uint64_t)cc[j] << 32) + (carry >> 32) + (lo >> 32);
  }
  top = (uint32_t)carry;
  final_sub(r, t, top);
}

template <int V> __global__ void chain(uint32_t* out, const uint32_t* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[NL], b[NL];
  for (int i = 0; i < NL; i++) { a[i] = in[tid * 16 + i]; b[i] = in[tid * 16 + 8 + i]; }
  for (int it = 0; it < iters; it++) { if (V == 0) mont_mul_A(a, a, b); else if (V == 1) mont_mul_B(a, a, b); else if (V == 3) mont_mul_D(a, a, b); else mont_mul_C(a, a, b); }
  for (int i = 0; i < NL; i++) out[tid * 8 + i] = a[i];
}
// 4 independent chains per lane (ILP)
template <int V> __global__ void chain4(uint32_t* out, const uint32_t* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[4][NL], b[NL];
  for (int i = 0; i < NL; i++) { b[i] = in[tid * 16 + 8 + i]; for (int c = 0; c < 4; c++) a[c][i] = in[tid * 16 + i] ^ (c * 77); a[0][7] &= 0x0fffffff; a[1][7] &= 0x0fffffff; a[2][7] &= 0x0fffffff; a[3][7] &= 0x0fffffff; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < 4; c++) { if (V == 0) mont_mul_A(a[c], a[c], b); else if (V == 1) mont_mul_B(a[c], a[c], b); else if (V == 3) mont_mul_D(a[c], a[c], b); else mont_mul_C(a[c], a[c], b); }
  }
  for (int i = 0; i < NL; i++) out[tid * 8 + i] = a[0][i] ^ a[1][i] ^ a[2][i] ^ a[3][i];
}

static uint64_t sm(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  const int blocks = cus * 8, threads = 256, n = blocks * threads;
  std::vector<uint32_t> h_in((size_t)n * 16), h_a((size_t)n * 8), h_b((size_t)n * 8);
  for (size_t i = 0; i < h_in.size(); i++) { h_in[i] = (uint32_t)sm(i); if (i % 8 == 7) h_in[i] &= 0x1fffffff; }
  uint32_t *d_in, *d_out;
  CHECK(hipMalloc(&d_in, h_in.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)n * 32));
  CHECK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  // correctness: 5 iterations, compare A (device), B (device), A (host)
  hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 5);
  CHECK(hipMemcpy(h_a.data(), d_out, (size_t)n * 32, hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 5);
  CHECK(hipMemcpy(h_b.data(), d_out, (size_t)n * 32, hipMemcpyDeviceToHost));
  std::vector<uint32_t> h_c((size_t)n * 8);
  hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 5);
  CHECK(hipMemcpy(h_c.data(), d_out, (size_t)n * 32, hipMemcpyDeviceToHost));
  size_t badC = 0; for (size_t i = 0; i < h_c.size(); i++) if (h_c[i] != h_a[i]) badC++;
  printf("correctness: C_dev vs A_dev mismatches %zu\n", badC);
  hipLaunchKernelGGL(chain<3>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 5);
  CHECK(hipMemcpy(h_c.data(), d_out, (size_t)n * 32, hipMemcpyDeviceToHost));
  size_t badD = 0; for (size_t i = 0; i < h_c.size(); i++) if (h_c[i] != h_a[i]) badD++;
  printf("correctness: D_dev vs A_dev mismatches %zu\n", badD);
  size_t badAB = 0, badH = 0;
  for (int t = 0; t < n; t++) {
    uint32_t a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = h_in[(size_t)t * 16 + i]; b[i] = h_in[(size_t)t * 16 + 8 + i]; }
    for (int it = 0; it < 5; it++) mont_mul_A(a, a, b);
    for (int i = 0; i < 8; i++) { if (h_a[(size_t)t * 8 + i] != h_b[(size_t)t * 8 + i]) badAB++; if (h_a[(size_t)t * 8 + i] != a[i]) badH++; }
  }
  printf("correctness: A_dev vs B_dev mismatches %zu ; A_dev vs A_host mismatches %zu (of %d words)\n", badAB, badH, n * 8);
  auto timeit = [&](const char* name, void (*k)(uint32_t*, const uint32_t*, int), int mulsPerIter, int blocks_per_cu = 8) {
    const int iters = 2000;
    const int blocks = cus * blocks_per_cu; const int n = blocks * threads;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)n * iters * mulsPerIter;
    printf("%-28s waves/SIMD %d  %8.3f ms  %8.2f G montmul/s\n", name, blocks_per_cu, ms, muls / (ms * 1e-3) * 1e-9);
  };
  timeit("A (C++ CIOS) chain", chain<0>, 1);
  timeit("B (asm product-scan) chain", chain<1>, 1);
  timeit("A x4 ILP", chain4<0>, 4);
  timeit("B x4 ILP", chain4<1>, 4);
  timeit("D (B + carry-set) chain", chain<3>, 1);
  timeit("D x4 ILP", chain4<3>, 4);
  timeit("C (column-parallel) chain", chain<2>, 1);
  timeit("C x4 ILP", chain4<2>, 4);
  for (int w : {1, 2, 3, 4}) { timeit("D chain", chain<3>, 1, w); timeit("B chain", chain<1>, 1, w); timeit("C chain", chain<2>, 1, w); timeit("A chain", chain<0>, 1, w); }
  return 0;
}
