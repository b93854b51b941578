// Reduced-radix Montgomery multiplication probe for gfx950: BN254 Fq as 9 limbs of 29 bits (R = 2^261).
//
// Why: with 32-bit limbs every partial product needs a carry instruction next to its v_mad_u64_u32
// (128 mad + 128 addc + ~47 mov per product; addc issues at 31 T lane-op/s, mad at 34 T, profiles/
// r1_ubench_instr_rate.txt).  With 29-bit limbs a column of <= 18 products of < 2^58 fits a 64-bit
// accumulator, so a column is a plain chain of v_mad_u64_u32 (81 + 81 = 162 per product, no carries),
// plus one mask / shift per column.  Squaring with a doubled operand needs 45 + 81.  Additions are 9
// independent v_add_u32 (lazy: limbs may grow to 31 bits, values to 2^261 >> p, so no conditional
// subtraction anywhere).
//
// Prints the chip-wide rate of (E) the 29-bit product, (F) the 29-bit square, next to (D) the shipped
// 32-bit product, after checking E and F against a host big-number evaluation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- 32-bit reference (variant D of montmul_rate.hip) ---------------------------------------------
#define NL 8
static constexpr uint32_t Pk[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#define INV32 0xe4866389u
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
  for (int i = 0; i < NL + 2; i++) t[i] = 0;
  for (int i = 0; i < NL; i++) {
    uint32_t carry = 0;
    for (int j = 0; j < NL; j++) { uint64_t acc = (uint64_t)a[j] * b[i] + t[j] + carry; t[j] = (uint32_t)acc; carry = (uint32_t)(acc >> 32); }
    uint64_t acc = (uint64_t)t[NL] + carry; t[NL] = (uint32_t)acc; t[NL + 1] = (uint32_t)(acc >> 32);
    uint32_t m = t[0] * INV32;
    acc = (uint64_t)m * Pk[0] + t[0]; carry = (uint32_t)(acc >> 32);
    for (int j = 1; j < NL; j++) { acc = (uint64_t)m * Pk[j] + t[j] + carry; t[j - 1] = (uint32_t)acc; carry = (uint32_t)(acc >> 32); }
    acc = (uint64_t)t[NL] + carry; t[NL - 1] = (uint32_t)acc; t[NL] = t[NL + 1] + (uint32_t)(acc >> 32);
  }
  final_sub(r, t, t[NL]);
}
#define MADC(acc, c2, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(x), "v"(y) : "vcc")
#define MADC_S(acc, c2, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(x), "s"(y) : "vcc")
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
    m[k] = (uint32_t)acc * INV32;
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

// ---- 29-bit limbs ------------------------------------------------------------------------------------
constexpr int N9 = 9;
constexpr uint32_t M29 = (1u << 29) - 1;
// p in 29-bit limbs, and -p^-1 mod 2^29, derived at compile time from Pk / INV32
struct P29 {
  uint32_t l[N9];
  uint32_t inv;
  constexpr P29() : l{}, inv(0) {
    for (int i = 0; i < N9; i++) {
      int bit = 29 * i;
      uint64_t v = 0;
      int w = bit / 32, o = bit % 32;
      if (w < 8) v = Pk[w];
      if (w + 1 < 8) v |= (uint64_t)Pk[w + 1] << 32;
      l[i] = (uint32_t)(v >> o) & M29;
    }
    inv = INV32 & M29;   // -p^-1 mod 2^32 reduced mod 2^29 is -p^-1 mod 2^29
  }
};
static constexpr P29 kP29{};

__device__ __forceinline__ void mul29(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t acc = 0;
  uint32_t m[N9];
#pragma unroll
  for (int k = 0; k < N9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * kP29.l[k - i];
    m[k] = ((uint32_t)acc * kP29.inv) & M29;
    acc += (uint64_t)m[k] * kP29.l[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = N9; k < 2 * N9 - 1; k++) {
#pragma unroll
    for (int i = k - N9 + 1; i < N9; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
    for (int i = k - N9 + 1; i < N9; i++) acc += (uint64_t)m[i] * kP29.l[k - i];
    r[k - N9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r[N9 - 1] = (uint32_t)acc;
}
// a^2 with the doubled operand: 45 products
__device__ __forceinline__ void sqr29(uint32_t* r, const uint32_t* a) {
  uint32_t a2[N9];
#pragma unroll
  for (int i = 0; i < N9; i++) a2[i] = a[i] << 1;
  uint64_t acc = 0;
  uint32_t m[N9];
#pragma unroll
  for (int k = 0; k < N9; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a[k - i];
    if (k % 2 == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * kP29.l[k - i];
    m[k] = ((uint32_t)acc * kP29.inv) & M29;
    acc += (uint64_t)m[k] * kP29.l[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = N9; k < 2 * N9 - 1; k++) {
#pragma unroll
    for (int i = k - N9 + 1; 2 * i < k; i++) acc += (uint64_t)a2[i] * a[k - i];
    if (k % 2 == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = k - N9 + 1; i < N9; i++) acc += (uint64_t)m[i] * kP29.l[k - i];
    r[k - N9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r[N9 - 1] = (uint32_t)acc;
}

template <int V> __global__ void chain29(uint32_t* out, const uint32_t* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[N9], b[N9];
  for (int i = 0; i < N9; i++) { a[i] = in[tid * 18 + i]; b[i] = in[tid * 18 + 9 + i]; }
  for (int it = 0; it < iters; it++) { if (V == 0) mul29(a, a, b); else { sqr29(a, a); } }
  for (int i = 0; i < N9; i++) out[tid * 9 + i] = a[i];
}
// a "mixed-add shaped" mix: 8 products + 2 squares + 7 lazy additions + 4 normalisations per iteration
__device__ __forceinline__ void norm29(uint32_t* a) {
#pragma unroll
  for (int i = 0; i < N9 - 1; i++) { a[i + 1] += a[i] >> 29; a[i] &= M29; }
}
__global__ void mix29(uint32_t* out, const uint32_t* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x[N9], y[N9], z[N9], w[N9], t[N9], u[N9];
  for (int i = 0; i < N9; i++) { x[i] = in[tid * 18 + i]; y[i] = in[tid * 18 + 9 + i]; z[i] = x[i] ^ 5; w[i] = y[i] ^ 9; }
  for (int it = 0; it < iters; it++) {
    mul29(t, x, z); mul29(u, y, w);
    for (int i = 0; i < N9; i++) { t[i] += x[i]; u[i] += y[i]; }
    norm29(t); norm29(u);
    sqr29(x, t); mul29(y, t, x); mul29(z, z, x); mul29(w, w, y); mul29(t, t, x);
    sqr29(u, u);
    for (int i = 0; i < N9; i++) { x[i] = u[i] + y[i] + t[i]; }
    norm29(x);
    for (int i = 0; i < N9; i++) { u[i] = t[i] + x[i]; }
    mul29(u, u, z); mul29(t, y, w);
    for (int i = 0; i < N9; i++) { y[i] = u[i] + t[i]; }
    norm29(y);
  }
  for (int i = 0; i < N9; i++) out[tid * 9 + i] = x[i] ^ y[i] ^ z[i] ^ w[i];
}
template <int V> __global__ void chain32(uint32_t* out, const uint32_t* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[NL], b[NL];
  for (int i = 0; i < NL; i++) { a[i] = in[tid * 18 + i]; b[i] = in[tid * 18 + 9 + i]; }
  a[7] &= 0x1fffffff; b[7] &= 0x1fffffff;
  for (int it = 0; it < iters; it++) mont_mul_D(a, a, b);
  for (int i = 0; i < NL; i++) out[tid * 9 + i] = a[i];
}

// ---- host check: out * 2^5 == a * b * 2^-256 (mod p) ---------------------------------------------------
static void pack32(uint32_t* o, const uint32_t* l29) {   // 9 x 29-bit (possibly unnormalised top) -> 9 x 32-bit words
  unsigned __int128 acc = 0; int bits = 0, w = 0;
  for (int i = 0; i < 9; i++) { acc |= (unsigned __int128)l29[i] << bits; bits += 29; while (bits >= 32 && w < 9) { o[w++] = (uint32_t)acc; acc >>= 32; bits -= 32; } }
  while (w < 9) { o[w++] = (uint32_t)acc; acc >>= 32; }
}
static void mod_p(uint32_t* v9) {   // v (9 words) mod p by repeated subtraction of shifted p (value < 2^264)
  for (int sh = 10; sh >= 0; sh--) {
    uint32_t ps[9] = {0};
    for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)Pk[i] << sh; ps[i] |= (uint32_t)t; ps[i + 1] |= (uint32_t)(t >> 32); }
    for (;;) {
      int ge = 1;
      for (int i = 8; i >= 0; i--) if (v9[i] != ps[i]) { ge = v9[i] > ps[i]; break; }
      if (!ge) break;
      uint64_t br = 0;
      for (int i = 0; i < 9; i++) { uint64_t d = (uint64_t)v9[i] - ps[i] - br; v9[i] = (uint32_t)d; br = (d >> 32) & 1; }
    }
  }
}
static uint64_t sm(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  const int threads = 256, blocks = cus * 8, n = blocks * threads;
  std::vector<uint32_t> h_in((size_t)n * 18), h_o((size_t)n * 9);
  for (size_t i = 0; i < h_in.size(); i++) { h_in[i] = (uint32_t)sm(i) & M29; if (i % 9 == 8) h_in[i] &= 0x3fffff; }   // value < 2^254
  uint32_t *d_in, *d_out;
  CHECK(hipMalloc(&d_in, h_in.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)n * 36));
  CHECK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  for (int v = 0; v < 2; v++) {
    if (v == 0) hipLaunchKernelGGL(chain29<0>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 1);
    else hipLaunchKernelGGL(chain29<1>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 1);
    CHECK(hipMemcpy(h_o.data(), d_out, (size_t)n * 36, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int t = 0; t < 20000; t++) {
      uint32_t a9[9], b9[9], o9[9];
      pack32(a9, &h_in[(size_t)t * 18]); pack32(b9, &h_in[(size_t)t * 18 + 9]); pack32(o9, &h_o[(size_t)t * 9]);
      mod_p(a9); mod_p(b9);
      uint32_t ref[8];
      mont_mul_A(ref, a9, v == 0 ? b9 : a9);          // a*b*2^-256 mod p, fully reduced
      // o * 32 mod p
      uint64_t c = 0;
      for (int i = 0; i < 9; i++) { uint64_t s = ((uint64_t)o9[i] << 5) | c; o9[i] = (uint32_t)s; c = s >> 32; }
      mod_p(o9);
      for (int i = 0; i < 8; i++) if (o9[i] != ref[i]) { bad++; break; }
    }
    printf("correctness: %s vs host big-number evaluation: %zu mismatches of 20000\n", v == 0 ? "mul29" : "sqr29", bad);
  }
  auto timeit = [&](const char* name, void (*k)(uint32_t*, const uint32_t*, int), double mulsPerIter, int blocks_per_cu, int iters) {
    const int blocks = cus * blocks_per_cu; const int n = blocks * threads;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)n * iters * mulsPerIter;
    printf("%-40s waves/SIMD %d  %8.3f ms  %8.2f G/s\n", name, blocks_per_cu, ms, muls / (ms * 1e-3) * 1e-9);
  };
  for (int w : {8, 4, 2, 1}) {
    timeit("D (32-bit limbs, shipped r1) product", chain32<0>, 1, w, 2000);
    timeit("E (29-bit limbs) product", chain29<0>, 1, w, 2000);
    timeit("F (29-bit limbs) square", chain29<1>, 1, w, 2000);
    timeit("mixed-add shaped mix (10 mul-equiv/iter)", mix29, 10, w, 200);
  }
  return 0;
}
