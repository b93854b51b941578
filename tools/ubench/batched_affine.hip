// Batched-affine bucket accumulation, MEASURED: cost per point addition of
//   (X) the shipped XYZZ mixed addition (8M + 2S, no inversion), and
//   (B) affine additions with ONE shared inversion per workgroup and round (Montgomery's trick: lane-local prefix
//       products over A independent additions per lane, a product tree over the lanes of the workgroup in LDS, one
//       Fermat inversion by one lane, the tree back down, back-substitution; then lambda, x3, y3: 5M + 1S per addition
//       + the shared part),
// on the library's own reduced-radix field (csrc/fp29.h, BN254 Fq), operands gathered from a table in HBM exactly as the
// accumulation kernel gathers them (random 64-byte rows), A = 1, 2, 4, 8 accumulators per lane, workgroups of 256 and
// 1024 lanes.  Output: JSON lines (ns per addition chip-wide, G additions/s) + a correctness check of (B) against (X).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../distributed-groth16_amd/csrc/consts_gen.h"
#include "../../distributed-groth16_amd/csrc/ec29.h"

using namespace dg16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

using P = bn254_fq_params;
using F = Fp<P>;
using T = RR<P>;
using E = Fe<P, 130, 1>;          // storage type of the affine accumulators (< ~2 p, normalised)
constexpr int N = T::N;

__device__ __forceinline__ E ld_e(const uint32_t* w) { return fe_from_words<P>(w).template as<130, 1>(); }

// a^(p - 2)
__device__ __noinline__ E fe_inv(E a) {
  E acc = fe_one<P>().template as<130, 1>();
  bool started = false;
  for (int li = P::NL - 1; li >= 0; li--) {
    const uint32_t w = F::pm2_limb(li);
    for (int b = 31; b >= 0; b--) {
      if (started) acc = fit<130>(sqr(acc));
      if ((w >> b) & 1) {
        acc = fit<130>(acc * a);
        started = true;
      }
    }
  }
  return acc;
}

// ---- (X) XYZZ mixed additions: lane = chain of `iters` gathers into one accumulator -----------------------------------
__global__ void __launch_bounds__(256) xyzz_kernel(const uint32_t* __restrict__ table, unsigned tmask, int iters,
                                                    uint32_t* __restrict__ out) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned idx = t * 2654435761u;
  XYZZ29<F> acc = XYZZ29<F>::inf();
  for (int j = 0; j < iters; j++) {
    idx = idx * 1664525u + 1013904223u;
    const Affine29<F> p = Affine29<F>::load(table + (size_t)(idx & tmask) * 16);
    acc = acc.madd(p, false);
  }
  const auto c = canon(acc.x * acc.zzz + acc.y * acc.zz);
  out[t] = c.l[0] ^ c.l[3];
}

// ---- (B) batched affine: A accumulators per lane, one inversion per workgroup and round -------------------------------
template <int A, int BLOCK>
__global__ void __launch_bounds__(BLOCK) affine_kernel(const uint32_t* __restrict__ table, unsigned tmask, int rounds,
                                                        uint32_t* __restrict__ out, uint32_t* __restrict__ check) {
  __shared__ uint32_t tree[2 * BLOCK][N];       // product tree over the lanes (node 1 = root, leaves BLOCK .. 2 BLOCK - 1)
  const unsigned lane = threadIdx.x;
  const unsigned t = blockIdx.x * BLOCK + lane;
  unsigned idx = t * 2654435761u;
  E ax[A], ay[A];
  auto st = [&](unsigned node, const E& v) {
#pragma unroll
    for (int i = 0; i < N; i++) tree[node][i] = v.l[i];
  };
  auto ldn = [&](unsigned node) {
    E v;
#pragma unroll
    for (int i = 0; i < N; i++) v.l[i] = tree[node][i];
    return v;
  };
#pragma unroll
  for (int k = 0; k < A; k++) {                 // accumulators start at table points (distinct x with overwhelming odds)
    idx = idx * 1664525u + 1013904223u;
    const uint32_t* row = table + (size_t)(idx & tmask) * 16;
    ax[k] = ld_e(row);
    ay[k] = ld_e(row + 8);
  }
  for (int r = 0; r < rounds; r++) {
    E px[A], py[A], d[A], pre[A];
#pragma unroll
    for (int k = 0; k < A; k++) {
      idx = idx * 1664525u + 1013904223u;
      const uint32_t* row = table + (size_t)(idx & tmask) * 16;
      px[k] = ld_e(row);
      py[k] = ld_e(row + 8);
      d[k] = fit<130>(px[k] - ax[k]);
      pre[k] = k ? fit<130>(pre[k - 1] * d[k]) : d[k];
    }
    // up-sweep over the lanes
    st(BLOCK + lane, pre[A - 1]);
    __syncthreads();
#pragma unroll 1
    for (unsigned width = BLOCK / 2; width >= 1; width >>= 1) {
      if (lane < width) st(width + lane, fit<130>(ldn(2 * (width + lane)) * ldn(2 * (width + lane) + 1)));
      __syncthreads();
    }
    if (lane == 0) st(1, fe_inv(ldn(1)));       // ONE inversion per workgroup and round
    __syncthreads();
    // down-sweep: node i holds the inverse of its subtree's product
#pragma unroll 1
    for (unsigned width = 1; width < BLOCK; width <<= 1) {
      if (lane < width) {
        const unsigned i = width + lane;
        const E inv = ldn(i), l = ldn(2 * i), rr = ldn(2 * i + 1);
        st(2 * i, fit<130>(inv * rr));
        st(2 * i + 1, fit<130>(inv * l));
      }
      __syncthreads();
    }
    E inv = ldn(BLOCK + lane);                  // 1 / (d_0 .. d_{A-1}) of this lane
    __syncthreads();
#pragma unroll
    for (int k = A - 1; k >= 0; k--) {
      const E dinv = k ? fit<130>(inv * pre[k - 1]) : inv;
      if (k) inv = fit<130>(inv * d[k]);
      const auto lam = fit<130>((py[k] - ay[k]) * dinv);
      const auto x3 = fit<130>(sqr(lam) - (ax[k] + px[k]));
      ay[k] = fit<130>(lam * (ax[k] - x3) - ay[k]);
      ax[k] = x3;
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < A; k++) {
    const auto c = canon(ax[k] + ay[k]);
    acc ^= c.l[0] ^ c.l[3];
    if (check && t == 0 && k == 0) {
      const auto cx = canon(ax[0]);
      const auto cy = canon(ay[0]);
      for (int i = 0; i < N; i++) { check[i] = cx.l[i]; check[N + i] = cy.l[i]; }
    }
  }
  out[t] = acc;
}

// reference for the check: the same chain of lane 0, accumulator 0 with XYZZ additions, converted to affine
template <int A>
__global__ void ref_kernel(const uint32_t* __restrict__ table, unsigned tmask, int rounds, uint32_t* __restrict__ check) {
  unsigned idx = 0;
  XYZZ29<F> acc = XYZZ29<F>::inf();
  for (int k = 0; k < A; k++) {
    idx = idx * 1664525u + 1013904223u;
    if (k == 0) acc = acc.madd(Affine29<F>::load(table + (size_t)(idx & tmask) * 16), false);
  }
  for (int r = 0; r < rounds; r++)
    for (int k = 0; k < A; k++) {
      idx = idx * 1664525u + 1013904223u;
      if (k == 0) acc = acc.madd(Affine29<F>::load(table + (size_t)(idx & tmask) * 16), false);
    }
  const E zi3 = fe_inv(fit<130>(acc.zzz));
  const auto zi2 = sqr(zi3 * acc.zz);
  const auto cx = canon(acc.x * zi2);
  const auto cy = canon(acc.y * zi3);
  for (int i = 0; i < N; i++) { check[i] = cx.l[i]; check[N + i] = cy.l[i]; }
}

// table of 2^log_rows on-curve points in the internal form: multiples of the generator by a walk (P_{i+1} = P_i + G)
__global__ void table_kernel(uint32_t* table, unsigned rows_per_lane) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  Affine<F> g;
  for (int i = 0; i < P::NL; i++) { g.x.l[i] = bn254_g1_consts::GX[i]; g.y.l[i] = bn254_g1_consts::GY[i]; }
  uint32_t k[2] = {t * rows_per_lane + 1, 0};
  XYZZ<F> cur = scalar_mul<F, 2>(XYZZ<F>::from_affine(g), k);
  for (unsigned r = 0; r < rows_per_lane; r++) {
    affine_to_internal(cur.to_affine(), table + ((size_t)t * rows_per_lane + r) * 16);
    cur = cur.madd(g, false);
  }
}

template <int A, int BLOCK>
static void run_b(const uint32_t* d_table, unsigned tmask, int cus, uint32_t* d_out, uint32_t* d_chk, double xyzz_ns) {
  const int rounds = 64 / A > 8 ? 64 / A : 8;
  const int blocks = cus * (1024 / BLOCK) * 2;          // two rounds of workgroups over the chip
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((affine_kernel<A, BLOCK>), dim3(blocks), dim3(BLOCK), 0, 0, d_table, tmask, rounds, d_out, d_chk);
  CHECK(hipDeviceSynchronize());
  uint32_t h[2][2 * N];
  CHECK(hipMemcpy(h[0], d_chk, sizeof h[0], hipMemcpyDeviceToHost));
  hipLaunchKernelGGL((ref_kernel<A>), dim3(1), dim3(1), 0, 0, d_table, tmask, rounds, d_chk);
  CHECK(hipMemcpy(h[1], d_chk, sizeof h[1], hipMemcpyDeviceToHost));
  bool ok = true;
  for (int i = 0; i < 2 * N; i++) ok = ok && h[0][i] == h[1][i];
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((affine_kernel<A, BLOCK>), dim3(blocks), dim3(BLOCK), 0, 0, d_table, tmask, rounds, d_out, nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double adds = (double)blocks * BLOCK * A * rounds;
  const double ns = best * 1e6 / adds;
  printf("{\"variant\": \"batched_affine\", \"adds_per_lane_per_inversion\": %d, \"workgroup\": %d, \"ms\": %.3f, \"additions\": %.0f, "
         "\"ns_per_addition_chipwide\": %.5f, \"G_additions_per_s\": %.2f, \"vs_xyzz\": %.2f, \"matches_xyzz_chain\": %s}\n",
         A, BLOCK, best, adds, ns, 1.0 / ns, ns / xyzz_ns, ok ? "true" : "false");
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const unsigned log_rows = 20, rows = 1u << log_rows;       // 64 MB of points: HBM / MALL gathers like a key table
  uint32_t *d_table, *d_out, *d_chk;
  CHECK(hipMalloc(&d_table, (size_t)rows * 64));
  CHECK(hipMalloc(&d_out, (size_t)cus * 2048 * 4 * 4));
  CHECK(hipMalloc(&d_chk, 256));
  hipLaunchKernelGGL(table_kernel, dim3(rows / 64 / 64), dim3(64), 0, 0, d_table, 64u);
  CHECK(hipDeviceSynchronize());
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"field\": \"bn254_fq (fp29.h)\", \"table_rows\": %u}\n", p.gcnArchName, cus, rows);
  // (X) baseline: chains of 16 mixed additions per lane, 4 waves per SIMD x 4 rounds
  double xyzz_ns = 0;
  {
    const int iters = 16, blocks = cus * 4 * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(xyzz_kernel, dim3(blocks), dim3(256), 0, 0, d_table, rows - 1, iters, d_out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    const double adds = (double)blocks * 256 * iters;
    xyzz_ns = best * 1e6 / adds;
    printf("{\"variant\": \"xyzz_mixed_add\", \"ms\": %.3f, \"additions\": %.0f, \"ns_per_addition_chipwide\": %.5f, "
           "\"G_additions_per_s\": %.2f}\n", best, adds, xyzz_ns, 1.0 / xyzz_ns);
  }
  run_b<1, 256>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  run_b<2, 256>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  run_b<4, 256>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  run_b<8, 256>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  run_b<2, 1024>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  run_b<4, 1024>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  run_b<8, 1024>(d_table, rows - 1, cus, d_out, d_chk, xyzz_ns);
  return 0;
}
