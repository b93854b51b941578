// Chip-wide rate of the LIBRARY's reduced-radix field product and square (distributed-groth16_amd/csrc/fp29.h, the
// arithmetic of the bucket kernels and the NTT) for every base field: a dependent chain per lane, waves/SIMD swept.
// Output is JSON on stdout (one object): bench.py reads the committed copy (profiles/r3_valu_constants.json) as the
// VALU roof of `valu_roofline` -- measured constants live in profiles/, not in the ABI.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../distributed-groth16_amd/csrc/consts_gen.h"
#include "../../distributed-groth16_amd/csrc/fp29.h"

using namespace dg16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class P, int V>
__global__ void __launch_bounds__(256) chain(uint32_t* out, const uint32_t* in, int iters) {
  using T = RR<P>;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fe<P, 68, 1> a, b;
  for (int i = 0; i < T::N; i++) {
    a.l[i] = in[(tid * 2) * 16 % 4096 + i] & T::MASK;
    b.l[i] = in[(tid * 2 + 1) * 16 % 4096 + i] & T::MASK;
  }
  a.l[T::N - 1] &= T::PTOP >> 1;
  b.l[T::N - 1] &= T::PTOP >> 1;
  for (int it = 0; it < iters; it++) {
    if (V == 0) {
      const auto c = a * b;
      a = b;
      b = c.template as<68, 1>();
    } else {
      a = sqr(a).template as<68, 1>();
    }
  }
  uint32_t acc = 0;
  for (int i = 0; i < T::N; i++) acc ^= a.l[i] ^ b.l[i];
  out[tid] = acc;
}

template <class P>
static void run(const char* name, int cus, uint32_t* d_out, const uint32_t* d_in, bool last) {
  using T = RR<P>;
  printf("  \"%s\": {\"limb_bits\": %d, \"limbs\": %d, \"mads_per_product\": %d, \"mads_per_square\": %d,\n", name, T::W,
         T::N, 2 * T::N * T::N, T::N * (T::N + 1) / 2 + T::N * T::N);
  const int iters = 2000;
  double best[2] = {0, 0};
  for (int v = 0; v < 2; v++) {
    printf("    \"%s_G_per_s_by_waves_per_simd\": {", v == 0 ? "product" : "square");
    bool first = true;
    for (int w : {8, 4, 2, 1}) {
      const int blocks = cus * w;
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      auto launch = [&](int it) {
        if (v == 0) hipLaunchKernelGGL((chain<P, 0>), dim3(blocks), dim3(256), 0, 0, d_out, d_in, it);
        else hipLaunchKernelGGL((chain<P, 1>), dim3(blocks), dim3(256), 0, 0, d_out, d_in, it);
      };
      launch(iters);                      // warm-up at full length: clocks and power state settle under this load
      CHECK(hipDeviceSynchronize());
      float ms = 1e30f;
      for (int rep = 0; rep < 3; rep++) {  // best of three
        CHECK(hipEventRecord(e0));
        launch(iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (t < ms) ms = t;
      }
      const double g = (double)blocks * 256 * iters / (ms * 1e-3) * 1e-9;
      if (g > best[v]) best[v] = g;
      printf("%s\"%d\": %.2f", first ? "" : ", ", w, g);
      first = false;
    }
    printf("},\n");
  }
  printf("    \"product_G_per_s\": %.2f, \"square_G_per_s\": %.2f}%s\n", best[0], best[1], last ? "" : ",");
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  std::vector<uint32_t> h(4096 + 64);
  uint64_t x = 88172645463325252ULL;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
  uint32_t *d_in, *d_out;
  CHECK(hipMalloc(&d_in, h.size() * 4));
  CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
  CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"source\": \"tools/ubench/fe_rate.hip (fp29.h products, dependent chain per lane)\",\n",
         p.gcnArchName, cus);
  hipLaunchKernelGGL((chain<bn254_fr_params, 0>), dim3(cus * 8), dim3(256), 0, 0, d_out, d_in, 20000);   // ~100 ms of load
  CHECK(hipDeviceSynchronize());
  run<bn254_fq_params>("bn254_fq", cus, d_out, d_in, false);
  run<bls12_381_fq_params>("bls12_381_fq", cus, d_out, d_in, false);
  run<bls12_377_fq_params>("bls12_377_fq", cus, d_out, d_in, false);
  run<bn254_fr_params>("bn254_fr", cus, d_out, d_in, false);
  run<bls12_381_fr_params>("bls12_381_fr", cus, d_out, d_in, false);
  run<bn254_fq_params>("bn254_fq_again", cus, d_out, d_in, true);     // order check: the first field measured once more
  printf("}\n");
  return 0;
}
