// Does a v_mov_b32_dpp that reads the result limbs of an inline-asm product (fp29_asm_gen.h) see them?  K wait states are
// placed between the product and the broadcast (0 = what hipcc emits on its own); the same limbs are also broadcast with
// v_readlane.  Prints, per K, the number of lanes whose DPP broadcast differs from the readlane one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I distributed-groth16_amd/csrc tools/ubench/dpp_hazard.hip -o /tmp/dpp_hazard
#include "msm_impl.h"
#include <cstdio>
using namespace dg16;
using P = bn254_fq_params;
using S = Fe<P, 448, 1>;
template <int K>
__global__ void __launch_bounds__(64) probe(const S* in, S* out) {
  const unsigned lane = threadIdx.x, slot = (lane & 15) >> 2;
  const S a = in[0], b = in[1], c = in[2];
  const S x = select(slot == 0, a, select(slot == 1, b, c));
  S t = fit<448>(x * x);                                   // one product per lane, three different values per row
  if constexpr (K > 0)
    asm volatile("s_nop %9" : "+v"(t.l[0]), "+v"(t.l[1]), "+v"(t.l[2]), "+v"(t.l[3]), "+v"(t.l[4]), "+v"(t.l[5]),
                 "+v"(t.l[6]), "+v"(t.l[7]), "+v"(t.l[8]) : "n"(K - 1));
  S d, r;
#pragma unroll
  for (int i = 0; i < 9; i++) d.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t.l[i], 0x150 + 4, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)t.l[i], 4);
  out[2 * lane] = d;
  out[2 * lane + 1] = r;
}
template <int K>
static void run(const S* din, S* dout) {
  hipLaunchKernelGGL(probe<K>, dim3(1), dim3(64), 0, 0, din, dout);
  S h[128];
  (void)hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int i = 0; i < 9; i++)
      if (h[2 * l].l[i] != h[2 * l + 1].l[i]) { bad++; break; }
  printf("wait states %d: %d of 64 lanes differ\n", K, bad);
}
int main() {
  S hin[3];
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < 9; i++) hin[k].l[i] = (0x1234567u * (k + 1) + 0x9e3779u * i) & 0x1fffffffu;
  for (int k = 0; k < 3; k++) hin[k].l[8] &= 0x3fffff;
  S *din, *dout;
  (void)hipMalloc(&din, sizeof hin);
  (void)hipMalloc(&dout, 128 * sizeof(S));
  (void)hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
  run<0>(din, dout); run<1>(din, dout); run<2>(din, dout); run<4>(din, dout); run<8>(din, dout);
  return 0;
}
