// What v_mov_b32_dpp row_newbcast:N does on this device, lane by lane (and next to a VALU write of its source).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) probe(int* out) {
  const int lane = threadIdx.x;
  int v = lane * 10;
  out[lane] = __builtin_amdgcn_update_dpp(0, v, 0x150 + 4, 0xf, 0xf, false);
  out[64 + lane] = __builtin_amdgcn_update_dpp(0, v, 0x150 + 12, 0xf, 0xf, false);
  int w = lane * 3 + 1;                         // freshly written source
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, w, 0x150 + 8, 0xf, 0xf, false);
}
int main() {
  int* d; hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int k = 0; k < 3; k++) { printf("set %d:", k); for (int i = 0; i < 64; i++) printf(" %d", h[64 * k + i]); printf("\n"); }
  return 0;
}
