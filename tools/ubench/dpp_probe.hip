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
  // the forms hipcc's DPP combiner makes of a broadcast feeding a subtraction: VOP2 with the DPP on src0
  int a = lane * 7 + 5, b = 100000 + lane, r1, r2;
  asm volatile("s_nop 1\n\tv_sub_u32_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r1) : "v"(a), "v"(b));
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r2) : "v"(a), "v"(b));
  out[192 + lane] = r1;                         // expected: a[lane 8 of the row] - b[lane]
  out[256 + lane] = r2;                         // expected: b[lane] - a[lane 4 of the row]
}
int main() {
  int* d; hipMalloc(&d, 320 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int k = 0; k < 3; k++) { printf("set %d:", k); for (int i = 0; i < 64; i++) printf(" %d", h[64 * k + i]); printf("\n"); }
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 64; i++) {
    const int row = i & ~15;
    if (h[192 + i] != ((row + 8) * 7 + 5) - (100000 + i)) bad1++;
    if (h[256 + i] != (100000 + i) - ((row + 4) * 7 + 5)) bad2++;
  }
  printf("v_sub_u32_dpp row_newbcast: %d of 64 lanes wrong; v_subrev_u32_dpp row_newbcast: %d of 64 lanes wrong\n", bad1, bad2);
  printf("v_sub_u32_dpp lanes 0..3: %d %d %d %d (expected %d %d %d %d)\n", h[192], h[193], h[194], h[195], 61 - 100000, 61 - 100001, 61 - 100002, 61 - 100003);
  return 0;
}
