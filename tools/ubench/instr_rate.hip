// Instruction-throughput microbenchmark for the integer/fp64 ops a 256-bit Montgomery multiply can
// be built from on gfx950.  Each kernel issues ITER x 16 independent instances of one instruction
// per lane; the host reports lane-ops per clock per CU.  Build: hipcc --offload-arch=gfx950 -O3
// instr_rate.hip -o instr_rate ; run on the GPU box.  Results are recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL64(NAME, ASMSTR)                                                          \
__global__ void NAME(uint64_t* out, int iters, uint32_t a, uint32_t b, long long* cyc) {   \
  uint64_t r[16];                                                                               \
  uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x; for (int i = 0; i < 16; i++) r[i] = (uint64_t)threadIdx.x * (i + 1);                                                                                       \
  long long t0 = clock64();                                                               \
  for (int it = 0; it < iters; it++) {                                                    \
    asm volatile(ASMSTR : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])   \
                 : "v"(x), "v"(y) : "vcc");                                                \
  }                                                                                       \
  long long t1 = clock64();                                                               \
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;                                \
  uint64_t s = 0; for (int i = 0; i < 16; i++) s ^= r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                                       \
}
#define KERNEL32(NAME, ASMSTR)                                                          \
__global__ void NAME(uint64_t* out, int iters, uint32_t a, uint32_t b, long long* cyc) {   \
  uint32_t r[16];                                                                               \
  uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x; for (int i = 0; i < 16; i++) r[i] = threadIdx.x * (i + 1);                                                                                       \
  long long t0 = clock64();                                                               \
  for (int it = 0; it < iters; it++) {                                                    \
    asm volatile(ASMSTR : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])   \
                 : "v"(x), "v"(y) : "vcc");                                                \
  }                                                                                       \
  long long t1 = clock64();                                                               \
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;                                \
  uint32_t s = 0; for (int i = 0; i < 16; i++) s ^= r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                                       \
}
#define KERNELF64(NAME, ASMSTR)                                                          \
__global__ void NAME(uint64_t* out, int iters, uint32_t a, uint32_t b, long long* cyc) {   \
  double r[16];                                                                               \
  double x = 1.0 + a * 1e-9 + threadIdx.x * 1e-6, y = 1.0 - b * 1e-9; for (int i = 0; i < 16; i++) r[i] = threadIdx.x * (i + 1);                                                                                       \
  long long t0 = clock64();                                                               \
  for (int it = 0; it < iters; it++) {                                                    \
    asm volatile(ASMSTR : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])   \
                 : "v"(x), "v"(y) : "vcc");                                                \
  }                                                                                       \
  long long t1 = clock64();                                                               \
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;                                \
  double s = 0; for (int i = 0; i < 16; i++) s += r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;                                                                                       \
}
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_mad_u64_u32 %1, vcc, %16, %17, %1\n\tv_mad_u64_u32 %2, vcc, %16, %17, %2\n\tv_mad_u64_u32 %3, vcc, %16, %17, %3\n\tv_mad_u64_u32 %4, vcc, %16, %17, %4\n\tv_mad_u64_u32 %5, vcc, %16, %17, %5\n\tv_mad_u64_u32 %6, vcc, %16, %17, %6\n\tv_mad_u64_u32 %7, vcc, %16, %17, %7\n\tv_mad_u64_u32 %8, vcc, %16, %17, %8\n\tv_mad_u64_u32 %9, vcc, %16, %17, %9\n\tv_mad_u64_u32 %10, vcc, %16, %17, %10\n\tv_mad_u64_u32 %11, vcc, %16, %17, %11\n\tv_mad_u64_u32 %12, vcc, %16, %17, %12\n\tv_mad_u64_u32 %13, vcc, %16, %17, %13\n\tv_mad_u64_u32 %14, vcc, %16, %17, %14\n\tv_mad_u64_u32 %15, vcc, %16, %17, %15")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0\n\tv_lshl_add_u64 %1, %1, 0, %1\n\tv_lshl_add_u64 %2, %2, 0, %2\n\tv_lshl_add_u64 %3, %3, 0, %3\n\tv_lshl_add_u64 %4, %4, 0, %4\n\tv_lshl_add_u64 %5, %5, 0, %5\n\tv_lshl_add_u64 %6, %6, 0, %6\n\tv_lshl_add_u64 %7, %7, 0, %7\n\tv_lshl_add_u64 %8, %8, 0, %8\n\tv_lshl_add_u64 %9, %9, 0, %9\n\tv_lshl_add_u64 %10, %10, 0, %10\n\tv_lshl_add_u64 %11, %11, 0, %11\n\tv_lshl_add_u64 %12, %12, 0, %12\n\tv_lshl_add_u64 %13, %13, 0, %13\n\tv_lshl_add_u64 %14, %14, 0, %14\n\tv_lshl_add_u64 %15, %15, 0, %15")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 29, %0\n\tv_lshrrev_b64 %1, 29, %1\n\tv_lshrrev_b64 %2, 29, %2\n\tv_lshrrev_b64 %3, 29, %3\n\tv_lshrrev_b64 %4, 29, %4\n\tv_lshrrev_b64 %5, 29, %5\n\tv_lshrrev_b64 %6, 29, %6\n\tv_lshrrev_b64 %7, 29, %7\n\tv_lshrrev_b64 %8, 29, %8\n\tv_lshrrev_b64 %9, 29, %9\n\tv_lshrrev_b64 %10, 29, %10\n\tv_lshrrev_b64 %11, 29, %11\n\tv_lshrrev_b64 %12, 29, %12\n\tv_lshrrev_b64 %13, 29, %13\n\tv_lshrrev_b64 %14, 29, %14\n\tv_lshrrev_b64 %15, 29, %15")
KERNEL32(k_and_b32_e32, "v_and_b32 %0, %16, %0\n\tv_and_b32 %1, %16, %1\n\tv_and_b32 %2, %16, %2\n\tv_and_b32 %3, %16, %3\n\tv_and_b32 %4, %16, %4\n\tv_and_b32 %5, %16, %5\n\tv_and_b32 %6, %16, %6\n\tv_and_b32 %7, %16, %7\n\tv_and_b32 %8, %16, %8\n\tv_and_b32 %9, %16, %9\n\tv_and_b32 %10, %16, %10\n\tv_and_b32 %11, %16, %11\n\tv_and_b32 %12, %16, %12\n\tv_and_b32 %13, %16, %13\n\tv_and_b32 %14, %16, %14\n\tv_and_b32 %15, %16, %15")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %16, %0\n\tv_mul_lo_u32 %1, %16, %1\n\tv_mul_lo_u32 %2, %16, %2\n\tv_mul_lo_u32 %3, %16, %3\n\tv_mul_lo_u32 %4, %16, %4\n\tv_mul_lo_u32 %5, %16, %5\n\tv_mul_lo_u32 %6, %16, %6\n\tv_mul_lo_u32 %7, %16, %7\n\tv_mul_lo_u32 %8, %16, %8\n\tv_mul_lo_u32 %9, %16, %9\n\tv_mul_lo_u32 %10, %16, %10\n\tv_mul_lo_u32 %11, %16, %11\n\tv_mul_lo_u32 %12, %16, %12\n\tv_mul_lo_u32 %13, %16, %13\n\tv_mul_lo_u32 %14, %16, %14\n\tv_mul_lo_u32 %15, %16, %15")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %16, %0\n\tv_mul_hi_u32 %1, %16, %1\n\tv_mul_hi_u32 %2, %16, %2\n\tv_mul_hi_u32 %3, %16, %3\n\tv_mul_hi_u32 %4, %16, %4\n\tv_mul_hi_u32 %5, %16, %5\n\tv_mul_hi_u32 %6, %16, %6\n\tv_mul_hi_u32 %7, %16, %7\n\tv_mul_hi_u32 %8, %16, %8\n\tv_mul_hi_u32 %9, %16, %9\n\tv_mul_hi_u32 %10, %16, %10\n\tv_mul_hi_u32 %11, %16, %11\n\tv_mul_hi_u32 %12, %16, %12\n\tv_mul_hi_u32 %13, %16, %13\n\tv_mul_hi_u32 %14, %16, %14\n\tv_mul_hi_u32 %15, %16, %15")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %16, %0\n\tv_mul_u32_u24 %1, %16, %1\n\tv_mul_u32_u24 %2, %16, %2\n\tv_mul_u32_u24 %3, %16, %3\n\tv_mul_u32_u24 %4, %16, %4\n\tv_mul_u32_u24 %5, %16, %5\n\tv_mul_u32_u24 %6, %16, %6\n\tv_mul_u32_u24 %7, %16, %7\n\tv_mul_u32_u24 %8, %16, %8\n\tv_mul_u32_u24 %9, %16, %9\n\tv_mul_u32_u24 %10, %16, %10\n\tv_mul_u32_u24 %11, %16, %11\n\tv_mul_u32_u24 %12, %16, %12\n\tv_mul_u32_u24 %13, %16, %13\n\tv_mul_u32_u24 %14, %16, %14\n\tv_mul_u32_u24 %15, %16, %15")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %16, %0\n\tv_mul_hi_u32_u24 %1, %16, %1\n\tv_mul_hi_u32_u24 %2, %16, %2\n\tv_mul_hi_u32_u24 %3, %16, %3\n\tv_mul_hi_u32_u24 %4, %16, %4\n\tv_mul_hi_u32_u24 %5, %16, %5\n\tv_mul_hi_u32_u24 %6, %16, %6\n\tv_mul_hi_u32_u24 %7, %16, %7\n\tv_mul_hi_u32_u24 %8, %16, %8\n\tv_mul_hi_u32_u24 %9, %16, %9\n\tv_mul_hi_u32_u24 %10, %16, %10\n\tv_mul_hi_u32_u24 %11, %16, %11\n\tv_mul_hi_u32_u24 %12, %16, %12\n\tv_mul_hi_u32_u24 %13, %16, %13\n\tv_mul_hi_u32_u24 %14, %16, %14\n\tv_mul_hi_u32_u24 %15, %16, %15")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %16, %17, %0\n\tv_mad_u32_u24 %1, %16, %17, %1\n\tv_mad_u32_u24 %2, %16, %17, %2\n\tv_mad_u32_u24 %3, %16, %17, %3\n\tv_mad_u32_u24 %4, %16, %17, %4\n\tv_mad_u32_u24 %5, %16, %17, %5\n\tv_mad_u32_u24 %6, %16, %17, %6\n\tv_mad_u32_u24 %7, %16, %17, %7\n\tv_mad_u32_u24 %8, %16, %17, %8\n\tv_mad_u32_u24 %9, %16, %17, %9\n\tv_mad_u32_u24 %10, %16, %17, %10\n\tv_mad_u32_u24 %11, %16, %17, %11\n\tv_mad_u32_u24 %12, %16, %17, %12\n\tv_mad_u32_u24 %13, %16, %17, %13\n\tv_mad_u32_u24 %14, %16, %17, %14\n\tv_mad_u32_u24 %15, %16, %17, %15")
KERNEL32(k_add_u32, "v_add_u32 %0, %16, %0\n\tv_add_u32 %1, %16, %1\n\tv_add_u32 %2, %16, %2\n\tv_add_u32 %3, %16, %3\n\tv_add_u32 %4, %16, %4\n\tv_add_u32 %5, %16, %5\n\tv_add_u32 %6, %16, %6\n\tv_add_u32 %7, %16, %7\n\tv_add_u32 %8, %16, %8\n\tv_add_u32 %9, %16, %9\n\tv_add_u32 %10, %16, %10\n\tv_add_u32 %11, %16, %11\n\tv_add_u32 %12, %16, %12\n\tv_add_u32 %13, %16, %13\n\tv_add_u32 %14, %16, %14\n\tv_add_u32 %15, %16, %15")
KERNEL32(k_add_co_u32, "v_add_co_u32 %0, vcc, %16, %0\n\tv_add_co_u32 %1, vcc, %16, %1\n\tv_add_co_u32 %2, vcc, %16, %2\n\tv_add_co_u32 %3, vcc, %16, %3\n\tv_add_co_u32 %4, vcc, %16, %4\n\tv_add_co_u32 %5, vcc, %16, %5\n\tv_add_co_u32 %6, vcc, %16, %6\n\tv_add_co_u32 %7, vcc, %16, %7\n\tv_add_co_u32 %8, vcc, %16, %8\n\tv_add_co_u32 %9, vcc, %16, %9\n\tv_add_co_u32 %10, vcc, %16, %10\n\tv_add_co_u32 %11, vcc, %16, %11\n\tv_add_co_u32 %12, vcc, %16, %12\n\tv_add_co_u32 %13, vcc, %16, %13\n\tv_add_co_u32 %14, vcc, %16, %14\n\tv_add_co_u32 %15, vcc, %16, %15")
KERNEL32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %16, %0, vcc\n\tv_addc_co_u32 %1, vcc, %16, %1, vcc\n\tv_addc_co_u32 %2, vcc, %16, %2, vcc\n\tv_addc_co_u32 %3, vcc, %16, %3, vcc\n\tv_addc_co_u32 %4, vcc, %16, %4, vcc\n\tv_addc_co_u32 %5, vcc, %16, %5, vcc\n\tv_addc_co_u32 %6, vcc, %16, %6, vcc\n\tv_addc_co_u32 %7, vcc, %16, %7, vcc\n\tv_addc_co_u32 %8, vcc, %16, %8, vcc\n\tv_addc_co_u32 %9, vcc, %16, %9, vcc\n\tv_addc_co_u32 %10, vcc, %16, %10, vcc\n\tv_addc_co_u32 %11, vcc, %16, %11, vcc\n\tv_addc_co_u32 %12, vcc, %16, %12, vcc\n\tv_addc_co_u32 %13, vcc, %16, %13, vcc\n\tv_addc_co_u32 %14, vcc, %16, %14, vcc\n\tv_addc_co_u32 %15, vcc, %16, %15, vcc")
KERNEL32(k_add3_u32, "v_add3_u32 %0, %16, %17, %0\n\tv_add3_u32 %1, %16, %17, %1\n\tv_add3_u32 %2, %16, %17, %2\n\tv_add3_u32 %3, %16, %17, %3\n\tv_add3_u32 %4, %16, %17, %4\n\tv_add3_u32 %5, %16, %17, %5\n\tv_add3_u32 %6, %16, %17, %6\n\tv_add3_u32 %7, %16, %17, %7\n\tv_add3_u32 %8, %16, %17, %8\n\tv_add3_u32 %9, %16, %17, %9\n\tv_add3_u32 %10, %16, %17, %10\n\tv_add3_u32 %11, %16, %17, %11\n\tv_add3_u32 %12, %16, %17, %12\n\tv_add3_u32 %13, %16, %17, %13\n\tv_add3_u32 %14, %16, %17, %14\n\tv_add3_u32 %15, %16, %17, %15")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %16, %0, 13\n\tv_alignbit_b32 %1, %16, %1, 13\n\tv_alignbit_b32 %2, %16, %2, 13\n\tv_alignbit_b32 %3, %16, %3, 13\n\tv_alignbit_b32 %4, %16, %4, 13\n\tv_alignbit_b32 %5, %16, %5, 13\n\tv_alignbit_b32 %6, %16, %6, 13\n\tv_alignbit_b32 %7, %16, %7, 13\n\tv_alignbit_b32 %8, %16, %8, 13\n\tv_alignbit_b32 %9, %16, %9, 13\n\tv_alignbit_b32 %10, %16, %10, 13\n\tv_alignbit_b32 %11, %16, %11, 13\n\tv_alignbit_b32 %12, %16, %12, 13\n\tv_alignbit_b32 %13, %16, %13, 13\n\tv_alignbit_b32 %14, %16, %14, 13\n\tv_alignbit_b32 %15, %16, %15, 13")
KERNEL32(k_and_or, "v_and_or_b32 %0, %16, %17, %0\n\tv_and_or_b32 %1, %16, %17, %1\n\tv_and_or_b32 %2, %16, %17, %2\n\tv_and_or_b32 %3, %16, %17, %3\n\tv_and_or_b32 %4, %16, %17, %4\n\tv_and_or_b32 %5, %16, %17, %5\n\tv_and_or_b32 %6, %16, %17, %6\n\tv_and_or_b32 %7, %16, %17, %7\n\tv_and_or_b32 %8, %16, %17, %8\n\tv_and_or_b32 %9, %16, %17, %9\n\tv_and_or_b32 %10, %16, %17, %10\n\tv_and_or_b32 %11, %16, %17, %11\n\tv_and_or_b32 %12, %16, %17, %12\n\tv_and_or_b32 %13, %16, %17, %13\n\tv_and_or_b32 %14, %16, %17, %14\n\tv_and_or_b32 %15, %16, %17, %15")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %16, %17, %0\n\tv_fma_f32 %1, %16, %17, %1\n\tv_fma_f32 %2, %16, %17, %2\n\tv_fma_f32 %3, %16, %17, %3\n\tv_fma_f32 %4, %16, %17, %4\n\tv_fma_f32 %5, %16, %17, %5\n\tv_fma_f32 %6, %16, %17, %6\n\tv_fma_f32 %7, %16, %17, %7\n\tv_fma_f32 %8, %16, %17, %8\n\tv_fma_f32 %9, %16, %17, %9\n\tv_fma_f32 %10, %16, %17, %10\n\tv_fma_f32 %11, %16, %17, %11\n\tv_fma_f32 %12, %16, %17, %12\n\tv_fma_f32 %13, %16, %17, %13\n\tv_fma_f32 %14, %16, %17, %14\n\tv_fma_f32 %15, %16, %17, %15")
KERNEL32(k_mad_i32_i24, "v_mad_i32_i24 %0, %16, %17, %0\n\tv_mad_i32_i24 %1, %16, %17, %1\n\tv_mad_i32_i24 %2, %16, %17, %2\n\tv_mad_i32_i24 %3, %16, %17, %3\n\tv_mad_i32_i24 %4, %16, %17, %4\n\tv_mad_i32_i24 %5, %16, %17, %5\n\tv_mad_i32_i24 %6, %16, %17, %6\n\tv_mad_i32_i24 %7, %16, %17, %7\n\tv_mad_i32_i24 %8, %16, %17, %8\n\tv_mad_i32_i24 %9, %16, %17, %9\n\tv_mad_i32_i24 %10, %16, %17, %10\n\tv_mad_i32_i24 %11, %16, %17, %11\n\tv_mad_i32_i24 %12, %16, %17, %12\n\tv_mad_i32_i24 %13, %16, %17, %13\n\tv_mad_i32_i24 %14, %16, %17, %14\n\tv_mad_i32_i24 %15, %16, %17, %15")
KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %16, %17, %0\n\tv_dot4_u32_u8 %1, %16, %17, %1\n\tv_dot4_u32_u8 %2, %16, %17, %2\n\tv_dot4_u32_u8 %3, %16, %17, %3\n\tv_dot4_u32_u8 %4, %16, %17, %4\n\tv_dot4_u32_u8 %5, %16, %17, %5\n\tv_dot4_u32_u8 %6, %16, %17, %6\n\tv_dot4_u32_u8 %7, %16, %17, %7\n\tv_dot4_u32_u8 %8, %16, %17, %8\n\tv_dot4_u32_u8 %9, %16, %17, %9\n\tv_dot4_u32_u8 %10, %16, %17, %10\n\tv_dot4_u32_u8 %11, %16, %17, %11\n\tv_dot4_u32_u8 %12, %16, %17, %12\n\tv_dot4_u32_u8 %13, %16, %17, %13\n\tv_dot4_u32_u8 %14, %16, %17, %14\n\tv_dot4_u32_u8 %15, %16, %17, %15")
KERNELF64(k_fma_f64, "v_fma_f64 %0, %16, %17, %0\n\tv_fma_f64 %1, %16, %17, %1\n\tv_fma_f64 %2, %16, %17, %2\n\tv_fma_f64 %3, %16, %17, %3\n\tv_fma_f64 %4, %16, %17, %4\n\tv_fma_f64 %5, %16, %17, %5\n\tv_fma_f64 %6, %16, %17, %6\n\tv_fma_f64 %7, %16, %17, %7\n\tv_fma_f64 %8, %16, %17, %8\n\tv_fma_f64 %9, %16, %17, %9\n\tv_fma_f64 %10, %16, %17, %10\n\tv_fma_f64 %11, %16, %17, %11\n\tv_fma_f64 %12, %16, %17, %12\n\tv_fma_f64 %13, %16, %17, %13\n\tv_fma_f64 %14, %16, %17, %14\n\tv_fma_f64 %15, %16, %17, %15")
KERNELF64(k_mul_f64, "v_mul_f64 %0, %16, %0\n\tv_mul_f64 %1, %16, %1\n\tv_mul_f64 %2, %16, %2\n\tv_mul_f64 %3, %16, %3\n\tv_mul_f64 %4, %16, %4\n\tv_mul_f64 %5, %16, %5\n\tv_mul_f64 %6, %16, %6\n\tv_mul_f64 %7, %16, %7\n\tv_mul_f64 %8, %16, %8\n\tv_mul_f64 %9, %16, %9\n\tv_mul_f64 %10, %16, %10\n\tv_mul_f64 %11, %16, %11\n\tv_mul_f64 %12, %16, %12\n\tv_mul_f64 %13, %16, %13\n\tv_mul_f64 %14, %16, %14\n\tv_mul_f64 %15, %16, %15")
KERNELF64(k_add_f64, "v_add_f64 %0, %16, %0\n\tv_add_f64 %1, %16, %1\n\tv_add_f64 %2, %16, %2\n\tv_add_f64 %3, %16, %3\n\tv_add_f64 %4, %16, %4\n\tv_add_f64 %5, %16, %5\n\tv_add_f64 %6, %16, %6\n\tv_add_f64 %7, %16, %7\n\tv_add_f64 %8, %16, %8\n\tv_add_f64 %9, %16, %9\n\tv_add_f64 %10, %16, %10\n\tv_add_f64 %11, %16, %11\n\tv_add_f64 %12, %16, %12\n\tv_add_f64 %13, %16, %13\n\tv_add_f64 %14, %16, %14\n\tv_add_f64 %15, %16, %15")

typedef void (*kern_t)(uint64_t*, int, uint32_t, uint32_t, long long*);

static long long* d_cyc;
static void run(const char* name, kern_t k, uint64_t* d_out, int cus, double mhz, int wpc) {
  const int blocks = cus * wpc / 4, threads = 256, iters = 4096;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 16, 3u, 5u, d_cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 3u, 5u, d_cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * threads * iters * 16;
  double per_s = ops / (ms * 1e-3);
  long long cyc; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
  // one wave issues iters*16 instructions in `cyc` shader clocks while wpc/4 waves share its SIMD
  double cyc_per_wave_instr = (double)cyc / (iters * 16.0) / (wpc / 4.0);
  printf("%-20s waves/CU %2d  %8.3f ms  %7.2f Tlane-op/s  %6.2f SIMD-cycles per wave64 instr  (eff clk %.0f MHz)\n", name, wpc, ms,
         per_s * 1e-12, cyc_per_wave_instr, cyc / (ms * 1e-3) * 1e-6);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount; double mhz = p.clockRate / 1000.0;
  printf("device %s CUs %d clock %.0f MHz\n", p.name, cus, mhz);
  uint64_t* d_out; CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
CHECK(hipMalloc(&d_cyc, 8));
#define R(k) run(#k, k, d_out, cus, mhz, 32); run(#k, k, d_out, cus, mhz, 8);
  R(k_add_u32) R(k_add_co_u32) R(k_addc_co_u32) R(k_add3_u32) R(k_alignbit) R(k_and_or) R(k_fma_f32)
  R(k_mul_lo_u32) R(k_mul_hi_u32) R(k_mad_u64_u32) R(k_lshl_add_u64) R(k_lshrrev_b64) R(k_and_b32_e32)
  R(k_mul_u32_u24) R(k_mul_hi_u32_u24) R(k_mad_u32_u24) R(k_mad_i32_i24) R(k_dot4_u32_u8)
  R(k_fma_f64) R(k_mul_f64) R(k_add_f64)
  return 0;
}
