// In-run calibration of the integer-VALU roof bench.py prices the MSM kernels against (libdg16_calib.so; NOT part of the
// C ABI of libdg16.so -- measurement infrastructure, loaded by bench.py only).
//
// The bound of the bucket accumulations is the chip-wide issue rate of v_mad_u64_u32.  Until round 5
// the bench line carried a constant measured once (34.4 T lane-op/s, profiles/r1_ubench_instr_rate.txt), so two boxes of
// the pool that differ by 5-8 % in ms_per_step showed different roofline fractions for the same binary and nothing on
// the line could say whether the box or the code was slower.  dg16_calib_mad_rate measures, in < 0.1 s and in the same
// process right before the timed loop:
//   * the issue rate itself: every lane runs `iters` rounds of 16 independent v_mad_u64_u32 (the kernel of
//     tools/ubench/instr_rate.hip), `waves_per_simd` waves per SIMD on every CU, timed with HIP events;
//   * the shader clock the chip holds UNDER that load: s_memtime ticks once per shader cycle, s_memrealtime at a
//     constant 100 MHz (MI355X_MICROARCH.md, cycle-constant table) -- their ratio over the kernel, taken by one lane of
//     every workgroup and averaged, is the clock in units of 100 MHz.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

#define MAD16                                                                                                         \
  "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_mad_u64_u32 %1, vcc, %16, %17, %1\n\tv_mad_u64_u32 %2, vcc, %16, %17, %2\n\t"  \
  "v_mad_u64_u32 %3, vcc, %16, %17, %3\n\tv_mad_u64_u32 %4, vcc, %16, %17, %4\n\tv_mad_u64_u32 %5, vcc, %16, %17, %5\n\t"  \
  "v_mad_u64_u32 %6, vcc, %16, %17, %6\n\tv_mad_u64_u32 %7, vcc, %16, %17, %7\n\tv_mad_u64_u32 %8, vcc, %16, %17, %8\n\t"  \
  "v_mad_u64_u32 %9, vcc, %16, %17, %9\n\tv_mad_u64_u32 %10, vcc, %16, %17, %10\n\tv_mad_u64_u32 %11, vcc, %16, %17, %11\n\t" \
  "v_mad_u64_u32 %12, vcc, %16, %17, %12\n\tv_mad_u64_u32 %13, vcc, %16, %17, %13\n\tv_mad_u64_u32 %14, vcc, %16, %17, %14\n\t" \
  "v_mad_u64_u32 %15, vcc, %16, %17, %15"

__global__ void __launch_bounds__(256) calib_mad_kernel(uint64_t* out, int iters, uint32_t a, uint32_t b,
                                                         unsigned long long* ticks /* [blocks][2]: shader, 100 MHz */) {
  uint64_t r[16];
  const uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
  for (int i = 0; i < 16; i++) r[i] = (uint64_t)threadIdx.x * (i + 1);
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    asm volatile(MAD16
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y)
                 : "vcc");
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    ticks[2 * blockIdx.x] = c1 - c0;
    ticks[2 * blockIdx.x + 1] = w1 - w0;
  }
  uint64_t s = 0;
  for (int i = 0; i < 16; i++) s ^= r[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

extern "C" {

// waves_per_simd: 1..8 (256-lane workgroups = one wave on each SIMD of a CU; that many workgroups per CU)
// out[0] = T v_mad_u64_u32 lane-op/s chip-wide (median of five runs), out[1] = shader clock under the load in MHz (0 if the
// counters do not separate), out[2] = kernel time in ms, out[3] = compute units, out[4] / out[5] = the fastest / slowest
// run's rate.  out: 6 doubles.  Returns 0, or a hipError_t.
int dg16_calib_mad_rate(int device, int waves_per_simd, int iters, double* out) {
  if (!out || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1) return -1;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t p;
  if ((e = hipGetDeviceProperties(&p, device)) != hipSuccess) return (int)e;
  const int cus = p.multiProcessorCount, blocks = cus * waves_per_simd, threads = 256;
  uint64_t* d_out = nullptr;
  unsigned long long* d_ticks = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  auto ok = [&](hipError_t x) { if (x != hipSuccess && !rc) rc = (int)x; return x == hipSuccess; };
  if (ok(hipMalloc((void**)&d_out, (size_t)blocks * threads * 8)) && ok(hipMalloc((void**)&d_ticks, (size_t)blocks * 16)) &&
      ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) && ok(hipEventCreate(&e0)) && ok(hipEventCreate(&e1))) {
    // DVFS makes single runs differ by up to 10 % (profiles/r6c_calibration_check.txt: 30.3 / 35.6 / 32.3 T for one kernel
    // length on one box): one warm-up, then the MEDIAN of five, each about as long as the accumulation kernels it prices
    constexpr int REPS = 5;
    double ms_v[REPS], mhz_v[REPS];
    int got = 0;
    for (int rep = 0; rep < REPS + 1 && !rc; rep++) {
      ok(hipEventRecord(e0, s));
      hipLaunchKernelGGL(calib_mad_kernel, dim3(blocks), dim3(threads), 0, s, d_out, iters, 12345u + rep, 6789u, d_ticks);
      ok(hipEventRecord(e1, s));
      ok(hipStreamSynchronize(s));
      float ms = 0;
      ok(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 0 || rc) continue;
      unsigned long long* h = new unsigned long long[(size_t)blocks * 2];
      ok(hipMemcpy(h, d_ticks, (size_t)blocks * 16, hipMemcpyDeviceToHost));
      double sc = 0, wc = 0;
      for (int i = 0; i < blocks; i++) { sc += (double)h[2 * i]; wc += (double)h[2 * i + 1]; }
      delete[] h;
      ms_v[got] = ms;
      mhz_v[got] = wc > 0 ? sc / wc * 100.0 : 0.0;
      got++;
    }
    double best_ms = 0, best_mhz = 0;
    if (!rc && got == REPS) {
      int idx[REPS];
      for (int i = 0; i < REPS; i++) idx[i] = i;
      for (int i = 0; i < REPS; i++)
        for (int j = i + 1; j < REPS; j++)
          if (ms_v[idx[j]] < ms_v[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
      best_ms = ms_v[idx[REPS / 2]];
      best_mhz = mhz_v[idx[REPS / 2]];
      out[4] = (double)blocks * threads * iters * 16.0 / (ms_v[idx[0]] * 1e-3) / 1e12;          // fastest
      out[5] = (double)blocks * threads * iters * 16.0 / (ms_v[idx[REPS - 1]] * 1e-3) / 1e12;   // slowest
    } else if (!rc) {
      rc = -2;
    }
    if (!rc) {
      out[0] = (double)blocks * threads * iters * 16.0 / (best_ms * 1e-3) / 1e12;
      out[1] = best_mhz;
      out[2] = best_ms;
      out[3] = cus;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (s) (void)hipStreamDestroy(s);
  if (d_ticks) (void)hipFree(d_ticks);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

}  // extern "C"
