// Limb-per-lane field / group arithmetic (csrc/lane29.h) against the one-product-per-lane chains it replaces
// (msm_impl.h: dbl_wave29 / add_wave29): semantics of the row primitives on this device, parity of products and group
// operations on random operands and on the special cases, and the time of a dependent chain in both forms.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I distributed-groth16_amd/csrc -I include tools/ubench/lane29_probe.hip -o tools/ubench/lane29_probe
#include "msm_impl.h"
#include "lane29.h"
#include <cstdio>
#include <vector>

using namespace dg16;
using F = bn254_fq;
using F2 = Fp2<bn254_fq>;
using P = typename FieldOf<F>::Params;
constexpr int BS = XYZZ29<F>::BS;
using S = Fe<P, BS, 1>;
using FO = lane29::Ops<F>;

__device__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// a pseudo-random element < 4 p (normalised limbs), the same in every lane
__device__ S rnd_fe(uint32_t seed) {
  S r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = mix(seed * 16u + i) & lane29::MASK;
  r.l[8] &= (1u << 23) - 1;       // p >> 232 ~ 2^21.6: below 4 p
  return r;
}
__device__ bool same(const S& a, const S& b) {
  const auto ca = canon(a), cb = canon(b);
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 9; i++) ok = ok && ca.l[i] == cb.l[i];
  return ok;
}

__global__ void __launch_bounds__(64) k_prims(uint32_t* out) {
  const uint32_t lane = threadIdx.x, v = 100 + lane, w = 1000 + lane;
  uint32_t r0, r1, r2, r3;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r0) : "v"(v));
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r1) : "v"(v));
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:7 row_mask:0xf bank_mask:0xf" : "=&v"(r2) : "v"(v));
  asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r3) : "v"(v), "v"(w));
  const auto s16 = __builtin_amdgcn_permlane16_swap(v, w, false, false);
  const auto s32 = __builtin_amdgcn_permlane32_swap(v, w, false, false);
  out[lane] = r0; out[64 + lane] = r1; out[128 + lane] = r2; out[192 + lane] = r3;
  out[256 + lane] = s16[0]; out[320 + lane] = s16[1]; out[384 + lane] = s32[0]; out[448 + lane] = s32[1];
}

__global__ void __launch_bounds__(64) k_check_mul(unsigned* bad) {
  FO::KT k;
  k.init();
  const S a = rnd_fe(2 * blockIdx.x + 1), b = rnd_fe(2 * blockIdx.x + 2);
  const S ref = fit<BS>(a * b);
  const uint32_t la = FO::from_regs<BS>(k, a), lb = FO::from_regs<BS>(k, b);
  const S got = FO::to_regs<BS>(lane29::full_norm(lane29::mul(k, la, lb)));
  // a subtraction and a doubled operand as well; then the conditional subtraction of p
  const S ref2 = fit<BS>((a - b) * dbl(b));
  const uint32_t l2 = lane29::mul(k, FO::sub<1>(k, la, lb), lb << 1);
  const S got2 = FO::to_regs<BS>(lane29::cond_sub_p(k, lane29::full_norm(l2)));
  if (!same(ref, got) && threadIdx.x == 0) atomicAdd(bad, 1u);
  if (!same(ref2, got2) && threadIdx.x == 0) atomicAdd(bad + 1, 1u);
}

template <class G> struct Rnd;
template <> struct Rnd<F> {
  static __device__ S get(uint32_t seed) { return rnd_fe(seed); }
  static __device__ bool eq(const S& a, const S& b) { return same(a, b); }
};
// the fourteen-limb base fields: random limbs, the top one below 4 p's
template <class QP, uint32_t TOPMASK>
struct Rnd14 {
  using S14 = typename FieldOf<Fp<QP>>::Store;
  static __device__ S14 get(uint32_t seed) {
    S14 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = mix(seed * 16u + i) & ((1u << 28) - 1);
    r.l[13] &= TOPMASK;
    return r;
  }
  static __device__ bool eq(const S14& a, const S14& b) {
    const auto ca = canon(a), cb = canon(b);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 14; i++) ok = ok && ca.l[i] == cb.l[i];
    return ok;
  }
};
using F381 = bls12_381_fq;
using F377 = bls12_377_fq;
template <> struct Rnd<F381> : Rnd14<bls12_381_fq_params, (1u << 18) - 1> {};
template <> struct Rnd<F377> : Rnd14<bls12_377_fq_params, (1u << 14) - 1> {};
template <class QP, uint32_t TOPMASK>
struct Rnd14x2 {
  using B = Rnd14<QP, TOPMASK>;
  using S2 = typename FieldOf<Fp2<Fp<QP>>>::Store;
  static __device__ S2 get(uint32_t seed) { return {B::get(2 * seed + 2000003u), B::get(2 * seed + 2000004u)}; }
  static __device__ bool eq(const S2& a, const S2& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
};
using F381x2 = Fp2<bls12_381_fq>;
using F377x2 = Fp2<bls12_377_fq>;
template <> struct Rnd<F381x2> : Rnd14x2<bls12_381_fq_params, (1u << 18) - 1> {};
template <> struct Rnd<F377x2> : Rnd14x2<bls12_377_fq_params, (1u << 14) - 1> {};
template <> struct Rnd<F2> {
  using S2 = typename FieldOf<F2>::Store;
  static __device__ S2 get(uint32_t seed) {
    const S a = rnd_fe(2 * seed + 1000003u), b = rnd_fe(2 * seed + 1000004u);   // (< 4 p: squeeze under the Fq2 storage bound)
    return {fit<XYZZ29<F2>::BS>(a * fe_one<P>()), fit<XYZZ29<F2>::BS>(b * fe_one<P>())};
  }
  static __device__ bool eq(const S2& a, const S2& b) {
    const auto ca0 = canon(a.c0), cb0 = canon(b.c0), ca1 = canon(a.c1), cb1 = canon(b.c1);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 9; i++) ok = ok && ca0.l[i] == cb0.l[i] && ca1.l[i] == cb1.l[i];
    return ok;
  }
};
template <class G>
__device__ XYZZ29<G> rnd_pt(uint32_t seed) {
  return {Rnd<G>::get(4 * seed), Rnd<G>::get(4 * seed + 1), Rnd<G>::get(4 * seed + 2), Rnd<G>::get(4 * seed + 3)};
}
template <class G>
__device__ bool same_pt(const XYZZ29<G>& a, const XYZZ29<G>& b) {
  if (a.is_inf() || b.is_inf()) return a.is_inf() == b.is_inf();
  return Rnd<G>::eq(a.x, b.x) && Rnd<G>::eq(a.y, b.y) && Rnd<G>::eq(a.zz, b.zz) && Rnd<G>::eq(a.zzz, b.zzz);
}
template <class G>
__global__ void __launch_bounds__(64) k_check_pt(unsigned* bad) {
  using GO = lane29::Ops<G>;
  using LPt = lane29::Pt<GO>;
  typename GO::KT k;
  k.init();
  const XYZZ29<G> p = rnd_pt<G>(2 * blockIdx.x + 1), o = rnd_pt<G>(2 * blockIdx.x + 2);
  const LPt lp = lane29::to_pt<G>(k, p), lo = lane29::to_pt<G>(k, o);
  const bool d_ok = same_pt(dbl_wave29(p), lane29::from_pt<G>(k, lane29::dbl_pt<GO>(k, lp)));
  const bool a_ok = same_pt(add_wave29(p, o), lane29::from_pt<G>(k, lane29::add_pt<GO>(k, lp, lo)));
  // special cases: p + p (the doubling branch), p + (-p) (the identity), identity operands
  const bool pp_ok = same_pt(add_wave29(p, p), lane29::from_pt<G>(k, lane29::add_pt<GO>(k, lp, lp)));
  const XYZZ29<G> n = p.neg_pt();
  const bool pn_ok = lane29::add_pt<GO>(k, lp, lane29::to_pt<G>(k, n)).inf && lane29::add_pt<GO>(k, lp, lane29::neg_pt<GO>(k, lp)).inf;
  const LPt inf = lane29::to_pt<G>(k, XYZZ29<G>::inf());
  const bool id_ok = same_pt(p, lane29::from_pt<G>(k, lane29::add_pt<GO>(k, lp, inf))) &&
                     same_pt(o, lane29::from_pt<G>(k, lane29::add_pt<GO>(k, inf, lo))) && lane29::dbl_pt<GO>(k, inf).inf;
  // a chain: 16 doublings and an addition (of o, then of -o), three times (bounds of the steady state)
  XYZZ29<G> r = p;
  LPt lr = lp;
  for (int it = 0; it < 3; it++) {
    for (int j = 0; j < 16; j++) { r = dbl_wave29(r); lr = lane29::dbl_pt<GO>(k, lr); }
    r = add_wave29(r, (it & 1) ? o.neg_pt() : o);
    lr = lane29::add_pt<GO>(k, lr, (it & 1) ? lane29::neg_pt<GO>(k, lo) : lo);
  }
  const bool c_ok = same_pt(r, lane29::from_pt<G>(k, lr));
  // the affine form (one inversion in lane form) of a point whose ZZ, ZZZ are a square and a cube: X / ZZ, Y / ZZZ
  XYZZ29<G> q = p;
  q.zz = fit<XYZZ29<G>::BS>(o.x * o.x);
  q.zzz = fit<XYZZ29<G>::BS>(q.zz * o.x);
  const Affine<G> want = q.to_xyzz32().to_affine(), got = lane29::to_affine<G>(k, lane29::to_pt<G>(k, q));
  const bool f_ok = want.x == got.x && want.y == got.y && lane29::to_affine<G>(k, inf).is_inf();
  if (threadIdx.x == 0) {
    if (!f_ok) atomicAdd(bad + 6, 1u);
    if (!d_ok) atomicAdd(bad, 1u);
    if (!a_ok) atomicAdd(bad + 1, 1u);
    if (!pp_ok) atomicAdd(bad + 2, 1u);
    if (!pn_ok) atomicAdd(bad + 3, 1u);
    if (!id_ok) atomicAdd(bad + 4, 1u);
    if (!c_ok) atomicAdd(bad + 5, 1u);
  }
}

__device__ uint32_t digest(const XYZZ29<F>& r) { return r.x.l[0] ^ r.y.l[3] ^ r.zz.l[1]; }
__device__ uint32_t digest(const XYZZ29<F2>& r) { return r.x.c0.l[0] ^ r.y.c1.l[3] ^ r.zz.c0.l[1]; }
__device__ uint32_t digest(const XYZZ29<F381x2>& r) { return r.x.c0.l[0] ^ r.y.c1.l[3] ^ r.zz.c0.l[1]; }
__device__ uint32_t digest(const XYZZ29<F377x2>& r) { return r.x.c0.l[0] ^ r.y.c1.l[3] ^ r.zz.c0.l[1]; }
__device__ uint32_t digest(const XYZZ29<F381>& r) { return r.x.l[0] ^ r.y.l[3] ^ r.zz.l[1]; }
__device__ uint32_t digest(const XYZZ29<F377>& r) { return r.x.l[0] ^ r.y.l[3] ^ r.zz.l[1]; }
template <class G, bool NEW>
__global__ void __launch_bounds__(64) k_chain(uint32_t* out, int iters, int dbls) {
  const XYZZ29<G> p = rnd_pt<G>(2 * blockIdx.x + 1), o = rnd_pt<G>(2 * blockIdx.x + 2);
  if constexpr (NEW) {
    using GO = lane29::Ops<G>;
    typename GO::KT k;
    k.init();
    lane29::Pt<GO> r = lane29::to_pt<G>(k, p);
    const lane29::Pt<GO> lo = lane29::to_pt<G>(k, o);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
      for (int j = 0; j < dbls; j++) r = lane29::dbl_pt<GO>(k, r);
      r = lane29::add_pt<GO>(k, r, lo);
    }
    const XYZZ29<G> res = lane29::from_pt<G>(k, r);
    if (threadIdx.x == 0) out[blockIdx.x] = digest(res);
  } else {
    XYZZ29<G> r = p;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
      for (int j = 0; j < dbls; j++) r = dbl_wave29(r);
      r = add_wave29(r, o);
    }
    if (threadIdx.x == 0) out[blockIdx.x] = digest(r);
  }
}
template <bool NEW>
__global__ void __launch_bounds__(64) k_mulchain(uint32_t* out, int iters) {
  S a = rnd_fe(2 * blockIdx.x + 1);
  const S b = rnd_fe(2 * blockIdx.x + 2);
  if constexpr (NEW) {
    FO::KT k;
    k.init();
    uint32_t x = FO::from_regs<BS>(k, a);
    const uint32_t y = FO::from_regs<BS>(k, b);
#pragma unroll 1
    for (int it = 0; it < iters; it++) x = lane29::mul(k, x, y);
    out[blockIdx.x * 64 + threadIdx.x] = x;
  } else {
#pragma unroll 1
    for (int it = 0; it < iters; it++) a = fit<BS>(a * b);
    if (threadIdx.x == 0) out[blockIdx.x] = a.l[0] ^ a.l[5];
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class Fn>
static float timed(Fn launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  uint32_t* d;
  CK(hipMalloc(&d, 1 << 20));
  CK(hipMemset(d, 0, 1 << 20));
  hipLaunchKernelGGL(k_prims, dim3(1), dim3(64), 0, 0, d);
  std::vector<uint32_t> h(512);
  CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const int r = l & 15, base = l & ~15;
    bad += h[l] != (r >= 3 ? 100u + l - 3 : 0u);
    bad += h[64 + l] != (r + 8 < 16 ? 100u + l + 8 : 0u);
    bad += h[128 + l] != 100u + base + ((r - 7 + 16) & 15);
    bad += h[192 + l] != (r >= 1 ? 100u + l - 1 : 0u) + 1000u + l;
    const int row = l >> 4;
    // permlane16_swap(v, w): odd rows of the first <-> even rows of the second
    bad += h[256 + l] != ((row & 1) ? 1000u + l - 16 : 100u + l);
    bad += h[320 + l] != ((row & 1) ? 1000u + l : 100u + l + 16);
    // permlane32_swap(v, w): upper half of the first <-> lower half of the second
    bad += h[384 + l] != (l >= 32 ? 1000u + l - 32 : 100u + l);
    bad += h[448 + l] != (l >= 32 ? 1000u + l : 100u + l + 32);
  }
  printf("row primitives (row_shr / row_shl / row_ror / add_dpp / permlane16_swap / permlane32_swap): %d mismatches\n", bad);
  if (bad) {
    for (int s = 0; s < 8; s++) { printf("set %d:", s); for (int l = 0; l < 64; l++) printf(" %u", h[64 * s + l]); printf("\n"); }
  }
  CK(hipMemset(d, 0, 64));
  hipLaunchKernelGGL(k_check_mul, dim3(4096), dim3(64), 0, 0, d);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), d, 64, hipMemcpyDeviceToHost));
  printf("products: %u of 4096 wrong; (a - b)(2 b): %u wrong\n", h[0], h[1]);
  const char* names[6] = {"BN254 G1", "BN254 G2", "BLS12-381 G1", "BLS12-377 G1", "BLS12-381 G2", "BLS12-377 G2"};
  for (int g = 0; g < 6; g++) {
    CK(hipMemset(d, 0, 64));
    if (g == 0) hipLaunchKernelGGL(k_check_pt<F>, dim3(1024), dim3(64), 0, 0, d);
    else if (g == 1) hipLaunchKernelGGL(k_check_pt<F2>, dim3(1024), dim3(64), 0, 0, d);
    else if (g == 2) hipLaunchKernelGGL(k_check_pt<F381>, dim3(1024), dim3(64), 0, 0, d);
    else if (g == 3) hipLaunchKernelGGL(k_check_pt<F377>, dim3(1024), dim3(64), 0, 0, d);
    else if (g == 4) hipLaunchKernelGGL(k_check_pt<F381x2>, dim3(256), dim3(64), 0, 0, d);
    else hipLaunchKernelGGL(k_check_pt<F377x2>, dim3(256), dim3(64), 0, 0, d);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 64, hipMemcpyDeviceToHost));
    printf("%s, of 1024 (256 for the 14-limb G2): doubling %u wrong, addition %u, p + p %u, p - p %u, identity operands %u, chain of 48 doublings + 3 additions %u, "
           "affine form %u\n", names[g], h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
  }
  for (int blocks : {1, 256, 1024}) {
    const int iters = 200;
    const float m_old = timed([&] { hipLaunchKernelGGL(k_mulchain<false>, dim3(blocks), dim3(64), 0, 0, d, 4 * iters); });
    const float m_new = timed([&] { hipLaunchKernelGGL(k_mulchain<true>, dim3(blocks), dim3(64), 0, 0, d, 4 * iters); });
    const float t_old = timed([&] { hipLaunchKernelGGL((k_chain<F, false>), dim3(blocks), dim3(64), 0, 0, d, iters, 16); });
    const float t_new = timed([&] { hipLaunchKernelGGL((k_chain<F, true>), dim3(blocks), dim3(64), 0, 0, d, iters, 16); });
    const float u_old = timed([&] { hipLaunchKernelGGL((k_chain<F2, false>), dim3(blocks), dim3(64), 0, 0, d, iters / 4, 16); });
    const float u_new = timed([&] { hipLaunchKernelGGL((k_chain<F2, true>), dim3(blocks), dim3(64), 0, 0, d, iters / 4, 16); });
    const float v_old = timed([&] { hipLaunchKernelGGL((k_chain<F377, false>), dim3(blocks), dim3(64), 0, 0, d, iters / 4, 16); });
    const float v_new = timed([&] { hipLaunchKernelGGL((k_chain<F377, true>), dim3(blocks), dim3(64), 0, 0, d, iters / 4, 16); });
    const float w_old = timed([&] { hipLaunchKernelGGL((k_chain<F377x2, false>), dim3(blocks), dim3(64), 0, 0, d, iters / 10, 16); });
    const float w_new = timed([&] { hipLaunchKernelGGL((k_chain<F377x2, true>), dim3(blocks), dim3(64), 0, 0, d, iters / 10, 16); });
    printf("%4d waves: BLS12-377 G2 16 doublings + 1 addition %.2f -> %.2f us (%.2fx)\n", blocks, 1e3 * w_old / (iters / 10),
           1e3 * w_new / (iters / 10), w_old / w_new);
    printf("%4d waves: dependent product %.3f -> %.3f us; 16 doublings + 1 addition: G1 %.2f -> %.2f us (%.2fx), G2 %.2f -> %.2f us (%.2fx), "
           "BLS12-377 G1 %.2f -> %.2f us (%.2fx)\n",
           blocks, 1e3 * m_old / (4 * iters), 1e3 * m_new / (4 * iters), 1e3 * t_old / iters, 1e3 * t_new / iters, t_old / t_new,
           1e3 * u_old / (iters / 4), 1e3 * u_new / (iters / 4), u_old / u_new, 1e3 * v_old / (iters / 4), 1e3 * v_new / (iters / 4),
           v_old / v_new);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
