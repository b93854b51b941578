// Groth16 prover glue on the GPU: the single-prover value every distributed run must equal.
//
// Restates `Groth16::create_proof_with_reduction_and_matrices` (third-party fork of ark-groth16,
// branch distributed-groth16, NOT vendored; call sites groth16/examples/sha256.rs:159,
// mpc-api/src/main.rs:393) on top of the kernels of this library, using the base-vector mapping of
// groth16/src/proving_key.rs:48-65 and the assembly of groth16/src/prove.rs:21-136:
//
//   h    = witness_map(a, b, c)                               ark-circom/src/circom/qap.rs:64-91
//   A    = alpha_g1 + a_query[0] + msm(a_query[1..], w[1..]) + r*delta_g1              prove.rs:21-46
//   B1   = beta_g1  + b_g1_query[0] + msm(b_g1_query[1..], w[1..]) + s*delta_g1  (only if r != 0)
//   B    = beta_g2  + b_g2_query[0] + msm(b_g2_query[1..], w[1..]) + s*delta_g2        prove.rs:62-85
//   C    = msm(l_query, w[ni..]) + msm(h_query, h) + s*A + r*B1 - r*s*delta_g1         prove.rs:106-136
//
// MI355X mapping: the r*delta / s*delta / -rs*delta terms ride as one extra (base, scalar) pair on
// the MSMs (the key is stored with delta appended), the five MSMs and the h-polynomial run on the
// three channels (HIP streams) like the reference's three multiplexed channels (prove.rs:119-125,
// ext_wit.rs:34-61), and the two scalar multiples s*A, r*B1 (serial double-and-add, one lane each)
// overlap with the L / H MSMs.  Everything stays in HBM; the only host traffic is r, s and the
// 3-point proof.
#pragma once
#include "ctx.h"
#include "types.h"

#include "pk.h"

namespace dg16 {

// scalars[n_w] = extra (the r / s / -rs slot that pairs with the appended delta base)
template <class Fr>
__global__ void prover_scalar_prep_kernel(const Fr* r_s, Fr* sc_a, Fr* sc_b1, Fr* sc_b2, Fr* sc_l, size_t n_ab,
                                          size_t n_l, int mont, int carries_delta) {
  // r_s[0] = r, r_s[1] = s in the same form as the witness (Montgomery iff mont).  Only the last
  // shard carries the delta pairs; the others multiply their delta slot by zero.
  Fr r = r_s[0], s = r_s[1];
  if (!carries_delta) { r = Fr::zero(); s = Fr::zero(); }
  Fr rm = mont ? r : r.to_mont(), sm = mont ? s : s.to_mont();
  Fr nrs = (rm * sm).neg();                 // Montgomery form of -(r*s)
  sc_a[n_ab] = r;
  sc_b1[n_ab] = s;
  sc_b2[n_ab] = s;
  sc_l[n_l] = mont ? nrs : nrs.from_mont();
}

// k * p, k canonical NL-limb integer, one lane
template <class F, class Fr>
__device__ XYZZ<F> mul_by_fr(const XYZZ<F>& p, const Fr& k_canon) {
  return scalar_mul<F, Fr::NL>(p, k_canon.l);
}

// stage 1 (after the A, B1, B2 MSMs): A, B and the two scalar multiples needed by C
template <class Fq, class Fq2, class Fr>
__global__ void __launch_bounds__(192) prover_stage1_kernel(const Jacobian<Fq>* msm_a, const Jacobian<Fq>* msm_b1,
                                                             const Jacobian<Fq2>* msm_b2, const Affine<Fq>* fixed_g1,
                                                             const Affine<Fq2>* fixed_g2, const Fr* r_s, int mont,
                                                             Jacobian<Fq>* out_a, Jacobian<Fq2>* out_b,
                                                             XYZZ<Fq>* s_a, XYZZ<Fq>* r_b1) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane != 0) return;
  Fr r = r_s[0], s = r_s[1];
  if (mont) { r = r.from_mont(); s = s.from_mont(); }
  if (wave == 0) {
    XYZZ<Fq> a = XYZZ<Fq>::from_jacobian(*msm_a).madd(fixed_g1[0], false).madd(fixed_g1[1], false);
    *out_a = a.to_jacobian();
    *s_a = mul_by_fr<Fq, Fr>(a, s);
  } else if (wave == 1) {
    XYZZ<Fq> b1 = XYZZ<Fq>::inf();
    if (!r.is_zero()) b1 = XYZZ<Fq>::from_jacobian(*msm_b1).madd(fixed_g1[2], false).madd(fixed_g1[3], false);
    *r_b1 = mul_by_fr<Fq, Fr>(b1, r);
  } else {
    XYZZ<Fq2> b = XYZZ<Fq2>::from_jacobian(*msm_b2).madd(fixed_g2[0], false).madd(fixed_g2[1], false);
    *out_b = b.to_jacobian();
  }
}

// stage 2 (after the L and H MSMs): C = L + H + s*A + r*B1   (-rs*delta is inside L)
template <class Fq>
__global__ void prover_stage2_kernel(const Jacobian<Fq>* msm_l, const Jacobian<Fq>* msm_h, const XYZZ<Fq>* s_a,
                                     const XYZZ<Fq>* r_b1, Jacobian<Fq>* out_c) {
  XYZZ<Fq> c = XYZZ<Fq>::from_jacobian(*msm_l).add(XYZZ<Fq>::from_jacobian(*msm_h)).add(*s_a).add(*r_b1);
  *out_c = c.to_jacobian();
}

// partial results of one shard: A, B1, L, H (G1 Jacobian) then B (G2 Jacobian)
template <int CURVE>
static size_t msm_results_bytes() {
  using CT = CurveTypes<CURVE>;
  return 4 * sizeof(Jacobian<typename CT::Fq>) + sizeof(Jacobian<typename CT::Fq2>);
}

// h-polynomial + the five MSMs of this key's shard.  res_out (device or host per dev_ptrs) receives
// msm_results_bytes() bytes.  Uses all three channels; returns with the results stream-ordered on
// channel 0.
template <int CURVE>
static void msms_typed(dg16_ctx* ctx, Call& k0, Call& k1, Call& k2, const PkDev& pk, const void* a, const void* b,
                       const void* c, const void* witness, const void* r_s_host, bool mont, bool dev_ptrs,
                       uint8_t* res_dev) {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  using Fr = typename CT::Fr;
  const size_t nv = pk.num_vars, ni = pk.num_inputs, m = pk.m;
  const size_t n_ab = pk.ab_hi - pk.ab_lo;   // this shard's slice of w[1..]
  const size_t n_l = pk.l_hi - pk.l_lo;      // ... of w[ni..]
  const size_t n_h = pk.h_hi - pk.h_lo;      // ... of h
  unsigned log_m = 0;
  while (((size_t)1 << log_m) < m) log_m++;
  const size_t g1j = sizeof(Jacobian<Fq>);

  const Fr* w_dev = (const Fr*)stage_in(k0, 18, witness, nv * sizeof(Fr), dev_ptrs);
  const void* a_dev = stage_in(k0, 19, a, m * sizeof(Fr), dev_ptrs);
  const void* b_dev = stage_in(k0, 20, b, m * sizeof(Fr), dev_ptrs);
  const void* c_dev = stage_in(k0, 21, c, m * sizeof(Fr), dev_ptrs);
  Fr* r_s = (Fr*)ws(k0.c, 22, 4096);
  Jacobian<Fq>* res_a = (Jacobian<Fq>*)res_dev;
  Jacobian<Fq>* res_b1 = res_a + 1;
  Jacobian<Fq>* res_l = res_a + 2;
  Jacobian<Fq>* res_h = res_a + 3;
  Jacobian<Fq2>* res_b2 = (Jacobian<Fq2>*)(res_dev + 4 * g1j);
  DG_HIP(hipMemcpyAsync(r_s, r_s_host, 2 * sizeof(Fr), hipMemcpyHostToDevice, k0.s()));
  // scalar vectors with the extra slot (one per MSM that carries a delta pair)
  Fr* sc_a = (Fr*)ws(k0.c, 23, ((n_ab + 1) * 3 + (n_l + 1)) * sizeof(Fr));
  Fr* sc_b1 = sc_a + (n_ab + 1);
  Fr* sc_b2 = sc_b1 + (n_ab + 1);
  Fr* sc_l = sc_b2 + (n_ab + 1);
  DG_HIP(hipMemcpyAsync(sc_a, w_dev + 1 + pk.ab_lo, n_ab * sizeof(Fr), hipMemcpyDeviceToDevice, k0.s()));
  DG_HIP(hipMemcpyAsync(sc_b1, w_dev + 1 + pk.ab_lo, n_ab * sizeof(Fr), hipMemcpyDeviceToDevice, k0.s()));
  DG_HIP(hipMemcpyAsync(sc_b2, w_dev + 1 + pk.ab_lo, n_ab * sizeof(Fr), hipMemcpyDeviceToDevice, k0.s()));
  DG_HIP(hipMemcpyAsync(sc_l, w_dev + ni + pk.l_lo, n_l * sizeof(Fr), hipMemcpyDeviceToDevice, k0.s()));
  hipLaunchKernelGGL(prover_scalar_prep_kernel<Fr>, dim3(1), dim3(1), 0, k0.s(), r_s, sc_a, sc_b1, sc_b2, sc_l,
                     n_ab, n_l, (int)mont, (int)(pk.shard + 1 == pk.nshards));
  DG_HIP(hipGetLastError());
  hipEvent_t ready, e1, e2;
  DG_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  DG_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  DG_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  DG_HIP(hipEventRecord(ready, k0.s()));
  DG_HIP(hipStreamWaitEvent(k1.s(), ready, 0));
  DG_HIP(hipStreamWaitEvent(k2.s(), ready, 0));

  // channel 2: B (G2) alone -- the longest chain (its serial Horner tail is ~10 ms);
  // channel 1: A, B1, L | channel 0: h-poly, H
  Fr* h_dev = (Fr*)ws(k0.c, 3, m * sizeof(Fr));
  msm_launch(k2, CURVE, 2, pk.b2_q, sc_b2, n_ab + 1, mont, false, res_b2);
  h_poly_launch(k0, CURVE, a_dev, b_dev, c_dev, log_m, h_dev);
  msm_launch(k1, CURVE, 1, pk.a_q, sc_a, n_ab + 1, mont, false, res_a);
  msm_launch(k0, CURVE, 1, pk.h_q, h_dev + pk.h_lo, n_h, true, false, res_h);
  msm_launch(k1, CURVE, 1, pk.b1_q, sc_b1, n_ab + 1, mont, false, res_b1);
  msm_launch(k1, CURVE, 1, pk.l_q, sc_l, n_l + 1, mont, false, res_l);
  DG_HIP(hipEventRecord(e1, k1.s()));
  DG_HIP(hipEventRecord(e2, k2.s()));
  DG_HIP(hipStreamWaitEvent(k0.s(), e1, 0));
  DG_HIP(hipStreamWaitEvent(k0.s(), e2, 0));
  DG_HIP(hipEventDestroy(ready));
  DG_HIP(hipEventDestroy(e1));
  DG_HIP(hipEventDestroy(e2));
}

// proof = assemble(sum of the shards' MSM results).  res_dev: msm_results_bytes() on the device.
template <int CURVE>
static void assemble_typed(Call& k0, const PkDev& pk, const uint8_t* res_dev, const void* r_s_host, bool mont,
                           uint8_t* proof_dev) {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  using Fr = typename CT::Fr;
  const size_t g1j = sizeof(Jacobian<Fq>), g2j = sizeof(Jacobian<Fq2>);
  uint8_t* small = (uint8_t*)ws(k0.c, 22, 4096);
  Fr* r_s = (Fr*)small;
  XYZZ<Fq>* s_a = (XYZZ<Fq>*)(small + 64);
  XYZZ<Fq>* r_b1 = s_a + 1;
  DG_HIP(hipMemcpyAsync(r_s, r_s_host, 2 * sizeof(Fr), hipMemcpyHostToDevice, k0.s()));
  const Jacobian<Fq>* res = (const Jacobian<Fq>*)res_dev;
  const Affine<Fq>* fixed_g1 = (const Affine<Fq>*)pk.fixed;
  const Affine<Fq2>* fixed_g2 = (const Affine<Fq2>*)((const uint8_t*)pk.fixed + 4 * sizeof(Affine<Fq>));
  hipLaunchKernelGGL((prover_stage1_kernel<Fq, Fq2, Fr>), dim3(1), dim3(192), 0, k0.s(), res, res + 1,
                     (const Jacobian<Fq2>*)(res_dev + 4 * g1j), fixed_g1, fixed_g2, r_s, (int)mont,
                     (Jacobian<Fq>*)proof_dev, (Jacobian<Fq2>*)(proof_dev + g1j), s_a, r_b1);
  hipLaunchKernelGGL(prover_stage2_kernel<Fq>, dim3(1), dim3(1), 0, k0.s(), res + 2, res + 3, s_a, r_b1,
                     (Jacobian<Fq>*)(proof_dev + g1j + g2j));
  DG_HIP(hipGetLastError());
}

template <int CURVE>
static void prove_typed(dg16_ctx* ctx, const PkDev& pk, const void* a, const void* b, const void* c,
                        const void* witness, const void* r_s_host, bool mont, bool dev_ptrs, void* proof_out) {
  using CT = CurveTypes<CURVE>;
  const size_t g1j = sizeof(Jacobian<typename CT::Fq>), g2j = sizeof(Jacobian<typename CT::Fq2>);
  DG_REQUIRE(pk.nshards == 1, DG16_ERR_BAD_ARG, "dg16_groth16_prove needs an unsharded key; use _msms + _assemble");
  Call k0(ctx, 0), k1(ctx, 1), k2(ctx, 2);
  uint8_t* buf = (uint8_t*)ws(k0.c, 16, 8192);
  uint8_t* res_dev = buf;
  uint8_t* proof_dev = buf + 4096;
  k0.begin_dominant();
  msms_typed<CURVE>(ctx, k0, k1, k2, pk, a, b, c, witness, r_s_host, mont, dev_ptrs, res_dev);
  assemble_typed<CURVE>(k0, pk, res_dev, r_s_host, mont, proof_dev);
  k0.end_dominant();
  stage_out(k0, proof_out, proof_dev, 2 * g1j + g2j, dev_ptrs);
  k0.finish();
  k1.finish();
  k2.finish();
  if (!dev_ptrs) DG_HIP(hipStreamSynchronize(k0.s()));
}

// out = sum of n Jacobian points (one lane; n is the number of GPUs)
template <class F>
__global__ void point_sum_kernel(const Jacobian<F>* in, size_t n, size_t stride_bytes, Jacobian<F>* out) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (size_t i = 0; i < n; i++)
    acc = acc.add(XYZZ<F>::from_jacobian(*(const Jacobian<F>*)((const uint8_t*)in + i * stride_bytes)));
  *out = acc.to_jacobian();
}

// gathered: n_shards records of msm_results_bytes(); out: one record with the per-MSM sums
template <int CURVE>
static void reduce_results_typed(Call& k, const uint8_t* gathered, size_t n_shards, uint8_t* out) {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  const size_t rec = msm_results_bytes<CURVE>(), g1j = sizeof(Jacobian<Fq>);
  // the 1-lane adds of the five results are independent: one block each
  for (int i = 0; i < 4; i++)
    hipLaunchKernelGGL(point_sum_kernel<Fq>, dim3(1), dim3(1), 0, k.s(), (const Jacobian<Fq>*)(gathered + i * g1j),
                       n_shards, rec, (Jacobian<Fq>*)(out + i * g1j));
  hipLaunchKernelGGL(point_sum_kernel<Fq2>, dim3(1), dim3(1), 0, k.s(), (const Jacobian<Fq2>*)(gathered + 4 * g1j),
                     n_shards, rec, (Jacobian<Fq2>*)(out + 4 * g1j));
  DG_HIP(hipGetLastError());
}

// a_query etc. are given as full arkworks vectors (element 0 included); delta is appended here
template <int CURVE>
static void pk_build(dg16_ctx* ctx, PkDev& d, const void* a_query, const void* b_g1_query,
                     const void* b_g2_query, const void* h_query, const void* l_query, const void* fixed_host,
                     bool dev_ptrs) {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  const size_t p1 = sizeof(Affine<Fq>), p2 = sizeof(Affine<Fq2>);
  const size_t nv = d.num_vars, ni = d.num_inputs, m = d.m;
  hipMemcpyKind kind = dev_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  auto slice = [&](size_t n, size_t& lo, size_t& hi) {
    lo = n * d.shard / d.nshards;
    hi = n * (d.shard + 1) / d.nshards;
  };
  slice(nv - 1, d.ab_lo, d.ab_hi);
  slice(nv - ni, d.l_lo, d.l_hi);
  slice(m, d.h_lo, d.h_hi);
  const size_t n_ab = d.ab_hi - d.ab_lo, n_l = d.l_hi - d.l_lo, n_h = d.h_hi - d.h_lo;
  DG_HIP(hipSetDevice(ctx->device));
  DG_HIP(hipMalloc(&d.a_q, (n_ab + 1) * p1));
  DG_HIP(hipMalloc(&d.b1_q, (n_ab + 1) * p1));
  DG_HIP(hipMalloc(&d.b2_q, (n_ab + 1) * p2));
  DG_HIP(hipMalloc(&d.l_q, (n_l + 1) * p1));
  DG_HIP(hipMalloc(&d.h_q, (n_h ? n_h : 1) * p1));
  DG_HIP(hipMalloc(&d.fixed, 4 * p1 + 2 * p2));
  // fixed_host layout: alpha_g1, beta_g1, delta_g1 (G1 affine) | beta_g2, delta_g2 (G2 affine)
  const uint8_t* fx = (const uint8_t*)fixed_host;
  const uint8_t* aq = (const uint8_t*)a_query;
  const uint8_t* b1 = (const uint8_t*)b_g1_query;
  const uint8_t* b2 = (const uint8_t*)b_g2_query;
  uint8_t* fixed = (uint8_t*)d.fixed;
  DG_HIP(hipMemcpy(d.a_q, aq + (1 + d.ab_lo) * p1, n_ab * p1, kind));
  DG_HIP(hipMemcpy((uint8_t*)d.a_q + n_ab * p1, fx + 2 * p1, p1, kind));            // delta_g1
  DG_HIP(hipMemcpy(d.b1_q, b1 + (1 + d.ab_lo) * p1, n_ab * p1, kind));
  DG_HIP(hipMemcpy((uint8_t*)d.b1_q + n_ab * p1, fx + 2 * p1, p1, kind));
  DG_HIP(hipMemcpy(d.b2_q, b2 + (1 + d.ab_lo) * p2, n_ab * p2, kind));
  DG_HIP(hipMemcpy((uint8_t*)d.b2_q + n_ab * p2, fx + 3 * p1 + p2, p2, kind));      // delta_g2
  DG_HIP(hipMemcpy(d.l_q, (const uint8_t*)l_query + d.l_lo * p1, n_l * p1, kind));
  DG_HIP(hipMemcpy((uint8_t*)d.l_q + n_l * p1, fx + 2 * p1, p1, kind));
  if (n_h) DG_HIP(hipMemcpy(d.h_q, (const uint8_t*)h_query + d.h_lo * p1, n_h * p1, kind));
  DG_HIP(hipMemcpy(fixed, fx, p1, kind));                     // alpha_g1
  DG_HIP(hipMemcpy(fixed + p1, aq, p1, kind));                // a_query[0]
  DG_HIP(hipMemcpy(fixed + 2 * p1, fx + p1, p1, kind));       // beta_g1
  DG_HIP(hipMemcpy(fixed + 3 * p1, b1, p1, kind));            // b_g1_query[0]
  DG_HIP(hipMemcpy(fixed + 4 * p1, fx + 3 * p1, p2, kind));   // beta_g2
  DG_HIP(hipMemcpy(fixed + 4 * p1 + p2, b2, p2, kind));       // b_g2_query[0]
}


}  // namespace dg16

