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

#include "msm_impl.h"
#include "pk.h"

namespace dg16 {

// Extra scalar slots that pair with the delta bases: sc[n] = r, sc[n + 1] = s, sc[n + 2] = -r*s (n = the shard's slice of
// w[1..]).  A, B1, B and L all take THIS scalar vector: the key stores their bases index-aligned (L's has the identity
// at the public-input positions) with the delta slots [d1,0,0] / [0,d1,0] / [0,d2,0] / [0,0,d1].
template <class Fr>
__global__ void prover_scalar_prep_kernel(const Fr* r_s, Fr* sc, size_t n, int mont, int carries_delta) {
  // r_s[0] = r, r_s[1] = s in the same form as the witness (Montgomery iff mont).  Only the last
  // shard carries the delta pairs; the others multiply their delta slots by zero.
  Fr r = r_s[0], s = r_s[1];
  if (!carries_delta) { r = Fr::zero(); s = Fr::zero(); }
  Fr rm = mont ? r : r.to_mont(), sm = mont ? s : s.to_mont();
  Fr nrs = (rm * sm).neg();                 // Montgomery form of -(r*s)
  sc[n] = r;
  sc[n + 1] = s;
  sc[n + 2] = mont ? nrs : nrs.from_mont();
}

// k * p, k canonical NL-limb integer, one lane
template <class F, class Fr>
__device__ XYZZ<F> mul_by_fr(const XYZZ<F>& p, const Fr& k_canon) {
  return scalar_mul<F, Fr::NL>(p, k_canon.l);
}

// Per-shard record (what the ranks all-gather): A', B1', L, H, s*A', r*B1' (G1 Jacobian) then B' (G2 Jacobian),
// where the primed values of shard 0 already contain the fixed points (alpha + a_query[0] etc.).  Everything
// after the gather is then a plain per-slot sum: C = sum L + sum H + sum s*A' + sum r*B1' (linearity), so the
// two serial scalar multiplications (~2.5 ms, one lane each) run BEFORE the exchange, hidden behind the other
// MSMs on a side stream, on every rank.
constexpr int kRecA = 0, kRecB1 = 1, kRecL = 2, kRecH = 3, kRecSA = 4, kRecRB1 = 5, kRecG1 = 6;

// which = 0: A' = msm (+ alpha_g1 + a_query[0] on shard 0), s*A';  which = 1: B1' (+ beta_g1 + b_g1_query[0]),
// r*B1'.  One wave each: the doublings and additions of the scalar multiple run wave-cooperatively (msm_impl.h:
// scalar_mul_wave29 -- round 6: 127 doublings + ~51 additions for BN254 instead of 254 + ~127).
template <class Fq, class Fr>
__global__ void __launch_bounds__(128) prover_stage1_g1_kernel(Jacobian<Fq>* rec, const Affine<Fq>* fixed_g1,
                                                               const Fr* r_s, int mont, int first_shard) {
  const int which = (int)blockIdx.x;     // two workgroups of one wave: s*A' and r*B1' side by side
  __builtin_amdgcn_s_setprio(DG16_CHAIN_PRIO);     // a serial chain on one wave: ahead of the accumulation waves it shares a SIMD with
  Fr r = r_s[0], s = r_s[1];
  if (mont) { r = r.from_mont(); s = s.from_mont(); }
  XYZZ<Fq> v = XYZZ<Fq>::inf();
  if (which == 0 || !r.is_zero()) {      // B1 only matters when r != 0 (prove.rs:106-136)
    v = XYZZ<Fq>::from_jacobian(rec[which == 0 ? kRecA : kRecB1]);
    if (first_shard) v = v.madd(fixed_g1[2 * which], false).madd(fixed_g1[2 * which + 1], false);
  }
  const Fr k = which == 0 ? s : r;
  // (the chain runs on the reduced-radix types, over width-4 NAF digits and -- BN254 -- the two halves of the endomorphism
  // split: msm_impl.h: scalar_mul_wave29)
  __shared__ ScalarMulLds<Fq> lds;
  XYZZ<Fq> kv;
  if constexpr (scalar_mul_splits<Fq>()) {       // two waves per chain: one half of the split scalar each (msm_impl.h)
    __shared__ XYZZ29<Fq> xchg;
    kv = scalar_mul_two_waves29<Fq, Fr::NL>(XYZZ29<Fq>::from_xyzz32(v), k.l, &lds, &xchg).to_xyzz32();
  } else {
    kv = scalar_mul_wave29<Fq, Fr::NL>(XYZZ29<Fq>::from_xyzz32(v), k.l, &lds).to_xyzz32();
  }
  if (threadIdx.x != 0) return;
  rec[which == 0 ? kRecA : kRecB1] = v.to_jacobian();
  rec[which == 0 ? kRecSA : kRecRB1] = kv.to_jacobian();
}
// B' = msm (+ beta_g2 + b_g2_query[0] on shard 0); one wave, wave-cooperative additions (a lone lane took 0.13-0.3 ms
// at the end of the longest chain of a proof)
template <class Fq2>
__global__ void __launch_bounds__(64) prover_stage1_g2_kernel(Jacobian<Fq2>* msm_b2, const Affine<Fq2>* fixed_g2,
                                                               int first_shard) {
  if (!first_shard) return;
  __builtin_amdgcn_s_setprio(DG16_CHAIN_PRIO);
  XYZZ<Fq2> b = XYZZ<Fq2>::from_jacobian(*msm_b2);
#pragma unroll 1
  for (int i = 0; i < 2; i++) b = add_wave(b, XYZZ<Fq2>::from_affine(fixed_g2[i]));
  if (threadIdx.x == 0) *msm_b2 = b.to_jacobian();
}

// ---- the assembly's chains: wave-cooperative operations on the reduced-radix types, products behind a call ----------
// The assembly runs ONCE per proof on a few waves: what it costs is dependent issue plus the fetch of cold code, so every
// level's product goes through one out-of-line copy per field (msm_impl.h: add_wave29<F, SlotMulCall>) and the kernel
// stays a few KB.
// (operands as 16-lane vectors: hipcc passes aggregates beyond 16 dwords per call through scratch, vectors in VGPRs)
typedef uint32_t limbs16_t __attribute__((ext_vector_type(16)));
template <class P, int B>
__device__ __attribute__((noinline)) limbs16_t fe_mul_call(limbs16_t a, limbs16_t b) {
  static_assert(RR<P>::N <= 16, "limbs per element");
  Fe<P, B, 1> x, y;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) { x.l[i] = a[i]; y.l[i] = b[i]; }
  const Fe<P, B, 1> r = fit<B>(x * y);
  limbs16_t o = a;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) o[i] = r.l[i];
  return o;
}
struct SlotMulCall {
  template <class P, int B>
  static __device__ __forceinline__ Fe<P, B, 1> mul(const Fe<P, B, 1>& a, const Fe<P, B, 1>& b) {
    limbs16_t x = {}, y = {};
#pragma unroll
    for (int i = 0; i < RR<P>::N; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
    const limbs16_t o = fe_mul_call<P, B>(x, y);
    Fe<P, B, 1> r;
#pragma unroll
    for (int i = 0; i < RR<P>::N; i++) r.l[i] = o[i];
    return r;
  }
  template <class P, int B>
  static __device__ __forceinline__ Fe2<P, B, 1> mul(const Fe2<P, B, 1>& a, const Fe2<P, B, 1>& b) {
    return slot_mul29_fe2<SlotMulCall>(a, b);
  }
};
// the words of an arkworks-form element re-sliced into the internal limbs (NO form change: still x R32), a constant in
// the storage type, and canonical words back
template <class P>
__device__ __forceinline__ typename FieldOf<Fp<P>>::Store raw29(const Fp<P>& a) {
  return fe_from_words<P>(a.l).template as<FieldOf<Fp<P>>::BS, 1>();
}
template <class P>
__device__ __forceinline__ typename FieldOf<Fp2<Fp<P>>>::Store raw29(const Fp2<Fp<P>>& a) {
  constexpr int BS = FieldOf<Fp2<Fp<P>>>::BS;
  return {fe_from_words<P>(a.c0.l).template as<BS, 1>(), fe_from_words<P>(a.c1.l).template as<BS, 1>()};
}
template <class P, class L>
__device__ __forceinline__ typename FieldOf<Fp<P>>::Store konst29(const Fp<P>*, const L& c) {
  return fe_const<P>(c).template as<FieldOf<Fp<P>>::BS, 1>();
}
template <class P, class L>
__device__ __forceinline__ typename FieldOf<Fp2<Fp<P>>>::Store konst29(const Fp2<Fp<P>>*, const L& c) {
  constexpr int BS = FieldOf<Fp2<Fp<P>>>::BS;
  return {fe_const<P>(c).template as<BS, 1>(), Fe<P, BS, 1>::zero()};
}
template <class P, int B>
__device__ __forceinline__ Fp<P> words32(const Fe<P, B, 1>& v) {
  Fp<P> r;
  fe_to_words<P>(canon(v), r.l);
  return r;
}
template <class P, int B>
__device__ __forceinline__ Fp2<Fp<P>> words32(const Fe2<P, B, 1>& v) { return {words32(v.c0), words32(v.c1)}; }
// a record's point (Jacobian, arkworks form; uniform across the wave) -> XYZZ29: the three form changes x R32 -> x R are
// ONE level of products (slots 0..2), zz = z^2 and zzz = z zz two more (ec.h: XYZZ::from_jacobian, ec29.h: from_xyzz32
// do the same on one lane: 4 + 2 dependent products for G1, three times that for G2)
template <class F>
__device__ __forceinline__ XYZZ29<F> record_to_xyzz29_wave(const Jacobian<F>& j) {
  using P = typename FieldOf<F>::Params;
  if (j.z.is_zero()) return XYZZ29<F>::inf();
  const unsigned slot = (__lane_id() & 15) >> 2;
  const auto w = select(slot == 0, raw29(j.x), select(slot == 1, raw29(j.y), raw29(j.z)));
  const auto r1 = SlotMulCall::mul(w, konst29((const F*)nullptr, RR<P>::FROM32));
  const auto z = bcast29<8>(r1);
  const auto zz = SlotMulCall::mul(z, z);
  return {bcast29<0>(r1), bcast29<4>(r1), zz, SlotMulCall::mul(z, zz)};
}
// ... and back: X = x zz | Y = y zzz | Z = zz (ec.h: XYZZ::to_jacobian), then x R -> x R32, canonical words
template <class F>
__device__ __forceinline__ Jacobian<F> xyzz29_to_record_wave(const XYZZ29<F>& a) {
  using FO = FieldOf<F>;
  using P = typename FO::Params;
  if (a.is_inf()) return {F::one(), F::one(), F::zero()};
  const unsigned slot = (__lane_id() & 15) >> 2;
  const auto a1 = select(slot == 0, a.x, select(slot == 1, a.y, a.zz));
  const auto b1 = select(slot == 0, a.zz, select(slot == 1, a.zzz, FO::one()));
  const auto r1 = SlotMulCall::mul(a1, b1);
  const auto r2 = SlotMulCall::mul(r1, konst29((const F*)nullptr, RR<P>::TO32));
  return {words32(bcast29<0>(r2)), words32(bcast29<4>(r2)), words32(bcast29<8>(r2))};
}

// after the gather: per-slot sums over the shards, C = L + H + s*A + r*B1 (-rs*delta is inside L).
// This kernel is the exposed tail of every proof (and, on N GPUs, of the all-gather), so each sum is a chain on its own
// WAVE: chain 0 sums A, chains 1 and 6 the two halves of B, chains 2..5 L, H, s*A, r*B1 into LDS, and behind the ONE barrier
// chain 1 adds B's other half, chain 2 the three other partials of C -- ceil(n_shards / 2) + 1 dependent G2 additions and
// n_shards + 3 G1 additions.  Two workgroups of four waves (block 0: chains 2..5 = C; block 1: A and B's halves), so that
// a wave has a SIMD's whole register file: seven waves in one workgroup left 256 registers each and the 14-limb curves
// spilled 536 B per lane into scratch.  Round 5: the chains run on the reduced-radix wave-cooperative operations (seven
// product levels per record: three to load it, four to add it; ~0.6 us a level for G1, ~1 us for G2) instead of the
// 32-bit-limb ones (~2.5 us per G1 addition, ~20 us per G2 addition: B's chain alone was 0.18 ms at 8 shards); emulated
// against the oracle in tests/test_kernel_emulation.py::test_proof_assembly_workgroup.
template <class Fq, class Fq2>
__global__ void __launch_bounds__(256, 1) prover_assemble_kernel(const uint8_t* gathered, size_t n_shards, size_t rec_bytes,
                                                                  Jacobian<Fq>* out_a, Jacobian<Fq2>* out_b,
                                                                  Jacobian<Fq>* out_c) {
  __shared__ XYZZ29<Fq> part[4];
  __shared__ XYZZ29<Fq2> part_b;
  __builtin_amdgcn_s_setprio(DG16_CHAIN_PRIO);     // exposed tail: ahead of whatever else is resident on the SIMD
  const unsigned w4 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x == 0 ? 2 + w4 : w4 == 2 ? 6u : w4;      // the chain this wave runs
  if (blockIdx.x == 1 && w4 == 3) return;                                 // (block 1 has three chains)
  if (wave == 1 || wave == 6) {
    const size_t half = (n_shards + 1) / 2;
    const size_t lo = wave == 1 ? 0 : half, cnt = wave == 1 ? half : n_shards - half;
    const size_t steps = cnt + (wave == 1 ? 1 : 0);
    XYZZ29<Fq2> b = XYZZ29<Fq2>::inf();
#pragma unroll 1
    for (size_t k = 0; k < steps; k++) {
      if (k == cnt) __syncthreads();               // (wave 1 only: wave 6 is at the barrier below)
      XYZZ29<Fq2> o;
      if (k < cnt)
        o = record_to_xyzz29_wave(*(const Jacobian<Fq2>*)(gathered + (lo + k) * rec_bytes + kRecG1 * sizeof(Jacobian<Fq>)));
      else
        o = part_b;
      b = add_wave29<Fq2, SlotMulCall>(b, o);
    }
    if (wave == 6) {
      if (lane == 0) part_b = b;
      __syncthreads();
      return;
    }
    const Jacobian<Fq2> r = xyzz29_to_record_wave(b);
    if (lane == 0) *out_b = r;
    return;
  }
  // one addition site for every G1 sum: n_shards records, then (wave 2 only, after the barrier) the other partials
  const int slot = wave == 0 ? kRecA : wave == 2 ? kRecL : wave == 3 ? kRecH : wave == 4 ? kRecSA : kRecRB1;
  XYZZ29<Fq> a = XYZZ29<Fq>::inf();
  const size_t steps = n_shards + (wave == 2 ? 3 : 0);
#pragma unroll 1
  for (size_t k = 0; k < steps; k++) {
    if (k == n_shards) __syncthreads();            // (wave 2 only: the other waves are past their loop, at the barrier below)
    XYZZ29<Fq> o;
    if (k < n_shards)
      o = record_to_xyzz29_wave(*((const Jacobian<Fq>*)(gathered + k * rec_bytes) + slot));
    else
      o = part[k - n_shards + 1];
    a = add_wave29<Fq, SlotMulCall>(a, o);
    if (k + 1 == n_shards && wave >= 3 && lane == 0) part[wave - 2] = a;
  }
  if (wave != 2) __syncthreads();
  if (wave == 0 || wave == 2) {
    const Jacobian<Fq> r = xyzz29_to_record_wave(a);
    if (lane == 0) *(wave == 0 ? out_a : out_c) = r;
  }
}

// bytes of one shard's record (layout above)
template <int CURVE>
static size_t msm_results_bytes() {
  using CT = CurveTypes<CURVE>;
  return kRecG1 * sizeof(Jacobian<typename CT::Fq>) + sizeof(Jacobian<typename CT::Fq2>);
}

// h-polynomial + the five MSMs of this key's shard.  res_out (device or host per dev_ptrs) receives
// msm_results_bytes() bytes.  Uses all three channels; returns with the results stream-ordered on
// channel 0.  Where h comes from:
//   h_given            the caller made it (this key's n_h scalars; a, b, c unused)
//   comm with > 1 rank  sharded h-polynomial (ntt.hip: h_poly_dist_launch): a, b, c are this rank's cyclic rows and
//                      the key is a DG16_F_H_CYCLIC shard
//   otherwise          the whole h-polynomial from the whole a, b, c; the key's slice of it is used
// overlap_tail (DG16_F_OVERLAP_TAIL, one GPU): H's bucket reduction -- the exposed tail of a proof, ~0.6 ms of latency-bound
// launches on an otherwise idle chip -- goes down channel 2's stream instead of channel 0's, so that what the caller
// enqueues next on channel 0 (the next proof's R1CS x witness and h-polynomial) runs under it.  Returns true when it did:
// H's result is then ordered on channel 2's stream, everything else on channel 0's as before.  The buffers the tail
// still reads when the next proof starts (H's buckets and digit-sort metadata, the results record) are fenced with
// pipe_ev[18] at their first reuse (marked "tail fence" below); other entry points wait in Call().
template <int CURVE>
static bool msms_typed(dg16_ctx* ctx, Call& k0, Call& k1, Call& k2, const PkDev& pk, const void* a, const void* b,
                       const void* c, const void* witness, const void* r_s_host, bool mont, bool dev_ptrs,
                       uint8_t* res_dev, const dg16_comm* comm = nullptr, const void* h_given = nullptr,
                       bool overlap_tail = false) {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  using Fr = typename CT::Fr;
  const size_t nv = pk.num_vars, ni = pk.num_inputs, m = pk.m;
  const size_t n_ab = pk.ab_hi - pk.ab_lo;   // this shard's slice of w[1..]
  const size_t n_h = pk.h_hi - pk.h_lo;      // ... of h
  unsigned log_m = 0;
  while (((size_t)1 << log_m) < m) log_m++;
  const size_t g1j = sizeof(Jacobian<Fq>);

  const bool dist = !h_given && comm && comm->n_ranks(comm->self) > 1;
  if (dist) {
    DG_REQUIRE(pk.h_cyclic && comm->n_ranks(comm->self) == pk.nshards && comm->rank(comm->self) == pk.shard,
               DG16_ERR_BAD_ARG, "distributed prove: the key must be the DG16_F_H_CYCLIC shard `rank` of `n_ranks`");
  } else if (!h_given) {
    DG_REQUIRE(!(pk.h_cyclic && pk.nshards > 1), DG16_ERR_BAD_ARG,
               "a DG16_F_H_CYCLIC shard needs the sharded h-polynomial (dg16_groth16_prove_dist / _msms_h)");
  }
  const size_t rows = dist ? m / pk.nshards : m;     // length of a, b, c
  const Fr* w_dev = (const Fr*)stage_in(k0, 18, witness, nv * sizeof(Fr), dev_ptrs);
  const void* a_dev = h_given ? nullptr : stage_in(k0, 19, a, rows * sizeof(Fr), dev_ptrs);
  const void* b_dev = h_given ? nullptr : stage_in(k0, 20, b, rows * sizeof(Fr), dev_ptrs);
  const void* c_dev = h_given ? nullptr : stage_in(k0, 21, c, rows * sizeof(Fr), dev_ptrs);
  const Fr* h_in = h_given ? (const Fr*)stage_in(k0, 19, h_given, n_h * sizeof(Fr), dev_ptrs) : nullptr;
  Fr* r_s = (Fr*)ws(k0.c, 22, 4096);
  Jacobian<Fq>* rec = (Jacobian<Fq>*)res_dev;
  Jacobian<Fq>* res_h = rec + kRecH;
  Jacobian<Fq2>* res_b2 = (Jacobian<Fq2>*)(res_dev + kRecG1 * g1j);
  const int first_shard = pk.shard == 0;
  DG_HIP(hipMemcpyAsync(r_s, r_s_host, 2 * sizeof(Fr), hipMemcpyHostToDevice, k0.s()));
  // ONE scalar vector for A, B1, B and L: w[1..] slice ++ [r, s, -rs]
  Fr* sc_ab = (Fr*)ws(k0.c, 23, (n_ab + 3) * sizeof(Fr));
  DG_HIP(hipMemcpyAsync(sc_ab, w_dev + 1 + pk.ab_lo, n_ab * sizeof(Fr), hipMemcpyDeviceToDevice, k0.s()));
  hipLaunchKernelGGL(prover_scalar_prep_kernel<Fr>, dim3(1), dim3(1), 0, k0.s(), r_s, sc_ab, n_ab, (int)mont,
                     (int)(pk.shard + 1 == pk.nshards));
  DG_HIP(hipGetLastError());
  DG_HIP(hipEventRecord(ctx->pipe_ev[8], k0.s()));
  // Scheduling.  Every saturating kernel of a proof (bucket accumulations, NTTs) goes down ONE stream (channel 0):
  // they are all VALU-bound, and co-scheduling them was measured equal at best and up to 1.7x worse from run to run.
  // The latency-bound bucket reductions and the serial s*A, r*B1 run on channels 1 / 2 behind the next accumulation;
  // only the last reduction is exposed.
  hipStream_t main = k0.s(), side = k1.s(), side2 = k2.s();
  hipEvent_t* ev = ctx->pipe_ev;   // persistent (see ctx.h)
  const Affine<Fq>* fixed_g1 = (const Affine<Fq>*)pk.fixed;
  const Affine<Fq2>* fixed_g2 = (const Affine<Fq2>*)((const uint8_t*)pk.fixed + 4 * sizeof(Affine<Fq>));
  // FOUR streams, not more: the runtime multiplexes streams onto 4 hardware queues by default (GPU_MAX_HW_QUEUES);
  // a fifth stream shares a queue with another one and its kernels serialise behind that one's (measured: +3 ms per
  // 2^20 proof with H's reduction on a fifth stream).  The exchanges of a distributed proof ride on `aux`.
  hipStream_t xch = ctx->aux[0];

  // Schedule.  Main stream (every saturating kernel, back to back):
  //     h-polynomial | B (G2) accumulation | A | B1 | L accumulations | H accumulation
  //   * A, B1, B and L share the digit sort of w[1..] ++ [r, s, -rs];
  //   * the digit sorts (ten small latency-bound launches each) run on `side`: the sort of w under the h-polynomial
  //     (which needs nothing but a, b, c and goes first), the sort of h under the G2 accumulation;
  //   * reductions: B's on `side2` behind its accumulation (the longest chain: it has all G1 accumulations to hide
  //     behind), A's and L's on `side`, B1's on `xch` followed by the two serial scalar multiples s*A', r*B1'; H's is
  //     the exposed tail (or, with overlap_tail, runs on `side2` under the next proof).
  // Measured and removed (rounds 2-4): A / B1 / L as three instances of ONE launch with one reduction
  // chain (12.5 vs 11.4 ms: the 4-ms launch starves B's reduction waves of SIMD slots), a second low-priority lane of
  // saturating kernels, the h-polynomial on a side stream, B's reduction after H's accumulation.
  // Distributed proof: the three stages of the sharded h-polynomial interleave with the accumulations so that each
  // all-to-all hides behind one:  stage 0 | a2a 1 || B | stage 1 | a2a 2 || A, B1, L | stage 2 | sort h | H.
  const unsigned n_ranks = dist ? comm->n_ranks(comm->self) : 1, rank = dist ? comm->rank(comm->self) : 0;
  const size_t xbytes = 3 * rows * sizeof(Fr);
  void* xbuf_a = dist ? ws(k0.c, 26, xbytes) : nullptr;
  void* xbuf_b = dist ? ws(k0.c, 27, xbytes) : nullptr;
  auto exchange = [&](int ev_ready, int ev_done) {        // xbuf_a (made on main) -> all-to-all on xch -> xbuf_b
    DG_HIP(hipEventRecord(ev[ev_ready], main));
    DG_HIP(hipStreamWaitEvent(xch, ev[ev_ready], 0));
    int rc = comm->all_to_all(comm->self, xbuf_a, xbuf_b, xbytes / n_ranks, xch);
    DG_REQUIRE(rc == DG16_OK, DG16_ERR_NET, "all-to-all of the sharded h-polynomial failed");
    DG_HIP(hipEventRecord(ev[ev_done], xch));
  };

  // side: the digit sort shared by A, B1, B and L; its buffers live in channel 1
  DG_HIP(hipStreamWaitEvent(side, ev[8], 0));
  MsmSort st_ab = msm_sort_on<Fr, CT::SCALAR_BITS>(side, k1.c, sc_ab, n_ab + 3, mont, true, pk.c_ab, pk.stride);
  DG_HIP(hipEventRecord(ev[13], side));
  MsmBuffers<Fq2> buf_b2 = msm_buffers<Fq2>(k2.c, st_ab.g);
  buf_b2.busy_chip = true;

  // main: h (whole, or stage 0 of the sharded form)
  Fr* h_dev = h_given ? nullptr : (Fr*)ws(k0.c, 3, rows * sizeof(Fr));
  const Fr* h_scalars = h_in;
  if (dist) {
    const void* rows_in[3] = {a_dev, b_dev, c_dev};
    h_poly_dist_stage(k0, CURVE, log_m, rank, n_ranks, 0, rows_in, xbuf_a);
    exchange(3, 4);
  } else if (!h_given) {
    h_poly_launch(k0, CURVE, a_dev, b_dev, c_dev, log_m, h_dev);
    h_scalars = h_dev + pk.h_lo;
  }
  MsmSort st_h;
  const bool tail_fence = overlap_tail && ctx->tail_pending.load(std::memory_order_acquire);
  if (!dist) {   // side: the sort of h, underneath the G2 accumulation
    DG_HIP(hipEventRecord(ev[14], main));
    DG_HIP(hipStreamWaitEvent(side, ev[14], 0));
    // tail fence: the last proof's H reduction reads the digit-sort metadata this sort rewrites (and, further down this
    // stream, A's and L's reductions write the results record its assembly reads)
    if (tail_fence) DG_HIP(hipStreamWaitEvent(side, ev[18], 0));
    st_h = msm_sort_on<Fr, CT::SCALAR_BITS>(side, k0.c, h_scalars, n_h, true, true, pk.c_h, pk.stride);
    DG_HIP(hipEventRecord(ev[15], side));
  }

  // main: B (G2).  (The timing events of channels 2 / 1 bracket the G2 / G1 accumulation ON THE STREAM THEY RUN ON, so
  // that dg16_last_kernel_ms(ctx, 2 or 1, 1) reports the dominant kernels of the proof that was just made.)
  DG_HIP(hipStreamWaitEvent(main, ev[13], 0));
  DG_HIP(hipEventRecord(k2.c.ev[2], main));
  buf_b2.acc_done = k2.c.ev[3];              // the accumulation kernel alone; its finalize follows on the same stream
  if (ctx->kclk) buf_b2.clk = ctx->kclk + 2 * 2;      // its clock under the kernel (dg16_last_kernel_ms(ctx, 2, 2))
  {
    // B's finalize (a throughput kernel at two waves per SIMD: 8 + 1 dependent Fq2 additions per lane) goes to B's reduction
    // stream, under A's accumulation, instead of holding the main stream: an 8-shard rank in a queue 2.48 -> 2.36-2.38 ms,
    // the 2^20 proof and config 4 within the run-to-run spread (profiles/r6oo_g2_finalize_side_ab.txt; =0 puts it back)
    // Nine-limb fields only: the 14-limb G2 finalize (a step loop at ONE wave per SIMD with the whole register file) next
    // to the 14-limb G1 accumulation costs a BLS12-381 2^20 proof 1 ms (21.7 -> 22.7 in a queue: profiles/r6qq_*).
    static const int fin_env = [] { const char* e = getenv("DG16_G2_FINALIZE_SIDE"); return e ? atoi(e) : -1; }();
    const bool side_fin = fin_env >= 0 ? fin_env != 0 : RR<typename FieldOf<Fq>::Params>::N == 9;
    if (side_fin) buf_b2.finalize_stream = side2;
  }
  msm_accumulate_phase<Fq2>(main, st_ab, buf_b2, pk.b2_q);
  k2.c.ev_valid[1] = true;
  DG_HIP(hipEventRecord(ev[2], main));
  DG_HIP(hipStreamWaitEvent(side2, ev[2], 0));
  msm_bucket_phase<Fq2>(side2, st_ab, buf_b2, false, res_b2);
  hipLaunchKernelGGL(prover_stage1_g2_kernel<Fq2>, dim3(1), dim3(64), 0, side2, res_b2, fixed_g2, first_shard);
  DG_HIP(hipEventRecord(ev[5], side2));
  if (dist) {
    const void* in1[1] = {xbuf_b};
    DG_HIP(hipStreamWaitEvent(main, ev[4], 0));
    h_poly_dist_stage(k0, CURVE, log_m, rank, n_ranks, 1, in1, xbuf_a);
    exchange(9, 11);
  }

  // main: A, B1, L -- three launches, three reductions on side streams (A's and L's on `side`, B1's on `xch` followed
  // by the two serial scalar multiples s*A', r*B1')
  // SHORT MSMs (a shard of a quarter or less of a 2^20 key, BASELINE config 4) run A, B1 and L as three INSTANCES of one
  // accumulation launch with one reduction chain (MsmBases): their launches are latency-bound -- one round of workgroups
  // of 8-entry segments each -- and three of them in a row on the main stream is three times that latency.  (At 2^20 the
  // merged launch was measured slower in round 3: one 4-ms launch starves B's reduction waves where three 1.3-ms launches
  // do not -- CHANGELOG.md.)
  static const int abl_env = [] { const char* e = getenv("DG16_ABL_MERGED"); return e ? atoi(e) : -1; }();
  const bool abl_merged = abl_env >= 0 ? abl_env != 0 : st_ab.g.region * st_ab.g.bw <= ((size_t)1 << 22);
  if (abl_merged) {
    MsmBuffers<Fq> buf3 = msm_buffers<Fq>(k0.c, st_ab.g, 3);
    buf3.busy_chip = true;
    const void* bases3[3] = {pk.a_q, pk.b1_q, pk.l_q};
    static_assert(kRecA == 0 && kRecB1 == 1 && kRecL == 2, "the three results land back to back in the record");
    DG_HIP(hipEventRecord(k1.c.ev[2], main));
    if (ctx->kclk) buf3.clk = ctx->kclk + 2 * 1;
    msm_accumulate_phase<Fq>(main, st_ab, buf3, bases3);
    DG_HIP(hipEventRecord(k1.c.ev[3], main));
    k1.c.ev_valid[1] = true;
    DG_HIP(hipEventRecord(ev[6], main));
    DG_HIP(hipStreamWaitEvent(side, ev[6], 0));
    if (tail_fence) DG_HIP(hipStreamWaitEvent(side, ev[18], 0));    // tail fence: the record the last assembly reads
    msm_bucket_phase<Fq>(side, st_ab, buf3, false, rec + kRecA);
    DG_HIP(hipEventRecord(ev[12], side));
    DG_HIP(hipStreamWaitEvent(xch, ev[12], 0));
    hipLaunchKernelGGL((prover_stage1_g1_kernel<Fq, Fr>), dim3(2), dim3(scalar_mul_splits<Fq>() ? 128 : 64), 0, xch, rec,
                       fixed_g1, r_s, (int)mont, first_shard);
    DG_HIP(hipEventRecord(ev[7], xch));
    DG_HIP(hipStreamWaitEvent(side, ev[7], 0));
    DG_HIP(hipEventRecord(ev[10], side));             // A, B1, L, s*A', r*B1' all done
  } else
  {
    MsmBuffers<Fq> buf_a = msm_buffers<Fq>(k0.c, st_ab.g);
    MsmBuffers<Fq> buf_b1 = msm_buffers<Fq>(k1.c, st_ab.g);
    MsmBuffers<Fq> buf_l = msm_buffers<Fq>(ctx->xws[1], st_ab.g);
    buf_a.busy_chip = buf_b1.busy_chip = buf_l.busy_chip = true;
    msm_accumulate_phase<Fq>(main, st_ab, buf_a, pk.a_q);
    DG_HIP(hipEventRecord(ev[0], main));
    // (the timing bracket and the clock probe of channel 1 sit on B1's launch: A's runs beside B's G2 finalize since that
    // moved to B's reduction stream, B1's beside the light row / top kernels only -- the launch that says what the kernel does)
    DG_HIP(hipEventRecord(k1.c.ev[2], main));
    if (ctx->kclk) buf_b1.clk = ctx->kclk + 2 * 1;    // (dg16_last_kernel_ms(ctx, 1, 2))
    msm_accumulate_phase<Fq>(main, st_ab, buf_b1, pk.b1_q);
    DG_HIP(hipEventRecord(k1.c.ev[3], main));
    k1.c.ev_valid[1] = true;
    DG_HIP(hipEventRecord(ev[1], main));
    msm_accumulate_phase<Fq>(main, st_ab, buf_l, pk.l_q);
    DG_HIP(hipEventRecord(ev[6], main));
    DG_HIP(hipStreamWaitEvent(side, ev[0], 0));
    if (tail_fence) DG_HIP(hipStreamWaitEvent(side, ev[18], 0));    // tail fence: rec[kRecA] / rec[kRecL] (the single-GPU form
                                                                     // has it in front of the sort of h already; the sharded
                                                                     // form sorts h on main)
    msm_bucket_phase<Fq>(side, st_ab, buf_a, false, rec + kRecA);
    DG_HIP(hipEventRecord(ev[12], side));
    DG_HIP(hipStreamWaitEvent(xch, ev[1], 0));
    if (tail_fence) DG_HIP(hipStreamWaitEvent(xch, ev[18], 0));     // tail fence: rec[kRecB1] is read by the last assembly
    msm_bucket_phase<Fq>(xch, st_ab, buf_b1, false, rec + kRecB1);
    DG_HIP(hipStreamWaitEvent(xch, ev[12], 0));
    hipLaunchKernelGGL((prover_stage1_g1_kernel<Fq, Fr>), dim3(2), dim3(scalar_mul_splits<Fq>() ? 128 : 64), 0, xch, rec,
                       fixed_g1, r_s, (int)mont, first_shard);
    DG_HIP(hipEventRecord(ev[7], xch));
    DG_HIP(hipStreamWaitEvent(side, ev[6], 0));
    msm_bucket_phase<Fq>(side, st_ab, buf_l, false, rec + kRecL);
    DG_HIP(hipStreamWaitEvent(side, ev[7], 0));
    DG_HIP(hipEventRecord(ev[10], side));             // A, B1, L, s*A', r*B1' all done
  }

  // main: the rest of a sharded h, the sort of h where it has not run yet, H and its reduction (the exposed tail)
  if (dist) {
    const void* in2[1] = {xbuf_b};
    DG_HIP(hipStreamWaitEvent(main, ev[11], 0));
    h_poly_dist_stage(k0, CURVE, log_m, rank, n_ranks, 2, in2, h_dev);
    if (tail_fence) DG_HIP(hipStreamWaitEvent(main, ev[18], 0));    // tail fence: the digit-sort metadata of h
    st_h = msm_sort_on<Fr, CT::SCALAR_BITS>(main, k0.c, h_dev, n_h, true, true, pk.c_h, pk.stride);
  } else {
    DG_HIP(hipStreamWaitEvent(main, ev[15], 0));
  }
  MsmBuffers<Fq> buf_h = msm_buffers<Fq>(ctx->xws[0], st_h.g);
  const bool tail_on_side2 = overlap_tail && !h_given;      // (round 6: the sharded proof too -- prove_dist_typed)
  // H's reduction on the main stream is the exposed tail of the proof: nothing saturating runs beside it, the 16-wave form
  // of the lane reduction serves (msm_reduce_impl.h); under the next proof's first kernels (side2) the one-wave form
  buf_h.busy_chip = tail_on_side2;
  if (tail_fence) DG_HIP(hipStreamWaitEvent(main, ev[18], 0));     // tail fence: H's buckets
  msm_accumulate_phase<Fq>(main, st_h, buf_h, pk.h_q);
  if (tail_on_side2) {
    DG_HIP(hipEventRecord(ev[17], main));
    DG_HIP(hipStreamWaitEvent(side2, ev[17], 0));
    msm_bucket_phase<Fq>(side2, st_h, buf_h, false, res_h);
    DG_HIP(hipStreamWaitEvent(side2, ev[10], 0));       // A, B1, L results, s*A, r*B1 (B's are in order on side2 itself)
  } else {
    msm_bucket_phase<Fq>(main, st_h, buf_h, false, res_h);
  }
  DG_HIP(hipStreamWaitEvent(main, ev[10], 0));          // A, B1, L results, s*A, r*B1
  DG_HIP(hipStreamWaitEvent(main, ev[5], 0));           // B result
  DG_HIP(hipGetLastError());
  return tail_on_side2;
}

// proof = assemble(sum of the shards' MSM results).  res_dev: msm_results_bytes() on the device.
template <int CURVE>
static void assemble_typed(Call& k0, const uint8_t* gathered_dev, size_t n_shards, uint8_t* proof_dev,
                           hipStream_t stream = nullptr) {
  using CT = CurveTypes<CURVE>;
  using Fq = typename CT::Fq;
  using Fq2 = typename CT::Fq2;
  const size_t g1j = sizeof(Jacobian<Fq>), g2j = sizeof(Jacobian<Fq2>);
  hipLaunchKernelGGL((prover_assemble_kernel<Fq, Fq2>), dim3(2), dim3(256), 0, stream ? stream : k0.s(), gathered_dev, n_shards,
                     msm_results_bytes<CURVE>(), (Jacobian<Fq>*)proof_dev, (Jacobian<Fq2>*)(proof_dev + g1j),
                     (Jacobian<Fq>*)(proof_dev + g1j + g2j));
  DG_HIP(hipGetLastError());
}

template <int CURVE>
static void prove_typed(dg16_ctx* ctx, const PkDev& pk, const void* a, const void* b, const void* c,
                        const void* witness, const void* r_s_host, bool mont, bool dev_ptrs, void* proof_out,
                        bool overlap_tail = false) {
  using CT = CurveTypes<CURVE>;
  const size_t g1j = sizeof(Jacobian<typename CT::Fq>), g2j = sizeof(Jacobian<typename CT::Fq2>);
  DG_REQUIRE(pk.nshards == 1, DG16_ERR_BAD_ARG, "dg16_groth16_prove needs an unsharded key; use _msms + _assemble");
  overlap_tail = overlap_tail && dev_ptrs;      // a host-pointer call ends in a synchronisation anyway
  Call k0(ctx, 0, overlap_tail), k1(ctx, 1, overlap_tail), k2(ctx, 2, overlap_tail);
  uint8_t* buf = (uint8_t*)ws(k0.c, 16, 8192);
  uint8_t* res_dev = buf;
  uint8_t* proof_dev = buf + 4096;
  k0.begin_dominant();
  const bool tail = msms_typed<CURVE>(ctx, k0, k1, k2, pk, a, b, c, witness, r_s_host, mont, dev_ptrs, res_dev, nullptr,
                                      nullptr, overlap_tail);
  if (tail) {
    // the proof is complete on channel 2's stream (dg16.h: DG16_F_OVERLAP_TAIL); channel 0 is free for the next one
    k0.end_dominant();
    assemble_typed<CURVE>(k0, res_dev, 1, proof_dev, k2.s());
    if (proof_out != proof_dev)
      DG_HIP(hipMemcpyAsync(proof_out, proof_dev, 2 * g1j + g2j, hipMemcpyDeviceToDevice, k2.s()));
    DG_HIP(hipEventRecord(ctx->pipe_ev[18], k2.s()));
    ctx->tail_pending.store(true, std::memory_order_release);
    k0.finish();
    k1.finish();
    k2.finish();
    return;
  }
  assemble_typed<CURVE>(k0, res_dev, 1, proof_dev);
  k0.end_dominant();
  stage_out(k0, proof_out, proof_dev, 2 * g1j + g2j, dev_ptrs);
  k0.finish();
  k1.finish();
  k2.finish();
  if (!dev_ptrs) DG_HIP(hipStreamSynchronize(k0.s()));
}

// The whole distributed proof on this rank: msms (with the sharded h-polynomial) -> all-gather of the records on
// channel 0's stream -> assembly.  No host synchronisation anywhere (device-pointer calls).
template <int CURVE>
static void prove_dist_typed(dg16_ctx* ctx, const PkDev& pk, const dg16_comm* comm, const void* a, const void* b,
                             const void* c, const void* witness, const void* r_s_host, bool mont, bool dev_ptrs,
                             void* proof_out, bool overlap_tail = false) {
  using CT = CurveTypes<CURVE>;
  const size_t g1j = sizeof(Jacobian<typename CT::Fq>), g2j = sizeof(Jacobian<typename CT::Fq2>);
  const unsigned n = comm ? comm->n_ranks(comm->self) : 1;
  DG_REQUIRE(pk.nshards == n, DG16_ERR_BAD_ARG, "distributed prove: key shards != ranks");
  overlap_tail = overlap_tail && dev_ptrs;      // a host-pointer call ends in a synchronisation anyway
  Call k0(ctx, 0, overlap_tail), k1(ctx, 1, overlap_tail), k2(ctx, 2, overlap_tail);
  uint8_t* buf = (uint8_t*)ws(k0.c, 16, 8192);
  uint8_t* res_dev = buf;
  uint8_t* proof_dev = buf + 4096;
  const size_t rec = msm_results_bytes<CURVE>();
  uint8_t* gathered = n > 1 ? (uint8_t*)ws(k0.c, 28, n * rec) : res_dev;
  k0.begin_dominant();
  const bool tail = msms_typed<CURVE>(ctx, k0, k1, k2, pk, a, b, c, witness, r_s_host, mont, dev_ptrs, res_dev, comm, nullptr,
                                      overlap_tail);
  // DG16_F_OVERLAP_TAIL (round 6: the sharded proof of a QUEUE): H's bucket reduction, the all-gather and the assembly --
  // the exposed tail of a rank, ~0.4 of its 3.0 ms at 8 shards -- are ordered on channel 2's stream, so the next proof's
  // first stage and its B accumulation start under them.  The all-gather is then issued on another stream than the
  // all-to-alls of the h-polynomial (the aux stream): every rank enqueues its collectives in the same program order, which
  // is what one communicator needs; the transport serialises them in that order.
  hipStream_t ts = tail ? k2.s() : k0.s();
  if (n > 1) {
    // the "all-reduce of bucket sums": RCCL has no user-defined reduction, so the N records (768 B each for BN254)
    // are gathered and every rank adds them (assemble)
    int rc = comm->all_gather(comm->self, res_dev, rec, gathered, ts);
    DG_REQUIRE(rc == DG16_OK, DG16_ERR_NET, "all-gather of the MSM records failed");
  }
  assemble_typed<CURVE>(k0, gathered, n, proof_dev, ts);
  k0.end_dominant();
  if (tail) {
    if (proof_out != proof_dev)
      DG_HIP(hipMemcpyAsync(proof_out, proof_dev, 2 * g1j + g2j, hipMemcpyDeviceToDevice, ts));
    DG_HIP(hipEventRecord(ctx->pipe_ev[18], ts));
    ctx->tail_pending.store(true, std::memory_order_release);
  } else {
    stage_out(k0, proof_out, proof_dev, 2 * g1j + g2j, dev_ptrs);
  }
  k0.finish();
  k1.finish();
  k2.finish();
  if (!dev_ptrs) DG_HIP(hipStreamSynchronize(k0.s()));
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
  // L rides on the digit sort of A / B1 / B: its bases are index-aligned with w[1..] -- position i of the slice holds
  // l_query[ab_lo + i - (ni - 1)], the identity where w[1 + ab_lo + i] is a public input (l_query has no such element)
  d.l_lo = (d.ab_lo > ni - 1 ? d.ab_lo : ni - 1) - (ni - 1);
  d.l_hi = (d.ab_hi > ni - 1 ? d.ab_hi : ni - 1) - (ni - 1);
  slice(m, d.h_lo, d.h_hi);
  if (d.h_cyclic) {
    DG_REQUIRE(!(d.nshards & (d.nshards - 1)) && (size_t)d.nshards * d.nshards <= m, DG16_ERR_BAD_ARG,
               "DG16_F_H_CYCLIC: n_shards must be a power of two with n_shards^2 <= domain_size");
    d.h_lo = 0;
    d.h_hi = m / d.nshards;
  }
  const size_t n_ab = d.ab_hi - d.ab_lo, n_l = d.l_hi - d.l_lo, n_h = d.h_hi - d.h_lo;
  DG_HIP(hipSetDevice(ctx->device));
  // plain arrays first (slice ++ delta slots), then the window tables that replace them
  // the plain arrays only live until the tables are built: freed on every path, including a DG_HIP throw mid-build
  // (pk_free releases d.* -- the tables built so far and d.fixed -- when the caller sees the error)
  void *a_plain = nullptr, *b1_plain = nullptr, *b2_plain = nullptr, *l_plain = nullptr, *h_plain = nullptr;
  struct PlainGuard {
    void **p[5];
    ~PlainGuard() {
      for (void** q : p)
        if (*q) { hipFree(*q); *q = nullptr; }
    }
  } plain_guard{{&a_plain, &b1_plain, &b2_plain, &l_plain, &h_plain}};
  DG_HIP(hipMalloc(&a_plain, (n_ab + 3) * p1));
  DG_HIP(hipMalloc(&b1_plain, (n_ab + 3) * p1));
  DG_HIP(hipMalloc(&b2_plain, (n_ab + 3) * p2));
  DG_HIP(hipMalloc(&l_plain, (n_ab + 3) * p1));
  DG_HIP(hipMalloc(&h_plain, (n_h ? n_h : 1) * p1));
  DG_HIP(hipMalloc(&d.fixed, 4 * p1 + 2 * p2));
  // fixed_host layout: alpha_g1, beta_g1, delta_g1 (G1 affine) | beta_g2, delta_g2 (G2 affine)
  const uint8_t* fx = (const uint8_t*)fixed_host;
  const uint8_t* aq = (const uint8_t*)a_query;
  const uint8_t* b1 = (const uint8_t*)b_g1_query;
  const uint8_t* b2 = (const uint8_t*)b_g2_query;
  uint8_t* fixed = (uint8_t*)d.fixed;
  DG_HIP(hipMemset((uint8_t*)a_plain + n_ab * p1, 0, 3 * p1));
  DG_HIP(hipMemset((uint8_t*)b1_plain + n_ab * p1, 0, 3 * p1));
  DG_HIP(hipMemset((uint8_t*)b2_plain + n_ab * p2, 0, 3 * p2));
  DG_HIP(hipMemset(l_plain, 0, (n_ab + 3) * p1));
  DG_HIP(hipMemcpy(a_plain, aq + (1 + d.ab_lo) * p1, n_ab * p1, kind));
  DG_HIP(hipMemcpy((uint8_t*)a_plain + n_ab * p1, fx + 2 * p1, p1, kind));                 // [delta_g1, 0]
  DG_HIP(hipMemcpy(b1_plain, b1 + (1 + d.ab_lo) * p1, n_ab * p1, kind));
  DG_HIP(hipMemcpy((uint8_t*)b1_plain + (n_ab + 1) * p1, fx + 2 * p1, p1, kind));          // [0, delta_g1]
  DG_HIP(hipMemcpy(b2_plain, b2 + (1 + d.ab_lo) * p2, n_ab * p2, kind));
  DG_HIP(hipMemcpy((uint8_t*)b2_plain + (n_ab + 1) * p2, fx + 3 * p1 + p2, p2, kind));     // [0, delta_g2]
  if (n_l)      // slice position of l_query[l_lo]: (ni - 1 + l_lo) - ab_lo
    DG_HIP(hipMemcpy((uint8_t*)l_plain + (ni - 1 + d.l_lo - d.ab_lo) * p1, (const uint8_t*)l_query + d.l_lo * p1, n_l * p1,
                     kind));
  DG_HIP(hipMemcpy((uint8_t*)l_plain + (n_ab + 2) * p1, fx + 2 * p1, p1, kind));          // [0, 0, delta_g1]
  if (d.h_cyclic)   // h_query[shard + nshards * j]: the layout the sharded h-polynomial leaves its output in
    DG_HIP(hipMemcpy2D(h_plain, p1, (const uint8_t*)h_query + d.shard * p1, (size_t)d.nshards * p1, p1, n_h, kind));
  else if (n_h) DG_HIP(hipMemcpy(h_plain, (const uint8_t*)h_query + d.h_lo * p1, n_h * p1, kind));
  {
    using CTc = CurveTypes<CURVE>;
    auto nwin_of = [](unsigned c) { return (unsigned)((CTc::SCALAR_BITS + 1 + c - 1) / c); };
    d.c_ab = msm_window_bits(n_ab + 3, true, CTc::SCALAR_BITS);
    d.c_l = d.c_ab;
    d.c_h = msm_window_bits(n_h ? n_h : 1, true, CTc::SCALAR_BITS);
    // (H's window width on its own was swept in round 4, profiles/r4w_table_window_sweep.txt: log2(n) - 3 for it too)
    // HBM budget (dg16_ctx_set_table_budget): one row stride for all five tables -- A, B1, B and L share a digit sort
    const size_t full_bytes = (size_t)nwin_of(d.c_ab) * (n_ab + 3) * (3 * p1 + p2) +
                              (size_t)nwin_of(d.c_h) * (n_h ? n_h : 1) * p1;
    const unsigned wmin = nwin_of(d.c_ab) < nwin_of(d.c_h) ? nwin_of(d.c_ab) : nwin_of(d.c_h);
    unsigned stride = 1;
    auto rows_of = [&](unsigned c) { return (nwin_of(c) + stride - 1) / stride; };
    auto bytes_at = [&] { return (size_t)rows_of(d.c_ab) * (n_ab + 3) * (3 * p1 + p2) + (size_t)rows_of(d.c_h) * (n_h ? n_h : 1) * p1; };
    if (ctx->table_budget)
      while (stride < wmin && bytes_at() > ctx->table_budget) stride++;
    d.stride = stride;
    (void)full_bytes;
    d.a_q = msm_build_table<Fq>(nullptr, a_plain, n_ab + 3, d.c_ab * stride, rows_of(d.c_ab));
    d.b1_q = msm_build_table<Fq>(nullptr, b1_plain, n_ab + 3, d.c_ab * stride, rows_of(d.c_ab));
    d.b2_q = msm_build_table<Fq2>(nullptr, b2_plain, n_ab + 3, d.c_ab * stride, rows_of(d.c_ab));
    d.l_q = msm_build_table<Fq>(nullptr, l_plain, n_ab + 3, d.c_ab * stride, rows_of(d.c_ab));
    d.h_q = msm_build_table<Fq>(nullptr, h_plain, n_h, d.c_h * stride, rows_of(d.c_h));
    d.table_bytes = bytes_at();
    DG_HIP(hipDeviceSynchronize());
  }
  DG_HIP(hipMemcpy(fixed, fx, p1, kind));                     // alpha_g1
  DG_HIP(hipMemcpy(fixed + p1, aq, p1, kind));                // a_query[0]
  DG_HIP(hipMemcpy(fixed + 2 * p1, fx + p1, p1, kind));       // beta_g1
  DG_HIP(hipMemcpy(fixed + 3 * p1, b1, p1, kind));            // b_g1_query[0]
  DG_HIP(hipMemcpy(fixed + 4 * p1, fx + 3 * p1, p2, kind));   // beta_g2
  DG_HIP(hipMemcpy(fixed + 4 * p1 + p2, b2, p2, kind));       // b_g2_query[0]
}


}  // namespace dg16

