// MSM / base generation / affine conversion / d_msm / packexp for ONE (curve, group), selected at compile time:
//   hipcc -DDG_CURVE=<0|1|2> -DDG_GROUP=<1|2> -DDG_NAME=<curve>_g<k> -c msm_group.hip -o msm_<curve>_g<k>.o
// (one object per pair so that the heavy template instantiations compile in parallel; msm.hip dispatches on the
// names).  The bucket reduction of the same pair is msm_reduce.hip.
#include "msm_impl.h"
#include "dist_impl.h"

#define DG_CAT2(a, b) a##b
#define DG_CAT(a, b) DG_CAT2(a, b)
#define DG_FN(stem) DG_CAT(stem, DG_NAME)

namespace dg16 {
using CT = CurveTypes<DG_CURVE>;
#if DG_GROUP == 1
using GF = CT::Fq;
using GC = CT::G1c;
#else
using GF = CT::Fq2;
using GC = CT::G2c;
#endif

// the host-side phases other translation units (the prover) call for this pair: instantiated here, `extern` there
template void msm_accumulate_phase<GF>(hipStream_t, const MsmSort&, const MsmBuffers<GF>&, const void* const*);
template void msm_finalize_phase<GF>(hipStream_t, const MsmSort&, const MsmBuffers<GF>&);
template void msm_tail_phase<GF>(hipStream_t, const MsmSort&, const MsmBuffers<GF>&, bool, void*);
template void* msm_build_table<GF>(hipStream_t, const void*, size_t, unsigned, unsigned);
template MsmSort msm_sort_on<CT::Fr, CT::SCALAR_BITS>(hipStream_t, Channel&, const void*, size_t, unsigned, bool, unsigned,
                                                      unsigned);

void DG_FN(msm_)(Call& k, const void* bases, const void* scalars, size_t n, unsigned mode, bool affine, void* out) {
  msm_run<GF, CT::Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mode, affine, out);
}
void DG_FN(gen_bases_)(Call& k, uint64_t seed, size_t n, void* out) { gen_bases_run<GF, GC>(k, seed, n, out); }
void DG_FN(to_affine_)(Call& k, const void* jac, void* out, size_t n) { to_affine_run<GF>(k, jac, out, n); }

// resident bases: the table of window multiples (internal form) for n points; c and the row count come back.
// budget != 0: at most that many bytes -- the table keeps every stride-th row (msm_geometry) so that it fits
void* DG_FN(bases_table_)(Call& k, const void* bases_dev, size_t n, size_t budget, unsigned* c_out, unsigned* rows_out,
                          unsigned* stride_out) {
  const unsigned c = msm_window_bits(n ? n : 1, true, CT::SCALAR_BITS);
  const unsigned nwin = (CT::SCALAR_BITS + 1 + c - 1) / c;
  const unsigned stride = table_stride_for((size_t)nwin * (n ? n : 1) * sizeof(Affine<GF>), budget, nwin);
  const unsigned rows = (nwin + stride - 1) / stride;
  *c_out = c;
  *rows_out = rows;
  *stride_out = stride;
  return msm_build_table<GF>(k.s(), bases_dev, n, c * stride, rows);
}
// MSM over a resident table: one digit sort in table mode; stride 1: one bucket set, no Horner tail
void DG_FN(msm_resident_)(Call& k, const void* table, size_t n, unsigned c, unsigned stride, const void* scalars,
                          bool mont, bool affine, void* out) {
  MsmSort st = msm_sort<CT::Fr, CT::SCALAR_BITS>(k, scalars, n, mont, true, c, stride);
  msm_reduce<GF>(k, st, table, affine, out);
}

// The king's combination "in the exponent" when there are FEW outputs (d_msm: one; unpackexp of a handful of elements):
// out[e][r] = sum_c M[r][c] P[e][c] with one WORKGROUP per output and one WAVE per term -- the scalar multiples run side
// by side as wave-cooperative width-4 NAF chains on the reduced-radix types (msm_impl.h: scalar_mul_wave29), then wave 0
// adds the terms.  Round 6: the lane-per-output kernel of dist_impl.h (matvec_points_kernel: right for thousands of
// outputs) ran d_msm's single output as EIGHT full scalar multiplications one after another on ONE lane on 32-bit limbs --
// ~80 ms for BLS12-377 G1 whatever the size of the MSM in front of it (bench.py's dmsm_sweep: 81-99 ms per round at
// 2^10 .. 2^17 points per party, profiles/r6d_bench_line.json).
template <class F, class Fr>
__global__ void __launch_bounds__(512) matvec_points_wave_kernel(const Fr* __restrict__ Mc, unsigned rows, unsigned cols,
                                                                   const Affine<F>* __restrict__ in,
                                                                   Affine<F>* __restrict__ out) {
  __shared__ ScalarMulLds<F> lds[8];
  __shared__ XYZZ29<F> part[8];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t e = blockIdx.x / rows;
  const unsigned r = blockIdx.x % rows;
  __builtin_amdgcn_s_setprio(DG16_CHAIN_PRIO);
  {
    const Affine<F> q = in[e * cols + wave];
    XYZZ29<F> v = XYZZ29<F>::inf();
    if (!q.is_inf()) v = XYZZ29<F>::from_xyzz32(XYZZ<F>::from_affine(q));
    const Fr k = Mc[(size_t)r * cols + wave];
    // the NAF recoding and the table of odd multiples of a chain live in THIS wave's slice of LDS; the barriers inside
    // scalar_mul_wave29 are workgroup barriers that every wave reaches the same number of times (the chain's length
    // differs between waves, its barriers do not)
    v = scalar_mul_wave29<F, Fr::NL, false>(v, k.l, &lds[wave]);
    if (lane == 0) part[wave] = v;
  }
  __syncthreads();
  if (wave != 0) return;
  const Affine<F> acc = sum_points_affine_wave<F>(part, cols);
  if (lane == 0) out[blockIdx.x] = acc;
}
// out[e][r] = sum_c M[r][c] P[e][c]: few outputs -> a workgroup each (<= 8 terms: two waves per SIMD, 256 registers each),
// many -> a lane each
template <class F, class Fr>
static void matvec_points(hipStream_t s, const Fr* Mc, unsigned rows, unsigned cols, const Affine<F>* in, Affine<F>* out,
                          size_t count) {
  if (count * rows <= 256 && cols >= 1 && cols <= 8)
    hipLaunchKernelGGL((matvec_points_wave_kernel<F, Fr>), dim3((unsigned)(count * rows)), dim3(64 * cols), 0, s, Mc, rows, cols,
                       in, out);
  else
    hipLaunchKernelGGL((matvec_points_kernel<F, Fr>), dim3((unsigned)((count * rows + 63) / 64)), dim3(64), 0, s, Mc, rows,
                       cols, in, out, count);
}

// d_msm (dist-primitives/src/dmsm/mod.rs:70-98): local MSM of the share vectors, gather to the king, who
// interpolates in the exponent (unpackexp, degree2) and sums the l secrets -- one n-term combination with
// the constant scalars v_j = sum_i unpack2[i][j] -- then sends the same point to every party.
// (table != nullptr: the base shares are resident, `bases` is unused)
void DG_FN(d_msm_)(Call& k, const dg16_pss* pp, const dg16_net* net, int sid, const void* bases, const void* scalars,
                   size_t n, unsigned mode, void* out_jac, const void* table, unsigned table_c, unsigned table_stride) {
  const bool mont = mode & 1u;
  using F = GF;
  using Fr = CT::Fr;
  const unsigned np = pp->n;
  Affine<F>* c_share = (Affine<F>*)ws(k.c, 18, sizeof(Affine<F>));
  if (table)
    DG_FN(msm_resident_)(k, table, n, table_c, table_stride, scalars, mont, true, c_share);
  else
    msm_run<F, Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mode, true, c_share);        // dmsm/mod.rs:82
  const bool king = net->party_id(net->self) == 0;
  Affine<F>* shares = king ? (Affine<F>*)ws(k.c, 19, np * sizeof(Affine<F>)) : nullptr;
  if (net->gather_to_king(net->self, sid, c_share, sizeof(Affine<F>), shares, k.s()) != DG16_OK)
    throw StatusError{DG16_ERR_NET, "send_to_king failed"};
  Affine<F>* send = nullptr;
  if (king) {
    send = (Affine<F>*)ws(k.c, 20, np * sizeof(Affine<F>));
    const Fr* v2 = (const Fr*)pp->mats + 6 * pp->n * pp->l;
    matvec_points<F, Fr>(k.s(), v2, 1u, np, shares, send, (size_t)1);
    for (unsigned p = 1; p < np; p++)   // vec![output; n_parties] (dmsm/mod.rs:94)
      DG_HIP(hipMemcpyAsync(send + p, send, sizeof(Affine<F>), hipMemcpyDeviceToDevice, k.s()));
    DG_HIP(hipGetLastError());
  }
  Affine<F>* got = (Affine<F>*)ws(k.c, 21, sizeof(Affine<F>));
  if (net->scatter_from_king(net->self, sid, send, sizeof(Affine<F>), got, k.s()) != DG16_OK)
    throw StatusError{DG16_ERR_NET, "recv_from_king failed"};
  hipLaunchKernelGGL(affine_to_jacobian_kernel<F>, dim3(1), dim3(1), 0, k.s(), got, (Jacobian<F>*)out_jac);
  DG_HIP(hipGetLastError());
}
// The clear-text group arithmetic around the d_msm calls of prove::A / B / C::compute (groth16/src/prove.rs:36-44,
// 76-83, 127-134): out = sum_i (mask bit i ? k_i * P_i : P_i) over <= 8 Jacobian terms.  One workgroup, one WAVE per
// term: the scalar multiples (N * r, K * s, A * s, M * r, h * r) are wave-cooperative double-and-add chains side by
// side, then wave 0 adds the terms in the order they were given.
template <class F, class Fr>
__global__ void __launch_bounds__(512) mpc_combine_kernel(const Jacobian<F>* __restrict__ terms,
                                                           const Fr* __restrict__ scalars, unsigned n_terms,
                                                           unsigned mask, int mont, Jacobian<F>* __restrict__ out) {
  __shared__ XYZZ29<F> part[8];
  __shared__ ScalarMulLds<F> lds[8];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  XYZZ29<F> v = XYZZ29<F>::from_xyzz32(XYZZ<F>::from_jacobian(terms[wave]));
  {
    // (every wave runs the chain -- its workgroup barriers must be reached by all -- an unmasked term with the scalar 1)
    Fr kk = Fr::zero();
    kk.l[0] = 1;
    if ((mask >> wave) & 1) {
      kk = scalars[wave];
      if (mont) kk = kk.from_mont();
    }
    // round 6: width-4 NAF chains on the reduced-radix types (254 doublings + ~51 additions at ~1.3 us a level instead of
    // 254 + ~127 on the 32-bit-limb wave operations at ~4 us); no endomorphism split: the terms are arbitrary points
    v = scalar_mul_wave29<F, Fr::NL, false>(v, kk.l, &lds[wave]);
  }
  if (lane == 0) part[wave] = v;
  __syncthreads();
  if (wave != 0) return;
  const XYZZ29<F> acc = sum_points_wave<F>(part, n_terms);
  if (lane == 0) *out = acc.to_xyzz32().to_jacobian();
}
// terms: n_terms Jacobian points; scalars: n_terms Fr (only those under `mask` are read); all device pointers
void DG_FN(mpc_combine_)(Call& k, const void* terms, const void* scalars, unsigned n_terms, unsigned mask, bool mont,
                         void* out_jac) {
  DG_REQUIRE(n_terms >= 1 && n_terms <= 8, DG16_ERR_BAD_ARG, "mpc_combine: 1..8 terms");
  hipLaunchKernelGGL((mpc_combine_kernel<GF, CT::Fr>), dim3(1), dim3(64 * n_terms), 0, k.s(), (const Jacobian<GF>*)terms,
                     (const CT::Fr*)scalars, n_terms, mask, (int)mont, (Jacobian<GF>*)out_jac);
  DG_HIP(hipGetLastError());
}
void DG_FN(affine_to_jac_)(Call& k, const void* aff, void* jac) {
  hipLaunchKernelGGL(affine_to_jacobian_kernel<GF>, dim3(1), dim3(1), 0, k.s(), (const Affine<GF>*)aff, (Jacobian<GF>*)jac);
  DG_HIP(hipGetLastError());
}
// packexp_from_public / unpackexp (dmsm/mod.rs:7-68): the constant n x l matrices applied to group elements
void DG_FN(packexp_)(Call& k, const dg16_pss* pp, int which, const void* in, size_t count, void* out) {
  using F = GF;
  using Fr = CT::Fr;
  const unsigned cols = which == 0 ? pp->l : pp->n, rows = which == 0 ? pp->n : pp->l;
  const Fr* Mc = (const Fr*)pp->mats + 3 * pp->n * pp->l + (size_t)which * pp->n * pp->l;
  matvec_points<F, Fr>(k.s(), Mc, rows, cols, (const Affine<F>*)in, (Affine<F>*)out, count);
  DG_HIP(hipGetLastError());
}
}  // namespace dg16
