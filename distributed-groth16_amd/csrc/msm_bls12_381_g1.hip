// MSM / base generation / affine conversion instantiated for curve id 1 (bls12_381), G1.
#include "msm_impl.h"
#include "dist_impl.h"

namespace dg16 {
using CT = CurveTypes<1>;
void msm_bls12_381_g1(Call& k, const void* bases, const void* scalars, size_t n, bool mont, bool affine, void* out) {
  msm_run<CT::Fq, CT::Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mont, affine, out);
}
void gen_bases_bls12_381_g1(Call& k, uint64_t seed, size_t n, void* out) { gen_bases_run<CT::Fq, CT::G1c>(k, seed, n, out); }
void to_affine_bls12_381_g1(Call& k, const void* jac, void* out, size_t n) { to_affine_run<CT::Fq>(k, jac, out, n); }

// d_msm (dist-primitives/src/dmsm/mod.rs:70-98): local MSM of the share vectors, gather to the king, who
// interpolates in the exponent (unpackexp, degree2) and sums the l secrets -- one n-term combination with
// the constant scalars v_j = sum_i unpack2[i][j] -- then sends the same point to every party.
void d_msm_bls12_381_g1(Call& k, const dg16_pss* pp, const dg16_net* net, int sid, const void* bases,
                          const void* scalars, size_t n, bool mont, void* out_jac) {
  using F = CT::Fq;
  using Fr = CT::Fr;
  const unsigned np = pp->n;
  Affine<F>* c_share = (Affine<F>*)ws(k.c, 18, sizeof(Affine<F>));
  msm_run<F, Fr, CT::SCALAR_BITS>(k, bases, scalars, n, mont, true, c_share);          // dmsm/mod.rs:82
  const bool king = net->party_id(net->self) == 0;
  Affine<F>* shares = king ? (Affine<F>*)ws(k.c, 19, np * sizeof(Affine<F>)) : nullptr;
  if (net->gather_to_king(net->self, sid, c_share, sizeof(Affine<F>), shares, k.s()) != DG16_OK)
    throw StatusError{DG16_ERR_NET, "send_to_king failed"};
  Affine<F>* send = nullptr;
  if (king) {
    send = (Affine<F>*)ws(k.c, 20, np * sizeof(Affine<F>));
    const Fr* v2 = (const Fr*)pp->mats + 6 * pp->n * pp->l;
    hipLaunchKernelGGL((matvec_points_kernel<F, Fr>), dim3(1), dim3(64), 0, k.s(), v2, 1u, np, shares, send, (size_t)1);
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
void packexp_bls12_381_g1(Call& k, const dg16_pss* pp, int which, const void* in, size_t count, void* out) {
  using F = CT::Fq;
  using Fr = CT::Fr;
  const unsigned cols = which == 0 ? pp->l : pp->n, rows = which == 0 ? pp->n : pp->l;
  const Fr* Mc = (const Fr*)pp->mats + 3 * pp->n * pp->l + (size_t)which * pp->n * pp->l;
  hipLaunchKernelGGL((matvec_points_kernel<F, Fr>), dim3((unsigned)((count * rows + 63) / 64)), dim3(64), 0, k.s(), Mc,
                     rows, cols, (const Affine<F>*)in, (Affine<F>*)out, count);
  DG_HIP(hipGetLastError());
}
}  // namespace dg16
