// The second form of the G2 bucket accumulation of a 14-limb curve: msm_accumulate_lds_kernel with the field products OUT
// OF LINE (this unit is compiled with -DDG29_OUTLINE_MUL; the kernel carries TU = 1 so that its symbol differs from the
// inline form's in msm_group.hip).  msm_accumulate_phase (msm_impl.h) times both on the device once and keeps the faster.
//   hipcc -DDG_CURVE=<1|2> -DDG29_OUTLINE_MUL -c msm_group_outl.hip -o msm_outl_<curve>_g2.o
#ifndef DG29_OUTLINE_MUL
#error "msm_group_outl.hip is the out-of-line form: compile it with -DDG29_OUTLINE_MUL"
#endif
#include "msm_impl.h"

namespace dg16 {
using CT = CurveTypes<DG_CURVE>;
using GF = CT::Fq2;
static_assert(sizeof(GF) > 64, "only the 14-limb curves have a second form");

template <>
void msm_accumulate_lds_outl<GF>(hipStream_t s, dim3 grid, MsmBases mb, size_t n, MsmGeom g, const unsigned* offsets,
                                 const unsigned* counts, const unsigned* seg_off, const unsigned* seg_total,
                                 const unsigned* entries, XYZZ29<GF>* seg_sum, XYZZ29<GF>* buckets) {
  constexpr int BLOCK = 1 << msm_acc_block_log<GF>();
  hipLaunchKernelGGL((msm_accumulate_lds_kernel<GF, BLOCK, 1>), grid, dim3(BLOCK), 0, s, mb, n, g, offsets, counts, seg_off,
                     seg_total, entries, seg_sum, buckets);
}
}  // namespace dg16
