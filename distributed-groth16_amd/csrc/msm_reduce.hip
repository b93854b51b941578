// Bucket reduction (phase B of the MSM) for ONE (curve, group), selected at compile time like msm_group.hip:
//   hipcc -DDG_CURVE=<0|1|2> -DDG_GROUP=<1|2> -DDG29_OUTLINE_MUL -c msm_reduce.hip -o msm_red_<curve>_g<k>.o
// DG29_OUTLINE_MUL (fp29.h) makes the field products of these latency-bound kernels calls instead of inline code.
#include "msm_reduce_impl.h"

namespace dg16 {
using CT = CurveTypes<DG_CURVE>;
#if DG_GROUP == 1
using GF = CT::Fq;
#else
using GF = CT::Fq2;
#endif
extern template void msm_finalize_phase<GF>(hipStream_t, const MsmSort&, const MsmBuffers<GF>&);   // msm_group.hip
extern template void msm_tail_phase<GF>(hipStream_t, const MsmSort&, const MsmBuffers<GF>&, bool, void*);
template void msm_bucket_phase<GF>(hipStream_t, const MsmSort&, const MsmBuffers<GF>&, bool, void*);
}  // namespace dg16
