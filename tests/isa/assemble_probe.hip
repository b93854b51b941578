// prover_assemble_kernel alone (the proof assembly behind the gather: per-slot sums of the shards' records on the
// reduced-radix wave-cooperative operations, products behind a call) -- seconds of hipcc, so that
// tests/test_kernel_emulation.py runs the shipped kernel body on the Workgroup emulator in the default suite.
#include "msm_impl.h"
namespace dg16 {
DG16_MSM_EXTERN(CurveTypes<DG_CURVE>)
}
#include "prover_impl.h"
namespace dg16 {
using AQ = CurveTypes<DG_CURVE>::Fq;
using AQ2 = CurveTypes<DG_CURVE>::Fq2;
template __global__ void prover_assemble_kernel<AQ, AQ2>(const uint8_t*, size_t, size_t, Jacobian<AQ>*, Jacobian<AQ2>*,
                                                         Jacobian<AQ>*);
}  // namespace dg16
