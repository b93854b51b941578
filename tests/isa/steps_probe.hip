// msm_accumulate_steps_kernel alone (the 14-limb G2 bucket accumulation: one Fq2-product site visited per step, its
// temporaries in accumulation registers) -- four seconds of hipcc instead of the minutes msm_group.hip takes for a G2, so
// that tests/test_kernel_emulation.py runs the shipped kernel body on the CPU in the default suite.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDG_CURVE=<1|2> -I distributed-groth16_amd/csrc --cuda-device-only -S ...
#include "msm_impl.h"
namespace dg16 {
using GF = CurveTypes<DG_CURVE>::Fq2;
template __global__ void msm_accumulate_steps_kernel<GF, 128>(MsmBases, size_t, MsmGeom, const unsigned*, const unsigned*,
                                                              const unsigned*, const unsigned*, const unsigned*,
                                                              XYZZ29<GF>*, XYZZ29<GF>*, unsigned long long*);
}  // namespace dg16
