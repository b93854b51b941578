// msm_finalize_lds_kernel of a 14-limb G2 alone (the throughput finalize: two lanes per bucket, XYZZ29::add_into on LDS
// columns) -- under a minute of hipcc instead of the minutes msm_group.hip takes for a G2, so that
// tests/test_kernel_emulation.py runs the shipped kernel body on the Workgroup emulator in the default suite.
#include "msm_impl.h"
namespace dg16 {
using GF = CurveTypes<DG_CURVE>::Fq2;
template __global__ void msm_finalize_lds_kernel<GF, 128>(MsmGeom, size_t, unsigned, unsigned, const unsigned*,
                                                          const unsigned*, const XYZZ29<GF>*, XYZZ29<GF>*, unsigned*,
                                                          unsigned*, unsigned);
}  // namespace dg16
