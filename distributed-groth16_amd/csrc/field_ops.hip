// Element-wise field kernels (parity probe for the Montgomery arithmetic; also the pointwise
// passes of the h-polynomial).  HBM-bound for add/sub, VALU-bound for mul.
#include "ctx.h"
#include "types.h"

namespace dg16 {

template <class F>
__global__ void __launch_bounds__(256) field_op_kernel(int op, const F* __restrict__ a,
                                                        const F* __restrict__ b, F* __restrict__ out,
                                                        size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    F x = a[i];
    F r;
    switch (op) {
      case DG16_OP_ADD: r = x + b[i]; break;
      case DG16_OP_SUB: r = x - b[i]; break;
      case DG16_OP_MUL: r = x * b[i]; break;
      case DG16_OP_SQR: r = x.sqr(); break;
      case DG16_OP_INV: r = x.inv(); break;
      case DG16_OP_TO_MONT: r = x.to_mont(); break;
      case DG16_OP_FROM_MONT: r = x.from_mont(); break;
      default: r = x.neg(); break;
    }
    out[i] = r;
  }
}

template <class F>
static void launch(Call& k, int op, const void* a, const void* b, void* out, size_t n) {
  size_t blocks = (n + 255) / 256;
  size_t cap = (size_t)k.ctx->compute_units * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(field_op_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, k.s(), op,
                     (const F*)a, (const F*)b, (F*)out, n);
  DG_HIP(hipGetLastError());
}

void field_op_launch(Call& k, int field_id, int op, const void* a, const void* b, void* out, size_t n) {
  switch (field_id) {
    case 0: launch<bn254_fq>(k, op, a, b, out, n); break;
    case 1: launch<bls12_381_fq>(k, op, a, b, out, n); break;
    case 2: launch<bls12_377_fq>(k, op, a, b, out, n); break;
    case 16: launch<bn254_fr>(k, op, a, b, out, n); break;
    case 17: launch<bls12_381_fr>(k, op, a, b, out, n); break;
    case 18: launch<bls12_377_fr>(k, op, a, b, out, n); break;
    default: throw StatusError{DG16_ERR_BAD_CURVE, "unknown field id"};
  }
}

}  // namespace dg16
