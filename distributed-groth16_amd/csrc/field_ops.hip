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

// ---- QAP evaluation vectors (groth16/src/qap.rs:44-91 == ark-circom/src/circom/qap.rs:34-62) --------------
// a[i] = <A_i, w>, b[i] = <B_i, w> over the CSR rows (evaluate_constraint), a[nc + j] = w[j] for the
// num_inputs instance variables, c = a o b on the constraint rows, zero above.  One lane per domain slot;
// HBM-bound gather: ~3 nonzeros x (32 B coefficient + 32 B witness element) per row.
template <class Fr>
__global__ void __launch_bounds__(256) qap_kernel(const unsigned* __restrict__ a_ptr, const unsigned* __restrict__ a_col,
                                                   const Fr* __restrict__ a_val, const unsigned* __restrict__ b_ptr,
                                                   const unsigned* __restrict__ b_col, const Fr* __restrict__ b_val,
                                                   const Fr* __restrict__ w, int w_mont, size_t nc, size_t ni, size_t nv,
                                                   size_t m, size_t row_start, size_t row_stride, Fr* __restrict__ a,
                                                   Fr* __restrict__ b, Fr* __restrict__ c,
                                                   unsigned* __restrict__ err_flag) {
  // output slot `slot` holds domain row i = row_start + row_stride * slot (row_stride = 1: the whole vectors)
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = row_start + row_stride * slot;
  if (i >= m) return;
  Fr av = Fr::zero(), bv = Fr::zero(), cv = Fr::zero();
  if (i < nc) {
    // Column indices from a key file are untrusted: an entry whose column is not a wire is skipped and reported
    // through the context's sticky flag (dg16_sync -> DG16_ERR_BAD_ARG); w[] is never read out of range.  The ROW
    // POINTERS of the device-pointer entry points are trusted (the kernel does not know the length of the column /
    // coefficient arrays: a row_ptr beyond them reads past their end) -- the host-pointer path and the file readers
    // validate row_ptr against nnz before anything reaches the device (capi.hip, formats.hip)
    bool bad = false;
    const unsigned a_lo = a_ptr[i], a_hi = a_ptr[i + 1], b_lo = b_ptr[i], b_hi = b_ptr[i + 1];
    if (a_hi < a_lo || b_hi < b_lo || a_hi - a_lo > nv || b_hi - b_lo > nv) bad = true;
    for (unsigned j = a_lo; !bad && j < a_hi; j++) {
      const unsigned col = a_col[j];
      if (col >= nv) { bad = true; break; }
      Fr x = w[col];
      if (!w_mont) x = x.to_mont();
      av = av + a_val[j] * x;
    }
    for (unsigned j = b_lo; !bad && j < b_hi; j++) {
      const unsigned col = b_col[j];
      if (col >= nv) { bad = true; break; }
      Fr x = w[col];
      if (!w_mont) x = x.to_mont();
      bv = bv + b_val[j] * x;
    }
    if (bad) {
      *(volatile unsigned*)err_flag = 1u;   // plain store: the word lives in pinned host memory
      av = Fr::zero();
      bv = Fr::zero();
    }
    cv = av * bv;
  } else if (i < nc + ni) {
    av = w[i - nc];
    if (!w_mont) av = av.to_mont();
  }
  a[slot] = av;
  b[slot] = bv;
  c[slot] = cv;
}

void qap_launch(Call& k, int curve, const unsigned* a_ptr, const unsigned* a_col, const void* a_val,
                const unsigned* b_ptr, const unsigned* b_col, const void* b_val, const void* w, bool w_mont, size_t nc,
                size_t ni, size_t nv, size_t m, size_t row_start, size_t row_stride, void* a, void* b, void* c) {
  unsigned blocks = (unsigned)((m / row_stride + 255) / 256);
#define QAP(F)                                                                                              \
  hipLaunchKernelGGL(qap_kernel<F>, dim3(blocks), dim3(256), 0, k.s(), a_ptr, a_col, (const F*)a_val, b_ptr, \
                     b_col, (const F*)b_val, (const F*)w, (int)w_mont, nc, ni, nv, m, row_start, row_stride, (F*)a, (F*)b, (F*)c, \
                     k.ctx->dev_flag)
  switch (curve) {
    case 0: QAP(bn254_fr); break;
    case 1: QAP(bls12_381_fr); break;
    default: QAP(bls12_377_fr); break;
  }
#undef QAP
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
