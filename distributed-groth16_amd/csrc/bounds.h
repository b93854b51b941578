// Bounds-checked build of the MSM's sort -> accumulate -> finalize -> giant hand-off (`make XFLAGS=-DDG16_BOUNDS`, or
// tools/build_variant.sh bounds "-DDG16_BOUNDS").  Every index these kernels DERIVE FROM DEVICE DATA (the sort's counts /
// offsets / segment offsets, the giant work list, entry references, LDS column partners) goes through DG_IDX(site, index,
// limit): in the checked build an index >= limit is recorded -- (site, index, limit, workgroup, lane) of the FIRST violation
// and a count, in a small device buffer -- and replaced by 0, so that the kernel does not fault and the host can name the
// access (dg16_sync copies the record back and returns DG16_ERR_HIP with it in dg16_last_error).  In the product build DG_IDX is
// the index itself: no instruction is added.  Written for DESIGN.md section 7.2 (an abort seen on some boxes of the pool
// behind msm_finalize_lds_kernel); MSM_INVARIANTS.md states the capacities the sites check.
//
// Sites (the number the record carries):
//   1  msm_segment            bucket slot of a segment                       < bw << log_nb
//   2  msm_segment            last entry of the segment inside its region    <= region
//   3  accumulation kernels   entry reference (table row * n + point)        < rows * n
//   4  accumulation kernels   segment-sum slot written                       < seg_cap
//   5  accumulation kernels   bucket written                                 < (instances * bw) << log_nb
//   6  finalize / stitch      the sort's bucket slot read (counts, seg_off)  < bw << log_nb
//   7  finalize / stitch      segment-sum slot read (msm_part_slot)          < seg_cap
//   8  finalize / stitch      giant id slot written                          < giant_cap
//   9  finalize / stitch      giant work item written                        < 2 * giant_cap
//  10  msm_finalize_lds       LDS column of the tree partner (lane + d)      < BLOCK
//  11  msm_giant(_fold)       giant id read from a work item                 < giant_cap (and the id itself < buckets)
//  12  msm_giant(_fold)       segment-sum slot read / written                < seg_cap
//  13  msm_part_place         entry position written                         < bw * region
//  14  msm_part_scatter       (ref, slot) pair position written              < W * n
//  15  msm_scatter (direct)   entry position written                         < bw * region
//  16  wg_bucket_tree         LDS column of the tree partner (a + d)         < BLOCK
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dg16 {

#ifdef DG16_BOUNDS
constexpr int kBoundsWords = 8;   // [0] violations, [1] site, [2..3] index, [4..5] limit, [6] blockIdx.x, [7] threadIdx.x
unsigned* bounds_sink_device();   // capi.hip: kBoundsWords zeroed device words, one buffer per process
static __device__ unsigned* dg_bounds_sink_dev;          // one copy per translation unit, bound by bounds_bind()
static void bounds_bind() {
  static bool done = false;                               // (static function: one flag per translation unit too)
  if (done) return;
  unsigned* p = bounds_sink_device();
  if (hipMemcpyToSymbol(HIP_SYMBOL(dg_bounds_sink_dev), &p, sizeof p) == hipSuccess) done = true;
}
__device__ __forceinline__ bool dg_in_bounds(unsigned site, size_t idx, size_t limit) {
  if (idx < limit) return true;
  unsigned* s = dg_bounds_sink_dev;
  if (s && atomicAdd(&s[0], 1u) == 0) {
    s[1] = site;
    s[2] = (unsigned)idx; s[3] = (unsigned)((uint64_t)idx >> 32);
    s[4] = (unsigned)limit; s[5] = (unsigned)((uint64_t)limit >> 32);
    s[6] = blockIdx.x; s[7] = threadIdx.x;
    __threadfence();
  }
  return false;
}
// DG_IDX: the index, or 0 after recording a violation;  DG_OK: whether the index is in range (recording if not)
#define DG_IDX(site, idx, limit) (::dg16::dg_in_bounds((site), (size_t)(idx), (size_t)(limit)) ? (idx) : 0)
#define DG_OK(site, idx, limit) (::dg16::dg_in_bounds((site), (size_t)(idx), (size_t)(limit)))
#define DG_BOUNDS_BIND() ::dg16::bounds_bind()
#else
#define DG_IDX(site, idx, limit) (idx)
#define DG_OK(site, idx, limit) (true)
#define DG_BOUNDS_BIND() ((void)0)
#endif

}  // namespace dg16
