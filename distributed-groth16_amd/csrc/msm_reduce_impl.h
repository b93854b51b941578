// Bucket reduction of the MSM (phase B): sum_b (b + 1) B_b from the segment partials of phase A, and the Horner tail.
// Included by msm_reduce.hip only, which is compiled once per (curve, group) with DG29_OUTLINE_MUL: these kernels are
// chains of a few dozen dependent group operations at ~1 wave per SIMD, so they want SMALL code (the instruction cache
// is 64 KB per two CUs; with every field product inlined a G1 addition was 45 KB and a BLS12-381 G2 addition 250 KB,
// and the kernels were instruction-fetch-bound: 55 us per dependent G1 addition against ~8 us of issue time).
#pragma once
#include "msm_impl.h"

// Priority of the reduction kernels of this file.  Rounds 3-5 ran them at s_setprio 3 (a latency chain ahead of the
// accumulation waves it shares a SIMD with); with the chains as short as round 6 left them the default priority measures
// the same or better everywhere -- a queued 2^20 proof 10.11-10.13 -> 10.00-10.10 ms, one at a time 10.22-10.32 -> 10.08-10.14,
// config 4 queued 1.60 -> 1.58 ms (profiles/r6ii_reduce_prio_ab.txt, same call, three times)
#ifndef DG16_REDUCE_PRIO
#define DG16_REDUCE_PRIO 0
#endif
namespace dg16 {

// Giant buckets (a boolean witness puts half of ALL entries into bucket 0; the short top window of a c that does
// not divide the scalar width does the same): two launches.  Stage 1 cuts the bucket's segment partials into
// <= kGiantSlices slices, one workgroup each, and leaves every slice's sum IN PLACE in the slice's first
// segment slot; stage 2 adds the slice sums.  (One workgroup per bucket chained 128 dependent additions per
// lane for a 2^20-bit witness: 2.5 ms for G1, far more for G2.)
// (kGiantSlices, kGiantSliceSegs, giant_geometry: msm_impl.h -- the throughput finalize there registers giants too)

// ---- 4b / 5: bucket reduction --------------------------------------------------------------------------------
// sum_b (b + 1) B_b over the 2^(c-1) buckets of a bucket-window, B_b = sum of the bucket's segment partials.
// Every phase is a short chain of dependent group operations (a lone lane needs ~5 us per G1 addition, ~15 us per
// G2 addition), so the work is arranged for the shortest chains, not the fewest additions:
//   finalize    one lane per bucket (msm_impl.h: msm_finalize_thr_kernel)
//   giants      buckets with > kGiantSegs partials (boolean witnesses, short top windows): device-side work list,
//               one workgroup per slice, then one fold per giant (unchanged idea, see giant_geometry)
//   row         one workgroup per ROW of 256 buckets: suffix scan S_c = sum_{c' >= c} B_c' (8 steps) gives the row
//               sum R = S_0, and the tree sum of the S_c (8 steps) gives W = sum_c (c + 1) B_c
//   top         one workgroup per bucket-window: sum_r W_r (tree) and sum_r r R_r (serial running sums over the
//               rows a lane owns, suffix scan + tree across lanes) on the two halves of the workgroup at once, then
//               total = sum W + 256 * sum r R: 8 doublings and one addition
// ~50 dependent operations per MSM instead of ~150 (finalize: ~16, 8-bucket chunks + 16-bit scalar multiple: ~40,
// two 256-wide sums: ~70, each in its own launch at < 1 wave per SIMD).
// The reduction kernels below are chains of a few dozen dependent group operations on operands that live in memory
// (LDS / global).  Each kernel is written as ONE loop over a step schedule with exactly one addition site and one
// doubling site (XYZZ29::add_mem / dbl_mem, inlined): an addition is ~4 000 (G1) to ~30 000 (BLS12-381 G2)
// instructions, so a copy per call site made the library take 10 minutes to build, and an out-of-line copy behind
// a call pays ~300 callee-saved register spills per call (and faulted for the largest type).
template <class T>
__device__ __forceinline__ T lane_xor_words(const T& v, int mask) {
  static_assert(sizeof(T) % 4 == 0, "word-sized type");
  T r;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&v);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) dst[i] = (uint32_t)__shfl_xor((int)src[i], mask);
  return r;
}

template <class F>
__global__ void __launch_bounds__(256) msm_giant_kernel(MsmGeom g, unsigned wg_log, const unsigned* __restrict__ counts,
                                                         const unsigned* __restrict__ seg_off,
                                                         XYZZ29<F>* __restrict__ seg_sum,
                                                         const unsigned* __restrict__ giant_count,
                                                         const unsigned* __restrict__ giant_list, unsigned giant_cap) {
  __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  __shared__ XYZZ29<F> sh[256];
  unsigned nwork = giant_count[1];
  if (nwork > 2 * giant_cap) nwork = 2 * giant_cap;          // (never: msm_register_giant)
  const unsigned* work = giant_list + giant_cap;
  for (unsigned wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const unsigned item = work[wi];
    const unsigned gid = giant_list[DG_IDX(11, item >> 6, giant_cap)], slice = item & 63;
    const unsigned wy = gid >> g.log_nb;                                   // instance * bw + bucket-window
    const size_t gs = ((size_t)(wy % g.bw) << g.log_nb) + (gid & ((1u << g.log_nb) - 1));   // the sort's bucket slot
    const unsigned k = (counts[gs] + (1u << g.seg_log) - 1) >> g.seg_log, first = seg_off[gs];
    const unsigned nseg = msm_nparts(first, k, wg_log);      // partials = one per accumulation workgroup the bucket spans
    unsigned slices, per;
    giant_geometry(nseg, slices, per);
    XYZZ29<F>* sp = seg_sum + (size_t)wy * g.seg_cap;
    const unsigned lo = slice * per;
    const unsigned hi = lo + per < nseg ? lo + per : nseg;
    sh[threadIdx.x] = XYZZ29<F>::inf();
    __syncthreads();
    // steps 0 .. iters-1: strided partials of the slice; then 8 tree steps
    const unsigned iters = (hi - lo + 255) >> 8;
#pragma unroll 1
    for (unsigned step = 0; step < iters + 8; step++) {
      const XYZZ29<F>* b;
      bool on;
      if (step < iters) {
        const unsigned s = lo + step * 256 + threadIdx.x;
        on = s < hi;
        b = &sp[DG_IDX(12, msm_part_slot(first, on ? s : lo, wg_log), g.seg_cap)];
      } else {
        const unsigned stride = 128u >> (step - iters);
        on = threadIdx.x < stride;
        b = &sh[(threadIdx.x + stride) & 255];
      }
      if (on) XYZZ29<F>::add_mem(&sh[threadIdx.x], &sh[threadIdx.x], b);
      __syncthreads();
    }
    if (threadIdx.x == 0) sp[DG_IDX(12, msm_part_slot(first, lo, wg_log), g.seg_cap)] = sh[0];
    __syncthreads();
  }
}

template <class F>
__global__ void __launch_bounds__(64) msm_giant_fold_kernel(MsmGeom g, unsigned wg_log, const unsigned* __restrict__ counts,
                                                             const unsigned* __restrict__ seg_off,
                                                             const XYZZ29<F>* __restrict__ seg_sum,
                                                             XYZZ29<F>* __restrict__ buckets,
                                                             const unsigned* __restrict__ giant_count,
                                                             const unsigned* __restrict__ giant_list,
                                                             unsigned giant_cap) {
  __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  __shared__ XYZZ29<F> sh[kGiantSlices];
  unsigned ng = *giant_count;
  if (ng > giant_cap) ng = giant_cap;
  for (unsigned gi = blockIdx.x; gi < ng; gi += gridDim.x) {
    const unsigned gid = giant_list[gi];
    const unsigned wy = gid >> g.log_nb;
    const size_t gs = ((size_t)(wy % g.bw) << g.log_nb) + (gid & ((1u << g.log_nb) - 1));
    const unsigned k = (counts[gs] + (1u << g.seg_log) - 1) >> g.seg_log, first = seg_off[gs];
    const unsigned nseg = msm_nparts(first, k, wg_log);
    unsigned slices, per;
    giant_geometry(nseg, slices, per);
    const XYZZ29<F>* sp = seg_sum + (size_t)wy * g.seg_cap;
    sh[threadIdx.x] = threadIdx.x < slices ? sp[DG_IDX(12, msm_part_slot(first, threadIdx.x * per, wg_log), g.seg_cap)] : XYZZ29<F>::inf();
    __syncthreads();
#pragma unroll 1
    for (unsigned stride = kGiantSlices / 2; stride > 0; stride >>= 1) {
      if (threadIdx.x < stride) XYZZ29<F>::add_mem(&sh[threadIdx.x], &sh[threadIdx.x], &sh[threadIdx.x + stride]);
      __syncthreads();
    }
    if (threadIdx.x == 0) buckets[gid] = sh[0];
    __syncthreads();
  }
}


// A narrow tree step, sh[base + i] += sh[base + i + d] for i < d <= 64, on ROWS OF 16 LANES: a lone lane takes ~10 us (G1) /
// ~35 us (G2) per dependent addition whatever the number of idle lanes around it; add_wave29 (msm_impl.h) spreads the four
// product levels of ONE addition over the 16 lanes of a row -- its operands only have to be uniform within the row: the
// broadcasts are row_newbcast, the Fq2 product's shuffles stay inside a quad -- ~3 us (G1) / ~8 us (G2, products behind
// calls in this unit).  Sixteen rows of a 256-lane half take additions i = row, row + 16, ..: steps of <= 16 additions
// are one pass.  Round 5: the G2 bucket reduction is the critical path of a small proof (BASELINE config 4: row 0.61 +
// top 0.63 ms of a 2.0-ms proof, sixteen and seventeen dependent steps) and of a shard's B chain.
// No barrier inside: the row that owns i is the only reader of sh[i] and sh[i + d] and the only writer of sh[i].
constexpr unsigned kCoopTreeMax = 64;
template <class F>
__device__ __forceinline__ void tree_step_coop(XYZZ29<F>* sh, unsigned base, unsigned d) {
  const unsigned row16 = (threadIdx.x >> 4) & 15u;      // row inside this 256-lane half
#pragma unroll 1
  for (unsigned i = row16; i < d; i += 16) {
    const XYZZ29<F> v = add_wave29(sh[base + i], sh[base + i + d]);
    if ((threadIdx.x & 15u) == 0) sh[base + i] = v;
  }
}

template <class F>
__global__ void __launch_bounds__(256) msm_row_kernel(MsmGeom g, RowGeom rg, const XYZZ29<F>* __restrict__ buckets,
                                                       XYZZ29<F>* __restrict__ row_w, XYZZ29<F>* __restrict__ row_r) {
  __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  __shared__ XYZZ29<F> sh[256];
  const unsigned c = threadIdx.x, row = 1u << rg.row_log;
  const size_t rid = ((size_t)blockIdx.y << rg.rows_log) + blockIdx.x;      // (bucket-window, row)
  sh[c] = c < row ? buckets[(rid << rg.row_log) + c] : XYZZ29<F>::inf();
  __syncthreads();
  // steps 0 .. row_log-1: inclusive suffix scan sh[c] <- sum_{c' >= c} B_c' (then sh[0] = R);
  // steps row_log .. 2 row_log - 1: tree sum of the suffix sums = sum_c (c + 1) B_c
#pragma unroll 1
  for (unsigned step = 0; step < 2 * rg.row_log; step++) {
    const bool scan = step < rg.row_log;
    const unsigned d = scan ? 1u << step : row >> (step - rg.row_log + 1);
    const bool on = scan ? c + d < row : c < d;
    if (step == rg.row_log && c == 0) row_r[rid] = sh[0];
    if (!scan && d <= kCoopTreeMax) {          // a NARROW tree step: the additions wave-cooperatively (tree_step_coop)
      __syncthreads();                         // (row_r's read of sh[0] above, the previous step's writes)
      tree_step_coop<F>(sh, 0u, d);
      __syncthreads();
      continue;
    }
    XYZZ29<F> v;
    if (on) XYZZ29<F>::add_mem(&v, &sh[c], &sh[(c + d) & 255]);
    __syncthreads();
    if (on) sh[c] = v;
    __syncthreads();
  }
  if (c == 0) {
    if (rg.row_log == 0) row_r[rid] = sh[0];
    row_w[rid] = sh[0];
  }
}

// Many rows (the plain MSM: one bucket set PER WINDOW, 2048 rows of 256 buckets at 2^20 points): msm_row_kernel's
// suffix scan does 8 x 256 additions per row on lanes that are mostly switched off -- 3.8 M full additions per G1 MSM, a
// third of the accumulation's work, 0.76 ms (G2: 3.3 ms of a 12-ms MSM; profiles/r4b_msm_g*_2e20_*).  Here a lane OWNS
// 2^k consecutive buckets and runs the work-efficient serial form over them (run += B_i from the top, acc += run:
// r_j = sum_i B_i, w_j = sum_i (i + 1) B_i; 2 additions per bucket), and the workgroup (LANES lanes = LANES 2^k buckets)
// finishes with ONE suffix scan over the r_j and two trees side by side on the two halves of the lanes:
//   bucket b = (sr LANES + j) 2^k + i:  sum (b + 1) B_b = sum_sr [ W1_sr + 2^k ( T_sr + LANES sr R_sr ) ],
//   W1 = sum_j w_j,  R = sum_j r_j,  T = sum_j j r_j = sum_{u >= 1} suffix_u(r)
// written as (W, run, local) = (W1, R, T) entries of `fold` for the top kernel's folded mode (per = LANES, final
// doublings k).  2 x 2^k + 2 log2 LANES dependent steps on 1/2^k of the workgroups.
template <class F, int LANES>
__global__ void __launch_bounds__(LANES) msm_rowchunk_kernel(unsigned log_nb, unsigned k_log,
                                                              const XYZZ29<F>* __restrict__ buckets,
                                                              XYZZ29<F>* __restrict__ fold /* [bw][3][256]: W1, R, T */) {
  __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  constexpr unsigned LL = LANES == 256 ? 8u : 7u;
  static_assert(LANES == 256 || LANES == 128, "workgroup of 256 or 128 lanes");
  __shared__ XYZZ29<F> s_r[LANES];      // run -> r_j -> suffix sums -> tree (T)
  __shared__ XYZZ29<F> s_w[LANES];      // acc -> w_j -> tree (W1)
  const unsigned j = threadIdx.x, K = 1u << k_log;
  const XYZZ29<F>* B = buckets + ((size_t)blockIdx.y << log_nb) + (((size_t)blockIdx.x * LANES + j) << k_log);
  XYZZ29<F>* out = fold + (size_t)blockIdx.y * 3 * 256;
  s_r[j] = XYZZ29<F>::inf();
  s_w[j] = XYZZ29<F>::inf();
  __syncthreads();
  const unsigned s_scan = 2 * K, s_tree = s_scan + LL, n_steps = s_tree + LL;
#pragma unroll 1
  for (unsigned step = 0; step < n_steps; step++) {
    const XYZZ29<F>*a, *b;
    XYZZ29<F>* dst;
    bool on = true;
    if (step < s_scan) {                       // even: run += B_i (i from the top); odd: acc += run
      const bool even = (step & 1) == 0;
      dst = even ? &s_r[j] : &s_w[j];
      a = dst;
      b = even ? &B[K - 1 - (step >> 1)] : &s_r[j];
    } else if (step < s_tree) {                // inclusive suffix scan of the r_j
      const unsigned d = 1u << (step - s_scan);
      on = j + d < (unsigned)LANES;
      dst = &s_r[j];
      a = dst;
      b = &s_r[(j + d) & (LANES - 1)];
    } else {                                   // two trees at once: T on the low half of the lanes, W1 on the high half
      if (step == s_tree) {
        if (j == 0) {
          out[1 * 256 + blockIdx.x] = s_r[0];  // R = suffix_0
          s_r[0] = XYZZ29<F>::inf();           // weight j starts at 0: drop suffix_0
        }
        __syncthreads();
      }
      const unsigned d = (unsigned)LANES >> (step - s_tree + 1);
      const unsigned c = j & (LANES / 2 - 1);
      XYZZ29<F>* arr = j < LANES / 2 ? s_r : s_w;
      on = c < d;
      dst = &arr[c];
      a = dst;
      b = &arr[(c + d) & (LANES - 1)];
    }
    XYZZ29<F> v;
    if (on) XYZZ29<F>::add_mem(&v, a, b);
    __syncthreads();
    if (on) *dst = v;
    __syncthreads();
  }
  if (j == 0) out[2 * 256 + blockIdx.x] = s_r[0];
  if (j == LANES / 2) out[0 * 256 + blockIdx.x] = s_w[0];
}

// Bucket-windows with more than 256 rows (tables with c >= 18): every lane of the top kernel owns `per` consecutive
// rows; their plain sums (W: half 0, R: half 1) and, for R, the lane-local weighted sum sum_j j R_j, are taken here
// (serial running sums: 2 per operations) so that the top kernel always starts from <= 256 entries per half.
template <class F>
__global__ void __launch_bounds__(64) msm_rowfold_kernel(RowGeom rg, const XYZZ29<F>* __restrict__ row_w,
                                                          const XYZZ29<F>* __restrict__ row_r,
                                                          XYZZ29<F>* __restrict__ fold /* [bw][3][256]: W, R, local */) {
  __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  __shared__ XYZZ29<F> st[64][2];      // [lane][run, acc]
  const unsigned gid = blockIdx.x * 64 + threadIdx.x;   // (half, t)
  const unsigned half = gid >> 8, t = gid & 255;
  const unsigned per_log = rg.rows_log - 8, per = 1u << per_log;
  const size_t base = ((size_t)blockIdx.y << rg.rows_log) + (size_t)t * per;
  XYZZ29<F>* run = &st[threadIdx.x][0];
  XYZZ29<F>* acc = &st[threadIdx.x][1];
  *run = XYZZ29<F>::inf();
  *acc = XYZZ29<F>::inf();
#pragma unroll 1
  for (unsigned k = 0; k < 2 * per; k++) {
    const unsigned j = per - 1 - (k >> 1);
    // even k: acc += run (only the weighted half, and not before the first row); odd k: run += row j
    const bool first = (k & 1) == 0;
    const bool on = first ? (half == 1 && j + 1 < per) : true;
    const XYZZ29<F>* b = first ? run : (half == 0 ? row_w : row_r) + base + j;
    if (on) XYZZ29<F>::add_mem(first ? acc : run, first ? acc : run, b);
  }
  XYZZ29<F>* out = fold + (size_t)blockIdx.y * 3 * 256;
  out[half * 256 + t] = *run;
  if (half == 1) out[2 * 256 + t] = *acc;
}

// one workgroup per bucket-window; HALVES = 2: 512 lanes, lanes 0..255 sum_r W_r and lanes 256..511 sum_r r R_r at
// the same time; HALVES = 1 (types whose 512 LDS slots exceed the 160 KiB: BLS12-381 G2): 256 lanes, one pass after
// the other.  window sum = sum W + 2^row_log * sum r R, written in the 32-bit arkworks form the tail / the C ABI read.
//   sum_r r R_r = sum_t local_t + per * sum_t t run_t,   sum_t t run_t = sum_{u >= 1} suffix_u(run)
// What the top kernel sums (host side: top_geometry below).  Plain rows: `lanes` = rows per bucket-window, entries
// (W_r, R_r) from row_w / row_r, total = sum W + 2^final_log sum r R.  Folded (msm_rowfold_kernel: > 256 rows; or
// msm_rowchunk_kernel: lanes that own 2^k buckets each): entries (W_t, run_t, local_t) from `fold`,
// sum_r r R_r = sum_t local_t + 2^per_log sum_t t run_t.
struct TopGeom {
  unsigned folded, per_log, lanes, final_log, rows_log;
};
template <class F, int HALVES>
__global__ void __launch_bounds__(256 * HALVES) msm_top_kernel(TopGeom tg, const XYZZ29<F>* __restrict__ row_w,
                                                                const XYZZ29<F>* __restrict__ row_r,
                                                                const XYZZ29<F>* __restrict__ fold,
                                                                XYZZ29<F>* __restrict__ window_sums) {
  __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  __shared__ XYZZ29<F> sh[256 * HALVES];
  __shared__ XYZZ29<F> keep;                       // HALVES == 1: sum W while the second pass runs
  const unsigned t = threadIdx.x & 255;
  const bool folded = tg.folded != 0;
  const unsigned per_log = tg.per_log;
  const unsigned lanes = tg.lanes;                                // entries per half
  const size_t base = (size_t)blockIdx.x << tg.rows_log;
  const XYZZ29<F>* fb = fold + (size_t)blockIdx.x * 3 * 256;
  XYZZ29<F>* me = &sh[threadIdx.x];
#pragma unroll 1
  for (int pass = 0; pass < 3 - HALVES; pass++) {
    const unsigned half = HALVES == 2 ? threadIdx.x >> 8 : (unsigned)pass;
    *me = t < lanes ? (folded ? fb[half * 256 + t] : (half == 0 ? row_w : row_r)[base + t]) : XYZZ29<F>::inf();
    __syncthreads();
    // schedule: 8 scan steps (weighted half) | drop suffix_0 | per_log doublings | + local (folded) | 8 tree steps;
    // then ONE wave: final_log doublings | + sum W  (total = sum W + 2^final_log * sum r R)
    const unsigned s_tree = 9 + per_log, s_fin = s_tree + 8;
#pragma unroll 1
    for (unsigned step = 0; step < s_fin; step++) {
      XYZZ29<F> v;
      bool on = false, is_dbl = false;
      const XYZZ29<F>* b = me;
      if (step < 8) {
        const unsigned d = 1u << step;
        on = half == 1 && t + d < 256;
        b = me + (on ? d : 0);
      } else if (step < 8 + per_log) {
        on = half == 1;
        is_dbl = true;
      } else if (step == 8 + per_log) {
        on = half == 1 && folded && t < lanes;     // (chunk mode fills only `lanes` entries of the fold array)
        b = fb + 2 * 256 + (on ? t : 0);
      } else {
        const unsigned stride = 128u >> (step - s_tree);
        if (stride <= kCoopTreeMax) {          // narrow: wave-cooperative additions on rows of 16 lanes, both halves at once
          tree_step_coop<F>(sh, (threadIdx.x >> 8) << 8, stride);
          __syncthreads();
          continue;
        }
        on = t < stride;
        b = me + (on ? stride : 0);
      }
      if (step == 8 && half == 1 && t == 0) *me = XYZZ29<F>::inf();   // weight t starts at 0: drop suffix_0
      if (on) {
        if (is_dbl) XYZZ29<F>::dbl_mem(&v, me);
        else XYZZ29<F>::add_mem(&v, me, b);
      }
      __syncthreads();
      if (on) *me = v;
      __syncthreads();
    }
    if (HALVES == 1 && pass == 0) {
      if (threadIdx.x == 0) keep = sh[0];
      __syncthreads();
    }
  }
  // The last final_log + 1 operations are a chain on ONE value: the wave that holds it runs them wave-cooperatively
  // (msm_impl.h: dbl_wave29 / add_wave29 -- ~1.5 us per G1 operation against ~11 us for a lone lane's, G2 3 against 25)
  constexpr unsigned RES = HALVES == 2 ? 256u : 0u;
  if ((threadIdx.x >> 6) == (RES >> 6)) {
    if constexpr (lane29::enabled<F>()) {        // limb-per-lane chain (lane29.h)
      using FO = lane29::Ops<F>;
      typename FO::KT kc;
      kc.init();
      lane29::Pt<FO> a = lane29::load_pt<F>(kc, &sh[RES]);
#pragma unroll 1
      for (unsigned k = 0; k < tg.final_log; k++) a = lane29::dbl_pt<FO>(kc, a);
      a = lane29::add_pt<FO>(kc, a, lane29::load_pt<F>(kc, HALVES == 2 ? &sh[0] : &keep));
      lane29::store_pt<F>(kc, &window_sums[blockIdx.x], a);
    } else {
      XYZZ29<F> acc = sh[RES];
#pragma unroll 1
      for (unsigned k = 0; k < tg.final_log; k++) acc = dbl_wave29(acc);
      acc = add_wave29(acc, HALVES == 2 ? sh[0] : keep);
      if ((threadIdx.x & 63) == 0) window_sums[blockIdx.x] = acc;
    }
  }
}


// ---- small bucket sets: sum_b (b + 1) B_b as a radix-2^RB reduction in LANE form (lane29.h) --------------------------
// The row / top kernels above take 2 log2(nb) dependent additions of ~9 us (G1; a lone lane's) to ~35 us (G2) each: 0.29 ms
// (G1) and 1.0 ms (G2) for the 2 048 buckets of BASELINE config 4, the critical path of a short shard.  Here an addition
// is a WAVE's (lane29::add_pt: ~1.1 us G1, ~2.6 us Fq2) and a workgroup of G = 2^g waves reduces G consecutive entries:
//     entries j = 0 .. G-1 carry (W_j, R_j) and stand u = 2^u_log buckets apart;   R' = sum_j R_j,
//     W' = sum_j W_j + u sum_j j R_j,        sum_j j R_j = sum_{t >= 1} suffix_t(R)
// (level 0: W_j = R_j = B_j, u = 1: W' = sum_j (j + 1) B_j) -- g steps of a suffix scan over the waves' R, g steps of
// two trees side by side (the suffix sums on waves [0, d), the W_j on waves [d, 2 d)), u_log doublings and one addition
// on wave 0: 2 g + 1 additions deep per level, log2(nb) / g levels, one launch each.  Between steps the waves exchange
// points through LDS in raw lane form (one barrier per step, two buffers).
constexpr size_t kLaneParallelMaxEntries = 4096;
template <class F>
constexpr int lane_reduce_radix_log() {
  return (FieldOf<F>::EXT || RR<typename FieldOf<F>::Params>::N > 9) ? 3 : 4;     // 8 waves x 256 registers, or 16 x 128
}
template <class F, int RB>
__global__ void __launch_bounds__(64 << RB) msm_lane_reduce_kernel(const XYZZ29<F>* __restrict__ in,
                                                                    const XYZZ29<F>* __restrict__ in2, unsigned in_per_bw_log,
                                                                    unsigned g_log, unsigned u_log, int mode, unsigned k_log,
                                                                    XYZZ29<F>* __restrict__ out,
                                                                    XYZZ29<F>* __restrict__ window_sums) {
  // mode: where entry t of bucket-window y comes from --
  //   0  a bucket: W = R = in[e]                                   (level 0 of a bucket set)
  //   1  a pair of the level below: W = in[2 e], R = in[2 e + 1]
  //   2  a row of msm_row_kernel: W = in[e] (row_w), R = in2[e] (row_r)
  //   3  a chunk of msm_rowchunk_kernel / msm_rowfold_kernel, fold[y][3][256]: W = fold[0][t] + 2^k_log fold[2][t] (the
  //      chunk's own weighted part), R = fold[1][t]
  // (2, 3: the work of msm_top_kernel -- 16 + per_log lone-lane steps, 0.16 ms behind a 2^20-point G1 MSM -- as two levels)
  const int level0 = mode == 0;
  if constexpr (lane29::enabled<F>()) {
    using FO = lane29::Ops<F>;
    using LPt = lane29::Pt<FO>;
    __shared__ XYZZ29<F> sA[2][1 << RB];     // the scan's R, then the suffix-sum tree
    __shared__ XYZZ29<F> sB[2][1 << RB];     // the W tree
    __builtin_amdgcn_s_setprio(DG16_REDUCE_PRIO);
    const unsigned w = threadIdx.x >> 6, G = 1u << g_log;
    typename FO::KT kc;
    kc.init();
    const size_t e = ((size_t)blockIdx.y << in_per_bw_log) + ((size_t)blockIdx.x << g_log) + w;
    LPt R, Y;
    if (mode == 0) {
      R = lane29::load_pt<F>(kc, &in[e]);
      Y = R;
    } else if (mode == 1) {
      Y = lane29::load_pt<F>(kc, &in[2 * e]);
      R = lane29::load_pt<F>(kc, &in[2 * e + 1]);
    } else if (mode == 2) {
      Y = lane29::load_pt<F>(kc, &in[e]);
      R = lane29::load_pt<F>(kc, &in2[e]);
    } else {
      const size_t t = ((size_t)blockIdx.x << g_log) + w;
      const XYZZ29<F>* f = in + (size_t)blockIdx.y * 3 * 256;
      LPt loc = lane29::load_pt<F>(kc, &f[2 * 256 + t]);
#pragma unroll 1
      for (unsigned k = 0; k < k_log; k++) loc = lane29::dbl_pt<FO>(kc, loc);
      Y = lane29::add_pt<FO>(kc, lane29::load_pt<F>(kc, &f[t]), loc);
      R = lane29::load_pt<F>(kc, &f[256 + t]);
    }
    unsigned buf = 0;
#pragma unroll 1
    for (unsigned d = 1; d < G; d <<= 1) {           // inclusive suffix scan: R_w <- sum_{j >= w} R_j
      lane29::store_pt_raw<F>(kc, &sA[buf][w], R);
      __syncthreads();
      if (w + d < G) R = lane29::add_pt<FO>(kc, R, lane29::load_pt<F>(kc, &sA[buf][w + d]));
      buf ^= 1;
    }
    LPt X = R;                                       // suffix_w; the weighted sum drops suffix_0 (= R')
    if (w == 0) X = lane29::inf_pt<FO>(kc);
    bool first = true;
#pragma unroll 1
    for (unsigned d = G >> 1; d >= 1; d >>= 1) {
      lane29::store_pt_raw<F>(kc, &sA[buf][w], X);
      if (!level0) {
        if (first) lane29::store_pt_raw<F>(kc, &sB[buf][w], Y);
        else if (w >= 2 * d && w < 4 * d) lane29::store_pt_raw<F>(kc, &sB[buf][w - 2 * d], Y);
      }
      __syncthreads();
      if (w < d) X = lane29::add_pt<FO>(kc, X, lane29::load_pt<F>(kc, &sA[buf][w + d]));
      else if (!level0 && w < 2 * d)
        Y = lane29::add_pt<FO>(kc, lane29::load_pt<F>(kc, &sB[buf][w - d]), lane29::load_pt<F>(kc, &sB[buf][w]));
      buf ^= 1;
      first = false;
    }
    // wave 0: X = sum_j j R_j, R = R'; wave 1 (G >= 2): Y = sum_j W_j
    if (!level0 && G >= 2) {
      if (w == 1) lane29::store_pt_raw<F>(kc, &sB[buf][0], Y);
      __syncthreads();
      if (w == 0) Y = lane29::load_pt<F>(kc, &sB[buf][0]);
    }
    if (w != 0) return;
#pragma unroll 1
    for (unsigned k = 0; k < u_log; k++) X = lane29::dbl_pt<FO>(kc, X);
    const LPt Wp = lane29::add_pt<FO>(kc, level0 ? R : Y, X);
    if (window_sums) {
      lane29::store_pt<F>(kc, &window_sums[blockIdx.y], Wp);
    } else {
      const size_t o = ((size_t)blockIdx.y << (in_per_bw_log - g_log)) + blockIdx.x;
      lane29::store_pt_raw<F>(kc, &out[2 * o], Wp);
      lane29::store_pt_raw<F>(kc, &out[2 * o + 1], R);
    }
  }
}
// The same reduction on ONE wave per group, for a chip that is busy with other streams' saturating kernels (the MSMs of
// a proof): a 16-wave workgroup waits for a whole compute unit to drain while 256-lane accumulation workgroups keep taking
// the slots that free up, and a wave per bucket costs ~6x the issue slots of the row kernels (measured: config 4 1.70 ->
// 1.86 ms, 8-shard rank 2.57 -> 2.79 with the form above inside proofs; profiles/r6s_lane_reduce_ab.txt).  Here the group's
// sums are running sums from the top entry down -- run += R_j; acc += run -- 2 G + 1 additions (3 G above level 0) on a
// 64-lane workgroup, entries prefetched one ahead.
template <class F>
__global__ void __launch_bounds__(64) msm_lane_reduce_serial_kernel(const XYZZ29<F>* __restrict__ in, unsigned in_per_bw_log,
                                                                     unsigned g_log, unsigned u_log, int level0,
                                                                     XYZZ29<F>* __restrict__ out,
                                                                     XYZZ29<F>* __restrict__ window_sums) {
  if constexpr (lane29::enabled<F>()) {
    using FO = lane29::Ops<F>;
    using LPt = lane29::Pt<FO>;
    // (default priority ON PURPOSE: inside a queue of proofs these reductions have slack and the accumulations they run
    // beside do not -- at s_setprio 3 an 8-shard rank took 2.56-2.59 ms per proof, at 0 2.50-2.51, a 4-shard rank 3.78 against
    // 3.63, config 4 in a queue 1.65 against 1.61: profiles/r6hh_serial_reduce_prio_ab.txt)
#ifdef DG16_SERIAL_REDUCE_PRIO
    __builtin_amdgcn_s_setprio(DG16_SERIAL_REDUCE_PRIO);
#endif
    const unsigned G = 1u << g_log;
    typename FO::KT kc;
    kc.init();
    const size_t e0 = ((size_t)blockIdx.y << in_per_bw_log) + ((size_t)blockIdx.x << g_log);
    const XYZZ29<F>* base = level0 ? in + e0 : in + 2 * e0;
    const unsigned stride = level0 ? 1u : 2u, roff = level0 ? 0u : 1u;
    LPt run = lane29::inf_pt<FO>(kc), acc = run, wsum = run;
    LPt rj = lane29::load_pt_words<F>(kc, base + (size_t)(G - 1) * stride + roff), wj = rj;
    if (!level0) wj = lane29::load_pt_words<F>(kc, base + (size_t)(G - 1) * stride);
#pragma unroll 1
    for (unsigned j = G - 1; j >= 1; j--) {
      // (one ahead: the words only -- the identity flag is taken when the entry is used)
      const LPt rn = lane29::load_pt_words<F>(kc, base + (size_t)(j - 1) * stride + roff);
      LPt wn = rn;
      if (!level0) wn = lane29::load_pt_words<F>(kc, base + (size_t)(j - 1) * stride);
      run = lane29::add_pt<FO>(kc, run, lane29::with_inf_flag<FO>(rj));
      acc = lane29::add_pt<FO>(kc, acc, run);
      if (!level0) wsum = lane29::add_pt<FO>(kc, wsum, lane29::with_inf_flag<FO>(wj));
      rj = rn;
      wj = wn;
    }
    rj = lane29::with_inf_flag<FO>(rj);
    wj = lane29::with_inf_flag<FO>(wj);
    const LPt Rp = lane29::add_pt<FO>(kc, run, rj);                  // R' = sum_j R_j;  acc = sum_j j R_j
    if (!level0) wsum = lane29::add_pt<FO>(kc, wsum, wj);
#pragma unroll 1
    for (unsigned k = 0; k < u_log; k++) acc = lane29::dbl_pt<FO>(kc, acc);
    const LPt Wp = lane29::add_pt<FO>(kc, level0 ? Rp : wsum, acc);
    if (window_sums) {
      lane29::store_pt<F>(kc, &window_sums[blockIdx.y], Wp);
    } else {
      const size_t o = ((size_t)blockIdx.y << (in_per_bw_log - g_log)) + blockIdx.x;
      lane29::store_pt_raw<F>(kc, &out[2 * o], Wp);
      lane29::store_pt_raw<F>(kc, &out[2 * o + 1], Rp);
    }
  }
}
// buckets[bwi][2^log_nb] -> window_sums[bwi]; false when the set is not one this path takes
template <class F>
bool msm_lane_reduce(hipStream_t s, const MsmGeom& g, const MsmBuffers<F>& b) {
  if constexpr (!lane29::enabled<F>()) return false;
  else {
    const unsigned bwi = g.bw * b.ninst;
    if (!b.lane_tmp || !lane_reduce_applies<F>((size_t)bwi << g.log_nb)) return false;
    static const bool busy_off = [] { const char* e = getenv("DG16_NO_LANE_REDUCE_BUSY"); return e && atoi(e) != 0; }();
    if (b.busy_chip && busy_off) return false;
    constexpr int RB = lane_reduce_radix_log<F>();
    // level 0 writes at most a quarter of the bucket count (pairs of an eighth), level 1 a 32nd, ... : two buffers in turn
    XYZZ29<F>* pong[2] = {b.lane_tmp, b.lane_tmp + (((size_t)bwi << g.log_nb) / 4 + 2)};
    const XYZZ29<F>* in = b.buckets;
    unsigned rem = g.log_nb, u_log = 0, per_log = g.log_nb, lvl = 0;
    do {
      const unsigned g_log = rem < (unsigned)RB ? rem : (unsigned)RB;
      const bool last = rem == g_log;
      XYZZ29<F>* out = pong[lvl & 1];
      // a wave per entry pays while the level's waves fit the chip a few times over: 8 192 buckets took 180 us there, the
      // issue slots of 74 000 wave-additions (profiles/r6x_timeline_config4.md); one wave per group beyond that
      const bool one_wave = b.busy_chip || ((size_t)bwi << per_log) > kLaneParallelMaxEntries;
      if (one_wave)
        hipLaunchKernelGGL((msm_lane_reduce_serial_kernel<F>), dim3(1u << (per_log - g_log), bwi), dim3(64), 0, s, in, per_log,
                           g_log, u_log, (int)(lvl == 0), out, last ? b.window_sums : (XYZZ29<F>*)nullptr);
      else
        hipLaunchKernelGGL((msm_lane_reduce_kernel<F, RB>), dim3(1u << (per_log - g_log), bwi), dim3(64u << g_log), 0, s, in,
                           (const XYZZ29<F>*)nullptr, per_log, g_log, u_log, lvl == 0 ? 0 : 1, 0u, out,
                           last ? b.window_sums : (XYZZ29<F>*)nullptr);
      in = out;
      rem -= g_log;
      per_log -= g_log;
      u_log += g_log;
      lvl++;
    } while (rem);
    return true;
  }
}

// What msm_top_kernel does, as lane-form levels over its <= 256 entries per bucket-window (modes 2 / 3 above); for an idle
// chip only (a plain MSM, H's exposed reduction): false when the path does not apply
template <class F>
bool msm_lane_top(hipStream_t s, const MsmGeom& g, const MsmBuffers<F>& b, const TopGeom& tg) {
  if constexpr (!lane29::enabled<F>()) return false;
  else {
    const unsigned bwi = g.bw * b.ninst;
    static const bool off = [] { const char* e = getenv("DG16_NO_LANE_TOP"); return e && atoi(e) != 0; }();
    if (off || !b.top_tmp || b.busy_chip || (size_t)bwi * tg.lanes > kLaneParallelMaxEntries) return false;
    unsigned ll = 0;
    while ((1u << ll) < tg.lanes) ll++;
    if ((1u << ll) != tg.lanes) return false;
    constexpr int RB = lane_reduce_radix_log<F>();
    XYZZ29<F>* pong[2] = {b.top_tmp, b.top_tmp + (size_t)bwi * 64 + 4};
    const XYZZ29<F>* in = tg.folded ? b.fold : b.row_w;
    const XYZZ29<F>* in2 = tg.folded ? nullptr : b.row_r;
    unsigned rem = ll, per_log = ll, lvl = 0;
    unsigned u_log = tg.folded ? tg.final_log + tg.per_log : tg.final_log;
    do {
      const unsigned g_log = rem < (unsigned)RB ? rem : (unsigned)RB;
      const bool last = rem == g_log;
      XYZZ29<F>* out = pong[lvl & 1];
      hipLaunchKernelGGL((msm_lane_reduce_kernel<F, RB>), dim3(1u << (per_log - g_log), bwi), dim3(64u << g_log), 0, s, in, in2,
                         per_log, g_log, u_log, lvl ? 1 : (tg.folded ? 3 : 2), tg.final_log, out,
                         last ? b.window_sums : (XYZZ29<F>*)nullptr);
      in = out;
      in2 = nullptr;
      rem -= g_log;
      per_log -= g_log;
      u_log += g_log;
      lvl++;
    } while (rem);
    return true;
  }
}

// Phase B (latency-bound, few waves): finalize -> giants -> rows -> top -> tail.  May run on another stream than
// phase A so that it hides behind the next MSM's accumulation.
// DG16_TRACE=1: synchronise after every launch of the bucket phase and print its wall time (debugging aid)
inline void trace_point(hipStream_t s, const char* what) {
  static const bool on = [] { const char* e = getenv("DG16_TRACE"); return e && atoi(e) != 0; }();
  if (!on) return;
  static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  hipError_t e = hipStreamSynchronize(s);
  auto now = std::chrono::steady_clock::now();
  fprintf(stderr, "[dg16 trace] %-24s %9.3f ms  %s\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
          e == hipSuccess ? "" : hipGetErrorString(e));
  last = now;
}

// (Round 5 measured this chain and the digit sort as hipGraphs -- captured once per argument set, replayed with one
// hipGraphLaunch: no gain, profiles/r5e_hipgraph_chains_ab.txt -- the host spends 0.3-0.4 ms enqueueing a whole proof
// outside a profiler; the 3 ms seen under rocprofv3 are the tracer's.  Removed.)
template <class F>
void msm_bucket_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b, bool out_affine, void* out_dev) {
  const MsmGeom& g = st.g;
  const unsigned bwi = g.bw * b.ninst;      // bucket-windows over all instances
  DG_BOUNDS_BIND();
  trace_point(s, "(before bucket phase)");
  if constexpr (FieldOf<F>::EXT) {
    // G2: the throughput finalize ran behind the accumulation, on its stream (msm_impl.h: msm_finalize_lds_phase)
  } else {
    DG_HIP(hipMemsetAsync(b.giant, 0, 8, s));
    msm_finalize_phase<F>(s, st, b);        // inline products (msm_group.hip)
  }
  trace_point(s, "finalize");
  // few workgroups striding over the device-side work list: nothing to do (the common case) costs ~10 us
  hipLaunchKernelGGL(msm_giant_kernel<F>, dim3(256), dim3(256), 0, s, g, msm_acc_wg_log<F>(), st.counts, st.seg_off,
                     b.seg_sum, b.giant, b.giant + 2, b.giant_cap);
  hipLaunchKernelGGL(msm_giant_fold_kernel<F>, dim3(64), dim3(64), 0, s, g, msm_acc_wg_log<F>(), st.counts, st.seg_off,
                     b.seg_sum, b.buckets, b.giant, b.giant + 2, b.giant_cap);
  trace_point(s, "giant + fold");
  if (msm_lane_reduce<F>(s, g, b)) {        // small bucket sets: the lane-form reduction (msm_lane_reduce_kernel)
    trace_point(s, "lane reduce");
    msm_tail_phase<F>(s, st, b, out_affine, out_dev);
    trace_point(s, "tail");
    DG_HIP(hipGetLastError());
    return;
  }
  // many rows over all bucket-windows (plain MSMs): lanes that own 2^k buckets each (msm_rowchunk_kernel), k so that
  // ~256 workgroups remain; few rows (one bucket set of a resident table): one lane per bucket, 16 steps
  constexpr int CH_LANES = sizeof(XYZZ29<F>) * 2 * 256 <= 150 * 1024 ? 256 : 128;
  constexpr unsigned CH_LL = CH_LANES == 256 ? 8u : 7u;
  unsigned k_log = 0;
  if (b.rg.row_log == kRowLog && g.log_nb >= CH_LL + 1) {
    const size_t total_rows = (size_t)bwi << b.rg.rows_log;
    while (k_log < 3 && (total_rows >> (k_log + 1)) >= 256 && g.log_nb >= CH_LL + k_log + 1) k_log++;
    if (total_rows < 1024) k_log = 0;
    // ... except for the 14-limb fields over a LARGE table (one bucket set of 2^16 buckets = 256 rows: the MSMs of a
    // 2^20 BLS12-381 proof): there the 16-step suffix scan + tree of msm_row_kernel is ~0.5 M full additions per MSM on
    // 65 536 lanes at 256 VGPRs -- work and SIMD slots taken from the accumulation running beside it -- while chunks of
    // 8 buckets do ~0.19 M on 8 192 lanes in 32 dependent steps that hide behind that accumulation anyway.  Same call,
    // chunks of 1 / 2 / 4 / 8 buckets: 25.6-25.7 / 25.6-25.7 / 24.9-25.0 / **24.4-24.6 ms** per BLS12-381 2^20 proof on one
    // box, 23.05-23.13 -> **22.19-22.23** on another; BN254 (9 limbs: a third of the work per addition) 10.54-10.60 /
    // 10.58-10.75 / 10.54-10.65: nothing, the rows stay (profiles/r5i_*, r5j_*).  Short shards keep the rows: their
    // reductions are the critical path, not hidden work.
    if (total_rows < 1024 && RR<typename FieldOf<F>::Params>::N > 9 && g.table && g.region >= ((size_t)1 << 23)) {
      k_log = g.log_nb - CH_LL < 3 ? g.log_nb - CH_LL : 3;
    }
    if (k_log > 3) k_log = 3;
    // the top kernel's folded mode takes <= 256 chunk workgroups per bucket-window: beyond that (one bucket set of
    // 2^20 buckets: the table of a 2^24-point key) the rows + msm_rowfold_kernel path serves
    if (k_log && g.log_nb - k_log - CH_LL > 8) k_log = 0;
  }
  TopGeom tg{};
  tg.rows_log = b.rg.rows_log;
  tg.final_log = b.rg.row_log;
  if (k_log) {
    const unsigned nsr_log = g.log_nb - k_log - CH_LL;          // chunk workgroups per bucket-window (<= 256: log_nb <= 19)
    DG_REQUIRE(nsr_log <= 8, DG16_ERR_UNSUPPORTED, "row chunks: more than 256 per bucket-window");
    hipLaunchKernelGGL((msm_rowchunk_kernel<F, CH_LANES>), dim3(1u << nsr_log, bwi), dim3(CH_LANES), 0, s, g.log_nb, k_log,
                       b.buckets, b.fold);
    trace_point(s, "row (chunks)");
    tg.folded = 1;
    tg.per_log = CH_LL;
    tg.lanes = 1u << nsr_log;
    tg.final_log = k_log;
  } else {
    hipLaunchKernelGGL(msm_row_kernel<F>, dim3(1u << b.rg.rows_log, bwi), dim3(256), 0, s, g, b.rg, b.buckets, b.row_w,
                       b.row_r);
    trace_point(s, "row");
    tg.folded = b.rg.rows_log > 8;
    tg.per_log = tg.folded ? b.rg.rows_log - 8 : 0;
    tg.lanes = tg.folded ? 256u : 1u << b.rg.rows_log;
    if (b.rg.rows_log > 8)
      hipLaunchKernelGGL(msm_rowfold_kernel<F>, dim3(8, bwi), dim3(64), 0, s, b.rg, b.row_w, b.row_r, b.fold);
  }
  constexpr int HALVES = sizeof(XYZZ29<F>) * 513 <= 160 * 1024 ? 2 : 1;
  if (!msm_lane_top<F>(s, g, b, tg))
    hipLaunchKernelGGL((msm_top_kernel<F, HALVES>), dim3(bwi), dim3(256 * HALVES), 0, s, tg, b.row_w, b.row_r,
                       b.fold, b.window_sums);
  trace_point(s, "top");
  msm_tail_phase<F>(s, st, b, out_affine, out_dev);     // msm_group.hip: inline products whatever this unit's are
  trace_point(s, "tail");
  DG_HIP(hipGetLastError());
}

}  // namespace dg16
