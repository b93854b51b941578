// Pippenger MSM for gfx950 -- replaces `G::msm(bases, scalars)` at dist-primitives/src/dmsm/mod.rs:82 (ark-ec
// VariableBaseMSM).  Any correct algorithm yields the same group element; parity is checked in affine form.
//
// Pipeline (DESIGN.md section 2.2 has the table; MSM_INVARIANTS.md every array's size, writer and capacity):
//   1-3 sort        scalar -> W signed c-bit digits, counted and placed per (bucket-window, bucket) by a two-level LDS-
//                   partitioned counting sort (msm_part_*; the direct atomic path msm_digits / msm_scatter for small inputs)
//   4   accumulate  one lane per SEGMENT of a bucket: XYZZ mixed additions on the reduced-radix types -- THE dominant kernels,
//                   VALU-issue-bound (msm_accumulate_kernel / _lds_kernel / _steps_kernel)
//   4b  finalize    bucket = sum of its partials (in-workgroup tree + stitch for BN254 G1, throughput finalize otherwise;
//                   buckets with > 64 partials through the giant work list)
//   5   reduce      sum_b (b + 1) B_b per bucket-window: rows of 256 buckets, then one workgroup per window (msm_reduce_impl.h)
//   6   tail        Horner over the window sums on ONE wave (fresh bases only; resident tables have one bucket set)
// What was measured and removed on the way is in CHANGELOG.md, not here.
#pragma once
#include <atomic>
#include <utility>
#include "glv.h"
#include <stdio.h>

#include <chrono>

#include "bounds.h"
#include "ctx.h"
#include "ec29.h"
#include "lane29.h"
#include "types.h"

#ifndef DG16_CHAIN_PRIO
#define DG16_CHAIN_PRIO 3       // priority of the one-wave chain kernels (Horner tail, scalar multiples, assembly)
#endif
namespace dg16 {

// Accumulation segments: a bucket of cnt entries is cut into k = ceil(cnt / 2^seg_log) segments of EQUAL length
// (floor / ceil of cnt / k), one lane each: the lanes of a wave run chains of nearly the same length (with fixed-length
// segments every bucket ended in a short one and its wave idled behind the long ones), and k -- the number of partials
// the finalize has to add per bucket -- is as small as the segment length allows.  seg_log follows the mean bucket
// occupancy, clamped by the lane count a launch needs.  The (segment -> bucket) map is not stored: a lane finds its
// bucket by binary search in the exclusive scan of the per-bucket segment counts.
constexpr unsigned kMinSegLog = 3, kMaxSegLog = 9;
constexpr unsigned kMinLanesLog = 18;    // want >= 2^18 segments (4 waves per SIMD) in an accumulation launch
constexpr unsigned kGiantSegs = 64;      // buckets with more segments are reduced by a whole workgroup
constexpr unsigned kGiantSlices = 64;    // ... in at most this many slices (one workgroup each) of about
constexpr unsigned kGiantSliceSegs = 512;   // ... this many partials (giant_geometry)

struct MsmGeom {
  unsigned c;        // window bits
  unsigned nwin;     // W: signed digits per scalar
  unsigned log_nb;   // log2 buckets per bucket-window = c - 1
  unsigned seg_log;  // log2 entries per accumulation segment
  unsigned seg_cap;  // segment slots per bucket-window
  unsigned bw;       // bucket-windows: W (plain), 1 (full table: all digits share one bucket set), or the row stride
                     // k of a table thinned to every k-th row (window w feeds bucket-window w % k through row w / k)
  unsigned table;    // 1: bases are a table T[r*n + i] = 2^(c*bw*r) * P_i (resident keys)
  unsigned rows;     // table rows R = ceil(W / bw) (1 in plain mode)
  size_t region;     // entries per bucket-window: rows * n
};

// window size: plain mode keeps ~32 points per bucket; table mode has a single bucket set of W*n entries:
// c = log2(n) - 3 keeps the bucket reduction at a few percent of the MSM (measured at 2^20: c = 17 beats
// both 16 and 20; again in round 4 with the last reduction off the critical path, profiles/r4w_table_window_sweep.txt:
// 17: 10.19 ms per proof, 18: 11.10, 19: 11.35, 20: 11.83; H alone at 19 / 20: 10.28 / 10.38)
inline unsigned msm_window_bits(size_t n, bool table, unsigned scalar_bits = 0) {
  unsigned lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  if (n > ((size_t)3 << lg) / 2) lg++;     // nearest power of two (2^20 - 5 points are "2^20")
  int c = table ? (int)lg - 3 : (int)lg - 4;
  // short tables (the shards of a multi-GPU key): the bucket reduction's latency does not shrink with the bucket
  // count, the number of bucket entries W*n does shrink with c -- measured on a 2^17-point shard: c = 14: 8.2 ms
  // per proof, 15: 6.2, 16: 6.2, 17: 6.3.  Round 6 (lane-form reductions; profiles/r6tt_*, r6uu_*, r6vv_*): on BN254 what
  // decides between neighbouring widths is the TOP window -- a width that leaves it two or three bits puts a quarter of the
  // key into a handful of giant buckets (255 digit bits: c = 12 or 14 cost a 2^15-point proof 1.8 / 1.45 ms against 1.36 at
  // 15 = 255 / 17; 16 leaves 15 bits too and costs 1.56 with twice the buckets).  So where 15 and 16 leave the same top
  // window (BN254, BLS12-377) 15 up to 2^16 points and 16 beyond; BLS12-381 (256 digit bits: 16 | 256) measured flat to
  // within 4 % between 14, 15 and 16 at 2^14..2^16 points and 12 % better at 14 for 2^13 -- it keeps lg + 1.
  if (table && c < 16) {
    c = (int)lg + 1 < 16 ? (int)lg + 1 : 16;
    if (scalar_bits && lg >= 13) {   // below 2^13 points nothing was measured: lg + 1 as before
      auto top = [&](int w) { const int bits = (int)scalar_bits + 1; return bits - ((bits + w - 1) / w - 1) * w; };
      const int t15 = top(15), t16 = top(16);
      if (t15 >= t16) c = (t15 > t16 || lg <= 16) ? 15 : 16;
    }
  }
  if (const char* e = getenv(table ? "DG16_MSM_TABLE_C" : "DG16_MSM_C")) c = atoi(e);
  int hi = table ? 20 : 16;
  if (c < 4) c = 4;
  if (c > hi) c = hi;
  return (unsigned)c;
}

// stride (table mode): 1 = every window has its table row; k > 1 = the table keeps every k-th row (HBM budget) and the
// MSM has k bucket sets combined by a Horner tail of (k - 1) * c doublings:
//   sum_w d_w 2^(c w) P = sum_{j < k} 2^(c j) sum_r d_{k r + j} (2^(c k r) P)
inline MsmGeom msm_geometry(size_t n, unsigned scalar_bits, bool table = false, unsigned c_fixed = 0, unsigned stride = 1) {
  MsmGeom g;
  g.c = c_fixed ? c_fixed : msm_window_bits(n, table);
  g.nwin = (scalar_bits + 1 + g.c - 1) / g.c;   // one spare bit absorbs the last carry
  g.log_nb = g.c - 1;
  g.table = table ? 1u : 0u;
  if (stride < 1) stride = 1;
  if (stride > g.nwin) stride = g.nwin;
  g.bw = table ? stride : g.nwin;
  g.rows = (g.nwin + g.bw - 1) / g.bw;
  g.region = (size_t)g.rows * n;
  {
    size_t mean = g.region >> g.log_nb;   // entries per bucket
    unsigned lm = 0;
    while (((size_t)2 << lm) <= mean) lm++;
    // Segment length: 16 entries while the launch has ~2^20 segments, 32 beyond -- short segments balance the last rounds of
    // a launch, long ones leave fewer partials per bucket for the tree / finalize, and which matters more is a matter of
    // how many rounds the launch runs.  Measured (profiles/r4seg_*, r5k_*, r5l_*, same call each): BN254 2^20 (2^23.9
    // entries) 16: 10.01-10.04 ms per proof, 32: 10.14-10.19; BLS12-381 2^20 (2^24 entries) 8: 22.4-23.1, **16: 21.2-21.6**,
    // 32: 22.3-22.4 (the rule before round 5 -- mean occupancy / 8 -- gave 32 there: 256 entries per bucket exactly),
    // 64: 24.2; BN254 2^22 (2^25.8) 16: 36.8-37.1, **32: 36.0-36.6**; BLS12-381 2^22 16: 83.5-83.8, 32: 82.0-82.1.
    unsigned le_all = 0;
    while (((size_t)2 << le_all) <= g.region * g.bw) le_all++;   // floor(log2(entries of the launch: all bucket-windows))
    int sl = (int)le_all - 20;
    if (sl < 4) sl = 4;
    if (sl > 5) sl = 5;
    if (sl > (int)lm - 2) sl = (int)lm - 2;                      // ... and at least four segments per mean bucket
    if (sl < 4) sl = 4;
    // ... but never so long that the launch runs out of lanes (a 2^17-point shard with 32-entry segments has
    // 1.2 waves per SIMD: measured 0.62 ms per G1 accumulation instead of 0.25)
    unsigned le = 0;
    while (((size_t)2 << le) <= (size_t)g.nwin * n) le++;
    int cap = (int)le - (int)kMinLanesLog;
    // ... except a PLAIN MSM of 2^20..2^21 entries (2^16 points after the GLV split): 8-entry segments are 1.25 rounds of the
    // 14-limb G1 accumulation's 2^17 resident lanes and ~4 partials per bucket for the wave-per-bucket finalize, 16-entry
    // ones are one round of the same length and half the partials -- BLS12-377 G1 2^16 1.51 -> 1.20 ms, BLS12-381 G1 1.26 ->
    // 1.13, BN254 G2 1.87 -> 1.64, BN254 G1 unchanged; a size down or up 16 is no better or worse
    // (profiles/r6xx_seg_log_small_plain_msm.txt: DG16_MSM_SEG_LOG sweep, same call)
    if (!table && le == 20 && cap < 4) cap = 4;
    if (sl > cap) sl = cap;
    g.seg_log = (unsigned)(sl < (int)kMinSegLog ? (int)kMinSegLog : sl > (int)kMaxSegLog ? (int)kMaxSegLog : sl);
    static const int seg_env = [] { const char* e = getenv("DG16_MSM_SEG_LOG"); return e ? atoi(e) : 0; }();   // (sweeps)
    if (seg_env >= (int)kMinSegLog && seg_env <= (int)kMaxSegLog) g.seg_log = (unsigned)seg_env;
  }
  g.seg_cap = (1u << g.log_nb) + (unsigned)((g.region + (1u << g.seg_log) - 1) >> g.seg_log);
  return g;
}

// atomicAdd(&ctr[idx], 1) for every active lane, returning the old value -- robust to heavy hitters.  Random
// digits almost never collide inside a wave, but the top window of any c (2 bits at c = 18: ALL scalars land
// in 3 buckets) and real witnesses (bits: half of all entries hit bucket 0 of window 0) serialise a million
// atomics on one address (measured: 23 -> 57 ms per proof at c = 18).  Cheap test first (does my neighbour
// lane hit the same counter?); only skewed waves pay the leader loop: one atomic per distinct counter.
__device__ __forceinline__ unsigned wave_atomic_inc(unsigned* __restrict__ ctr, unsigned idx, bool active) {
  const unsigned lane = __lane_id();
  const unsigned nb = __shfl_down(idx, 1);
  const bool nb_active = __shfl_down((int)active, 1);
  const unsigned long long like = __ballot(active && nb_active && nb == idx && lane < 63);
  unsigned old = 0;
  if (__popcll(like) < 8) {
    if (active) old = atomicAdd(&ctr[idx], 1u);
    return old;
  }
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned lidx = __shfl(idx, leader);
    const unsigned long long grp = __ballot(active && idx == lidx) & todo;
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(&ctr[lidx], (unsigned)__popcll(grp));
    base = __shfl(base, leader);
    if ((grp >> lane) & 1) old = base + (unsigned)__popcll(grp & ((1ull << lane) - 1));
    todo &= ~grp;
  }
  return old;
}

// The shader clock UNDER a kernel, measured by the kernel (round 6): s_memtime ticks once per shader cycle, s_memrealtime at
// a constant 100 MHz (MI355X_MICROARCH.md); lane 0 of every workgroup adds its two deltas to clk[0], clk[1] (two atomics per
// workgroup, outside every loop), so clk[0] / clk[1] x 100 MHz is the duration-weighted clock the chip held while the
// kernel ran.  bench.py prices the accumulations against 16 lanes x 4 SIMDs x CUs x THAT clock (dg16_last_kernel_ms,
// which = 2) next to the calibrated issue rate -- a cycle-based utilisation that does not move with DVFS.  clk may be null.
struct ClkProbe {
  unsigned long long c0 = 0, w0 = 0;
  __device__ __forceinline__ void begin(const unsigned long long* clk) {
    if (clk && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
  }
  __device__ __forceinline__ void end(unsigned long long* clk) const {
    if (clk && threadIdx.x == 0) {
      atomicAdd(&clk[0], (unsigned long long)__builtin_readcyclecounter() - c0);
      atomicAdd(&clk[1], (unsigned long long)wall_clock64() - w0);
    }
  }
};

// ---- 1: digits + histogram -------------------------------------------------------------------
template <class Fr>
__global__ void __launch_bounds__(256) msm_digits_kernel(const Fr* __restrict__ scalars, size_t n, int mont,
                                                          MsmGeom g, int* __restrict__ digits,
                                                          unsigned* __restrict__ counts) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;          // no early exit: the wave-aggregated histogram needs every lane
  Fr s = Fr::zero();
  if (live) {
    s = scalars[i];
    if (mont & 1) s = s.from_mont();
  }
  const bool flip = (mont & 2) && (s.l[Fr::NL - 1] >> 31);   // bit 255 = "negate this scalar": ONLY for the halves glv.h makes
  const unsigned c = g.c;
  const unsigned half = 1u << (c - 1);
  unsigned carry = 0;
  for (unsigned w = 0; w < g.nwin; w++) {
    unsigned bit = w * c;
    unsigned limb = bit >> 5, off = bit & 31;
    uint64_t v = 0;
    if (limb < (unsigned)Fr::NL) {
      v = s.l[limb];
      if (limb + 1 < (unsigned)Fr::NL) v |= (uint64_t)s.l[limb + 1] << 32;
      v >>= off;
    }
    int d = (int)((unsigned)v & ((1u << c) - 1)) + (int)carry;
    if ((unsigned)d > half) { d -= (int)(1u << c); carry = 1; } else { carry = 0; }
    if (flip) d = -d;
    if (live) digits[(size_t)w * n + i] = d;
    unsigned b = d ? (unsigned)(d < 0 ? -d : d) - 1 : 0u;
    unsigned slot = ((w % g.bw) << g.log_nb) + b;
    wave_atomic_inc(counts, slot, live && d != 0);
  }
}

// ---- 2: per-bucket-window exclusive scans (entry offsets and segment offsets), three small launches ---
// (kernels are templated on a dummy so that every translation unit carries its own copy)
constexpr unsigned kScanBlock = 4096;   // buckets per workgroup (1024 threads x 4)
template <int TU>
__global__ void __launch_bounds__(1024) msm_scan_local_kernel(const unsigned* __restrict__ counts,
                                                               unsigned* __restrict__ offsets,
                                                               unsigned* __restrict__ seg_off,
                                                               unsigned* __restrict__ block_tot, unsigned log_nb,
                                                               unsigned seg_log) {
  __shared__ unsigned sh[1024];
  __shared__ unsigned sh2[1024];
  const unsigned nb = 1u << log_nb;
  const unsigned seg_round = (1u << seg_log) - 1;
  const size_t base = (size_t)blockIdx.y << log_nb;
  const unsigned lo = blockIdx.x * kScanBlock + threadIdx.x * 4;
  unsigned cn[4], sum = 0, ssum = 0;
#pragma unroll
  for (unsigned j = 0; j < 4; j++) {
    cn[j] = (lo + j < nb) ? counts[base + lo + j] : 0;
    sum += cn[j];
    ssum += (cn[j] + seg_round) >> seg_log;
  }
  sh[threadIdx.x] = sum;
  sh2[threadIdx.x] = ssum;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {
    unsigned v = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    unsigned v2 = threadIdx.x >= d ? sh2[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    sh2[threadIdx.x] += v2;
    __syncthreads();
  }
  unsigned run = sh[threadIdx.x] - sum, srun = sh2[threadIdx.x] - ssum;
#pragma unroll
  for (unsigned j = 0; j < 4; j++)
    if (lo + j < nb) {
      offsets[base + lo + j] = run;
      seg_off[base + lo + j] = srun;
      run += cn[j];
      srun += (cn[j] + seg_round) >> seg_log;
    }
  if (threadIdx.x == 1023) {
    size_t t = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    block_tot[t] = sh[1023];
    block_tot[t + 1] = sh2[1023];
  }
}
// one workgroup per bucket-window: exclusive scan of the (<= 1024) block totals
template <int TU>
__global__ void __launch_bounds__(1024) msm_scan_tops_kernel(unsigned* __restrict__ block_tot, unsigned nblocks,
                                                              unsigned* __restrict__ seg_total) {
  __shared__ unsigned sh[1024];
  __shared__ unsigned sh2[1024];
  unsigned* t = block_tot + (size_t)blockIdx.x * nblocks * 2;
  unsigned a = threadIdx.x < nblocks ? t[threadIdx.x * 2] : 0;
  unsigned b = threadIdx.x < nblocks ? t[threadIdx.x * 2 + 1] : 0;
  sh[threadIdx.x] = a;
  sh2[threadIdx.x] = b;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {
    unsigned v = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    unsigned v2 = threadIdx.x >= d ? sh2[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    sh2[threadIdx.x] += v2;
    __syncthreads();
  }
  if (threadIdx.x < nblocks) {
    t[threadIdx.x * 2] = sh[threadIdx.x] - a;
    t[threadIdx.x * 2 + 1] = sh2[threadIdx.x] - b;
  }
  if (threadIdx.x == 1023) seg_total[blockIdx.x] = sh2[1023];
}
template <int TU>
__global__ void __launch_bounds__(1024) msm_scan_fix_kernel(unsigned* __restrict__ offsets,
                                                             unsigned* __restrict__ seg_off,
                                                             unsigned* __restrict__ cursor,
                                                             const unsigned* __restrict__ block_tot, unsigned log_nb) {
  const unsigned nb = 1u << log_nb;
  const size_t base = (size_t)blockIdx.y << log_nb;
  const size_t t = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
  const unsigned add = block_tot[t], sadd = block_tot[t + 1];
  const unsigned lo = blockIdx.x * kScanBlock + threadIdx.x * 4;
#pragma unroll
  for (unsigned j = 0; j < 4; j++)
    if (lo + j < nb) {
      offsets[base + lo + j] += add;
      seg_off[base + lo + j] += sadd;
      cursor[base + lo + j] = 0;
    }
}

// ---- 3: scatter ---------------------------------------------------------------------------------
template <int TU>
__global__ void __launch_bounds__(256) msm_scatter_kernel(const int* __restrict__ digits, size_t n, MsmGeom g,
                                                           const unsigned* __restrict__ offsets,
                                                           const unsigned* __restrict__ seg_off,
                                                           unsigned* __restrict__ cursor,
                                                           unsigned* __restrict__ entries) {
  // one thread per (scalar, window) -- blockIdx.y = window: the rank comes back from an atomic, and a thread that walked
  // its scalar's windows paid W dependent round trips: 34 -> 4 us of a 0.55-ms MSM at 2^10 points (at 2^13 the 2^17.6
  // returning atomics are the bound either way: 45 -> 38 us; the digits kernel's atomics return nothing and gain nothing
  // from the same split: profiles/r6zv)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const unsigned w = blockIdx.y;
  int d = live ? digits[(size_t)w * n + i] : 0;
  const bool act = d != 0;
  unsigned b = act ? (unsigned)(d < 0 ? -d : d) - 1 : 0u;
  const unsigned bwin = w % g.bw;
  unsigned slot = (bwin << g.log_nb) + b;
  unsigned rank = wave_atomic_inc(cursor, slot, act);
  if (!act) return;
  unsigned ref = (unsigned)((size_t)(w / g.bw) * n + i);   // table row w / bw = 2^(c*bw*(w/bw)) * P_i (plain: row 0)
  entries[DG_IDX(15, (size_t)bwin * g.region + offsets[slot] + rank, (size_t)g.bw * g.region)] = ref | (d < 0 ? 0x80000000u : 0u);
}

// ---- 1'-3': partitioned digit sort (large MSMs) ------------------------------------------------------
// The atomic path above issues one device-scope atomic per entry twice (histogram, rank) and scatters 4-byte
// entries at random: ~31 G atomics/s and a 64-byte memory transaction per entry, 1.5 ms per 2^20-scalar sort.
// Here the slot index (bucket-window, bucket) is split into a partition (high bits, <= 256 of them) and a bin
// (low bits, <= 4096).  Pass 1 recomputes the digits twice instead of storing them: (a) per-workgroup LDS
// histogram over partitions, (b) after a scan, (ref, slot) pairs go to their partition at LDS-ranked
// positions.  Pass 2 walks each partition in tiles: (a) LDS histogram over bins -> bucket counts (one global
// atomic per non-empty bin and workgroup instead of one per entry), (b) after the usual bucket scans, LDS ranks
// inside the tile + one returning atomic per bin and tile give every entry its final position.
constexpr unsigned kPartScalars = 1024;   // scalars per workgroup in pass 1
constexpr unsigned kPartMax = 256;        // partitions
constexpr unsigned kPartMaxLowBits = 12;  // bins per partition <= 4096
constexpr unsigned kPartTileLog = 11;     // entries per tile in pass 2 (8 per lane)
constexpr unsigned kPartBlocks = 32;      // workgroups striding over one partition's tiles

struct PartGeom {
  unsigned low_bits, nparts, nblk1;
};

template <class Fr>
__device__ __forceinline__ int msm_digit(const Fr& s, unsigned w, unsigned c, unsigned& carry) {
  const unsigned half = 1u << (c - 1);
  unsigned bit = w * c;
  unsigned limb = bit >> 5, off = bit & 31;
  uint64_t v = 0;
  if (limb < (unsigned)Fr::NL) {
    v = s.l[limb];
    if (limb + 1 < (unsigned)Fr::NL) v |= (uint64_t)s.l[limb + 1] << 32;
    v >>= off;
  }
  int d = (int)((unsigned)v & ((1u << c) - 1)) + (int)carry;
  if ((unsigned)d > half) { d -= (int)(1u << c); carry = 1; } else { carry = 0; }
  return d;
}

template <class Fr>
__global__ void __launch_bounds__(256) msm_part_hist_kernel(const Fr* __restrict__ scalars, size_t n, int mont,
                                                             MsmGeom g, PartGeom pg,
                                                             unsigned* __restrict__ blockhist) {
  __shared__ unsigned hist[kPartMax];
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (unsigned k = 0; k < kPartScalars / 256; k++) {
    size_t i = (size_t)blockIdx.x * kPartScalars + k * 256 + threadIdx.x;
    if (i >= n) continue;
    Fr s = scalars[i];
    if (mont & 1) s = s.from_mont();
    unsigned carry = 0;
    for (unsigned w = 0; w < g.nwin; w++) {
      int d = msm_digit(s, w, g.c, carry);
      if (d == 0) continue;
      unsigned slot = ((w % g.bw) << g.log_nb) + (unsigned)(d < 0 ? -d : d) - 1;
      atomicAdd(&hist[slot >> pg.low_bits], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < pg.nparts) blockhist[(size_t)threadIdx.x * pg.nblk1 + blockIdx.x] = hist[threadIdx.x];
}

template <class Fr>
__global__ void __launch_bounds__(256) msm_part_scatter_kernel(const Fr* __restrict__ scalars, size_t n, int mont,
                                                                MsmGeom g, PartGeom pg,
                                                                const unsigned* __restrict__ blockoff,
                                                                uint2* __restrict__ part) {
  __shared__ unsigned cur[kPartMax];
  if (threadIdx.x < pg.nparts) cur[threadIdx.x] = blockoff[(size_t)threadIdx.x * pg.nblk1 + blockIdx.x];
  __syncthreads();
  for (unsigned k = 0; k < kPartScalars / 256; k++) {
    size_t i = (size_t)blockIdx.x * kPartScalars + k * 256 + threadIdx.x;
    if (i >= n) continue;
    Fr s = scalars[i];
    if (mont & 1) s = s.from_mont();
    const bool flip = (mont & 2) && (s.l[Fr::NL - 1] >> 31);   // bit 255 = "negate this scalar": ONLY for the halves glv.h makes
    unsigned carry = 0;
    for (unsigned w = 0; w < g.nwin; w++) {
      int d = msm_digit(s, w, g.c, carry);
      if (d == 0) continue;
      unsigned slot = ((w % g.bw) << g.log_nb) + (unsigned)(d < 0 ? -d : d) - 1;
      unsigned ref = (unsigned)((size_t)(w / g.bw) * n + i);   // table row w / bw (plain mode: bw = W, row 0)
      unsigned pos = atomicAdd(&cur[slot >> pg.low_bits], 1u);
      part[DG_IDX(14, pos, (size_t)g.nwin * n)] = make_uint2(ref | (((d < 0) != flip) ? 0x80000000u : 0u), slot);
    }
  }
}

// generic in-place exclusive scan of a[0..len): chunk scan -> scan of chunk totals -> add back
template <int TU>
__global__ void __launch_bounds__(1024) scan_chunk_kernel(unsigned* __restrict__ a, size_t len,
                                                           unsigned* __restrict__ tot) {
  __shared__ unsigned sh[1024];
  const size_t lo = (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
  unsigned v[4], sum = 0;
#pragma unroll
  for (unsigned j = 0; j < 4; j++) {
    v[j] = lo + j < len ? a[lo + j] : 0;
    sum += v[j];
  }
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {
    unsigned t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned run = sh[threadIdx.x] - sum;
#pragma unroll
  for (unsigned j = 0; j < 4; j++)
    if (lo + j < len) {
      a[lo + j] = run;
      run += v[j];
    }
  if (threadIdx.x == 1023) tot[blockIdx.x] = sh[1023];
}
template <int TU>
__global__ void __launch_bounds__(1024) scan_tops_kernel(unsigned* __restrict__ tot, unsigned nchunks) {
  __shared__ unsigned sh[1024];
  const unsigned per = (nchunks + 1023) / 1024;
  const unsigned lo = threadIdx.x * per;
  unsigned sum = 0;
  for (unsigned j = 0; j < per; j++)
    if (lo + j < nchunks) sum += tot[lo + j];
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {
    unsigned t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned run = sh[threadIdx.x] - sum;
  for (unsigned j = 0; j < per; j++)
    if (lo + j < nchunks) {
      unsigned t = tot[lo + j];
      tot[lo + j] = run;
      run += t;
    }
}
template <int TU>
__global__ void __launch_bounds__(1024) scan_add_kernel(unsigned* __restrict__ a, size_t len,
                                                         const unsigned* __restrict__ tot) {
  const size_t lo = (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
  const unsigned add = tot[blockIdx.x];
#pragma unroll
  for (unsigned j = 0; j < 4; j++)
    if (lo + j < len) a[lo + j] += add;
}

template <int TU>
__global__ void __launch_bounds__(256) msm_part_count_kernel(const uint2* __restrict__ part,
                                                              const unsigned* __restrict__ blockoff, PartGeom pg,
                                                              unsigned* __restrict__ counts) {
  __shared__ unsigned hist[1u << kPartMaxLowBits];
  const unsigned p = blockIdx.y;
  const unsigned lo = blockoff[(size_t)p * pg.nblk1];
  const unsigned size = blockoff[(size_t)(p + 1) * pg.nblk1] - lo;
  if (((size_t)blockIdx.x << kPartTileLog) >= size) return;
  const unsigned nlow = 1u << pg.low_bits;
  for (unsigned b = threadIdx.x; b < nlow; b += 256) hist[b] = 0;
  __syncthreads();
  for (size_t t = (size_t)blockIdx.x << kPartTileLog; t < size; t += (size_t)gridDim.x << kPartTileLog)
    for (unsigned j = 0; j < (1u << kPartTileLog) / 256; j++) {
      size_t idx = t + j * 256 + threadIdx.x;
      if (idx < size) atomicAdd(&hist[part[lo + idx].y & (nlow - 1)], 1u);
    }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < nlow; b += 256)
    if (hist[b]) atomicAdd(&counts[((size_t)p << pg.low_bits) + b], hist[b]);
}

template <int TU>
__global__ void __launch_bounds__(256) msm_part_place_kernel(const uint2* __restrict__ part,
                                                              const unsigned* __restrict__ blockoff, PartGeom pg,
                                                              MsmGeom g, const unsigned* __restrict__ offsets,
                                                              const unsigned* __restrict__ seg_off,
                                                              unsigned* __restrict__ cursor,
                                                              unsigned* __restrict__ entries) {
  __shared__ unsigned cnt[1u << kPartMaxLowBits];    // entries of this tile per bin
  __shared__ unsigned rank0[1u << kPartMaxLowBits];  // rank of the tile's first entry inside its bucket
  __shared__ unsigned dst0[1u << kPartMaxLowBits];   // position of the bucket's first entry
  const unsigned p = blockIdx.y;
  const unsigned lo = blockoff[(size_t)p * pg.nblk1];
  const unsigned size = blockoff[(size_t)(p + 1) * pg.nblk1] - lo;
  if (((size_t)blockIdx.x << kPartTileLog) >= size) return;
  const unsigned nlow = 1u << pg.low_bits;
  constexpr unsigned PER = (1u << kPartTileLog) / 256;
  for (size_t t = (size_t)blockIdx.x << kPartTileLog; t < size; t += (size_t)gridDim.x << kPartTileLog) {
    for (unsigned b = threadIdx.x; b < nlow; b += 256) cnt[b] = 0;
    __syncthreads();
    uint2 e[PER];
    unsigned lr[PER];
#pragma unroll
    for (unsigned j = 0; j < PER; j++) {
      size_t idx = t + j * 256 + threadIdx.x;
      e[j] = make_uint2(0u, 0xFFFFFFFFu);
      if (idx < size) {
        e[j] = part[lo + idx];
        lr[j] = atomicAdd(&cnt[e[j].y & (nlow - 1)], 1u);
      }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < nlow; b += 256)
      if (cnt[b]) {
        const size_t slot = ((size_t)p << pg.low_bits) + b;
        rank0[b] = atomicAdd(&cursor[slot], cnt[b]);
        dst0[b] = (unsigned)((slot >> g.log_nb) * g.region) + offsets[slot];
      }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < PER; j++) {
      if (e[j].y == 0xFFFFFFFFu) continue;
      const unsigned slot = e[j].y, b = slot & (nlow - 1);
      const unsigned rank = rank0[b] + lr[j];
      entries[DG_IDX(13, dst0[b] + rank, (size_t)g.bw * g.region)] = e[j].x;
    }
    __syncthreads();
  }
}

// ---- 4: segment accumulation -------------------------------------------------------------------
// Bases arrive in the library's INTERNAL form (msm_to_internal_kernel / msm_table_kernel): x || y, each coordinate
// x R mod p of the reduced-radix representation (fp29.h) packed into the arkworks word count, identity = zeros.
// The mixed additions run on 29/28-bit limbs with lazy bounds (ec29.h: 162 v_mad_u64_u32 per Fq product and no
// carry or compare instructions, against 128 mad + 128 addc + ~70 others for the 32-bit product); segment sums stay
// in that representation for the bucket reduction below.
template <class F>
__device__ __forceinline__ Affine29<F> load_internal(const uint32_t* __restrict__ bases, unsigned idx) {
  constexpr int PW = 2 * FieldOf<F>::WORDS;               // words per point
  uint32_t w[PW];
  const uint4* src = reinterpret_cast<const uint4*>(bases + (size_t)idx * PW);
#pragma unroll
  for (int i = 0; i < PW / 4; i++) {
    const uint4 v = src[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  return Affine29<F>::load(w);
}

// The same load in two halves -- the raw packed words now, the limbs when the addition needs them -- for a loop that
// keeps the NEXT point's words in registers while it adds the current one (msm_accumulate_lds_kernel).
template <class F>
struct RawPoint {
  uint4 v[2 * FieldOf<F>::WORDS / 4];
};
template <class F>
__device__ __forceinline__ RawPoint<F> load_raw(const uint32_t* __restrict__ bases, unsigned idx) {
  constexpr int PW = 2 * FieldOf<F>::WORDS;
  RawPoint<F> r;
  const uint4* src = reinterpret_cast<const uint4*>(bases + (size_t)idx * PW);
#pragma unroll
  for (int i = 0; i < PW / 4; i++) r.v[i] = src[i];
  return r;
}
template <class F>
__device__ __forceinline__ Affine29<F> unpack_raw(const RawPoint<F>& r) {
  constexpr int PW = 2 * FieldOf<F>::WORDS;
  uint32_t w[PW];
#pragma unroll
  for (int i = 0; i < PW / 4; i++) {
    w[4 * i] = r.v[i].x; w[4 * i + 1] = r.v[i].y; w[4 * i + 2] = r.v[i].z; w[4 * i + 3] = r.v[i].w;
  }
  return Affine29<F>::load(w);
}

// Segment t of bucket-window w -> its bucket and its range of the bucket's entries.  seg_off is the exclusive scan of
// the per-bucket segment counts k_b = ceil(cnt_b / 2^seg_log): the bucket is the LAST b with seg_off[b] <= t (empty
// buckets share their offset with their successor and are skipped by construction); segment j of k covers the ranks
// [j cnt / k, (j + 1) cnt / k).
struct SegRange {
  size_t bslot;        // (w << log_nb) + bucket
  unsigned first, cnt; // rank of the segment's first entry inside the bucket, entries in the segment
  unsigned j, k;       // this is segment j of the bucket's k
};
__device__ __forceinline__ SegRange msm_segment(const MsmGeom& g, unsigned w, unsigned t,
                                                const unsigned* __restrict__ counts,
                                                const unsigned* __restrict__ seg_off) {
  const unsigned* so = seg_off + ((size_t)w << g.log_nb);
  unsigned lo = 0, hi = 1u << g.log_nb;          // invariant: so[lo] <= t, (hi == nb or so[hi] > t)
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (so[mid] <= t) lo = mid; else hi = mid;
  }
  SegRange r;
  r.bslot = DG_IDX(1, ((size_t)w << g.log_nb) + lo, (size_t)g.bw << g.log_nb);
  const unsigned c = counts[r.bslot];
  const unsigned k = (c + (1u << g.seg_log) - 1) >> g.seg_log;
  const unsigned j = t - so[lo];
  r.first = (unsigned)(((uint64_t)j * c) / k);
  r.cnt = (unsigned)(((uint64_t)(j + 1) * c) / k) - r.first;
  r.j = j;
  r.k = k;
  return r;
}

// ---- in-workgroup bucket tree ---------------------------------------------------------------------------------
// The lanes of an accumulation workgroup hold the partial sums of CONSECUTIVE segments, i.e. runs of lanes belong to
// one bucket (~15 lanes per bucket of a 2^20-point table MSM).  Instead of writing one partial per segment and
// summing them in a separate, latency-bound finalize launch (a million full additions per MSM, 3 ms for G2 inside a
// proof), the workgroup adds the partials of every run in a tree, in place, in LDS columns ([coordinate word][lane]).
// The additions of a round are COMPACTED onto the low lanes (ballot + prefix counts), so whole waves drop out:
// 128 + 64 + 32 + 16 additions of a 256-lane workgroup are 2 + 1 + 1 + 1 wave-level additions, ~12 % on top of the
// 4 x 16 mixed additions of the accumulation itself.  A run that covers its whole bucket writes the BUCKET; a bucket
// that crosses a workgroup boundary leaves one partial per workgroup in the segment-sum array, at the slot of the
// run's first lane -- the bucket's first segment slot, then the first slot of every further workgroup:
__device__ __forceinline__ unsigned msm_nparts(unsigned first_slot, unsigned k, unsigned wg_log) {
  return k ? ((first_slot + k - 1) >> wg_log) - (first_slot >> wg_log) + 1 : 0;
}
__device__ __forceinline__ unsigned msm_part_slot(unsigned first_slot, unsigned s, unsigned wg_log) {
  return s ? ((first_slot >> wg_log) + s) << wg_log : first_slot;
}
template <class F, int BLOCK>
struct ColAcc {       // one lane's XYZZ29 in the LDS columns
  using S = typename FieldOf<F>::Store;
  static constexpr int WORDS = sizeof(S) / 4;
  uint32_t (*sh)[BLOCK];
  unsigned lane;
  __device__ __forceinline__ S get(int coord) const {
    S v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < WORDS; i++) w[i] = sh[coord * WORDS + i][lane];
    return v;
  }
  __device__ __forceinline__ void put(int coord, const S& v) const {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < WORDS; i++) sh[coord * WORDS + i][lane] = w[i];
  }
};
// Round-6 experiment switches for the tree's addition (same-call A/B in profiles/r6g_g1_tree_experiments.txt; none is the
// default): DG16_TREE_ADD_OUTLINE = the addition behind a call (the loop then compiles without the tree's 304 B of scratch),
// DG16_TREE_ADD_INTO = XYZZ29::add_into (U1 / S1 overwrite X1 / Y1 in place: the smallest live set).
template <class F, int BLOCK>
#ifdef DG16_TREE_ADD_OUTLINE
__device__ __attribute__((noinline))
#else
__device__ __forceinline__
#endif
void wg_tree_add(uint32_t (*sh)[BLOCK], unsigned a, unsigned b) {
#ifdef DG16_TREE_ADD_INTO
  XYZZ29<F>::add_into(ColAcc<F, BLOCK>{sh, a}, ColAcc<F, BLOCK>{sh, b});
#else
  XYZZ29<F>::add_acc(ColAcc<F, BLOCK>{sh, a}, ColAcc<F, BLOCK>{sh, a}, ColAcc<F, BLOCK>{sh, b});
#endif
}
// q: my index inside my run, el: end (exclusive, a lane index) of my run; lanes outside every run pass q = 0, el = lane + 1
template <class F, int BLOCK>
__device__ __forceinline__ void wg_bucket_tree(uint32_t (*sh)[BLOCK], unsigned short* list, unsigned* wcnt,
                                                         unsigned lane, unsigned q, unsigned el) {
  constexpr unsigned NW = BLOCK / 64;
  if (lane == 0) wcnt[NW] = 0;
  __syncthreads();
  atomicMax(&wcnt[NW], el - (lane - q));
  __syncthreads();
  const unsigned maxlen = wcnt[NW];
#pragma unroll 1
  for (unsigned d = 1; d < maxlen; d <<= 1) {
    const bool act = (q & (2 * d - 1)) == 0 && lane + d < el;
    const unsigned long long m = __ballot(act);
    const unsigned wv = lane >> 6;
    if ((lane & 63) == 0) wcnt[wv] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (unsigned i = 0; i < NW; i++) {
      const unsigned c = wcnt[i];
      base += i < wv ? c : 0u;
      total += c;
    }
    if (act) list[base + (unsigned)__popcll(m & ((1ull << (lane & 63)) - 1))] = (unsigned short)lane;
    __syncthreads();
    if (lane < total) {
      const unsigned a = list[lane];
      wg_tree_add<F, BLOCK>(sh, a, (unsigned)DG_IDX(16, a + d, BLOCK));
    }
    __syncthreads();
  }
}

// Several MSMs over the SAME scalars (the A, B1 and L queries of a proof share one digit sort) run as INSTANCES of one
// launch: blockIdx.y = inst * bw + w; the sort's arrays are indexed by the bucket-window w, everything an instance
// owns (segment sums, buckets, rows, window sums) by wy = blockIdx.y.  One launch = one ramp-down at the end instead
// of three, and the bucket reduction behind it is ONE chain of launches for all instances.
constexpr unsigned kMaxInst = 4;
struct MsmBases {
  const uint32_t* p[kMaxInst];
};
template <class F>
constexpr bool msm_acc_tree();
// log2 of the accumulation workgroup of coordinate field F (msm_accumulate_phase)
template <class F>
constexpr unsigned msm_acc_block_log() {
  if constexpr (sizeof(F) > 48) return sizeof(typename FieldOf<F>::Store) * 4 * 256 <= 80 * 1024 ? 8u : 7u;
  else if constexpr (!msm_acc_tree<F>()) return 8u;
#ifdef DG16_G1_BLOCK_LOG     // (round-6 experiment: workgroups of 128 lanes -- a smoother last round, more buckets to stitch)
  else return DG16_G1_BLOCK_LOG;
#else
  else return sizeof(typename FieldOf<F>::Store) * 4 * 256 <= 40 * 1024 ? 8u : 7u;   // G1: four workgroups' trees per CU
#endif
}
// Does the accumulation kernel of F add the partials of a bucket inside the workgroup (wg_bucket_tree)?  Coordinate fields
// up to this size do: the G1 of BN254.  G2 (the tree's Fq2 addition next to a loop at its register limit) and the 48-byte
// G1 fields were measured with it and are slower (CHANGELOG.md: rounds 3-4, profiles/r4d_ab.md): there every lane writes its
// partial and a throughput finalize adds the ~15 of a bucket.
#ifndef DG16_TREE_MAX_BYTES
#define DG16_TREE_MAX_BYTES 32
#endif
template <class F>
constexpr bool msm_acc_tree() {
  return sizeof(F) <= DG16_TREE_MAX_BYTES;
}
// log2 of the span of segment slots that share ONE partial (msm_part_slot): the workgroup with the tree, one slot without
template <class F>
constexpr unsigned msm_acc_wg_log() { return msm_acc_tree<F>() ? msm_acc_block_log<F>() : 0u; }

// Waves per SIMD the accumulation of a 48-byte coordinate field is compiled for: 2 = up to 256 VGPRs (the loop with the
// fused Y3 takes 178, no scratch).  Three (168 VGPRs) needs the unfused Y3 and was 4 % slower (fp29.h: rr_fuse_mul_sub).
#ifndef DG16_ACC48_WAVES
#define DG16_ACC48_WAVES 2
#endif
// (waves per SIMD = 4 caps the kernel at 128 VGPRs: the loop needs 108; what the tree's full addition needs beyond
// that is spilled INSIDE the tree, which a workgroup runs five times, not inside the loop it runs 16 x 4 times)
template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK, (sizeof(F) > 32 ? DG16_ACC48_WAVES : 4))
msm_accumulate_kernel(MsmBases bases, size_t n,
                                                              MsmGeom g, const unsigned* __restrict__ offsets,
                                                              const unsigned* __restrict__ counts,
                                                              const unsigned* __restrict__ seg_off,
                                                              const unsigned* __restrict__ seg_total,
                                                              const unsigned* __restrict__ entries,
                                                              XYZZ29<F>* __restrict__ seg_sum,
                                                              XYZZ29<F>* __restrict__ buckets,
                                                              unsigned long long* __restrict__ clk) {
  ClkProbe probe;
  probe.begin(clk);
  const unsigned w = blockIdx.y % g.bw;
  const uint32_t* __restrict__ base_tab = bases.p[blockIdx.y / g.bw];
  const unsigned lane = threadIdx.x;
  const unsigned t = blockIdx.x * BLOCK + lane;
  const bool live = t < seg_total[w];
  SegRange sr{};
  XYZZ29<F> acc = XYZZ29<F>::inf();
  if (live) {
    sr = msm_segment(g, w, t, counts, seg_off);
    const unsigned cnt = DG_OK(2, (size_t)offsets[sr.bslot] + sr.first + sr.cnt, g.region + 1) ? sr.cnt : 0u;
    const unsigned* e = entries + (size_t)w * g.region + offsets[sr.bslot] + sr.first;
    // Latency hiding: several waves per SIMD cover the dependent (entry -> point) gathers; only the 4-byte entry
    // index is fetched one iteration ahead (a second point in registers costs the whole 128-register budget of four
    // waves and six scratch accesses per iteration: measured equal, profiles/r4b_ab_variants.md -- removed).
    unsigned cur = e[0];
    for (unsigned j = 0; j < cnt; j++) {
      unsigned nxt = (j + 1 < cnt) ? e[j + 1] : 0u;
      const Affine29<F> p = load_internal<F>(base_tab, DG_IDX(3, cur & 0x7fffffffu, g.region));
      acc = acc.madd(p, cur >> 31);
      cur = nxt;
    }
  }
  const size_t bucket_slot = ((size_t)blockIdx.y << g.log_nb) + (sr.bslot & (((size_t)1 << g.log_nb) - 1));
  if constexpr (msm_acc_tree<F>()) {
    using CA = ColAcc<F, BLOCK>;
    __shared__ uint32_t sh[4 * CA::WORDS][BLOCK];           // the partials of the bucket tree: 36 KiB for a 254-bit field
    __shared__ unsigned short list[BLOCK];
    __shared__ unsigned wcnt[BLOCK / 64 + 1];
    const CA me{sh, lane};
    me.put(0, acc.x); me.put(1, acc.y); me.put(2, acc.zz); me.put(3, acc.zzz);
    // my run: the lanes of this workgroup that hold segments of my bucket
    const unsigned hl = live ? (lane > sr.j ? lane - sr.j : 0u) : lane;
    const unsigned el = live ? (lane - sr.j + sr.k < (unsigned)BLOCK ? lane + sr.k - sr.j : (unsigned)BLOCK) : lane + 1;
    wg_bucket_tree<F, BLOCK>(sh, list, wcnt, lane, lane - hl, el);
    if (live && lane == hl) {
      const XYZZ29<F> v{me.get(0), me.get(1), me.get(2), me.get(3)};
      if (sr.j == 0 && lane + sr.k <= (unsigned)BLOCK) buckets[DG_IDX(5, bucket_slot, (size_t)gridDim.y << g.log_nb)] = v;   // the whole bucket
      else seg_sum[(size_t)blockIdx.y * g.seg_cap + DG_IDX(4, t, g.seg_cap)] = v;      // one partial per (bucket, workgroup): msm_part_slot
    }
  } else if (live) {
    if (sr.k == 1) buckets[DG_IDX(5, bucket_slot, (size_t)gridDim.y << g.log_nb)] = acc;   // a one-segment bucket needs no finalize
    else seg_sum[(size_t)blockIdx.y * g.seg_cap + DG_IDX(4, t, g.seg_cap)] = acc;
  }
  probe.end(clk);
}

// ---- 4 (G2): the same segment accumulation with the accumulator staged through LDS -------------------------
// An Fq2 mixed addition with its four accumulator coordinates in registers needs more than 256 VGPRs (one wave per
// SIMD, AGPR spills); with the coordinates in LDS between uses (layout [coordinate word][lane]: consecutive lanes ->
// consecutive banks, conflict-free ds_read/write_b32) the live set is the loaded point and ~6 temporaries.
// 4 coordinates x 2 N words x BLOCK lanes = 72 KiB for BN254 Fq2 at BLOCK = 256 (two workgroups per CU, 160 KiB LDS).
// (9-limb Fq2 -- BN254 -- only: the 14-limb curves run msm_accumulate_steps_kernel below)
template <class F, int BLOCK, int TU = 0>
__global__ void __launch_bounds__(BLOCK, (BLOCK == 256 ? 2 : 1))
msm_accumulate_lds_kernel(MsmBases bases, size_t n, MsmGeom g,
                          const unsigned* __restrict__ offsets, const unsigned* __restrict__ counts,
                          const unsigned* __restrict__ seg_off, const unsigned* __restrict__ seg_total,
                          const unsigned* __restrict__ entries, XYZZ29<F>* __restrict__ seg_sum,
                          XYZZ29<F>* __restrict__ buckets, unsigned long long* __restrict__ clk) {
  using FO = FieldOf<F>;
  using S = typename FO::Store;
  constexpr int BS = FO::BS;
  constexpr int WORDS = sizeof(S) / 4;
  __shared__ uint32_t sh[4 * WORDS][BLOCK];
  ClkProbe probe;
  probe.begin(clk);
  const unsigned lane = threadIdx.x;
  auto ld = [&](int coord) {
    S v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < WORDS; i++) w[i] = sh[coord * WORDS + i][lane];
    return v;
  };
  auto st = [&](int coord, const S& v) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < WORDS; i++) sh[coord * WORDS + i][lane] = w[i];
  };
#define DG_STAGE() asm volatile("" ::: "memory")   /* keep LDS reloads where they are written */
  const unsigned w = blockIdx.y % g.bw;
  const uint32_t* __restrict__ base_tab = bases.p[blockIdx.y / g.bw];
  const unsigned t = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = t < seg_total[w];
  SegRange sr{};
  if (live) sr = msm_segment(g, w, t, counts, seg_off);
  const unsigned cnt = live && DG_OK(2, (size_t)offsets[sr.bslot] + sr.first + sr.cnt, g.region + 1) ? sr.cnt : 0u;
  const unsigned* e = entries + (size_t)w * g.region + (live ? offsets[sr.bslot] + sr.first : 0u);
  bool inf = true;
  // Gather latency.  This kernel runs two waves per SIMD (LDS-bound) with registers to spare (175 of 256 for BN254), so
  // for 64-byte coordinates the NEXT point's 32 packed words are gathered while the current addition runs (its entry
  // index was fetched an iteration earlier, the index after it is fetched now): the dependent entry -> point load no
  // longer sits in front of every addition.  An index past the segment is 0 (a valid row).  48-byte-field Fq2 (252
  // registers, one wave) has no room for it and loads at the top of the iteration as before.
  // Measured against the plain loop in round 4 (profiles/r4a_ab_variants.md): 2.871-2.883 ms per launch against
  // 2.877-2.909, same box, same call -- inside the noise, ahead on both passes: kept, the build switch is gone.
  constexpr bool PREFETCH = sizeof(F) <= 64;
  unsigned cur = cnt ? e[0] : 0u;
  unsigned nxt = cnt > 1 ? e[1] : 0u;
  RawPoint<F> raw_cur{};
  if (PREFETCH && cnt) raw_cur = load_raw<F>(base_tab, DG_IDX(3, cur & 0x7fffffffu, g.region));
  for (unsigned j = 0; j < cnt; j++) {
    const unsigned nn = (j + 2 < cnt) ? e[j + 2] : 0u;
    RawPoint<F> raw_nxt{};
    if (PREFETCH) raw_nxt = load_raw<F>(base_tab, DG_IDX(3, nxt & 0x7fffffffu, g.region));
    else raw_cur = load_raw<F>(base_tab, DG_IDX(3, cur & 0x7fffffffu, g.region));
    const Affine29<F> q = unpack_raw<F>(raw_cur);
    const bool negate = cur >> 31;
    cur = nxt;
    nxt = nn;
    if (PREFETCH) raw_cur = raw_nxt;
    if (q.is_inf()) continue;
    const auto nqy = neg(q.y);
    const auto qy = select(negate, nqy, q.y.template as<decltype(nqy)::Bound, decltype(nqy)::Limb>());
    if (inf) {
      st(0, q.x.template as<BS, 1>()); st(1, fit<BS>(qy)); st(2, FO::one()); st(3, FO::one());
      inf = false;
      continue;
    }
    const auto p_ = norm(q.x * ld(2) - ld(0));          // U2 - X1
    DG_STAGE();
    const auto r_ = norm(qy * ld(3) - ld(1));           // S2 - Y1
    DG_STAGE();
    if (is_zero(p_)) {
      if (is_zero(r_)) {
        const XYZZ29<F> d = XYZZ29<F>::dbl_affine(q.x, qy);
        st(0, d.x); st(1, d.y); st(2, d.zz); st(3, d.zzz);
      } else {
        inf = true;
      }
      continue;
    }
    const auto pp = sqr(p_);
    const auto ppp = p_ * pp;
    DG_STAGE();
    st(2, fit<BS>(ld(2) * pp));
    DG_STAGE();
    st(3, fit<BS>(ld(3) * ppp));
    DG_STAGE();
    const auto q_ = ld(0) * pp;
    DG_STAGE();
    const auto x3 = fit<BS>(sqr(r_) - (ppp + dbl(q_)));
    st(0, x3);
    DG_STAGE();
    const auto y3 = fit<BS>(mul_sub(r_, q_ - x3, ppp, ld(1)));
    st(1, y3);
    DG_STAGE();
  }
  if (live) {
    // (no in-workgroup tree here: msm_acc_tree) one partial per segment; a one-segment bucket is written directly
    XYZZ29<F> out = XYZZ29<F>::inf();
    if (!inf) out = XYZZ29<F>{ld(0), ld(1), ld(2), ld(3)};
    if (sr.k == 1) buckets[((size_t)blockIdx.y << g.log_nb) + (sr.bslot & (((size_t)1 << g.log_nb) - 1))] = out;
    else seg_sum[(size_t)blockIdx.y * g.seg_cap + DG_IDX(4, t, g.seg_cap)] = out;
  }
  probe.end(clk);
#undef DG_STAGE
}

// ---- 4 (G2 of the 14-limb curves): the same accumulation as a STEP LOOP over three product sites ---------------------
// Inlined, an Fq2 mixed addition of a 14-limb curve is a 100-KB loop (eleven Fq2 products) run by one wave per SIMD against
// the 64-KB instruction cache two CUs share: 9 ms per 2^20-point launch on some boxes of the pool, 18 on others, same
// binary; products behind calls cost a dozen scratch accesses each (14.4 ms everywhere).  Here a mixed addition is a loop
// of NINE steps over THREE sites -- one Fq2 product (visited six times), one Fq2 square (twice) and the fused
// Y3 = R (Q - X3) - PPP Y1 (once) -- with a wave-uniform switch in front of a site to route its operands and one behind it
// to route the result:
//     0  P = x2 ZZ - X1      1  R = y2 ZZZ - Y1      2  PP = P^2      3  PPP = P PP      4  ZZ <- ZZ PP
//     5  ZZZ <- ZZZ PPP      6  Q = X1 PP            7  X3 = R^2 - PPP - 2 Q (-> X1), T = Q - X3      8  Y1 <- R T - PPP Y1
// The same 10 584 v_mad_u64_u32 per addition as the straight-line form, in a loop that stays in the instruction cache;
// values and the order of operations inside a product are those of the straight-line form (parity tests unchanged).
// The accumulator (X1, Y1, ZZ, ZZZ) lives in LDS columns; the four temporaries (P -> Q, R, PP -> T, PPP) in a FILE of
// accumulation registers at FIXED numbers a[kAccFileBase + 28 slot + i] named in asm statements (gfx950: 256 AGPRs next to
// the 256 VGPRs of a wave at one wave per SIMD) -- machine state the compiler does not see: as C++ values (in VGPRs, or in
// AGPRs through "=a" / "+a" operands) the step switch turned them into phis that hipcc merged with 270-330 copies per visit
// of a site against the 84 the routing needs.
constexpr int kAccFileBase = 144;
// Round 6 -- what round 5's abort was: a clobber list is NOT a reservation.  The first form named
// two registers ("a144", "a255": enough for the resource accounting) and hipcc, which needed 160 spill registers in the
// step-loop form of the 14-limb G2 FINALIZE, put sixteen of its own values -- hoisted operand addresses -- into
// a[144..159]; acc_set<0> then overwrote them and the next reload used field limbs as an address
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; reproduced at the first call on the all-equal-points shape of
// dmsm/mod.rs:155-159, profiles/r6a_*).  tests/test_kernel_isa.py could not see it: in a disassembly the compiler's
// v_accvgpr_write looks like acc_set's.  Now (i) every write NAMES its register as clobbered, so the compiler never keeps a
// value of its own in a file register across an acc_set; (ii) the kernel declares all 112; (iii) tools/check_agpr_file.py
// reads the compiler's assembly (-save-temps), where the asm statements are bracketed by ASMSTART / ASMEND, and FAILS THE
// BUILD (csrc/Makefile) if any instruction of the compiler's own touches a[144..255] in a kernel that uses the file.
#define DG_ACC_REGS_LO(X) X(144) X(145) X(146) X(147) X(148) X(149) X(150) X(151) X(152) X(153) X(154) X(155) X(156) X(157) X(158) X(159) X(160) X(161) X(162) X(163) X(164) X(165) X(166) X(167) X(168) X(169) X(170) X(171) X(172) X(173) X(174) X(175) X(176) X(177) X(178) X(179) X(180) X(181) X(182) X(183) X(184) X(185) X(186) X(187) X(188) X(189) X(190) X(191) X(192) X(193) X(194) X(195) X(196) X(197) X(198) X(199)
#define DG_ACC_REGS_HI(X) X(200) X(201) X(202) X(203) X(204) X(205) X(206) X(207) X(208) X(209) X(210) X(211) X(212) X(213) X(214) X(215) X(216) X(217) X(218) X(219) X(220) X(221) X(222) X(223) X(224) X(225) X(226) X(227) X(228) X(229) X(230) X(231) X(232) X(233) X(234) X(235) X(236) X(237) X(238) X(239) X(240) X(241) X(242) X(243) X(244) X(245) X(246) X(247) X(248) X(249) X(250) X(251) X(252) X(253) X(254)
// -DDG16_ACC_CLOBBER_R5: round 5's declaration (two registers named, nothing on the writes) -- the NEGATIVE CONTROL of
// tools/abort_hunt.sh and tests/test_kernel_isa.py: built that way the 14-limb G2 finalize collides again, and
// tools/check_agpr_file.py must say so (the Makefile then refuses the object: pass AGPR_CHECK=../../tools/true.py to get the
// library anyway).
#ifdef DG16_ACC_CLOBBER_R5
#define DG_ACC_WRITE_CLOBBER(n)
#else
#define DG_ACC_WRITE_CLOBBER(n) : "a" #n
#endif
template <int R> struct AccReg;
#define X(n)                                                                                             \
  template <> struct AccReg<n> {                                                                         \
    static __device__ __forceinline__ void w(uint32_t v) {                                               \
      asm volatile("v_accvgpr_write_b32 a" #n ", %0" ::"v"(v) DG_ACC_WRITE_CLOBBER(n));                  \
    }                                                                                                    \
    static __device__ __forceinline__ uint32_t r() {                                                     \
      uint32_t v;                                                                                        \
      asm volatile("v_accvgpr_read_b32 %0, a" #n : "=v"(v));                                             \
      return v;                                                                                          \
    }                                                                                                    \
  };
DG_ACC_REGS_LO(X) DG_ACC_REGS_HI(X) X(255)
#undef X
// all registers of the file, for the kernel's one declaration (resource accounting: the wave is allocated them)
#define X(n) "a" #n,
#ifdef DG16_ACC_CLOBBER_R5
#define DG_ACC_FILE_CLOBBERS "a144", "a255"
#else
#define DG_ACC_FILE_CLOBBERS DG_ACC_REGS_LO(X) DG_ACC_REGS_HI(X) "a255"
#endif
template <int BASE, int N, class P, int B, int... I>
__device__ __forceinline__ void acc_set_seq(const Fe2<P, B, 1>& v, std::integer_sequence<int, I...>) {
  ((AccReg<BASE + I>::w(v.c0.l[I]), AccReg<BASE + N + I>::w(v.c1.l[I])), ...);
}
template <int BASE, int N, class P, int B, int... I>
__device__ __forceinline__ void acc_get_seq(Fe2<P, B, 1>& v, std::integer_sequence<int, I...>) {
  ((v.c0.l[I] = AccReg<BASE + I>::r(), v.c1.l[I] = AccReg<BASE + N + I>::r()), ...);
}
template <int SLOT, class P, int B>
__device__ __forceinline__ void acc_set(const Fe2<P, B, 1>& v) {
  constexpr int N = RR<P>::N;
  static_assert(kAccFileBase + 2 * N * (SLOT + 1) <= 256, "slot inside the file");
  acc_set_seq<kAccFileBase + 2 * N * SLOT, N>(v, std::make_integer_sequence<int, N>{});
}
template <int SLOT, class P, int B>
__device__ __forceinline__ Fe2<P, B, 1> acc_get() {
  constexpr int N = RR<P>::N;
  Fe2<P, B, 1> v;
  acc_get_seq<kAccFileBase + 2 * N * SLOT, N>(v, std::make_integer_sequence<int, N>{});
  return v;
}
// d += b (full XYZZ addition, XYZZ29::add_into) as a STEP LOOP over the same three product sites: the 14-limb G2 finalize
// (msm_finalize_lds_kernel: two lanes per bucket summing the bucket's partials) inlined a 144-KB addition -- 35 900
// instructions with its doubling branch -- and ran at 16 % of its issue rate on the slow boxes of the pool.
//     0  U1 = X1 ZZ2 -> X1      1  S1 = Y1 ZZZ2 -> Y1      2  P = X2 ZZ1 - U1      3  R = Y2 ZZZ1 - S1
//     4  PP = P^2               5  PPP = P PP              6  T = ZZ1 ZZ2          7  ZZ3 = T PP
//     8  T = ZZZ1 ZZZ2          9  ZZZ3 = T PPP           10  Q = U1 PP           11  X3 = R^2 - PPP - 2 Q
//    12  Y3 = R (Q - X3) - PPP S1
// d: accumulator in LDS columns (get / put); b: read-only operand behind an accessor (memory or LDS), intact throughout, so
// the rare d == b case doubles b.  Needs the accumulation-register file of the calling kernel (kAccFileBase).
template <class F, class D, class B>
__device__ __forceinline__ void xyzz_add_into_steps(const D& d, const B& b_) {
  using FO = FieldOf<F>;
  using P = typename FO::Params;
  constexpr int BS = FO::BS;
  constexpr int BG = 640;
  using G = Fe2<P, BG, 1>;
  if (limbs_all_zero(b_.get(2))) return;
  if (limbs_all_zero(d.get(2))) {
    d.put(0, b_.get(0)); d.put(1, b_.get(1)); d.put(2, b_.get(2)); d.put(3, b_.get(3));
    return;
  }
  B b = b_;
  auto dg = [&](int c) { return d.get(c).template as<BG, 1>(); };
  auto bg = [&](int c) { return b.get(c).template as<BG, 1>(); };
  int special = 0;
  bool p_zero = false;
#pragma unroll 1
  for (int step = 0; step < 13; step++) {
    asm volatile("" : "+s"(step));          // opaque: the sites must not be cloned per step
    b.launder();                            // ... and the operand's 112 word addresses not hoisted out of the loop (they
                                            // were: 224 registers of pointers, 932 B of scratch per lane)
    if (step == 4 || step == 11) {
      const G a = step == 4 ? acc_get<0, P, BG>() : acc_get<1, P, BG>();
      const auto c = sqr(a);
      if (step == 4) {
        acc_set<2>(c.template as<BG, 1>());                             // PP
      } else {
        const auto ppp = acc_get<3, P, 128>(), q_ = acc_get<0, P, 128>();
        const auto x3 = fit<BS>(c - (ppp + dbl(q_)));
        d.put(0, x3);
        acc_set<2>(fit<BG>(q_ - x3));                                   // Q - X3
      }
    } else if (step == 12) {
      const auto r_ = acc_get<1, P, BG>(), t_ = acc_get<2, P, BG>();
      const auto ppp = acc_get<3, P, 128>();
      d.put(1, fit<BS>(mul_sub(r_, t_, ppp, d.get(1))));                // R (Q - X3) - PPP S1
    } else {
      G a, bb;
      switch (step) {
        case 0: a = dg(0); bb = bg(2); break;                            // X1 ZZ2
        case 1: a = dg(1); bb = bg(3); break;                            // Y1 ZZZ2
        case 2: a = bg(0); bb = dg(2); break;                            // X2 ZZ1
        case 3: a = bg(1); bb = dg(3); break;                            // Y2 ZZZ1
        case 5: a = acc_get<0, P, BG>(); bb = acc_get<2, P, BG>(); break;   // P PP
        case 6: a = dg(2); bb = bg(2); break;                            // ZZ1 ZZ2
        case 7: a = acc_get<0, P, BG>(); bb = acc_get<2, P, BG>(); break;   // (ZZ1 ZZ2) PP
        case 8: a = dg(3); bb = bg(3); break;                            // ZZZ1 ZZZ2
        case 9: a = acc_get<0, P, BG>(); bb = acc_get<3, P, BG>(); break;   // (ZZZ1 ZZZ2) PPP
        default: a = dg(0); bb = acc_get<2, P, BG>(); break;             // U1 PP
      }
      const auto c = a * bb;
      switch (step) {
        case 0: d.put(0, c.template as<BS, 1>()); break;                 // U1
        case 1: d.put(1, c.template as<BS, 1>()); break;                 // S1
        case 2: {
          const auto p_ = fit<BG>(c - d.get(0));
          p_zero = is_zero_compact(p_);
          acc_set<0>(p_);
          break;
        }
        case 3: {
          const auto r_ = fit<BG>(c - d.get(1));
          if (p_zero) special = is_zero_compact(r_) ? 1 : 2;
          acc_set<1>(r_);
          break;
        }
        case 5: acc_set<3>(c.template as<BG, 1>()); break;               // PPP
        case 6: acc_set<0>(c.template as<BG, 1>()); break;
        case 7: d.put(2, c.template as<BS, 1>()); break;                 // ZZ3
        case 8: acc_set<0>(c.template as<BG, 1>()); break;
        case 9: d.put(3, c.template as<BS, 1>()); break;                 // ZZZ3
        default: acc_set<0>(c.template as<BG, 1>()); break;              // Q
      }
      if (special) break;
    }
  }
  if (special == 1) {
    const XYZZ29<F> t = XYZZ29<F>{b.get(0), b.get(1), b.get(2), b.get(3)}.dbl_pt();
    d.put(0, t.x); d.put(1, t.y); d.put(2, t.zz); d.put(3, t.zzz);
  } else if (special == 2) {
    d.put(2, FO::zero());                                              // the identity: zz = 0
  }
}

template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK, 1)
msm_accumulate_steps_kernel(MsmBases bases, size_t n, MsmGeom g,
                            const unsigned* __restrict__ offsets, const unsigned* __restrict__ counts,
                            const unsigned* __restrict__ seg_off, const unsigned* __restrict__ seg_total,
                            const unsigned* __restrict__ entries, XYZZ29<F>* __restrict__ seg_sum,
                            XYZZ29<F>* __restrict__ buckets, unsigned long long* __restrict__ clk) {
  ClkProbe probe;
  probe.begin(clk);
  using FO = FieldOf<F>;
  using P = typename FO::Params;
  using S = typename FO::Store;
  constexpr int BS = FO::BS;
  constexpr int BG = 640;                    // every operand of a site is below 10 p (P, R, Q - X3: < 9.3 p)
  using G = Fe2<P, BG, 1>;
  static_assert(kAccFileBase + 4 * 2 * RR<P>::N <= 256, "four temporaries in the accumulation registers");
  constexpr int WORDS = sizeof(S) / 4;
  __shared__ uint32_t sh[4 * WORDS][BLOCK];
  const unsigned lane = threadIdx.x;
  auto ld = [&](int coord) {
    S v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < WORDS; i++) w[i] = sh[coord * WORDS + i][lane];
    return v;
  };
  auto ldg = [&](int coord) { return ld(coord).template as<BG, 1>(); };
  auto st = [&](int coord, const S& v) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < WORDS; i++) sh[coord * WORDS + i][lane] = w[i];
  };
  const unsigned w = blockIdx.y % g.bw;
  const uint32_t* __restrict__ base_tab = bases.p[blockIdx.y / g.bw];
  const unsigned t = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = t < seg_total[w];
  SegRange sr{};
  if (live) sr = msm_segment(g, w, t, counts, seg_off);
  const unsigned cnt = live && DG_OK(2, (size_t)offsets[sr.bslot] + sr.first + sr.cnt, g.region + 1) ? sr.cnt : 0u;
  const unsigned* e = entries + (size_t)w * g.region + (live ? offsets[sr.bslot] + sr.first : 0u);
  asm volatile("" ::: DG_ACC_FILE_CLOBBERS);   // the temporaries' registers belong to this wave (acc_set / acc_get)
  bool inf = true;
  unsigned cur = cnt ? e[0] : 0u;
  for (unsigned j = 0; j < cnt; j++) {
    const unsigned nxt = (j + 1 < cnt) ? e[j + 1] : 0u;
    const unsigned ent = cur;
    cur = nxt;
    const bool negate = ent >> 31;
    {
      const Affine29<F> q = load_internal<F>(base_tab, DG_IDX(3, ent & 0x7fffffffu, g.region));
      if (q.is_inf()) continue;
      const auto nqy = neg(q.y);
      const auto qy = select(negate, nqy, q.y.template as<decltype(nqy)::Bound, decltype(nqy)::Limb>());
      if (inf) {
        st(0, q.x.template as<BS, 1>()); st(1, fit<BS>(qy)); st(2, FO::one()); st(3, FO::one());
        inf = false;
        continue;
      }
      acc_set<0>(q.x.template as<BG, 1>());
      acc_set<1>(fit<BG>(qy));
    }
    int special = 0;                          // 1: the same point again (double it), 2: its inverse (identity)
    bool p_zero = false;
#pragma unroll 1
    for (int step = 0; step < 9; step++) {
      asm volatile("" : "+s"(step));          // opaque: the sites must not be cloned per step
      if (step == 2 || step == 7) {
        // ---- the squaring site: PP = P^2, then X3 = R^2 - PPP - 2 Q
        const G a = step == 2 ? acc_get<0, P, BG>() : acc_get<1, P, BG>();
        const auto c = sqr(a);
        if (step == 2) {
          acc_set<2>(c.template as<BG, 1>());
        } else {
          const auto ppp = acc_get<3, P, 128>(), q_ = acc_get<0, P, 128>();   // products: below 2 p
          const auto x3 = fit<BS>(c - (ppp + dbl(q_)));
          st(0, x3);
          acc_set<2>(fit<BG>(q_ - x3));                                 // Q - X3
        }
      } else if (step == 8) {
        // ---- the fused site: Y3 = R (Q - X3) - PPP Y1, one reduction per component
        const auto r_ = acc_get<1, P, BG>(), d_ = acc_get<2, P, BG>();
        const auto ppp = acc_get<3, P, 128>();
        st(1, fit<BS>(mul_sub(r_, d_, ppp, ld(1))));
      } else {
        // ---- the product site
        G a, b;
        switch (step) {
          case 0: a = acc_get<0, P, BG>(); b = ldg(2); break;              // x2 ZZ
          case 1: a = acc_get<1, P, BG>(); b = ldg(3); break;              // y2 ZZZ
          case 3: a = acc_get<0, P, BG>(); b = acc_get<2, P, BG>(); break;  // P PP
          case 4: a = ldg(2); b = acc_get<2, P, BG>(); break;              // ZZ PP
          case 5: a = ldg(3); b = acc_get<3, P, BG>(); break;              // ZZZ PPP
          default: a = ldg(0); b = acc_get<2, P, BG>(); break;             // X1 PP
        }
        const auto c = a * b;
        switch (step) {
          case 0: {
            const auto p_ = fit<BG>(c - ld(0));                           // P = U2 - X1
            p_zero = is_zero_compact(p_);
            acc_set<0>(p_);
            break;
          }
          case 1: {
            const auto r_ = fit<BG>(c - ld(1));                           // R = S2 - Y1
            if (p_zero) special = is_zero_compact(r_) ? 1 : 2;
            acc_set<1>(r_);
            break;
          }
          case 3: acc_set<3>(c.template as<BG, 1>()); break;            // PPP
          case 4: st(2, c.template as<BS, 1>()); break;                   // ZZ3
          case 5: st(3, c.template as<BS, 1>()); break;                   // ZZZ3
          default: acc_set<0>(c.template as<BG, 1>()); break;           // Q
        }
        if (special) break;
      }
    }
    if (special == 1) {
      const Affine29<F> q2 = load_internal<F>(base_tab, ent & 0x7fffffffu);
      const auto nq2 = neg(q2.y);
      const auto qy2 = select(negate, nq2, q2.y.template as<decltype(nq2)::Bound, decltype(nq2)::Limb>());
      const XYZZ29<F> d = XYZZ29<F>::dbl_affine(q2.x, qy2);
      st(0, d.x); st(1, d.y); st(2, d.zz); st(3, d.zzz);
    } else if (special == 2) {
      inf = true;
    }
  }
  if (live) {
    XYZZ29<F> out = XYZZ29<F>::inf();
    if (!inf) out = XYZZ29<F>{ld(0), ld(1), ld(2), ld(3)};
    if (sr.k == 1) buckets[((size_t)blockIdx.y << g.log_nb) + (sr.bslot & (((size_t)1 << g.log_nb) - 1))] = out;
    else seg_sum[(size_t)blockIdx.y * g.seg_cap + DG_IDX(4, t, g.seg_cap)] = out;
  }
  probe.end(clk);
}

// arkworks-form bases (C ABI) -> internal form for the accumulation kernels (plain dg16_msm: one pass per call,
// 2 field products per point against ~10 W in the accumulation; resident keys convert once, in the table builder)
template <class F>
__global__ void __launch_bounds__(256) msm_to_internal_kernel(const Affine<F>* __restrict__ in, size_t n,
                                                               uint32_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int PW = 2 * FieldOf<F>::WORDS;
  uint32_t w[PW];
  affine_to_internal(in[i], w);
  uint4* dst = reinterpret_cast<uint4*>(out + i * PW);
#pragma unroll
  for (int k = 0; k < PW / 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// ---- wave-cooperative group operations (single-chain phases: Horner tail, s*A / r*B1) ------------------------
// A lone lane takes ~10 us (G1) / ~40 us (G2) per dependent group operation; these phases are chains of such
// operations with little parallelism, so one WAVE runs each chain and spreads the independent products of an
// operation over its lanes: the operands are uniform across the wave, slot = lane / 4 picks the product, an Fq2
// product is itself split over three lanes of the quad (Karatsuba), results are shared with readlane.
template <class P>
__device__ __forceinline__ Fp<P> lane_bcast(const Fp<P>& v, int src) {   // src: wave-uniform lane index
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < Fp<P>::NL; i++) r.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)v.l[i], src);
  return r;
}
template <class F>
__device__ __forceinline__ Fp2<F> lane_bcast(const Fp2<F>& v, int src) {
  return {lane_bcast(v.c0, src), lane_bcast(v.c1, src)};
}
template <class P>
__device__ __forceinline__ Fp<P> lane_get(const Fp<P>& v, int src) {     // src: per-lane index
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < Fp<P>::NL; i++) r.l[i] = (uint32_t)__shfl((int)v.l[i], src);
  return r;
}
// product per slot (slot = lane / 4; the operands must be equal across the quad).  The product is a CALL
// (Fp::mul_call): these chains run once per proof on one wave, so their cost is dependent issue + instruction fetch
// of cold code -- with every product inlined the s*A kernel was 180 KB and the proof assembly 210 KB of straight-line
// code (0.46 ms for ~40 us of arithmetic); a call keeps an addition at ~4 KB.
template <class P>
__device__ __forceinline__ Fp<P> slot_mul(const Fp<P>& a, const Fp<P>& b) { return Fp<P>::mul_call(a, b); }
template <class F>
__device__ __forceinline__ Fp2<F> slot_mul(const Fp2<F>& a, const Fp2<F>& b) {
  const unsigned q = __lane_id() & 3;
  const F x = F::select(q == 0, a.c0, F::select(q == 1, a.c1, a.c0 + a.c1));
  const F y = F::select(q == 0, b.c0, F::select(q == 1, b.c1, b.c0 + b.c1));
  const F t = F::mul_call(x, y);
  const int base = (int)(__lane_id() & ~3u);
  const F t0 = lane_get(t, base), t1 = lane_get(t, base + 1), t2 = lane_get(t, base + 2);
  return {t0 - fq2_beta_mul(t1), t2 - t0 - t1};      // u^2 = -BETA (fp2.h)
}
// 2 * p with p (and the result) uniform across the wave                 (dbl-2008-s-1, a = 0)
template <class F>
__device__ __forceinline__ XYZZ<F> dbl_wave(const XYZZ<F>& p) {
  if (p.is_inf()) return p;
  const unsigned slot = __lane_id() >> 2;
  const F u = p.y.dbl();
  // level 1: v = u^2 | xx = x^2
  const F a1 = F::select(slot == 0, u, p.x);
  const F r1 = slot_mul(a1, a1);
  const F v = lane_bcast(r1, 0), xx = lane_bcast(r1, 4);
  const F m = xx.dbl() + xx;
  // level 2: w = u v | s = x v | m^2 | zz' = v zz
  const F a2 = F::select(slot == 0, u, F::select(slot == 1, p.x, F::select(slot == 2, m, v)));
  const F b2 = F::select(slot <= 1, v, F::select(slot == 2, m, p.zz));
  const F r2 = slot_mul(a2, b2);
  const F w = lane_bcast(r2, 0), sv = lane_bcast(r2, 4), mm = lane_bcast(r2, 8), zz3 = lane_bcast(r2, 12);
  const F x3 = mm - sv.dbl();
  // level 3: m (s - x3) | w y | zzz' = w zzz
  const F a3 = F::select(slot == 0, m, w);
  const F b3 = F::select(slot == 0, sv - x3, F::select(slot == 1, p.y, p.zzz));
  const F r3 = slot_mul(a3, b3);
  const F y3 = lane_bcast(r3, 0) - lane_bcast(r3, 4);
  return {x3, y3, zz3, lane_bcast(r3, 8)};
}

// p + o, both (and the result) uniform across the wave: 14 products in 4 levels       (add-2008-s)
template <class F>
__device__ __forceinline__ XYZZ<F> add_wave(const XYZZ<F>& p, const XYZZ<F>& o) {
  if (o.is_inf()) return p;
  if (p.is_inf()) return o;
  const unsigned slot = __lane_id() >> 2;
  // level 1: u1 = x1 zz2 | u2 = x2 zz1 | s1 = y1 zzz2 | s2 = y2 zzz1
  const F a1 = F::select(slot == 0, p.x, F::select(slot == 1, o.x, F::select(slot == 2, p.y, o.y)));
  const F b1 = F::select(slot == 0, o.zz, F::select(slot == 1, p.zz, F::select(slot == 2, o.zzz, p.zzz)));
  const F r1 = slot_mul(a1, b1);
  const F u1 = lane_bcast(r1, 0), u2 = lane_bcast(r1, 4), s1 = lane_bcast(r1, 8), s2 = lane_bcast(r1, 12);
  const F pd = u2 - u1, rd = s2 - s1;
  if (pd.is_zero()) {
    if (rd.is_zero()) return dbl_wave(p);
    return XYZZ<F>::inf();
  }
  // level 2: pp = p^2 | rr = r^2 | zz1 zz2 | zzz1 zzz2
  const F a2 = F::select(slot == 0, pd, F::select(slot == 1, rd, F::select(slot == 2, p.zz, p.zzz)));
  const F b2 = F::select(slot == 0, pd, F::select(slot == 1, rd, F::select(slot == 2, o.zz, o.zzz)));
  const F r2 = slot_mul(a2, b2);
  const F pp = lane_bcast(r2, 0), rr = lane_bcast(r2, 4), zzp = lane_bcast(r2, 8), zzzp = lane_bcast(r2, 12);
  // level 3: ppp = p pp | q = u1 pp | zz3 = (zz1 zz2) pp
  const F a3 = F::select(slot == 0, pd, F::select(slot == 1, u1, zzp));
  const F r3 = slot_mul(a3, pp);
  const F ppp = lane_bcast(r3, 0), q = lane_bcast(r3, 4), zz3 = lane_bcast(r3, 8);
  const F x3 = rr - ppp - q.dbl();
  // level 4: r (q - x3) | s1 ppp | zzz3 = (zzz1 zzz2) ppp
  const F a4 = F::select(slot == 0, rd, F::select(slot == 1, s1, zzzp));
  const F b4 = F::select(slot == 0, q - x3, ppp);
  const F r4 = slot_mul(a4, b4);
  return {x3, lane_bcast(r4, 0) - lane_bcast(r4, 4), zz3, lane_bcast(r4, 8)};
}
// k * p by double-and-add on one wave; k = NW little-endian 32-bit words (plain integer), uniform
template <class F, int NW>
__device__ __forceinline__ XYZZ<F> scalar_mul_wave(const XYZZ<F>& p, const uint32_t* k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = NW * 32 - 1; i >= 0; i--) {
    acc = dbl_wave(acc);
    if ((k[i / 32] >> (i % 32)) & 1) acc = add_wave(acc, p);
  }
  return acc;
}

// ---- the same wave-cooperative operations on the reduced-radix types (XYZZ29, internal Montgomery form) -------------
// A level is ONE 162-mad (392 for 14 limbs) column-chain product (fp29_asm_gen.h) per lane instead of the 454-slot out-of-line
// 32-bit product of the forms above; coordinates stay below the storage bound BS p between levels (fit<BS>), so every
// slot's operand has the same static type.
// bcast29<SRC>: the value lane SRC (< 16) of every row of 16 lanes holds -> all lanes of the row, v_mov_b32_dpp
// row_newbcast:SRC, one VALU instruction per limb (the operands of these chains are uniform across the wave and every row
// holds the same four slots, so a row-local broadcast is a wave-wide one).  The moves are ONE OPAQUE asm statement per
// element ON PURPOSE -- two wait states first (a DPP read needs them after the VALU write of its source and hipcc cannot
// see a DPP inside an asm), then a v_mov_b32_dpp per limb: through __builtin_amdgcn_update_dpp hipcc's DPP combiner folds
// the broadcast into a consuming subtraction, v_subrev_u32_dpp ... row_newbcast, which does not compute S1 - dpp(S0) on
// gfx950 (the v_readlane form before it ran the glue on the scalar unit: CHANGELOG.md, round 4).
template <int SRC, class P, int B>
__device__ __forceinline__ Fe<P, B, 1> bcast29(const Fe<P, B, 1>& v) {
  static_assert(SRC >= 0 && SRC < 16, "row_newbcast takes a lane of the row");
  static_assert(RR<P>::N == 9 || RR<P>::N == 14, "limb count");
  Fe<P, B, 1> r;
  if constexpr (RR<P>::N == 9) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %0, %9 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %10 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %11 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %12 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %4, %13 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %5, %14 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %6, %15 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %7, %16 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %8, %17 row_newbcast:%18 row_mask:0xf bank_mask:0xf"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]),
          "=&v"(r.l[7]), "=&v"(r.l[8])
        : "v"(v.l[0]), "v"(v.l[1]), "v"(v.l[2]), "v"(v.l[3]), "v"(v.l[4]), "v"(v.l[5]), "v"(v.l[6]), "v"(v.l[7]),
          "v"(v.l[8]), "n"(SRC));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %0, %14 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %15 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %16 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %17 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %4, %18 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %5, %19 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %6, %20 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %7, %21 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %8, %22 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %9, %23 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %10, %24 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %11, %25 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %12, %26 row_newbcast:%28 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %13, %27 row_newbcast:%28 row_mask:0xf bank_mask:0xf"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]),
          "=&v"(r.l[7]), "=&v"(r.l[8]), "=&v"(r.l[9]), "=&v"(r.l[10]), "=&v"(r.l[11]), "=&v"(r.l[12]), "=&v"(r.l[13])
        : "v"(v.l[0]), "v"(v.l[1]), "v"(v.l[2]), "v"(v.l[3]), "v"(v.l[4]), "v"(v.l[5]), "v"(v.l[6]), "v"(v.l[7]),
          "v"(v.l[8]), "v"(v.l[9]), "v"(v.l[10]), "v"(v.l[11]), "v"(v.l[12]), "v"(v.l[13]), "n"(SRC));
  }
  return r;
}
template <int SRC, class P, int B>
__device__ __forceinline__ Fe2<P, B, 1> bcast29(const Fe2<P, B, 1>& v) {
  return {bcast29<SRC>(v.c0), bcast29<SRC>(v.c1)};
}
template <class P, int B>
__device__ __forceinline__ Fe<P, B, 1> lane_get29(const Fe<P, B, 1>& v, int src) {   // src: per-lane index
  Fe<P, B, 1> r;
#pragma unroll
  for (int i = 0; i < RR<P>::N; i++) r.l[i] = (uint32_t)__shfl((int)v.l[i], src);
  return r;
}
// product per slot (slot = lane / 4; operands equal across the quad); an Fq2 product is three base-field products on
// three lanes of the quad (Karatsuba), joined with ds_bpermute
template <class P, int B>
__device__ __forceinline__ Fe<P, B, 1> slot_mul29(const Fe<P, B, 1>& a, const Fe<P, B, 1>& b) { return fit<B>(a * b); }
// (BM: the base-field product of the three lanes -- inline, or behind a call: SlotMulCall in prover_impl.h)
template <class BM, class P, int B>
__device__ __forceinline__ Fe2<P, B, 1> slot_mul29_fe2(const Fe2<P, B, 1>& a, const Fe2<P, B, 1>& b) {
  constexpr int BETA = Fq2Beta<P>::value;
  const unsigned q = __lane_id() & 3;
  const Fe<P, B, 1> sa = fit<B>(a.c0 + a.c1), sb = fit<B>(b.c0 + b.c1);
  const Fe<P, B, 1> x = select(q == 0, a.c0, select(q == 1, a.c1, sa));
  const Fe<P, B, 1> y = select(q == 0, b.c0, select(q == 1, b.c1, sb));
  const Fe<P, B, 1> t = BM::mul(x, y);
  const int base = (int)(__lane_id() & ~3u);
  const Fe<P, B, 1> t0 = lane_get29(t, base), t1 = lane_get29(t, base + 1), t2 = lane_get29(t, base + 2);
  if constexpr (BETA == 1) return {fit<B>(t0 - t1), fit<B>(t2 - (t0 + t1))};          // u^2 = -BETA (fp2.h)
  else return {fit<B>(t0 - mul_small<BETA>(t1)), fit<B>(t2 - (t0 + t1))};
}
// M: how a level's product is issued -- inline (the chains that loop: Horner tail, scalar multiples, tree steps) or behind
// a call (prover_impl.h: SlotMulCall -- chains that run ONCE per proof, whose cost is the fetch of cold code)
struct SlotMulInline {
  template <class P, int B>
  static __device__ __forceinline__ Fe<P, B, 1> mul(const Fe<P, B, 1>& a, const Fe<P, B, 1>& b) { return fit<B>(a * b); }
  template <class P, int B>
  static __device__ __forceinline__ Fe2<P, B, 1> mul(const Fe2<P, B, 1>& a, const Fe2<P, B, 1>& b) {
    return slot_mul29_fe2<SlotMulInline>(a, b);
  }
};
template <class P, int B>
__device__ __forceinline__ Fe2<P, B, 1> slot_mul29(const Fe2<P, B, 1>& a, const Fe2<P, B, 1>& b) {
  return slot_mul29_fe2<SlotMulInline>(a, b);
}
// 2 p, p (and the result) uniform across the wave                       (dbl-2008-s-1, a = 0)
template <class F, class M = SlotMulInline>
__device__ __forceinline__ XYZZ29<F> dbl_wave29(const XYZZ29<F>& p) {
  constexpr int BS = XYZZ29<F>::BS;
  if (p.is_inf()) return p;
  const unsigned slot = (__lane_id() & 15) >> 2;     // four slots per row of 16 lanes (the same in every row)
  const auto u = fit<BS>(dbl(p.y));
  // level 1: v = u^2 | xx = x^2
  const auto a1 = select(slot == 0, u, p.x);
  const auto r1 = M::mul(a1, a1);
  const auto v = bcast29<0>(r1), xx = bcast29<4>(r1);
  const auto m = fit<BS>(dbl(xx) + xx);
  // level 2: w = u v | s = x v | m^2 | zz' = v zz
  const auto a2 = select(slot == 0, u, select(slot == 1, p.x, select(slot == 2, m, v)));
  const auto b2 = select(slot <= 1, v, select(slot == 2, m, p.zz));
  const auto r2 = M::mul(a2, b2);
  const auto w = bcast29<0>(r2), sv = bcast29<4>(r2), mm = bcast29<8>(r2), zz3 = bcast29<12>(r2);
  const auto x3 = fit<BS>(mm - dbl(sv));
  // level 3: m (s - x3) | w y | zzz' = w zzz
  const auto a3 = select(slot == 0, m, w);
  const auto b3 = select(slot == 0, fit<BS>(sv - x3), select(slot == 1, p.y, p.zzz));
  const auto r3 = M::mul(a3, b3);
  const auto y3 = fit<BS>(bcast29<0>(r3) - bcast29<4>(r3));
  return {x3, y3, zz3, bcast29<8>(r3)};
}
// p + o, both (and the result) uniform across the wave: 14 products in 4 levels       (add-2008-s)
template <class F, class M = SlotMulInline>
__device__ __forceinline__ XYZZ29<F> add_wave29(const XYZZ29<F>& p, const XYZZ29<F>& o) {
  constexpr int BS = XYZZ29<F>::BS;
  if (o.is_inf()) return p;
  if (p.is_inf()) return o;
  const unsigned slot = (__lane_id() & 15) >> 2;     // four slots per row of 16 lanes (the same in every row)
  // level 1: u1 = x1 zz2 | u2 = x2 zz1 | s1 = y1 zzz2 | s2 = y2 zzz1
  const auto a1 = select(slot == 0, p.x, select(slot == 1, o.x, select(slot == 2, p.y, o.y)));
  const auto b1 = select(slot == 0, o.zz, select(slot == 1, p.zz, select(slot == 2, o.zzz, p.zzz)));
  const auto r1 = M::mul(a1, b1);
  const auto u1 = bcast29<0>(r1), u2 = bcast29<4>(r1), s1 = bcast29<8>(r1), s2 = bcast29<12>(r1);
  const auto pd = fit<BS>(u2 - u1), rd = fit<BS>(s2 - s1);
  if (is_zero(pd)) {
    if (is_zero(rd)) return dbl_wave29<F, M>(p);
    return XYZZ29<F>::inf();
  }
  // level 2: pp = p^2 | rr = r^2 | zz1 zz2 | zzz1 zzz2
  const auto a2 = select(slot == 0, pd, select(slot == 1, rd, select(slot == 2, p.zz, p.zzz)));
  const auto b2 = select(slot == 0, pd, select(slot == 1, rd, select(slot == 2, o.zz, o.zzz)));
  const auto r2 = M::mul(a2, b2);
  const auto pp = bcast29<0>(r2), rr = bcast29<4>(r2), zzp = bcast29<8>(r2), zzzp = bcast29<12>(r2);
  // level 3: ppp = p pp | q = u1 pp | zz3 = (zz1 zz2) pp
  const auto a3 = select(slot == 0, pd, select(slot == 1, u1, zzp));
  const auto r3 = M::mul(a3, pp);
  const auto ppp = bcast29<0>(r3), q = bcast29<4>(r3), zz3 = bcast29<8>(r3);
  const auto x3 = fit<BS>(rr - (ppp + dbl(q)));
  // level 4: r (q - x3) | s1 ppp | zzz3 = (zzz1 zzz2) ppp
  const auto a4 = select(slot == 0, rd, select(slot == 1, s1, zzzp));
  const auto b4 = select(slot == 0, fit<BS>(q - x3), ppp);
  const auto r4 = M::mul(a4, b4);
  return {x3, fit<BS>(bcast29<0>(r4) - bcast29<4>(r4)), zz3, bcast29<8>(r4)};
}
// ---- k p on one wave: interleaved width-4 NAFs, and the endomorphism split where the group allows it -----------------
// The plain double-and-add chain (round 4) ran NW * 32 doublings and ~NW * 16 additions -- 254 x 3 + 127 x 4 = 1 270
// dependent product levels for s A' / r B1' of a proof: 0.52 ms of an 8-shard rank's 3.0 ms.  Here the scalar is recoded
// as a width-4 NAF (digits 0, +-1, +-3, +-5, +-7, one nonzero digit in five on average) over the odd multiples P, 3P, 5P,
// 7P kept in LDS, and for a group of cofactor one (BN254 G1: phi(P) = LAMBDA P holds for EVERY point of the curve;
// GlvCofactorOne below) k is first split k = k1 + k2 LAMBDA with 127-bit halves (glv.h) whose NAFs are interleaved
// over (P, phi P) (Straus): 127 doublings + ~51 additions + the table = ~600 levels.  Groups with a cofactor keep the
// unsplit scalar (a key's A' / B1' are in the order-r subgroup only if the key is valid, and a proof must equal
// arkworks' for any key): 254 doublings + ~51 additions = ~980 levels.
template <class F> struct GlvOf;
template <class F> struct GlvCofactorOne;
constexpr int kNafMax = 8 * 32 + 8;        // digits of one NAF (an NW-word integer has at most NW * 32 + 1)
template <class F>
struct ScalarMulLds {                      // per chain (one wave)
  XYZZ29<F> tab[8];                        // (2 j + 1) P, j < 4; then phi of them
  signed char naf[2][kNafMax];
};
// width-4 NAF of the NB-bit integer k[0 .. NW) (little-endian words): out[i] in {0, +-1, +-3, +-5, +-7}, i <= NB; returns
// the number of digits (highest nonzero position + 1).  One bit of carry instead of a multi-word subtraction: the window
// at a set bit is taken with the carry added, a window value >= 8 becomes value - 16 and carries into the bit after it.
template <int NW>
__device__ __forceinline__ int wnaf4_words(const uint32_t* k, int nbits, bool negate, signed char* out) {
  auto bits = [&](int at, int cnt) -> unsigned {      // k[at .. at + cnt), cnt <= 4 (bits past the top are zero)
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      lo = (at >> 5) == j ? k[j] : lo;
      hi = (at >> 5) + 1 == j ? k[j] : hi;
    }
    const uint64_t v = ((uint64_t)hi << 32 | lo) >> (at & 31);
    return (unsigned)v & ((1u << cnt) - 1u);
  };
  int len = 0;
  unsigned carry = 0;
  for (int i = 0; i <= nbits + 4; i++) out[i] = 0;
  int bit = 0;
  while (bit <= nbits) {
    if (bits(bit, 1) == carry) { bit++; continue; }
    int word = (int)(bits(bit, 4) + carry);
    carry = (unsigned)(word >> 3) & 1u;
    word -= (int)(carry << 4);
    out[bit] = (signed char)(negate ? -word : word);
    len = bit + 1;
    bit += 4;
  }
  return len;
}
// k p; p, k (NW canonical little-endian words) and the result uniform across the wave; `lds` is this wave's alone
// ALLOW_SPLIT = false: never split (a caller whose points need not be in the order-r subgroup of a cofactor-one group
// either -- there is none today -- or that wants one code path for all groups).
// Several waves of one workgroup may run chains side by side, each on its own `lds`: the two barriers below are WORKGROUP
// barriers, reached by every wave exactly twice whatever its point and scalar (no early return in front of them).
template <class F, int NW, bool ALLOW_SPLIT>
__device__ __forceinline__ XYZZ29<F> scalar_mul_lane29(const XYZZ29<F>& p_in, const uint32_t* k, ScalarMulLds<F>* lds);
template <class F, int NW>
__device__ __forceinline__ XYZZ29<F> scalar_mul_two_waves_lane29(const XYZZ29<F>& p_in, const uint32_t* k,
                                                                 ScalarMulLds<F>* lds, XYZZ29<F>* xchg);
template <class F, int NW, bool ALLOW_SPLIT = true>
__device__ __forceinline__ XYZZ29<F> scalar_mul_wave29(const XYZZ29<F>& p_in, const uint32_t* k, ScalarMulLds<F>* lds) {
  if constexpr (lane29::enabled<F>()) return scalar_mul_lane29<F, NW, ALLOW_SPLIT>(p_in, k, lds);   // (lane29.h)
  constexpr int BS = XYZZ29<F>::BS;
  constexpr bool SPLIT = ALLOW_SPLIT && GlvOf<F>::enabled && GlvCofactorOne<F>::value && NW == 8;
  const unsigned lane = __lane_id();
  const bool p_inf = p_in.is_inf();
  // (the identity runs the chain on a stand-in so that the barriers are reached; the result is discarded)
  XYZZ29<F> p = p_in;
  if (p_inf) { p.x = FieldOf<F>::one(); p.y = FieldOf<F>::one(); p.zz = FieldOf<F>::one(); p.zzz = FieldOf<F>::one(); }
  // the table of odd multiples (a doubling and three additions on the wave)
  {
    const XYZZ29<F> p2 = dbl_wave29(p);
    XYZZ29<F> m = p;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      if (j) m = add_wave29(m, p2);
      if (lane == 0) lds->tab[j] = m;
    }
  }
  int len = 0;
  if constexpr (SPLIT) {
    using GC = typename GlvOf<F>::C;
    uint32_t h[2][8];
    glv::split<GC>(k, h[0], h[1]);
    // lanes 0 and 1 recode one half each; phi(x, y) = (BETA x, y): x_affine = X / ZZ, so only X changes
    if (lane < 2) {
      uint32_t w[5];
#pragma unroll
      for (int i = 0; i < 4; i++) w[i] = lane ? h[1][i] : h[0][i];
      w[4] = 0;
      const bool neg_half = ((lane ? h[1][7] : h[0][7]) >> 31) != 0;
      len = wnaf4_words<5>(w, 128, neg_half, lds->naf[lane]);
    }
    __syncthreads();
    {
      Fp<typename FieldOf<F>::Params> beta32;
#pragma unroll
      for (int i = 0; i < Fp<typename FieldOf<F>::Params>::NL; i++) beta32.l[i] = GC::BETA[i];
      const auto beta = FieldOf<F>::from32(beta32);
      const unsigned slot = (lane & 15) >> 2;
      const XYZZ29<F> t = lds->tab[slot];
      const auto bx = fit<BS>(t.x * beta);
      if ((lane & 3) == 0 && lane < 16) {
        XYZZ29<F> e = t;
        e.x = bx;
        lds->tab[4 + slot] = e;
      }
    }
    len = max(__shfl(len, 0), __shfl(len, 1));
  } else {
    if (lane == 0) len = wnaf4_words<NW>(k, NW * 32, false, lds->naf[0]);
    len = __shfl(len, 0);
  }
  __syncthreads();
  XYZZ29<F> acc = XYZZ29<F>::inf();
#pragma unroll 1
  for (int i = len - 1; i >= 0; i--) {
    acc = dbl_wave29(acc);
#pragma unroll 1
    for (int hf = 0; hf < (SPLIT ? 2 : 1); hf++) {
      const int d = lds->naf[hf][i];
      if (d == 0) continue;
      XYZZ29<F> o = lds->tab[4 * hf + ((d < 0 ? -d : d) >> 1)];
      const auto ny = fit<BS>(neg(o.y));
      o.y = select(d < 0, ny, o.y);
      acc = add_wave29(acc, o);
    }
  }
  return p_inf ? p_in : acc;
}

// The same product on TWO waves of one workgroup, for a group whose scalars split (GlvOf + GlvCofactorOne): wave h runs the
// NAF chain of half h alone -- 127 doublings + ~25 additions each instead of 127 + ~51 on one wave -- over its own table (wave
// 1's entries are phi of wave 0's: X times BETA), and wave 0 adds the two results.  The chain was the critical path of a
// small proof (BASELINE config 4: prover_stage1_g1_kernel 0.61 of 1.95 ms, profiles/r6f_timeline_config4.md).
// Both waves call this with the same arguments; the result is valid in wave 0 (threadIdx.x < 64).
template <class F>
constexpr bool scalar_mul_splits() {
#ifdef DG16_STAGE1_ONE_WAVE      // (A/B switch of round 6: both halves interleaved on one wave, profiles/r6h_*)
  return false;
#else
  return GlvOf<F>::enabled && GlvCofactorOne<F>::value;
#endif
}
template <class F, int NW>
__device__ __forceinline__ XYZZ29<F> scalar_mul_two_waves29(const XYZZ29<F>& p_in, const uint32_t* k, ScalarMulLds<F>* lds,
                                                            XYZZ29<F>* xchg) {
  static_assert(NW == 8, "eight-word scalars");
  if constexpr (lane29::enabled<F>()) return scalar_mul_two_waves_lane29<F, NW>(p_in, k, lds, xchg);   // (lane29.h)
  constexpr int BS = XYZZ29<F>::BS;
  using GC = typename GlvOf<F>::C;
  const unsigned lane = __lane_id(), h = (threadIdx.x >> 6) & 1u;
  const bool p_inf = p_in.is_inf();
  XYZZ29<F> p = p_in;      // (the identity runs the chain on a stand-in so that the barriers are reached)
  if (p_inf) { p.x = FieldOf<F>::one(); p.y = FieldOf<F>::one(); p.zz = FieldOf<F>::one(); p.zzz = FieldOf<F>::one(); }
  {
    Fp<typename FieldOf<F>::Params> beta32;
#pragma unroll
    for (int i = 0; i < Fp<typename FieldOf<F>::Params>::NL; i++) beta32.l[i] = GC::BETA[i];
    const auto beta = FieldOf<F>::from32(beta32);
    const XYZZ29<F> p2 = dbl_wave29(p);
    XYZZ29<F> m = p;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      if (j) m = add_wave29(m, p2);
      XYZZ29<F> e = m;
      const auto bx = fit<BS>(m.x * beta);            // phi(x, y) = (BETA x, y): x_affine = X / ZZ, so only X changes
      e.x = select(h != 0, bx, m.x);
      if (lane == 0) lds->tab[4 * h + j] = e;
    }
  }
  uint32_t hv[2][8];
  glv::split<GC>(k, hv[0], hv[1]);
  int len = 0;
  if (lane == 0) {
    uint32_t w[5];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = h ? hv[1][i] : hv[0][i];
    w[4] = 0;
    len = wnaf4_words<5>(w, 128, ((h ? hv[1][7] : hv[0][7]) >> 31) != 0, lds->naf[h]);
  }
  len = __shfl(len, 0);
  __syncthreads();
  XYZZ29<F> acc = XYZZ29<F>::inf();
#pragma unroll 1
  for (int i = len - 1; i >= 0; i--) {
    acc = dbl_wave29(acc);
    const int d = lds->naf[h][i];
    if (d == 0) continue;
    XYZZ29<F> o = lds->tab[4 * h + ((d < 0 ? -d : d) >> 1)];
    const auto ny = fit<BS>(neg(o.y));
    o.y = select(d < 0, ny, o.y);
    acc = add_wave29(acc, o);
  }
  if (h == 1 && lane == 0) *xchg = acc;
  __syncthreads();
  if (h == 0) acc = add_wave29(acc, *xchg);
  return p_inf ? p_in : acc;
}

// ---- the same chains in LIMB-PER-LANE form (lane29.h) for the nine-limb base fields ----------------------------------
// One register per coordinate, a column-parallel product on each row of 16 lanes, the four products of a level on the four
// rows: a doubling 0.85 us instead of 2.1, an addition ~1.1 instead of 3.2 (profiles/r6l_lane29_probe.txt).  The table
// of odd multiples lives in the same LDS slots in raw lane form (lane29::store_pt_raw); p_in, k and the result are what
// the forms above take and return.
template <class F, int NW, bool ALLOW_SPLIT>
__device__ __forceinline__ XYZZ29<F> scalar_mul_lane29(const XYZZ29<F>& p_in, const uint32_t* k, ScalarMulLds<F>* lds) {
  using P = typename FieldOf<F>::Params;
  using FO = lane29::Ops<F>;
  using LPt = lane29::Pt<FO>;
  constexpr bool SPLIT = ALLOW_SPLIT && GlvOf<F>::enabled && GlvCofactorOne<F>::value && NW == 8 && !FO::EXT;
  const unsigned lane = __lane_id();
  typename FO::KT kc;
  kc.init();
  const bool p_inf = p_in.is_inf();
  LPt p = lane29::to_pt<F>(kc, p_in);
  if (p_inf) p = {FO::one(kc), FO::one(kc), FO::one(kc), FO::one(kc), false};   // (stand-in: the barriers below must be reached)
  {
    const LPt p2 = lane29::dbl_pt<FO>(kc, p);
    LPt m = p;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      if (j) m = lane29::add_pt<FO>(kc, m, p2);
      lane29::store_pt_raw<F>(kc, &lds->tab[j], m);
    }
  }
  int len = 0;
  if constexpr (SPLIT) {
    using GC = typename GlvOf<F>::C;
    uint32_t h[2][8];
    glv::split<GC>(k, h[0], h[1]);
    if (lane < 2) {
      uint32_t w[5];
#pragma unroll
      for (int i = 0; i < 4; i++) w[i] = lane ? h[1][i] : h[0][i];
      w[4] = 0;
      const bool neg_half = ((lane ? h[1][7] : h[0][7]) >> 31) != 0;
      len = wnaf4_words<5>(w, 128, neg_half, lds->naf[lane]);
    }
    __syncthreads();
    {
      // phi(x, y) = (BETA x, y): row j of the wave takes entry j -- one product for the four entries
      Fp<P> beta32;
#pragma unroll
      for (int i = 0; i < Fp<P>::NL; i++) beta32.l[i] = GC::BETA[i];
      const uint32_t beta = FO::template from_regs<XYZZ29<F>::BS>(kc, FieldOf<F>::from32(beta32));
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&lds->tab[kc.row]);
      uint32_t* dst = reinterpret_cast<uint32_t*>(&lds->tab[4 + kc.row]);
      const bool on = kc.l16 < 9u;
      const unsigned i = on ? kc.l16 : 0u;
      const uint32_t bx = lane29::mul(kc, on ? src[i] : 0u, beta);
      if (on) {
        dst[i] = bx;
        dst[9 + i] = src[9 + i];
        dst[18 + i] = src[18 + i];
        dst[27 + i] = src[27 + i];
      }
    }
    len = max(__shfl(len, 0), __shfl(len, 1));
  } else {
    if (lane == 0) len = wnaf4_words<NW>(k, NW * 32, false, lds->naf[0]);
    len = __shfl(len, 0);
  }
  __syncthreads();
  LPt acc = lane29::inf_pt<FO>(kc);
#pragma unroll 1
  for (int i = len - 1; i >= 0; i--) {
    acc = lane29::dbl_pt<FO>(kc, acc);
#pragma unroll 1
    for (int hf = 0; hf < (SPLIT ? 2 : 1); hf++) {
      const int d = lds->naf[hf][i];
      if (d == 0) continue;
      LPt o = lane29::load_pt<F>(kc, &lds->tab[4 * hf + ((d < 0 ? -d : d) >> 1)]);
      if (d < 0) o = lane29::neg_pt<FO>(kc, o);
      acc = lane29::add_pt<FO>(kc, acc, o);
    }
  }
  return p_inf ? p_in : lane29::from_pt<F>(kc, acc);
}
template <class F, int NW>
__device__ __forceinline__ XYZZ29<F> scalar_mul_two_waves_lane29(const XYZZ29<F>& p_in, const uint32_t* k,
                                                                 ScalarMulLds<F>* lds, XYZZ29<F>* xchg) {
  using P = typename FieldOf<F>::Params;
  using GC = typename GlvOf<F>::C;
  using FO = lane29::Ops<F>;
  using LPt = lane29::Pt<FO>;
  static_assert(!FO::EXT, "the endomorphism split of a cofactor-one G1");
  const unsigned lane = __lane_id(), h = (threadIdx.x >> 6) & 1u;
  typename FO::KT kc;
  kc.init();
  const bool p_inf = p_in.is_inf();
  LPt p = lane29::to_pt<F>(kc, p_in);
  if (p_inf) p = {FO::one(kc), FO::one(kc), FO::one(kc), FO::one(kc), false};
  {
    Fp<P> beta32;
#pragma unroll
    for (int i = 0; i < Fp<P>::NL; i++) beta32.l[i] = GC::BETA[i];
    const uint32_t beta = FO::template from_regs<XYZZ29<F>::BS>(kc, FieldOf<F>::from32(beta32));
    if (h) p.x = lane29::mul(kc, p.x, beta);          // wave 1 runs its chain over phi(P) = (BETA x, y)
    const LPt p2 = lane29::dbl_pt<FO>(kc, p);
    LPt m = p;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      if (j) m = lane29::add_pt<FO>(kc, m, p2);
      lane29::store_pt_raw<F>(kc, &lds->tab[4 * h + j], m);
    }
  }
  uint32_t hv[2][8];
  glv::split<GC>(k, hv[0], hv[1]);
  int len = 0;
  if (lane == 0) {
    uint32_t w[5];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = h ? hv[1][i] : hv[0][i];
    w[4] = 0;
    len = wnaf4_words<5>(w, 128, ((h ? hv[1][7] : hv[0][7]) >> 31) != 0, lds->naf[h]);
  }
  len = __shfl(len, 0);
  __syncthreads();
  LPt acc = lane29::inf_pt<FO>(kc);
#pragma unroll 1
  for (int i = len - 1; i >= 0; i--) {
    acc = lane29::dbl_pt<FO>(kc, acc);
    const int d = lds->naf[h][i];
    if (d == 0) continue;
    LPt o = lane29::load_pt<F>(kc, &lds->tab[4 * h + ((d < 0 ? -d : d) >> 1)]);
    if (d < 0) o = lane29::neg_pt<FO>(kc, o);
    acc = lane29::add_pt<FO>(kc, acc, o);
  }
  if (h == 1) lane29::store_pt_raw<F>(kc, xchg, acc);
  __syncthreads();
  if (h == 0) acc = lane29::add_pt<FO>(kc, acc, lane29::load_pt<F>(kc, xchg));
  return p_inf ? p_in : lane29::from_pt<F>(kc, acc);
}
// sum of n points in memory (proper XYZZ29s), uniform result: the chains that only add (the king's combination, the
// terms of prove::A / B / C)
template <class F>
__device__ __forceinline__ XYZZ29<F> sum_points_wave(const XYZZ29<F>* pts, unsigned n) {
  if constexpr (lane29::enabled<F>()) {
    using FO = lane29::Ops<F>;
    typename FO::KT kc;
    kc.init();
    lane29::Pt<FO> acc = lane29::inf_pt<FO>(kc);
#pragma unroll 1
    for (unsigned i = 0; i < n; i++) acc = lane29::add_pt<FO>(kc, acc, lane29::load_pt<F>(kc, &pts[i]));
    return lane29::from_pt<F>(kc, acc);
  } else {
    XYZZ29<F> acc = XYZZ29<F>::inf();
#pragma unroll 1
    for (unsigned i = 0; i < n; i++) acc = add_wave29(acc, pts[i]);
    return acc;
  }
}

// the same sum as an affine point (the king's combination hands affine points to the parties)
template <class F>
__device__ __forceinline__ Affine<F> sum_points_affine_wave(const XYZZ29<F>* pts, unsigned n) {
  if constexpr (lane29::enabled<F>()) {
    using FO = lane29::Ops<F>;
    typename FO::KT kc;
    kc.init();
    lane29::Pt<FO> acc = lane29::inf_pt<FO>(kc);
#pragma unroll 1
    for (unsigned i = 0; i < n; i++) acc = lane29::add_pt<FO>(kc, acc, lane29::load_pt<F>(kc, &pts[i]));
    return lane29::to_affine<F>(kc, acc);
  } else {
    return sum_points_wave<F>(pts, n).to_xyzz32().to_affine();
  }
}

// rows of 2^kRowLog buckets
constexpr unsigned kRowLog = 8;
struct RowGeom {
  unsigned row_log;    // log2 buckets per row (<= kRowLog)
  unsigned rows_log;   // log2 rows per bucket-window
};
inline RowGeom row_geometry(const MsmGeom& g) {
  RowGeom r;
  r.row_log = g.log_nb < kRowLog ? g.log_nb : kRowLog;
  r.rows_log = g.log_nb - r.row_log;
  return r;
}

// Result of the scalar-side passes (digits, scan, scatter): shared by every MSM over the same scalars.
struct MsmSort {
  MsmGeom g;
  size_t n = 0;
  int* digits = nullptr;
  unsigned *entries = nullptr, *counts = nullptr, *offsets = nullptr, *seg_off = nullptr, *cursor = nullptr;
  unsigned* seg_total = nullptr;
};

// scalars_mont: bit 0 = Montgomery form, bit 1 = bit 255 of a scalar is a sign flag (the halves of glv.h)
template <class Fr, int SCALAR_BITS>
MsmSort msm_sort_on(hipStream_t s, Channel& wsch, const void* scalars, size_t n, unsigned scalars_mont, bool table,
                    unsigned c_fixed, unsigned stride);
template <class Fr, int SCALAR_BITS>
MsmSort msm_sort(Call& k, const void* scalars, size_t n, unsigned scalars_mont, bool table, unsigned c_fixed = 0,
                 unsigned stride = 1) {
  return msm_sort_on<Fr, SCALAR_BITS>(k.s(), k.c, scalars, n, scalars_mont, table, c_fixed, stride);
}
// sort on stream `s` with the buffers of channel `wsch`
template <class Fr, int SCALAR_BITS>
MsmSort msm_sort_on(hipStream_t s, Channel& wsch, const void* scalars, size_t n, unsigned scalars_mont, bool table,
                    unsigned c_fixed, unsigned stride) {
  MsmSort r;
  DG_BOUNDS_BIND();
  r.n = n;
  r.g = msm_geometry(n ? n : 1, SCALAR_BITS, table, c_fixed, stride);
  const MsmGeom& g = r.g;
  const size_t nbw = (size_t)g.bw << g.log_nb;
  DG_REQUIRE((size_t)g.nwin * n < ((size_t)1 << 31), DG16_ERR_BAD_ARG, "W * n must be < 2^31");
  // large sorts: LDS-partitioned passes; small ones: the direct atomic path (fewer launches)
  unsigned lg_nbw = 0;
  while (((size_t)1 << lg_nbw) < nbw) lg_nbw++;
  static const int force_path = [] { const char* e = getenv("DG16_MSM_SORT"); return e ? atoi(e) : 0; }();  // 1 atomic, 2 partitioned
  PartGeom pg;
  pg.low_bits = lg_nbw > 8 ? lg_nbw - 8 : 0;
  const bool partitioned = pg.low_bits <= kPartMaxLowBits &&
                           (force_path == 2 || (force_path != 1 && (size_t)g.nwin * n >= ((size_t)1 << 18)));
  pg.nparts = (unsigned)((nbw + ((size_t)1 << pg.low_bits) - 1) >> pg.low_bits);
  pg.nblk1 = (unsigned)((n + kPartScalars - 1) / kPartScalars);
  r.digits = (int*)ws(wsch, 4, (size_t)g.nwin * n * (partitioned ? 8 : 4));
  r.entries = (unsigned*)ws(wsch, 5, (size_t)g.nwin * n * 4);
  unsigned* tabs = (unsigned*)ws(wsch, 6, (nbw * 4 + g.bw) * 4);
  r.counts = tabs;
  r.offsets = r.counts + nbw;
  r.seg_off = r.offsets + nbw;
  r.cursor = r.seg_off + nbw;
  r.seg_total = r.cursor + nbw;
  unsigned* blockoff = nullptr;
  if (n && partitioned) {
    const size_t len = (size_t)pg.nparts * pg.nblk1 + 1;
    blockoff = (unsigned*)ws(wsch, 25, (len + (unsigned)((len + 4095) / 4096)) * 4);
  }
  const unsigned scan_nblocks = ((1u << g.log_nb) + kScanBlock - 1) / kScanBlock;
  unsigned* block_tot = (unsigned*)ws(wsch, 9, (size_t)g.bw * scan_nblocks * 2 * 4);
  DG_HIP(hipMemsetAsync(r.counts, 0, nbw * 4, s));
  uint2* part = (uint2*)r.digits;
  if (n && partitioned) {
    const size_t len = (size_t)pg.nparts * pg.nblk1 + 1;      // + sentinel = total entries
    const unsigned nchunks = (unsigned)((len + 4095) / 4096);
    unsigned* tot = blockoff + len;
    DG_HIP(hipMemsetAsync(blockoff + len - 1, 0, 4, s));
    hipLaunchKernelGGL(msm_part_hist_kernel<Fr>, dim3(pg.nblk1), dim3(256), 0, s, (const Fr*)scalars, n,
                       (int)scalars_mont, g, pg, blockoff);
    hipLaunchKernelGGL(scan_chunk_kernel<0>, dim3(nchunks), dim3(1024), 0, s, blockoff, len, tot);
    hipLaunchKernelGGL(scan_tops_kernel<0>, dim3(1), dim3(1024), 0, s, tot, nchunks);
    hipLaunchKernelGGL(scan_add_kernel<0>, dim3(nchunks), dim3(1024), 0, s, blockoff, len, tot);
    hipLaunchKernelGGL(msm_part_scatter_kernel<Fr>, dim3(pg.nblk1), dim3(256), 0, s, (const Fr*)scalars, n,
                       (int)scalars_mont, g, pg, blockoff, part);
    hipLaunchKernelGGL(msm_part_count_kernel<0>, dim3(kPartBlocks, pg.nparts), dim3(256), 0, s, part, blockoff, pg,
                       r.counts);
  } else if (n) {
    hipLaunchKernelGGL(msm_digits_kernel<Fr>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const Fr*)scalars,
                       n, (int)scalars_mont, g, r.digits, r.counts);
  }
  {
    const unsigned nblocks = ((1u << g.log_nb) + kScanBlock - 1) / kScanBlock;   // <= 512 for c <= 22
    hipLaunchKernelGGL(msm_scan_local_kernel<0>, dim3(nblocks, g.bw), dim3(1024), 0, s, r.counts, r.offsets,
                       r.seg_off, block_tot, g.log_nb, g.seg_log);
    hipLaunchKernelGGL(msm_scan_tops_kernel<0>, dim3(g.bw), dim3(1024), 0, s, block_tot, nblocks, r.seg_total);
    hipLaunchKernelGGL(msm_scan_fix_kernel<0>, dim3(nblocks, g.bw), dim3(1024), 0, s, r.offsets, r.seg_off, r.cursor,
                       block_tot, g.log_nb);
  }
  if (n && partitioned)
    hipLaunchKernelGGL(msm_part_place_kernel<0>, dim3(kPartBlocks, pg.nparts), dim3(256), 0, s, part, blockoff, pg, g,
                       r.offsets, r.seg_off, r.cursor, r.entries);
  else if (n)
    hipLaunchKernelGGL(msm_scatter_kernel<0>, dim3((unsigned)((n + 255) / 256), g.nwin), dim3(256), 0, s, r.digits, n, g,
                       r.offsets, r.seg_off, r.cursor, r.entries);
  DG_HIP(hipGetLastError());
  return r;
}

// Workspace of one MSM's bucket phases (lives in `wsch`'s slots 7, 17, 15, 10 until the reduction is done).
// Small bucket sets (a short shard, BASELINE config 4, a plain MSM of <= 2^15 points) are reduced by the radix-16 / radix-8
// lane-form kernel of msm_reduce_impl.h (msm_lane_reduce_kernel): one WAVE per bucket at the first level
constexpr size_t kLaneReduceMaxBuckets = 32768;
template <class F>
inline bool lane_reduce_applies(size_t buckets_over_all_windows) {
  if constexpr (!lane29::enabled<F>()) return false;
  else {
    static const bool off = [] { const char* e = getenv("DG16_NO_LANE_REDUCE"); return e && atoi(e) != 0; }();
    return !off && buckets_over_all_windows <= kLaneReduceMaxBuckets;
  }
}
template <class F>
struct MsmBuffers {
  XYZZ29<F>* buckets;
  XYZZ29<F>* seg_sum;
  XYZZ29<F>* row_w;
  XYZZ29<F>* row_r;
  XYZZ29<F>* fold;
  XYZZ29<F>* window_sums;    // internal form: the tail's wave-cooperative chain runs on the reduced-radix types
  XYZZ29<F>* lane_tmp;       // (W, R) pairs between the levels of msm_lane_reduce_kernel, or null (msm_reduce_impl.h)
  XYZZ29<F>* top_tmp;        // the same for the lane-form levels that stand in for msm_top_kernel (msm_lane_top), or null
  hipStream_t finalize_stream = nullptr;   // G2: the throughput finalize goes to this stream (behind acc_done) instead of
                                           // following the accumulation on its own (the prover: B's reduction stream)
  bool busy_chip = false;    // the reduction runs beside saturating kernels of other streams (a proof's MSMs): small
                             // workgroups only (msm_lane_reduce_serial_kernel instead of the 16-wave form)
  unsigned* giant;
  unsigned giant_cap;
  size_t nbw, nrows;     // over all instances
  unsigned ninst;        // MSMs sharing the sort (msm_accumulate_kernel): bucket-window index wy = inst * bw + w
  RowGeom rg;
  unsigned long long* clk = nullptr;   // ClkProbe counters of the accumulation kernel (two device words), or null
  hipEvent_t acc_done = nullptr;   // recorded right behind the accumulation KERNEL (in front of the G2 finalize that
                                   // msm_accumulate_phase launches after it): the end of dg16_last_kernel_ms's bracket
};

template <class F>
MsmBuffers<F> msm_buffers(Channel& wsch, const MsmGeom& g, unsigned ninst = 1) {
  MsmBuffers<F> b;
  DG_REQUIRE(ninst >= 1 && ninst <= kMaxInst, DG16_ERR_BAD_ARG, "1..4 MSM instances per sort");
  b.ninst = ninst;
  const size_t bwi = (size_t)g.bw * ninst;
  b.nbw = bwi << g.log_nb;
  const size_t nseg_slots = bwi * g.seg_cap;
  b.giant_cap = (unsigned)(nseg_slots / kGiantSegs + 1);
  // capacities of the giant work list (msm_register_giant): ids < giant_cap, work items <= 2 giant_cap
  DG_REQUIRE(nseg_slots / kGiantSegs + 1 < ((size_t)1 << 26), DG16_ERR_BAD_ARG, "giant list: id slot must fit 26 bits");
  static_assert(kGiantSlices <= 64, "a work item keeps its slice in six bits");
  static_assert(kGiantSliceSegs >= kGiantSegs, "work items <= 2 giant_cap needs slices no shorter than kGiantSegs partials");
  DG_REQUIRE(nseg_slots / kGiantSliceSegs + 1 + b.giant_cap <= 2 * (size_t)b.giant_cap + 1, DG16_ERR_BAD_ARG,
             "giant list: work-item capacity");
  b.buckets = (XYZZ29<F>*)ws(wsch, 7, b.nbw * sizeof(XYZZ29<F>));
  b.seg_sum = (XYZZ29<F>*)ws(wsch, 17, nseg_slots * sizeof(XYZZ29<F>));
  b.rg = row_geometry(g);
  b.nrows = bwi << b.rg.rows_log;
  const size_t nfold = bwi * 3 * 256;
  const size_t nlane = lane_reduce_applies<F>(b.nbw) ? b.nbw / 2 + 4 : 0;
  const size_t ntop = lane29::enabled<F>() ? bwi * 96 + 8 : 0;      // <= 256 entries per bucket-window: 64 + 8 pairs' slots
  uint8_t* p15 = (uint8_t*)ws(wsch, 15, (2 * b.nrows + nfold + bwi + nlane + ntop) * sizeof(XYZZ29<F>));
  b.row_w = (XYZZ29<F>*)p15;
  b.row_r = b.row_w + b.nrows;
  b.fold = b.row_r + b.nrows;
  b.window_sums = b.fold + nfold;
  b.lane_tmp = nlane ? b.window_sums + bwi : nullptr;
  b.top_tmp = ntop ? b.window_sums + bwi + nlane : nullptr;
  // [0] giants, [1] work items, then giant_cap bucket ids, then <= 2 * giant_cap (giant, slice) work items
  b.giant = (unsigned*)ws(wsch, 10, ((size_t)b.giant_cap * 3 + 2) * 4);
  return b;
}

inline int msm_finalize_lds_lpb();
template <class F>
struct MsmBuffers;
template <class F>
void msm_finalize_lds_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b);
// Phase A (saturates the GPU): segment accumulation.  `bases` is the array of n points or, in table mode, the
// table of W*n points -- in INTERNAL form (msm_to_internal_kernel / msm_table_kernel).
// bases: b.ninst tables (or plain base arrays), one per instance
template <class F>
void msm_accumulate_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b, const void* const* bases) {
  const MsmGeom& g = st.g;
  DG_BOUNDS_BIND();
  MsmBases mb{};
  for (unsigned i = 0; i < b.ninst; i++) mb.p[i] = (const uint32_t*)bases[i];
  if (b.clk) DG_HIP(hipMemsetAsync(b.clk, 0, 16, s));
  if constexpr (sizeof(F) > 48) {
    // G2 (Fq2 coordinates): LDS-staged accumulator; two workgroups per CU must fit the 160 KiB of LDS
    constexpr int BLOCK = 1 << msm_acc_block_log<F>();
    const dim3 grid((g.seg_cap + BLOCK - 1) / BLOCK, g.bw * b.ninst);
    if constexpr (sizeof(F) > 64) {
      // 14-limb Fq2: ONE form -- three product sites visited by a step loop (msm_accumulate_steps_kernel: a loop that fits
      // the instruction cache).  Measured against round 4's straight-line loop, same call: 8.84-8.90 ms against 8.85-8.93
      // per 2^20-point launch on a fast box of the pool, 8.93-9.17 against 18.0-18.1 on a slow one
      // (profiles/r5b_*, r5c_*); both round-4 forms and the timing-based choice between them are gone.
      hipLaunchKernelGGL((msm_accumulate_steps_kernel<F, BLOCK>), grid, dim3(BLOCK), 0, s, mb, st.n, g, st.offsets,
                         st.counts, st.seg_off, st.seg_total, st.entries, b.seg_sum, b.buckets, b.clk);
    } else {
      hipLaunchKernelGGL((msm_accumulate_lds_kernel<F, BLOCK>), grid, dim3(BLOCK), 0, s, mb, st.n, g, st.offsets, st.counts,
                         st.seg_off, st.seg_total, st.entries, b.seg_sum, b.buckets, b.clk);
    }
    if (b.acc_done) DG_HIP(hipEventRecord(b.acc_done, s));
    if (b.finalize_stream && b.acc_done) {
      DG_HIP(hipStreamWaitEvent(b.finalize_stream, b.acc_done, 0));
      msm_finalize_lds_phase<F>(b.finalize_stream, st, b);
    } else {
      msm_finalize_lds_phase<F>(s, st, b);
    }
  } else {
    constexpr int BLOCK = 1 << msm_acc_block_log<F>();
    hipLaunchKernelGGL((msm_accumulate_kernel<F, BLOCK>), dim3((g.seg_cap + BLOCK - 1) / BLOCK, g.bw * b.ninst),
                       dim3(BLOCK), 0, s, mb, st.n, g, st.offsets, st.counts, st.seg_off, st.seg_total, st.entries,
                       b.seg_sum, b.buckets, b.clk);
    if (b.acc_done) DG_HIP(hipEventRecord(b.acc_done, s));
  }
  DG_HIP(hipGetLastError());
}
template <class F>
void msm_accumulate_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b, const void* bases) {
  const void* one[1] = {bases};
  msm_accumulate_phase<F>(s, st, b, one);
}

// ---- 6: Horner tail ---------------------------------------------------------------------------------
// W*c dependent doublings: inherently serial in the group, but not inside one doubling.  One wave runs the
// chain; the 9 multiplications of an XYZZ doubling form 3 dependency levels (2 | 4 | 3 products), each level
// is evaluated by different lanes at once and shared with readlane.  An Fq2 product is itself spread over three
// lanes of a quad (Karatsuba).  One thread per MSM took 2.5 ms (G1) / 10.2 ms (G2) for the 256 doublings of a
// 2^20-point MSM, as long as the bucket accumulation itself.  (Round 4: the chain runs on the reduced-radix types.)
template <class F>
__global__ void __launch_bounds__(64) msm_tail_kernel(const XYZZ29<F>* __restrict__ window_sums, MsmGeom g,
                                                       int affine, F* __restrict__ out) {
  __builtin_amdgcn_s_setprio(DG16_CHAIN_PRIO);   // latency-bound chain: issue ahead of the accumulation waves sharing the SIMD
  // one wave per MSM instance (blockIdx.x), every lane carries the same running total (internal form: dbl_wave29)
  window_sums += (size_t)blockIdx.x * g.bw;
  out += (size_t)blockIdx.x * (affine ? 2 : 3);
  XYZZ29<F> acc = XYZZ29<F>::inf();
  if constexpr (lane29::enabled<F>()) {          // limb-per-lane chain (lane29.h): 0.85 us per doubling instead of 2.1
    using FO = lane29::Ops<F>;
    typename FO::KT kc;
    kc.init();
    lane29::Pt<FO> a = lane29::inf_pt<FO>(kc);
#pragma unroll 1
    for (int w = (int)g.bw - 1; w >= 0; w--) {
#pragma unroll 1
      for (unsigned k = 0; k < g.c; k++) a = lane29::dbl_pt<FO>(kc, a);
      a = lane29::add_pt<FO>(kc, a, lane29::load_pt<F>(kc, &window_sums[w]));
    }
    if (affine) {            // (X / ZZ, Y / ZZZ) with the inversion in lane form as well (lane29::to_affine)
      const Affine<F> r = lane29::to_affine<F>(kc, a);
      if (threadIdx.x == 0) {
        out[0] = r.x;
        out[1] = r.y;
      }
      return;
    }
    acc = lane29::from_pt<F>(kc, a);
  } else {
#pragma unroll 1
    for (int w = (int)g.bw - 1; w >= 0; w--) {
#pragma unroll 1
      for (unsigned k = 0; k < g.c; k++) acc = dbl_wave29(acc);
      acc = add_wave29(acc, window_sums[w]);
    }
  }
  if (threadIdx.x != 0) return;
  using FO = FieldOf<F>;
  if (affine) {
    Affine<F> a = acc.to_xyzz32().to_affine();
    out[0] = a.x;
    out[1] = a.y;
  } else if (acc.is_inf()) {
    out[0] = F::one();
    out[1] = F::one();
    out[2] = F::zero();
  } else {
    // (X ZZ, Y ZZZ, ZZ) is the same point in Jacobian coordinates with Z = ZZ (ec.h: XYZZ::to_jacobian)
    out[0] = FO::to32(fit<FO::BS>(acc.x * acc.zz));
    out[1] = FO::to32(fit<FO::BS>(acc.y * acc.zzz));
    out[2] = FO::to32(acc.zz);
  }
}

// Launched by msm_bucket_phase (msm_reduce.hip) but INSTANTIATED in msm_group.hip: the chain's products stay inline for
// every group (a call per level cost the G2 tail 8 us per operation against 3 for G1's inline form).
template <class F>
void msm_tail_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b, bool out_affine, void* out_dev) {
  hipLaunchKernelGGL(msm_tail_kernel<F>, dim3(b.ninst), dim3(64), 0, s, b.window_sums, st.g, (int)out_affine, (F*)out_dev);
}

// ---- 4b: bucket = sum of its segment partials, as a throughput kernel -------------------------------------------
// Giant buckets (a boolean witness puts half of ALL entries into bucket 0; the short top window of a c that does
// not divide the scalar width does the same) go on a device-side work list: stage 1 cuts the bucket's segment
// partials into <= kGiantSlices slices, one workgroup each; stage 2 adds the slice sums (msm_reduce_impl.h).
__device__ __forceinline__ void giant_geometry(unsigned nseg, unsigned& slices, unsigned& per) {
  slices = (nseg + kGiantSliceSegs - 1) / kGiantSliceSegs;
  if (slices > kGiantSlices) slices = kGiantSlices;
  per = (nseg + slices - 1) / slices;
  slices = (nseg + per - 1) / per;
}
// A giant bucket (np > kGiantSegs partials) goes on the device-side work list: giant[0] = giants, giant[1] = work items,
// giant_list[0 .. giant_cap) = bucket ids, giant_list[giant_cap .. 3 giant_cap) = (id slot << 6 | slice) items.  Both
// capacities hold by construction (msm_buffers asserts the arithmetic; MSM_INVARIANTS.md): the giants of a launch own
// disjoint sets of > kGiantSegs of its <= nseg_slots partials, so there are < nseg_slots / kGiantSegs + 1 = giant_cap of
// them, and their slices number sum ceil(np / per) <= sum (np / kGiantSliceSegs + 1) < nseg_slots / kGiantSliceSegs +
// giant_cap <= 2 giant_cap.  The guards below keep a violated invariant from writing outside the list anyway.
__device__ __forceinline__ void msm_register_giant(unsigned gid, unsigned np, unsigned* __restrict__ giant_count,
                                                   unsigned* __restrict__ giant_list, unsigned giant_cap) {
  const unsigned slot = atomicAdd(giant_count, 1u);
  if (!DG_OK(8, slot, giant_cap) || slot >= giant_cap) return;
  giant_list[slot] = gid;
  unsigned slices, per;
  giant_geometry(np, slices, per);
  const unsigned wb = atomicAdd(giant_count + 1, slices);   // work items: (giant, slice)
  if (!DG_OK(9, (size_t)wb + slices, 2 * (size_t)giant_cap + 1) || (size_t)wb + slices > 2 * (size_t)giant_cap) return;
  unsigned* work = giant_list + giant_cap;
  for (unsigned i = 0; i < slices; i++) work[wb + i] = (slot << 6) | i;
}


// What is left of the finalize after the in-workgroup bucket tree: one lane per bucket adds the partials of the
// buckets that CROSS an accumulation workgroup (one bucket in ~17 for a 2^20 table MSM: one addition; a bucket held
// by one workgroup was written by it, an empty one is set to the identity here).  (Round 2 / early round 3 summed
// ~15 per-segment partials per bucket here, a million full additions per MSM at 5-18x the issue time of an
// addition: profiles/r3_finalize_experiments.md.)
// TU: 0 = instantiated in msm_group.hip (products inline), 1 = in msm_reduce.hip (for G2 compiled with out-of-line
// products, DG29_OUTLINE_MUL: an inlined Fq2 addition + doubling is 123 KB of code for BN254 and 250 KB for BLS12-381,
// against the 64 KB instruction cache two CUs share -- a BLS12-381 2^20 proof took 38 ms instead of 26 with it).
// Distinct symbols, so that both variants can live in one library.
// `total` = (instances * bw) << log_nb buckets; the sort's arrays are indexed by the bucket-window w = wy % bw.
template <class F, int TU = 0>
__global__ void __launch_bounds__(256) msm_finalize_thr_kernel(MsmGeom g, size_t total, unsigned wg_log,
                                                                const unsigned* __restrict__ counts,
                                                                const unsigned* __restrict__ seg_off,
                                                                const XYZZ29<F>* __restrict__ seg_sum,
                                                                XYZZ29<F>* __restrict__ buckets,
                                                                unsigned* __restrict__ giant_count,
                                                                unsigned* __restrict__ giant_list, unsigned giant_cap) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const unsigned wy = (unsigned)(gid >> g.log_nb);
  const size_t gs = ((size_t)(wy % g.bw) << g.log_nb) + (gid & (((size_t)1 << g.log_nb) - 1));   // the sort's bucket slot
  const unsigned k = (counts[gs] + (1u << g.seg_log) - 1) >> g.seg_log;
  const unsigned first = seg_off[gs];
  const unsigned np = msm_nparts(first, k, wg_log);     // partials the accumulation workgroups left for this bucket
  if (np == 0) {
    buckets[gid] = XYZZ29<F>::inf();
    return;
  }
  if (np == 1) return;                                  // one workgroup held the whole bucket and wrote it
  if (np > kGiantSegs) {
    msm_register_giant((unsigned)gid, np, giant_count, giant_list, giant_cap);
    return;
  }
  const XYZZ29<F>* sp = seg_sum + (size_t)wy * g.seg_cap;
  XYZZ29<F> acc = sp[DG_IDX(7, first, g.seg_cap)];
#pragma unroll 1
  for (unsigned s = 1; s < np; s++) acc = acc.add(sp[DG_IDX(7, msm_part_slot(first, s, wg_log), g.seg_cap)]);
  buckets[gid] = acc;
}
// The same sums with a WAVE per bucket in lane form (lane29.h), for the launches where the lone-lane form above is at its
// worst: a 14-limb G1 addition inlined is ~60 KB of code against the 64-KB instruction cache two CUs share, and the
// serial loop above ran at ~125 us per dependent addition -- 0.89 ms of a 1.5-ms plain MSM at 2^13 BLS12-377 points,
// 0.92 of 1.97 at 2^16 (profiles/r6kk_timeline_bls12_377_g1_2e13.md).  A wave per bucket costs ~8x the issue slots of a lane
// per bucket, so this form takes the launches with at most kLaneFinalizeMaxPartials partial sums in all.
constexpr size_t kLaneFinalizeMaxPartials = (size_t)1 << 18;
template <class F>
__global__ void __launch_bounds__(64) msm_finalize_lane_kernel(MsmGeom g, unsigned wg_log,
                                                                const unsigned* __restrict__ counts,
                                                                const unsigned* __restrict__ seg_off,
                                                                const XYZZ29<F>* __restrict__ seg_sum,
                                                                XYZZ29<F>* __restrict__ buckets,
                                                                unsigned* __restrict__ giant_count,
                                                                unsigned* __restrict__ giant_list, unsigned giant_cap) {
  if constexpr (lane29::enabled<F>()) {
    using FO = lane29::Ops<F>;
    using LPt = lane29::Pt<FO>;
    const size_t gid = blockIdx.x;                       // one wave per bucket: everything below is uniform across it
    const unsigned wy = (unsigned)(gid >> g.log_nb);
    const size_t gs = ((size_t)(wy % g.bw) << g.log_nb) + (gid & (((size_t)1 << g.log_nb) - 1));
    const unsigned k = (counts[gs] + (1u << g.seg_log) - 1) >> g.seg_log;
    const unsigned first = seg_off[gs];
    const unsigned np = msm_nparts(first, k, wg_log);
    if (np == 0) {
      if (threadIdx.x == 0) buckets[gid] = XYZZ29<F>::inf();
      return;
    }
    if (np == 1) return;
    if (np > kGiantSegs) {
      if (threadIdx.x == 0) msm_register_giant((unsigned)gid, np, giant_count, giant_list, giant_cap);
      return;
    }
    typename FO::KT kc;
    kc.init();
    const XYZZ29<F>* sp = seg_sum + (size_t)wy * g.seg_cap;
    LPt acc = lane29::load_pt<F>(kc, &sp[DG_IDX(7, first, g.seg_cap)]);
    LPt nx = lane29::load_pt_words<F>(kc, &sp[DG_IDX(7, msm_part_slot(first, 1, wg_log), g.seg_cap)]);
#pragma unroll 1
    for (unsigned s = 1; s < np; s++) {
      const LPt cur = lane29::with_inf_flag<FO>(nx);
      if (s + 1 < np) nx = lane29::load_pt_words<F>(kc, &sp[DG_IDX(7, msm_part_slot(first, s + 1, wg_log), g.seg_cap)]);
      acc = lane29::add_pt<FO>(kc, acc, cur);
    }
    lane29::store_pt<F>(kc, &buckets[gid], acc);
  }
}
// The same finalize as a THROUGHPUT kernel (G2): LPB lanes per bucket, each summing its share of the bucket's partials
// into an accumulator that lives in LDS columns between the products (the layout and register budget of
// msm_accumulate_lds_kernel: two workgroups per CU, ~175 VGPRs), then a log2(LPB)-step tree over neighbouring columns.
// A 2^20-point table MSM is one round of 2 waves per SIMD with 8 + 1 additions per lane; runs on the accumulation's own
// stream, right behind it (msm_accumulate_phase).  ONE addition site: the serial partials (global memory) and the tree
// partners (LDS columns) go through the same accessor, told apart at run time -- one site per operand kind was 227 KB of
// code against the 64-KB instruction cache (CHANGELOG.md, round 4).
template <class F, int BLOCK>
struct PartialAcc {      // an XYZZ29 behind the accessor interface of XYZZ29::add_into: memory if p, else LDS column
  using S = typename FieldOf<F>::Store;
  const XYZZ29<F>* p;
  ColAcc<F, BLOCK> col;
  __device__ __forceinline__ S get(int coord) const {
    if (p) return coord == 0 ? p->x : coord == 1 ? p->y : coord == 2 ? p->zz : p->zzz;
    return col.get(coord);
  }
  // the pointer as a value the compiler cannot trace (xyzz_add_into_steps: keeps address arithmetic inside the step)
  __device__ __forceinline__ void launder() { asm volatile("" : "+v"(p)); }
};
template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK, (BLOCK == 256 ? 2 : 1))
msm_finalize_lds_kernel(MsmGeom g, size_t total, unsigned wg_log, unsigned lpb_log,
                        const unsigned* __restrict__ counts,
                        const unsigned* __restrict__ seg_off, const XYZZ29<F>* __restrict__ seg_sum,
                        XYZZ29<F>* __restrict__ buckets, unsigned* __restrict__ giant_count,
                        unsigned* __restrict__ giant_list, unsigned giant_cap) {
  using FO = FieldOf<F>;
  constexpr int WORDS = sizeof(typename FO::Store) / 4;
  __shared__ uint32_t sh[4 * WORDS][BLOCK];
  __shared__ unsigned max_serial;
  if constexpr (sizeof(F) > 64) asm volatile("" ::: DG_ACC_FILE_CLOBBERS);   // xyzz_add_into_steps' temporaries (acc_set / acc_get)
  const unsigned LPB = 1u << lpb_log;
  const unsigned lane = threadIdx.x, sub = lane & (LPB - 1);
  const size_t gid = ((size_t)blockIdx.x * BLOCK + lane) >> lpb_log;
  const ColAcc<F, BLOCK> me{sh, lane};
  unsigned np = 0, first = 0;
  unsigned wy = 0;
  if (lane == 0) max_serial = 0;
  if (gid < total) {
    wy = (unsigned)(gid >> g.log_nb);
    const size_t gs = DG_IDX(6, ((size_t)(wy % g.bw) << g.log_nb) + (gid & (((size_t)1 << g.log_nb) - 1)), (size_t)g.bw << g.log_nb);
    const unsigned k = (counts[gs] + (1u << g.seg_log) - 1) >> g.seg_log;
    first = seg_off[gs];
    np = msm_nparts(first, k, wg_log);
    if (sub == 0) {
      if (np == 0) buckets[gid] = XYZZ29<F>::inf();
      if (np > kGiantSegs) msm_register_giant((unsigned)gid, np, giant_count, giant_list, giant_cap);
    }
  }
  const bool work = np >= 2 && np <= kGiantSegs;      // np == 1: the accumulation wrote the bucket itself
  const XYZZ29<F>* sp = seg_sum + (size_t)wy * g.seg_cap;
  const unsigned lo = work ? (unsigned)(((uint64_t)sub * np) >> lpb_log) : 0u;
  const unsigned hi = work ? (unsigned)(((uint64_t)(sub + 1) * np) >> lpb_log) : 0u;
  if (lo < hi) {
    const XYZZ29<F>* q = &sp[DG_IDX(7, msm_part_slot(first, lo, wg_log), g.seg_cap)];
    me.put(0, q->x); me.put(1, q->y); me.put(2, q->zz); me.put(3, q->zzz);
  } else {
    me.put(2, FO::zero());                              // the identity for add_into: zz = 0
  }
  const unsigned nser = lo < hi ? hi - lo - 1 : 0u;     // this lane's serial additions
  __syncthreads();
  atomicMax(&max_serial, nser);
  __syncthreads();
  const unsigned ms = max_serial;
  // steps 0 .. ms - 1: my share of the bucket's partials, one after another; then lpb_log tree steps over
  // neighbouring columns (behind a barrier each)
#pragma unroll 1
  for (unsigned step = 0; step < ms + lpb_log; step++) {
    const bool tree = step >= ms;
    if (tree) __syncthreads();
    const unsigned d = tree ? 1u << (step - ms) : 0u;
    const bool on = tree ? (work && (sub & (2 * d - 1)) == 0) : step < nser;
    if (on) {
      const PartialAcc<F, BLOCK> b{tree ? nullptr : &sp[DG_IDX(7, msm_part_slot(first, lo + 1 + step, wg_log), g.seg_cap)],
                                   ColAcc<F, BLOCK>{sh, (unsigned)DG_IDX(10, lane + d, BLOCK)}};
      // 14-limb Fq2: the addition as a step loop over the accumulation's three product sites (35 900 -> 22 700
      // instructions; 21.8 -> 21.5 ms per BLS12-381 2^20 proof, same call, twice: profiles/r6b_finalize_steps_ab.txt).
      // Withdrawn in round 5 behind an HSA aperture violation, back in round 6 with its cause removed: hipcc had put
      // sixteen of its own spills into the temporaries' register file (kAccFileBase).
      if constexpr (sizeof(F) > 64) xyzz_add_into_steps<F>(me, b);
      else XYZZ29<F>::add_into(me, b);
    }
  }
  if (work && sub == 0) {
    XYZZ29<F> out = XYZZ29<F>::inf();
    if (!limbs_all_zero(me.get(2))) out = XYZZ29<F>{me.get(0), me.get(1), me.get(2), me.get(3)};
    buckets[gid] = out;
  }
}
// G2 finalize as a throughput kernel behind the accumulation, two lanes per bucket (measured in round 4 against one and
// four lanes and against the one-lane-per-bucket kernel on the reduction stream: profiles/r4r_finalize_lpb_ab.txt,
// r3b_finalize_lds_ab.txt -- the switches are gone)
inline int msm_finalize_lds_lpb() { return 2; }
template <class F>
void msm_finalize_lds_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b) {
  constexpr int BLOCK = 1 << msm_acc_block_log<F>();
  const int lpb = msm_finalize_lds_lpb();
  const unsigned lpb_log = lpb == 4 ? 2u : lpb == 2 ? 1u : 0u;
  DG_HIP(hipMemsetAsync(b.giant, 0, 8, s));
  const unsigned blocks = (unsigned)((b.nbw * (size_t)lpb + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL((msm_finalize_lds_kernel<F, BLOCK>), dim3(blocks), dim3(BLOCK), 0, s, st.g, b.nbw,
                     msm_acc_wg_log<F>(), lpb_log, st.counts, st.seg_off, b.seg_sum, b.buckets, b.giant, b.giant + 2,
                     b.giant_cap);
}

// With the in-workgroup tree the finalize shrinks to a STITCH: one lane per accumulation-workgroup BOUNDARY (a few
// thousand lanes, not one per bucket) looks at the bucket that straddles it and, if this is the first boundary that
// bucket crosses, adds the partials its workgroups left.  Buckets held by one workgroup were written by it, empty ones
// are zeroed by msm_empty_buckets_kernel (all-zero limbs ARE the identity: zz = 0).  A 65 536-lane finalize with nothing
// to do still took 0.2-0.5 ms inside a proof, waiting for wave slots next to the accumulation.
template <class F, int TU = 0>
__global__ void __launch_bounds__(256) msm_stitch_kernel(MsmGeom g, unsigned wg_log, const unsigned* __restrict__ counts,
                                                          const unsigned* __restrict__ seg_off,
                                                          const unsigned* __restrict__ seg_total,
                                                          const XYZZ29<F>* __restrict__ seg_sum,
                                                          XYZZ29<F>* __restrict__ buckets,
                                                          unsigned* __restrict__ giant_count,
                                                          unsigned* __restrict__ giant_list, unsigned giant_cap) {
  const unsigned wy = blockIdx.y, w = wy % g.bw;
  const unsigned bd = blockIdx.x * blockDim.x + threadIdx.x + 1;      // boundary between workgroups bd - 1 and bd
  const unsigned slot = bd << wg_log;
  if (slot >= seg_total[w]) return;
  const unsigned* so = seg_off + ((size_t)w << g.log_nb);
  unsigned lo = 0, hi = 1u << g.log_nb;                                 // the bucket whose segments contain `slot`
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (so[mid] <= slot) lo = mid; else hi = mid;
  }
  const unsigned first = so[lo];
  if (first == slot || (first >> wg_log) != bd - 1) return;             // starts here, or crossed an earlier boundary
  const size_t gs = ((size_t)w << g.log_nb) + lo;
  const size_t gid = ((size_t)wy << g.log_nb) + lo;
  const unsigned k = (counts[gs] + (1u << g.seg_log) - 1) >> g.seg_log;
  const unsigned np = msm_nparts(first, k, wg_log);
  if (np > kGiantSegs) {
    msm_register_giant((unsigned)gid, np, giant_count, giant_list, giant_cap);
    return;
  }
  const XYZZ29<F>* sp = seg_sum + (size_t)wy * g.seg_cap;
  XYZZ29<F> acc = sp[DG_IDX(7, first, g.seg_cap)];
#pragma unroll 1
  for (unsigned s = 1; s < np; s++) acc = acc.add(sp[DG_IDX(7, msm_part_slot(first, s, wg_log), g.seg_cap)]);
  buckets[gid] = acc;
}
// ... and the empty buckets are set to the identity by a kernel of a dozen registers per lane (it fits next to any
// accumulation wave; a memset on the main stream in front of the accumulation cost a launch boundary per MSM: +0.2 ms
// per proof, measured)
template <class F>
__global__ void __launch_bounds__(256) msm_empty_buckets_kernel(MsmGeom g, size_t total, const unsigned* __restrict__ counts,
                                                                 XYZZ29<F>* __restrict__ buckets) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const unsigned wy = (unsigned)(gid >> g.log_nb);
  const size_t gs = ((size_t)(wy % g.bw) << g.log_nb) + (gid & (((size_t)1 << g.log_nb) - 1));
  if (counts[gs]) return;
  uint4* dst = reinterpret_cast<uint4*>(buckets + gid);
  static_assert(sizeof(XYZZ29<F>) % 16 == 0, "vector stores");
#pragma unroll
  for (unsigned i = 0; i < sizeof(XYZZ29<F>) / 16; i++) dst[i] = make_uint4(0u, 0u, 0u, 0u);   // zz = 0: the identity
}
template <class F>
void msm_finalize_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b) {
  DG_BOUNDS_BIND();
  if constexpr (msm_acc_tree<F>()) {
    hipLaunchKernelGGL(msm_empty_buckets_kernel<F>, dim3((unsigned)((b.nbw + 255) / 256)), dim3(256), 0, s, st.g, b.nbw,
                       st.counts, b.buckets);
    const unsigned nbd = (st.g.seg_cap >> msm_acc_wg_log<F>()) + 1;
    hipLaunchKernelGGL(msm_stitch_kernel<F>, dim3((nbd + 255) / 256, st.g.bw * b.ninst), dim3(256), 0, s, st.g,
                       msm_acc_wg_log<F>(), st.counts, st.seg_off, st.seg_total, b.seg_sum, b.buckets, b.giant,
                       b.giant + 2, b.giant_cap);
  } else {
    bool lane = false;
    if constexpr (lane29::enabled<F>()) {
      static const bool off = [] { const char* e = getenv("DG16_NO_LANE_FINALIZE"); return e && atoi(e) != 0; }();
      const size_t partials = ((st.g.region * st.g.bw * b.ninst) >> st.g.seg_log) + b.nbw;      // (an upper bound)
      lane = !off && !b.busy_chip && partials <= kLaneFinalizeMaxPartials && b.nbw < ((size_t)1 << 31);
    }
    if (lane)
      hipLaunchKernelGGL(msm_finalize_lane_kernel<F>, dim3((unsigned)b.nbw), dim3(64), 0, s, st.g, msm_acc_wg_log<F>(),
                         st.counts, st.seg_off, b.seg_sum, b.buckets, b.giant, b.giant + 2, b.giant_cap);
    else
      hipLaunchKernelGGL(msm_finalize_thr_kernel<F>, dim3((unsigned)((b.nbw + 255) / 256)), dim3(256), 0, s, st.g, b.nbw,
                         msm_acc_wg_log<F>(), st.counts, st.seg_off, b.seg_sum, b.buckets, b.giant, b.giant + 2,
                         b.giant_cap);
  }
  DG_HIP(hipGetLastError());
}

// Phase B (latency-bound, few waves): finalize -> giants -> rows -> top -> tail.  May run on another stream than
// phase A so that it hides behind the next MSM's accumulation.  Defined in msm_reduce_impl.h and instantiated once per
// (curve, group) in msm_reduce.hip -- a translation unit of its own because its kernels are compiled with out-of-line
// field products (DG29_OUTLINE_MUL, fp29.h).
// out_dev: b.ninst results back to back (Jacobian x, y, z -- or affine x, y -- of instance 0, then instance 1, ..)
template <class F>
void msm_bucket_phase(hipStream_t s, const MsmSort& st, const MsmBuffers<F>& b, bool out_affine, void* out_dev);

// both phases on the call's own stream and workspace
template <class F>
void msm_reduce(Call& k, const MsmSort& st, const void* bases, bool out_affine, void* out_dev) {
  MsmBuffers<F> b = msm_buffers<F>(k.c, st.g);
  k.begin_dominant();
  b.acc_done = k.c.ev[3];                    // = end_dominant(), but in front of the G2 finalize
  if (k.ctx->kclk) b.clk = k.ctx->kclk + 2 * (&k.c - k.ctx->ch);
  msm_accumulate_phase<F>(k.s(), st, b, bases);
  k.c.ev_valid[1] = true;
  msm_bucket_phase<F>(k.s(), st, b, out_affine, out_dev);
}

// ---- GLV for the plain G1 MSM (glv.h): 2n points (P_i, phi(P_i)), 127-bit half scalars, half the windows ------------
template <class F> struct GlvOf { static constexpr bool enabled = false; };
// phi(P) = LAMBDA P (psi(P) = LAMBDA P) holds for P in the order-r subgroup ONLY.  A group of cofactor one is that subgroup
// (BN254 G1); for every other group the split needs the caller's word that the bases are in it
// (DG16_F_BASES_IN_SUBGROUP) -- an on-curve point outside the subgroup (decoded with validate = 0, say) must still give
// the group element VariableBaseMSM::msm gives, so without the flag those groups run the unsplit path.
template <class F> struct GlvCofactorOne { static constexpr bool value = false; };
template <> struct GlvCofactorOne<Fp<bn254_fq_params>> { static constexpr bool value = true; };
// G1 of the three curves (j = 0): phi(x, y) = (BETA x, y)
template <class P, class GC>
struct GlvG1 {
  static constexpr bool enabled = true;
  static constexpr int DIM = 2;
  using C = GC;
  DG_HD static void endo(Affine<Fp<P>>& p) {
    Fp<P> beta;
#pragma unroll
    for (int k = 0; k < Fp<P>::NL; k++) beta.l[k] = GC::BETA[k];
    p.x = p.x * beta;
  }
};
template <> struct GlvOf<Fp<bn254_fq_params>> : GlvG1<bn254_fq_params, bn254_glv_consts> {};
template <> struct GlvOf<Fp<bls12_381_fq_params>> : GlvG1<bls12_381_fq_params, bls12_381_glv_consts> {};
template <> struct GlvOf<Fp<bls12_377_fq_params>> : GlvG1<bls12_377_fq_params, bls12_377_glv_consts> {};
// G2 of the three curves: psi(x, y) = (GAMMA_X conj(x), GAMMA_Y conj(y)) = LAMBDA (x, y) (untwist, Frobenius, twist),
// LAMBDA a root of x^4 - x^2 + 1 mod r.  DIM = 4: the four-dimensional split (glv.h: split4) -- 4n points P, psi P,
// psi^2 P, psi^3 P and quarters of at most 65 bits; DIM = 2: split() over psi alone.
template <class P, class GC, int D>
struct GlvG2 {
  static constexpr bool enabled = true;
  static constexpr int DIM = D;
  using C = GC;
  using Fq = Fp<P>;
  DG_HD static void endo(Affine<Fp2<Fq>>& p) {
    if (p.is_inf()) return;
    Fp2<Fq> gx, gy;
#pragma unroll
    for (int k = 0; k < Fq::NL; k++) {
      gx.c0.l[k] = GC::GAMMA_X_C0[k]; gx.c1.l[k] = GC::GAMMA_X_C1[k];
      gy.c0.l[k] = GC::GAMMA_Y_C0[k]; gy.c1.l[k] = GC::GAMMA_Y_C1[k];
    }
    p.x = Fp2<Fq>{p.x.c0, p.x.c1.neg()} * gx;
    p.y = Fp2<Fq>{p.y.c0, p.y.c1.neg()} * gy;
  }
};
// BN254: LAMBDA ~ 2^127, so the TWO-dimensional split over psi alone is balanced too, and it is the faster one there
// (2^20 points: 6.93 ms against 7.24 for the four-dimensional form, same call -- twice the points to sort and convert
// and a fifth, nearly empty window cost more than the shorter tail saves: profiles/r4n_glv4_ab.txt).  A BLS12 curve has
// q = u mod r, 64 bits: only the four-dimensional form is balanced (BLS12-381 2^20: 19.8 -> 16.2 ms).
template <> struct GlvOf<Fp2<Fp<bn254_fq_params>>> : GlvG2<bn254_fq_params, bn254_g2_glv_consts, 2> {};
template <> struct GlvOf<Fp2<Fp<bls12_381_fq_params>>> : GlvG2<bls12_381_fq_params, bls12_381_g2_glv4_consts, 4> {};
template <> struct GlvOf<Fp2<Fp<bls12_377_fq_params>>> : GlvG2<bls12_377_fq_params, bls12_377_g2_glv4_consts, 4> {};
constexpr int kGlvBits = 127;      // |k1|, |k2| < 2^127 (measured bound: 0.81 x 2^127 over all 255-bit inputs; tests/test_host_arith.py)
constexpr int kGlv4Bits = 65;      // the quarters of split4: < 2^65 (BLS12-381: 0.52 x 2^64 for canonical scalars; one spare bit for non-canonical 255-bit inputs)

template <class Fr, class GC>
__global__ void __launch_bounds__(256) glv_split_kernel(const Fr* __restrict__ scalars, size_t n, int mont,
                                                         Fr* __restrict__ halves /* [2 n]: |k1| .., then |k2| .. */) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = scalars[i];
  if (mont) s = s.from_mont();
  Fr h1, h2;
  glv::split<GC>(s.l, h1.l, h2.l);
  halves[i] = h1;
  halves[n + i] = h2;
}
template <class Fr, class GC>
__global__ void __launch_bounds__(256) glv_split4_kernel(const Fr* __restrict__ scalars, size_t n, int mont,
                                                          Fr* __restrict__ quarters /* [4 n]: |k0| .., |k1| .., .. */) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = scalars[i];
  if (mont) s = s.from_mont();
  Fr h0, h1, h2, h3;
  glv::split4<GC>(s.l, h0.l, h1.l, h2.l, h3.l);
  quarters[i] = h0;
  quarters[n + i] = h1;
  quarters[2 * n + i] = h2;
  quarters[3 * n + i] = h3;
}
// bases -> internal form, DIM times: P_i at i, its images under the endomorphism at n + i, 2n + i, .. (the identity maps to itself)
template <class F>
__global__ void __launch_bounds__(256) msm_to_internal_glv_kernel(const Affine<F>* __restrict__ in, size_t n,
                                                                   uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int PW = 2 * FieldOf<F>::WORDS;
  Affine<F> p = in[i];
  uint32_t w[PW];
  affine_to_internal(p, w);
  uint4* dst = reinterpret_cast<uint4*>(out + i * PW);
#pragma unroll
  for (int k = 0; k < PW / 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
#pragma unroll 1
  for (int img = 1; img < GlvOf<F>::DIM; img++) {
    GlvOf<F>::endo(p);
    affine_to_internal(p, w);
    dst = reinterpret_cast<uint4*>(out + ((size_t)img * n + i) * PW);
#pragma unroll
    for (int k = 0; k < PW / 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  }
}

// mode: bit 0 = scalars in Montgomery form, bit 1 = every base is in the order-r subgroup (ctx.h: msm_mode)
template <class F, class Fr, int SCALAR_BITS>
void msm_run(Call& k, const void* bases, const void* scalars, size_t n, unsigned mode, bool out_affine,
             void* out_dev) {
  const bool scalars_mont = mode & 1u;
  if (GlvOf<F>::enabled && (GlvCofactorOne<F>::value || (mode & 2u))) {
   if constexpr (GlvOf<F>::enabled) {
    // Split every scalar with the curve's endomorphism.  Same number of bucket
    // entries (2n points x half the windows), half the windows: half the dependent doublings of the Horner tail, half the
    // bucket sets to reduce, twice the entries per bucket (longer, better balanced accumulation segments).
    constexpr size_t DIM = GlvOf<F>::DIM;
    if (n && DIM * n * 40 < ((size_t)1 << 31)) {
      using GC = typename GlvOf<F>::C;
      Fr* halves = (Fr*)ws(k.c, 30, DIM * n * sizeof(Fr));
      MsmSort st;
      if constexpr (DIM == 2) {
        hipLaunchKernelGGL((glv_split_kernel<Fr, GC>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k.s(),
                           (const Fr*)scalars, n, (int)scalars_mont, halves);
        // Window width at SMALL sizes (round 6, profiles/r6b_msm_window_sweep.txt): the halves have 127 + 1 bits, and a
        // width of 7 or 9 (what log2(2 n) - 4 gives at n = 2^10 / 2^12) leaves a top window of two bits whose few buckets
        // turn giant -- 8 divides 128: BN254 G1 2^10 0.663 -> 0.606 ms, 2^12 0.685 -> 0.649; G2 1.68 -> 1.47, 1.92 -> 1.62
        // (same call).  The 14-limb G1 groups measured the other way (2^12: 1.33 -> 1.49 ms) and keep the rule.
        // ... and at LARGE sizes (end of round 6, profiles/r6zu_*, r6zt_*): 16 divides 128 as well -- eight windows instead of
        // the ten / nine of log2(2 n) - 4 = 14 / 15 (whose top windows are 2 / 8 bits), i.e. a fifth fewer additions per
        // point, and the bucket reductions are cheap enough since the lane forms to take 2^15 buckets per window: 2^17
        // points BN254 G1 1.24 -> 0.93 ms, BLS12-377 1.70 -> 1.60, BLS12-381 1.78 -> 1.59; 2^18 points 1.17 -> 1.15 / 2.33 ->
        // 2.06 / 2.64 -> 2.07 (same call, twice).  The G2 groups (four 64-bit quarters) measured mixed and keep the rule.
        unsigned c_small = 0;
        {
          const unsigned c0 = msm_window_bits(2 * n, false);
          if constexpr (RR<typename FieldOf<F>::Params>::N == 9)
            if ((c0 == 7 || c0 == 9) && !getenv("DG16_MSM_C")) c_small = 8;
          if ((c0 == 14 || c0 == 15) && !getenv("DG16_MSM_C")) c_small = 16;
        }
        st = msm_sort<Fr, kGlvBits>(k, halves, 2 * n, 2u, false, c_small);
      } else {
        hipLaunchKernelGGL((glv_split4_kernel<Fr, GC>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k.s(),
                           (const Fr*)scalars, n, (int)scalars_mont, halves);
        st = msm_sort<Fr, kGlv4Bits>(k, halves, 4 * n, 2u, false);
      }
      uint32_t* internal = (uint32_t*)ws(k.c, 24, DIM * n * sizeof(Affine<F>));
      hipLaunchKernelGGL(msm_to_internal_glv_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k.s(),
                         (const Affine<F>*)bases, n, internal);
      msm_reduce<F>(k, st, internal, out_affine, out_dev);
      return;
    }
   }
  }
  MsmSort st = msm_sort<Fr, SCALAR_BITS>(k, scalars, n, scalars_mont ? 1u : 0u, false);
  uint32_t* internal = (uint32_t*)ws(k.c, 24, (n ? n : 1) * sizeof(Affine<F>));
  if (n)
    hipLaunchKernelGGL(msm_to_internal_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k.s(),
                       (const Affine<F>*)bases, n, internal);
  // (ONE accumulation launch, then the reduction chain: the two-launch pipeline that overlapped the upper windows' chain with
  // the lower windows' accumulation was slower -- CHANGELOG.md round 4, profiles/r4e_msm_pipeline_ab.md)
  msm_reduce<F>(k, st, internal, out_affine, out_dev);
}

// ---- table of window multiples for resident bases: T[r*n + i] = 2^(c_step*r) * P_i (affine), r < rows -----------
// (c_step = c * stride: a full table has stride 1 and one row per window; a thinned one keeps every stride-th row)
constexpr unsigned kMaxTableWin = 64;
template <class F>
__global__ void __launch_bounds__(64) msm_table_kernel(const Affine<F>* __restrict__ bases, size_t n, unsigned c,
                                                        unsigned nwin, Affine<F>* __restrict__ table) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = bases[i];
  // rows are stored in the accumulation kernels' internal form (same size; see msm_to_internal_kernel)
  auto put = [&](size_t at, const Affine<F>& v) { affine_to_internal(v, reinterpret_cast<uint32_t*>(table + at)); };
  put(i, p);
  if (p.is_inf()) {
    for (unsigned w = 1; w < nwin; w++) put((size_t)w * n + i, p);
    return;
  }
  // rows 1..nwin-1 by repeated doubling; one shared inversion (Montgomery's trick over the rows)
  XYZZ<F> pts[kMaxTableWin];
  F pref[kMaxTableWin];
  XYZZ<F> cur = XYZZ<F>::from_affine(p);
  F run = F::one();
  for (unsigned w = 1; w < nwin; w++) {
    for (unsigned j = 0; j < c; j++) cur = cur.dbl();
    pts[w] = cur;
    pref[w] = run;
    // a point of odd prime order never doubles to the identity; tolerate small-order inputs anyway
    run = run * (cur.is_inf() ? F::one() : cur.zzz);
  }
  F inv = run.inv();
  for (unsigned w = nwin - 1; w >= 1; w--) {
    if (pts[w].is_inf()) { put((size_t)w * n + i, Affine<F>::inf()); continue; }
    F zi3 = inv * pref[w];
    inv = inv * pts[w].zzz;
    F zi2 = (zi3 * pts[w].zz).sqr();
    put((size_t)w * n + i, Affine<F>{pts[w].x * zi2, pts[w].y * zi3});
  }
}

// Row stride of a table under an HBM budget: the smallest k such that ceil(nwin / k) rows fit (0 = no budget -> 1).
inline unsigned table_stride_for(size_t full_bytes, size_t budget, unsigned nwin) {
  if (!budget || full_bytes <= budget || nwin <= 1) return 1;
  const size_t row = full_bytes / nwin;
  size_t rows_fit = budget / (row ? row : 1);
  if (rows_fit < 1) rows_fit = 1;                         // (one row -- the bases themselves -- is the floor)
  unsigned k = (unsigned)((nwin + rows_fit - 1) / rows_fit);
  return k < 1 ? 1 : k > nwin ? nwin : k;
}
// returns a device table of nwin*n affine points (caller owns it) for window size c
template <class F>
void* msm_build_table(hipStream_t s, const void* bases, size_t n, unsigned c, unsigned nwin) {
  DG_REQUIRE(nwin <= kMaxTableWin, DG16_ERR_BAD_ARG, "too many table windows");
  void* t = nullptr;
  DG_HIP(hipMalloc(&t, (size_t)nwin * (n ? n : 1) * sizeof(Affine<F>)));
  if (n)
    hipLaunchKernelGGL(msm_table_kernel<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const Affine<F>*)bases, n,
                       c, nwin, (Affine<F>*)t);
  DG_HIP(hipGetLastError());
  return t;
}

// ---- synthetic bases: P_i = (k0 + i*k1) * G -----------------------------------------------------------
constexpr unsigned kGenChunk = 64;

template <class F, class C>
__global__ void gen_setup_kernel(const uint32_t* k1_words, Affine<F>* d_out) {
  Affine<F> G = GenLoader<F, C>::get();
  XYZZ<F> d = scalar_mul<F, 4>(XYZZ<F>::from_affine(G), k1_words);
  *d_out = d.to_affine();
}

template <class F, class C>
__global__ void __launch_bounds__(64) gen_bases_kernel(const uint32_t* k0_words, const uint32_t* k1_words,
                                                        const Affine<F>* d_ptr, size_t n, Affine<F>* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t lo = t * kGenChunk;
  if (lo >= n) return;
  size_t hi = lo + kGenChunk < n ? lo + kGenChunk : n;
  // k = k0 + lo * k1  (128-bit * 64-bit + 128-bit  <  2^193)
  uint32_t kk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  {
    uint32_t lo_w[2] = {(uint32_t)lo, (uint32_t)((uint64_t)lo >> 32)};
    uint64_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 2; j++) {
        uint64_t pr = (uint64_t)k1_words[i] * lo_w[j];
        acc[i + j] += (uint32_t)pr;
        acc[i + j + 1] += pr >> 32;
      }
    for (int i = 0; i < 4; i++) acc[i] += k0_words[i];
    uint64_t carry = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t v = acc[i] + carry;
      kk[i] = (uint32_t)v;
      carry = v >> 32;
    }
  }
  Affine<F> G = GenLoader<F, C>::get();
  Affine<F> D = *d_ptr;
  XYZZ<F> cur = scalar_mul<F, 7>(XYZZ<F>::from_affine(G), kk);
  // walk the chunk; batch-invert zzz with Montgomery's trick (scratch arrays live in private memory)
  XYZZ<F> pts[kGenChunk];
  F pref[kGenChunk];
  F run = F::one();
  size_t cnt = hi - lo;
  for (size_t i = 0; i < cnt; i++) {
    pts[i] = cur;
    pref[i] = run;
    run = run * cur.zzz;
    cur = cur.madd(D, false);
  }
  F inv = run.inv();
  for (size_t i = cnt; i-- > 0;) {
    F zi3 = inv * pref[i];          // 1 / zzz_i
    inv = inv * pts[i].zzz;
    F zi2 = (zi3 * pts[i].zz).sqr();
    out[lo + i] = {pts[i].x * zi2, pts[i].y * zi3};
  }
}

template <class F, class C>
void gen_bases_run(Call& k, uint64_t seed, size_t n, void* out_dev);

template <class F>
__global__ void __launch_bounds__(64) to_affine_kernel(const Jacobian<F>* __restrict__ in, Affine<F>* __restrict__ out,
                                                        size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = XYZZ<F>::from_jacobian(in[i]).to_affine();
}

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

template <class F, class C>
void gen_bases_run(Call& k, uint64_t seed, size_t n, void* out_dev) {
  // same (k0, k1) derivation as the checker uses, so generated bases can be compared bit for bit
  uint64_t k0[2] = {splitmix64(seed ^ 0xA5A5), splitmix64(seed ^ 0x5A5A)};
  uint64_t k1[2] = {splitmix64(seed ^ 0x1234) | 1, splitmix64(seed ^ 0x4321)};
  uint32_t host_words[8] = {(uint32_t)k0[0], (uint32_t)(k0[0] >> 32), (uint32_t)k0[1], (uint32_t)(k0[1] >> 32),
                            (uint32_t)k1[0], (uint32_t)(k1[0] >> 32), (uint32_t)k1[1], (uint32_t)(k1[1] >> 32)};
  uint8_t* scratch = (uint8_t*)ws(k.c, 16, 64 + sizeof(Affine<F>));
  uint32_t* words = (uint32_t*)scratch;
  Affine<F>* d = (Affine<F>*)(scratch + 64);
  DG_HIP(hipMemcpyAsync(words, host_words, sizeof host_words, hipMemcpyHostToDevice, k.s()));
  DG_HIP(hipStreamSynchronize(k.s()));   // host_words is a stack buffer
  hipLaunchKernelGGL((gen_setup_kernel<F, C>), dim3(1), dim3(1), 0, k.s(), words + 4, d);
  size_t threads = (n + kGenChunk - 1) / kGenChunk;
  hipLaunchKernelGGL((gen_bases_kernel<F, C>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, k.s(), words,
                     words + 4, d, n, (Affine<F>*)out_dev);
  DG_HIP(hipGetLastError());
}

template <class F>
void to_affine_run(Call& k, const void* jac, void* out, size_t n) {
  hipLaunchKernelGGL(to_affine_kernel<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, k.s(),
                     (const Jacobian<F>*)jac, (Affine<F>*)out, n);
  DG_HIP(hipGetLastError());
}

}  // namespace dg16

// Translation units that only CALL the MSM phases of a curve (the prover) declare them extern so that the kernels are
// compiled once, in msm_group.hip / msm_reduce.hip:  namespace dg16 { DG16_MSM_EXTERN(CurveTypes<0>) }
#define DG16_MSM_EXTERN_GROUP(F)                                                                                  \
  extern template void msm_accumulate_phase<F>(hipStream_t, const MsmSort&, const MsmBuffers<F>&, const void* const*); \
  extern template void msm_finalize_phase<F>(hipStream_t, const MsmSort&, const MsmBuffers<F>&);                   \
  extern template void msm_tail_phase<F>(hipStream_t, const MsmSort&, const MsmBuffers<F>&, bool, void*);         \
  extern template void* msm_build_table<F>(hipStream_t, const void*, size_t, unsigned, unsigned);
#define DG16_MSM_EXTERN(CT)                                                                                       \
  DG16_MSM_EXTERN_GROUP(CT::Fq)                                                                                   \
  DG16_MSM_EXTERN_GROUP(CT::Fq2)                                                                                  \
  extern template MsmSort msm_sort_on<CT::Fr, CT::SCALAR_BITS>(hipStream_t, Channel&, const void*, size_t, unsigned, bool, \
                                                              unsigned, unsigned);

