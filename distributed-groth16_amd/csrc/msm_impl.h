// Pippenger MSM for gfx950 -- replaces `G::msm(bases, scalars)` at
// dist-primitives/src/dmsm/mod.rs:82 (ark-ec VariableBaseMSM).  Any correct algorithm yields the
// same group element; parity is checked in affine form.
//
// Pipeline (all on the GPU, one stream):
//   1 digits      scalar -> W signed c-bit digits (carry iff digit > 2^(c-1)); per-(window,bucket)
//                 histogram with global atomics.                               HBM: 32 B/scalar in
//   2 scan        exclusive prefix sums of the histogram (entry offsets) and of the per-bucket
//                 segment counts ceil(cnt/16), one workgroup per window
//   3 scatter     counting-sort placement of (point index | sign) into per-window bucket order;
//                 the lane that lands on rank 0, 16, 32, .. of a bucket also records the segment
//   4 accumulate  one lane per SEGMENT (<= 16 consecutive entries of one bucket): XYZZ mixed
//                 additions (8M+2S).  Load-balanced for any digit distribution (the top window and
//                 real witnesses are heavily skewed; lane-per-bucket ran 4.7x below the VALU rate).
//                 THE dominant kernel: ~N*W*10 Montgomery multiplications, VALU-bound; the
//                 64 B/point gather is served mostly from the 256 MiB Infinity Cache
//   4b finalize   per bucket: sum of its segment partials (giant buckets: one workgroup each)
//   5 reduce      sum_b (b+1)*B[w][b]: chunks of 8 buckets by running sums + a <=16-bit scalar
//                 multiple per chunk, then an LDS tree per window
//   6 tail        Horner over the W window sums (W*c doublings, inherently serial), -> Jacobian
//
// Data layout in HBM: bases n x (x||y) Montgomery as handed over; digits/entries int32 [W][n];
// histogram/offsets uint32 [W][2^(c-1)]; segment map uint32 and segment sums XYZZ [W][2^(c-1)+n/16];
// buckets XYZZ [W][2^(c-1)].
#pragma once
#include "ctx.h"
#include "types.h"

namespace dg16 {

constexpr unsigned kSegLog = 4;          // entries per accumulation segment = 16
constexpr unsigned kSeg = 1u << kSegLog;
constexpr unsigned kGiantSegs = 32;      // buckets with more segments are reduced by a whole workgroup

struct MsmGeom {
  unsigned c;        // window bits
  unsigned nwin;     // W
  unsigned log_nb;   // log2 buckets per window = c - 1
  unsigned seg_cap;  // segment slots per window = 2^log_nb + ceil(n / kSeg)
};

inline MsmGeom msm_geometry(size_t n, unsigned scalar_bits) {
  unsigned lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = (int)lg - 4;
  if (const char* e = getenv("DG16_MSM_C")) c = atoi(e);
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  MsmGeom g;
  g.c = (unsigned)c;
  g.nwin = (scalar_bits + 1 + g.c - 1) / g.c;   // one spare bit absorbs the last carry
  g.log_nb = g.c - 1;
  g.seg_cap = (1u << g.log_nb) + (unsigned)((n + kSeg - 1) >> kSegLog);
  return g;
}

// ---- 1: digits + histogram -------------------------------------------------------------------
template <class Fr>
__global__ void __launch_bounds__(256) msm_digits_kernel(const Fr* __restrict__ scalars, size_t n, int mont,
                                                          MsmGeom g, int* __restrict__ digits,
                                                          unsigned* __restrict__ counts) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = scalars[i];
  if (mont) s = s.from_mont();
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
    digits[(size_t)w * n + i] = d;
    if (d != 0) {
      unsigned b = (unsigned)(d < 0 ? -d : d) - 1;
      atomicAdd(&counts[((size_t)w << g.log_nb) + b], 1u);
    }
  }
}

// ---- 2: per-window exclusive scan (one 1024-thread workgroup per window) ------------------------
// (templated only so that every translation unit carries its own copy of the kernel)
template <int TU>
__global__ void __launch_bounds__(1024) msm_scan_kernel(const unsigned* __restrict__ counts,
                                                         unsigned* __restrict__ offsets,
                                                         unsigned* __restrict__ seg_off,
                                                         unsigned* __restrict__ seg_total,
                                                         unsigned* __restrict__ cursor, unsigned log_nb) {
  __shared__ unsigned sh[1024];
  __shared__ unsigned sh2[1024];
  const unsigned nb = 1u << log_nb;
  const unsigned ipt = (nb + 1023) / 1024;
  const size_t base = (size_t)blockIdx.x << log_nb;
  const unsigned lo = threadIdx.x * ipt;
  unsigned sum = 0, ssum = 0;
  for (unsigned j = 0; j < ipt; j++)
    if (lo + j < nb) {
      unsigned cn = counts[base + lo + j];
      sum += cn;
      ssum += (cn + kSeg - 1) >> kSegLog;
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
  unsigned run = sh[threadIdx.x] - sum;   // exclusive prefix of this thread's segment
  unsigned srun = sh2[threadIdx.x] - ssum;
  for (unsigned j = 0; j < ipt; j++)
    if (lo + j < nb) {
      unsigned cn = counts[base + lo + j];
      offsets[base + lo + j] = run;
      seg_off[base + lo + j] = srun;
      cursor[base + lo + j] = 0;
      run += cn;
      srun += (cn + kSeg - 1) >> kSegLog;
    }
  if (threadIdx.x == 1023) seg_total[blockIdx.x] = sh2[1023];
}

// ---- 3: scatter ---------------------------------------------------------------------------------
template <int TU>
__global__ void __launch_bounds__(256) msm_scatter_kernel(const int* __restrict__ digits, size_t n, MsmGeom g,
                                                           const unsigned* __restrict__ offsets,
                                                           const unsigned* __restrict__ seg_off,
                                                           unsigned* __restrict__ cursor,
                                                           unsigned* __restrict__ entries,
                                                           unsigned* __restrict__ seg_bucket) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (unsigned w = 0; w < g.nwin; w++) {
    int d = digits[(size_t)w * n + i];
    if (d == 0) continue;
    unsigned b = (unsigned)(d < 0 ? -d : d) - 1;
    size_t slot = ((size_t)w << g.log_nb) + b;
    unsigned rank = atomicAdd(&cursor[slot], 1u);
    entries[(size_t)w * n + offsets[slot] + rank] = (unsigned)i | (d < 0 ? 0x80000000u : 0u);
    if ((rank & (kSeg - 1)) == 0) seg_bucket[(size_t)w * g.seg_cap + seg_off[slot] + (rank >> kSegLog)] = b;
  }
}

// ---- 4: segment accumulation -------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const Affine<F>* __restrict__ bases, size_t n,
                                                              MsmGeom g, const unsigned* __restrict__ offsets,
                                                              const unsigned* __restrict__ counts,
                                                              const unsigned* __restrict__ seg_off,
                                                              const unsigned* __restrict__ seg_total,
                                                              const unsigned* __restrict__ seg_bucket,
                                                              const unsigned* __restrict__ entries,
                                                              XYZZ<F>* __restrict__ seg_sum) {
  const unsigned w = blockIdx.y;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= seg_total[w]) return;
  const size_t sslot = (size_t)w * g.seg_cap + t;
  const unsigned b = seg_bucket[sslot];
  const size_t bslot = ((size_t)w << g.log_nb) + b;
  const unsigned first = (t - seg_off[bslot]) << kSegLog;   // rank of this segment's first entry
  unsigned cnt = counts[bslot] - first;
  if (cnt > kSeg) cnt = kSeg;
  const unsigned* e = entries + (size_t)w * n + offsets[bslot] + first;
  // No software prefetch of the point: holding a second Affine<F> costs 16..64 VGPRs (occupancy for
  // G1; for G2 hipcc parked both copies in scratch and serialised every 16-byte piece behind
  // vmcnt(0)).  Only the 4-byte entry index is fetched one iteration ahead.
  unsigned cur = e[0];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (unsigned j = 0; j < cnt; j++) {
    unsigned nxt = (j + 1 < cnt) ? e[j + 1] : 0u;
    Affine<F> p = bases[cur & 0x7fffffffu];
    acc = acc.madd(p, cur >> 31);
    cur = nxt;
  }
  seg_sum[sslot] = acc;
}

// ---- 4b: bucket = sum of its segment partials -----------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) msm_finalize_kernel(MsmGeom g, const unsigned* __restrict__ counts,
                                                            const unsigned* __restrict__ seg_off,
                                                            const XYZZ<F>* __restrict__ seg_sum,
                                                            XYZZ<F>* __restrict__ buckets,
                                                            unsigned* __restrict__ giant_count,
                                                            unsigned* __restrict__ giant_list, unsigned giant_cap) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)g.nwin << g.log_nb;
  if (gid >= total) return;
  const unsigned w = (unsigned)(gid >> g.log_nb);
  const unsigned nseg = (counts[gid] + kSeg - 1) >> kSegLog;
  if (nseg == 0) { buckets[gid] = XYZZ<F>::inf(); return; }
  const XYZZ<F>* sp = seg_sum + (size_t)w * g.seg_cap + seg_off[gid];
  if (nseg == 1) { buckets[gid] = sp[0]; return; }
  if (nseg > kGiantSegs) {
    unsigned slot = atomicAdd(giant_count, 1u);
    if (slot < giant_cap) { giant_list[slot] = (unsigned)gid; return; }
    // list full (cannot happen: giant_cap >= total segments / kGiantSegs): fall through, serial
  }
  XYZZ<F> acc = sp[0];
  for (unsigned s = 1; s < nseg; s++) acc = acc.add(sp[s]);
  buckets[gid] = acc;
}

template <class F>
__global__ void __launch_bounds__(256) msm_giant_kernel(MsmGeom g, const unsigned* __restrict__ counts,
                                                         const unsigned* __restrict__ seg_off,
                                                         const XYZZ<F>* __restrict__ seg_sum,
                                                         XYZZ<F>* __restrict__ buckets,
                                                         const unsigned* __restrict__ giant_count,
                                                         const unsigned* __restrict__ giant_list, unsigned giant_cap) {
  __shared__ XYZZ<F> sh[256];
  unsigned ng = *giant_count;
  if (ng > giant_cap) ng = giant_cap;
  for (unsigned gi = blockIdx.x; gi < ng; gi += gridDim.x) {
    const unsigned gid = giant_list[gi];
    const unsigned w = gid >> g.log_nb;
    const unsigned nseg = (counts[gid] + kSeg - 1) >> kSegLog;
    const XYZZ<F>* sp = seg_sum + (size_t)w * g.seg_cap + seg_off[gid];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (unsigned s = threadIdx.x; s < nseg; s += 256) acc = acc.add(sp[s]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned stride = 128; stride > 0; stride >>= 1) {
      if (threadIdx.x < stride) sh[threadIdx.x] = sh[threadIdx.x].add(sh[threadIdx.x + stride]);
      __syncthreads();
    }
    if (threadIdx.x == 0) buckets[gid] = sh[0];
    __syncthreads();
  }
}

// small * p, small < 2^32
template <class F>
__device__ XYZZ<F> mul_small(const XYZZ<F>& p, unsigned k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  if (k == 0) return acc;
  int top = 31 - __clz(k);
  for (int i = top; i >= 0; i--) {
    acc = acc.dbl();
    if ((k >> i) & 1) acc = acc.add(p);
  }
  return acc;
}

// ---- 5a: chunks of 2^kChunkLog buckets -> one weighted partial each -----------------------------
constexpr unsigned kChunkLog = 3;
template <class F>
__global__ void __launch_bounds__(256) msm_chunk_kernel(const XYZZ<F>* __restrict__ buckets, MsmGeom g,
                                                         XYZZ<F>* __restrict__ partial) {
  const unsigned log_chunks = g.log_nb > kChunkLog ? g.log_nb - kChunkLog : 0;
  const unsigned L = 1u << (g.log_nb - log_chunks);
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)g.nwin << log_chunks;
  if (gid >= total) return;
  size_t w = gid >> log_chunks;
  unsigned ch = (unsigned)(gid & (((size_t)1 << log_chunks) - 1));
  unsigned lo = ch * L;
  const XYZZ<F>* b = buckets + (w << g.log_nb) + lo;
  XYZZ<F> run = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
  for (int j = (int)L - 1; j >= 0; j--) {
    run = run.add(b[j]);
    acc = acc.add(run);
  }
  // sum_j (lo + j + 1) * B[lo + j] = acc + lo * run
  if (lo) acc = acc.add(mul_small<F>(run, lo));
  partial[gid] = acc;
}

// ---- 5b: per-window sum of the partials (one workgroup per window) --------------------------------
template <class F>
__global__ void __launch_bounds__(256) msm_window_sum_kernel(const XYZZ<F>* __restrict__ partial, unsigned count,
                                                              XYZZ<F>* __restrict__ window_sums) {
  __shared__ XYZZ<F> sh[256];
  const XYZZ<F>* p = partial + (size_t)blockIdx.x * count;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (unsigned i = threadIdx.x; i < count; i += 256) acc = acc.add(p[i]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (unsigned stride = 128; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) sh[threadIdx.x] = sh[threadIdx.x].add(sh[threadIdx.x + stride]);
    __syncthreads();
  }
  if (threadIdx.x == 0) window_sums[blockIdx.x] = sh[0];
}

// ---- 6: Horner tail ---------------------------------------------------------------------------------
template <class F>
__global__ void msm_tail_kernel(const XYZZ<F>* __restrict__ window_sums, MsmGeom g, int affine, F* __restrict__ out) {
  XYZZ<F> total = XYZZ<F>::inf();
  for (int w = (int)g.nwin - 1; w >= 0; w--) {
    for (unsigned k = 0; k < g.c; k++) total = total.dbl();
    total = total.add(window_sums[w]);
  }
  if (affine) {
    Affine<F> a = total.to_affine();
    out[0] = a.x;
    out[1] = a.y;
  } else {
    Jacobian<F> j = total.to_jacobian();
    out[0] = j.x;
    out[1] = j.y;
    out[2] = j.z;
  }
}

template <class F, class Fr, int SCALAR_BITS>
void msm_run(Call& k, const void* bases, const void* scalars, size_t n, bool scalars_mont, bool out_affine,
             void* out_dev) {
  hipStream_t s = k.s();
  MsmGeom g = msm_geometry(n ? n : 1, SCALAR_BITS);
  const size_t nbw = (size_t)g.nwin << g.log_nb;   // buckets over all windows
  const size_t nseg_slots = (size_t)g.nwin * g.seg_cap;
  const unsigned giant_cap = (unsigned)(nseg_slots / kGiantSegs + 1);
  int* digits = (int*)ws(k.c, 4, (size_t)g.nwin * n * 4);
  unsigned* entries = (unsigned*)ws(k.c, 5, (size_t)g.nwin * n * 4);
  // u32 tables: counts | offsets | seg_off | cursor | seg_total[W] | giant_count | giant_list | seg_bucket
  unsigned* tabs = (unsigned*)ws(k.c, 6, (nbw * 4 + g.nwin + 1 + giant_cap + nseg_slots) * 4);
  unsigned* counts = tabs;
  unsigned* offsets = counts + nbw;
  unsigned* seg_off = offsets + nbw;
  unsigned* cursor = seg_off + nbw;
  unsigned* seg_total = cursor + nbw;
  unsigned* giant_count = seg_total + g.nwin;
  unsigned* giant_list = giant_count + 1;
  unsigned* seg_bucket = giant_list + giant_cap;
  XYZZ<F>* buckets = (XYZZ<F>*)ws(k.c, 7, nbw * sizeof(XYZZ<F>));
  XYZZ<F>* seg_sum = (XYZZ<F>*)ws(k.c, 17, nseg_slots * sizeof(XYZZ<F>));
  const unsigned log_chunks = g.log_nb > kChunkLog ? g.log_nb - kChunkLog : 0;
  const size_t nchunks = (size_t)g.nwin << log_chunks;
  XYZZ<F>* partial = (XYZZ<F>*)ws(k.c, 15, (nchunks + g.nwin) * sizeof(XYZZ<F>));
  XYZZ<F>* window_sums = partial + nchunks;

  DG_HIP(hipMemsetAsync(counts, 0, nbw * 4, s));
  DG_HIP(hipMemsetAsync(giant_count, 0, 4, s));
  if (n) {
    hipLaunchKernelGGL(msm_digits_kernel<Fr>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       (const Fr*)scalars, n, (int)scalars_mont, g, digits, counts);
  }
  hipLaunchKernelGGL(msm_scan_kernel<0>, dim3(g.nwin), dim3(1024), 0, s, counts, offsets, seg_off, seg_total,
                     cursor, g.log_nb);
  if (n) {
    hipLaunchKernelGGL(msm_scatter_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, digits, n, g,
                       offsets, seg_off, cursor, entries, seg_bucket);
  }
  k.begin_dominant();
  hipLaunchKernelGGL(msm_accumulate_kernel<F>, dim3((g.seg_cap + 255) / 256, g.nwin), dim3(256), 0, s,
                     (const Affine<F>*)bases, n, g, offsets, counts, seg_off, seg_total, seg_bucket, entries,
                     seg_sum);
  k.end_dominant();
  hipLaunchKernelGGL(msm_finalize_kernel<F>, dim3((unsigned)((nbw + 255) / 256)), dim3(256), 0, s, g, counts,
                     seg_off, seg_sum, buckets, giant_count, giant_list, giant_cap);
  {
    unsigned gblocks = giant_cap < 1024 ? giant_cap : 1024;
    hipLaunchKernelGGL(msm_giant_kernel<F>, dim3(gblocks), dim3(256), 0, s, g, counts, seg_off, seg_sum, buckets,
                       giant_count, giant_list, giant_cap);
  }
  hipLaunchKernelGGL(msm_chunk_kernel<F>, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, buckets, g,
                     partial);
  hipLaunchKernelGGL(msm_window_sum_kernel<F>, dim3(g.nwin), dim3(256), 0, s, partial, 1u << log_chunks,
                     window_sums);
  hipLaunchKernelGGL(msm_tail_kernel<F>, dim3(1), dim3(1), 0, s, window_sums, g, (int)out_affine, (F*)out_dev);
  DG_HIP(hipGetLastError());
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
