// Radix-2 NTT over the scalar field for gfx950, natural order in and out
// (ark-poly Radix2EvaluationDomain semantics as used at ark-circom/src/circom/qap.rs:64-85 and
// dist-primitives/src/dfft/mod.rs:40,78-81).
//
// Decomposition (generalised Cooley-Tukey, no separate bit-reversal pass): N = N1*N2*N3 (1..3
// steps of <= 2^9 points).  With input index n = N2N3*n1 + N3*n2 + n3 and output index
// k = k1 + N1*k2 + N1N2*k3, step j views the array as [A][Nj][B] and, for every (a, b), does a
// length-Nj DFT along the middle axis entirely inside LDS, multiplies by the inter-step twiddle
// w_{Nj*B}^(b*k) and stores in place; the last step (B = 1) stores transposed to natural order.
// A workgroup owns a tile of 2^s points x T lines (T contiguous in memory => T*32-byte coalesced
// segments), TILE = 1024 elements = 32 KiB of LDS, 256 threads, 2 butterflies/thread/level.
//
// Arithmetic: the reduced-radix representation of fp29.h (9 limbs of 29 bits, lazy bounds: a butterfly is one
// 162-mad product + 18 plain additions, no carries or compares).  Data stays in the arkworks Montgomery form
// (x R32) end to end: mont29(x R32, w R29) = x w R32, so only the TWIDDLE tables are held in the internal form.
// The twiddles of the in-LDS butterflies are staged in LDS once per workgroup; two levels are taken per LDS round
// trip (radix-4 steps: 4 products per 4 elements like radix-2, half the traffic and barriers).
//
// Roofline: 64 B/element algorithmic traffic, but ~10 Montgomery multiplications per element -- the kernel is
// VALU-bound, not HBM-bound.
#include <cstdio>
#include <cstdlib>
#include "ctx.h"
#include "fp29.h"
#include "types.h"

namespace dg16 {

// DG16_VERBOSE=1: say so when an OPTIONAL table could not be allocated (the transform then runs its composed-twiddle
// form: correct, a few percent slower) -- otherwise that path is indistinguishable from the fast one
static void optional_table_skipped(const char* what, size_t bytes) {
  static const bool verbose = [] { const char* e = getenv("DG16_VERBOSE"); return e && atoi(e) != 0; }();
  if (verbose) fprintf(stderr, "[dg16] %s (%zu bytes) not built: hipMalloc failed; using the composed form\n", what, bytes);
}


constexpr unsigned kTileLog = 10;         // 1024 elements per workgroup
constexpr unsigned kTile = 1u << kTileLog;
constexpr unsigned kMaxStepLog = 10;      // sub-FFT size limit per step: 2^20 = 2^10 x 2^10 is TWO passes over the data
// (2^21 / 2^22 as two passes on a 2048-element tile -- 72 KB of LDS + 32 KB of packed twiddles, one 512-lane workgroup
// per CU, steps of 2^11 points whose tile lines are single 32-byte elements -- was built and measured in round 5: 2^22
// 0.757 ms against 0.516-0.522 for the three passes below, 2^21 0.326 against 0.274, same call
// (profiles/r5h_ntt_two_pass_2048_tile_ab.txt).  The pass it saves is 11 % of the instructions; one workgroup per CU and
// uncoalesced lines cost more.  Removed.)
constexpr unsigned kLoBits = 11;          // twiddle table split

template <class F>
struct StepArgs {
  const F* src[3];           // up to three independent transforms per launch (blockIdx.y): the a, b, c of a proof
  F* dst[3];
  unsigned log_n, s, log_a, log_b;
  unsigned last;             // 1: B == 1, transposed store to natural order
  unsigned log_n1, log_n2;   // last step: a = k1*N2 + k2, out = k1 + N1*k2 + N1N2*k
  const F* small;            // w_{2^sm}^t, t < 2^(sm-1)
  unsigned sm;
  const F* tw_lo;            // inter-step twiddles: w^e = lo[e & mask] * hi[e >> lb]
  const F* tw_hi;
  unsigned lb;
  const F* tw_full;          // the same twiddles as ONE table in the step's own layout, [k][b] = w^((b k) << log_a) (times
                             // n^-1 where tw_hi would carry it), or null: a 32-byte coalesced read instead of a product
  const F* post_full;        // post_lo / post_hi as one table indexed by the output position, or null
  const F* pre_lo;           // optional: x[i] *= g^i on load (first step)
  const F* pre_hi;
  const F* post_lo;          // optional: out[o] *= g^o on store (last step)
  const F* post_hi;
  unsigned plb;
  const F* scale;            // optional: out *= *scale on store (last step)
  // sharded transforms: the vector is exchanged in 2^piece_log-element pieces that sit piece_stride elements apart in
  // the exchange buffer (piece q of this vector at q * piece_stride); 0 = contiguous.  src_*: first-step loads,
  // dst_*: last-step stores.
  unsigned src_piece_log, dst_piece_log;
  size_t src_piece_stride, dst_piece_stride;
};

__device__ __forceinline__ unsigned bitrev(unsigned v, unsigned bits) {
  return bits ? (__brev(v) >> (32 - bits)) : 0;
}

// Bound (in p / 64) of a tile element between butterfly steps.  The static type of the tile cannot carry a bound that
// grows per loop iteration, so the loops re-label their results as El = Fe<P, kNttBound> (Fe::unsafe_assume).  What makes
// that sound is checked AT COMPILE TIME, on the types of the very expressions the kernel evaluates (NttBounds below and
// the static_asserts next to each butterfly):
//   * every butterfly output is ONE tile input plus products and subtraction constants whose bounds do not depend on
//     the input's value, so a step raises the TRUE bound of an element by at most
//     (static bound of the output) - (static bound of the input) -- kFirstOut for the first radix-4 step on fresh
//     loads, kStepGrowth per later radix-4 step, kOddGrowth for the odd last level;
//   * a pass has one first step, at most (kMaxStepLog - 2) / 2 later radix-4 steps and at most one odd level:
//     kFirstOut + (kMaxStepLog - 2) / 2 * kStepGrowth + kOddGrowth <= kNttBound;
//   * the products and subtraction constants inside a step are sized for inputs up to kNttBound (their static type),
//     which the induction above guarantees.
constexpr int kNttBound = 48 * 64;
constexpr int kNttIn = 128;               // fresh loads: canonical data, or a product with a coset / shift table (< 2 p)
constexpr int kNttFirstOut = 9 * 64;      // first radix-4 step (three trivial twiddles): sums of four fresh loads
constexpr int kNttStepGrowth = 6 * 64 + 32;   // a later radix-4 step: x + t, then x - t + 3 p, twice (t a product < 1.5 p)
constexpr int kNttOddGrowth = 3 * 64 + 32;    // the odd last level: x - t + 3 p
static_assert(kNttFirstOut + (int)((kMaxStepLog - 2) / 2) * kNttStepGrowth + kNttOddGrowth <= kNttBound,
              "NTT tile bound: a pass of kMaxStepLog levels can exceed kNttBound");

template <class F>
__global__ void __launch_bounds__(256) ntt_step_kernel(StepArgs<F> p) {
  using P = typename F::Params;
  using T = RR<P>;
  constexpr int N = T::N;
  using El = Fe<P, kNttBound, 1>;
  using Tw = Fe<P, 64, 1>;
  __shared__ uint32_t tile_w[kTile * N];                               // [element][limb]: stride 9 words, conflict-free
  constexpr int NLW = T::NL;                                           // packed words per element (memory / table form)
  // w_{2^s}^t, t < 2^(s-1): this step's butterfly twiddles.  16 KB next to the 36 KB tile = three workgroups per CU.
  // 2^9 twiddles (s = 10) only fit PACKED (8 words each, unpacked on every read: ~25 VALU next to a 200-instruction
  // product); up to 2^8 (s <= 9) are staged unpacked, 9 limbs each.
  __shared__ uint32_t tw_w[(1u << (kMaxStepLog - 1)) * NLW];
  static_assert((1u << (kMaxStepLog - 2)) * N <= (1u << (kMaxStepLog - 1)) * NLW, "unpacked twiddles of the shorter steps fit");
  const unsigned tid = threadIdx.x;
  const unsigned s = p.s;
  const unsigned nj = 1u << s;
  const unsigned log_t = (p.log_n < kTileLog ? p.log_n : kTileLog) - s;  // lines per tile
  const unsigned Tn = 1u << log_t;
  const unsigned elems = nj << log_t;
  const size_t tile_id = blockIdx.x;
  // element e lives at slot e ^ ((e >> 2) & 31): with the odd word stride N every access pattern of the butterfly
  // loops (32 lanes on 32 elements whose indices skip the two butterfly bits) and of the linear passes is free of
  // bank conflicts (checked exhaustively for s = 4..9, tools/lds_swizzle_check.py)
  auto slot = [](unsigned e) { return (e ^ ((e >> 2) & 31u)) * (unsigned)N; };
  auto ld_tile = [&](unsigned i) {
    El v;
    const unsigned o = slot(i);
#pragma unroll
    for (int j = 0; j < N; j++) v.l[j] = tile_w[o + j];
    return v;
  };
  auto st_tile = [&](unsigned i, const El& v) {
    const unsigned o = slot(i);
#pragma unroll
    for (int j = 0; j < N; j++) tile_w[o + j] = v.l[j];
  };
  const bool tw_packed = p.s == kMaxStepLog;
  auto ld_tw = [&](unsigned i) {
    if (tw_packed) return fe_from_words<P>(&tw_w[i * NLW]);
    Tw v;
#pragma unroll
    for (int j = 0; j < N; j++) v.l[j] = tw_w[i * N + j];
    return v;
  };
  auto ld_packed = [&](const F* ptr) { return fe_from_words<P>(ptr->l); };   // a table entry (internal form) or data

  // tile origin
  size_t a = 0, b0 = 0, k1_0 = 0, k2 = 0;
  if (!p.last) {
    const unsigned tiles_per_a_log = p.log_b - log_t;
    a = tile_id >> tiles_per_a_log;
    b0 = (tile_id & (((size_t)1 << tiles_per_a_log) - 1)) << log_t;
  } else {
    // lines a_t = (k1_0 + t) * N2 + k2
    const unsigned k1_tiles_log = p.log_n1 - log_t;  // requires N1 >= T (host guarantees)
    k2 = tile_id >> k1_tiles_log;
    k1_0 = (tile_id & (((size_t)1 << k1_tiles_log) - 1)) << log_t;
  }

  // ---- stage this step's butterfly twiddles: w_{2^s}^t = small[t << (sm - s)] ----
  if (s >= 1)
    for (unsigned t = tid; t < (nj >> 1); t += 256) {
      const F* w = p.small + ((size_t)t << (p.sm - s));
      if (tw_packed) {
#pragma unroll
        for (int j = 0; j < NLW; j++) tw_w[t * NLW + j] = w->l[j];
      } else {
        const Tw u = ld_packed(w);
#pragma unroll
        for (int j = 0; j < N; j++) tw_w[t * N + j] = u.l[j];
      }
    }

  // ---- load (bit-reversed rows so that in-place DIT yields natural order) ----
  for (unsigned idx = tid; idx < elems; idx += 256) {
    unsigned n, t;
    size_t g;
    if (!p.last) {
      t = idx & (Tn - 1);
      n = idx >> log_t;
      g = ((((a << s) + n) << p.log_b) + b0 + t);
    } else {
      n = idx & (nj - 1);
      t = idx >> s;
      g = ((((k1_0 + t) << p.log_n2) + k2) << s) + n;
    }
    const size_t gs = p.src_piece_stride ? (g >> p.src_piece_log) * p.src_piece_stride + (g & (((size_t)1 << p.src_piece_log) - 1)) : g;
    const Tw x = ld_packed(p.src[blockIdx.y] + gs);
    El v = x.template as<kNttBound, 1>();
    if (p.pre_lo) {
      const auto w = ld_packed(p.pre_lo + (g & ((1u << p.plb) - 1))) * ld_packed(p.pre_hi + (g >> p.plb));
      v = (x * w).template as<kNttBound, 1>();
    }
    st_tile((bitrev(n, s) << log_t) + t, v);
  }
  __syncthreads();

  // ---- s radix-2 DIT levels in LDS, two per round trip (radix-4 steps), one plain level at the end when s is odd ----
  // a radix-4 step on (xa, xb, xc, xd) at distance `half` lines: level lv pairs (a, b) and (c, d) with w_{2 half}^k,
  // level lv + 1 pairs (a, c) with w_{4 half}^k and (b, d) with w_{4 half}^(k + half)
  unsigned lv = 1;
  if (s >= 2) {
    // first step: half = 1, k = 0 -- three of the four twiddles are 1 and the inputs are fresh loads (< 2 p: canonical
    // data, or the (1 + 3/64) p of a product with the coset table), so the sums are formed without products
    using In = Fe<P, kNttIn, 1>;
    const Tw w3 = ld_tw(1u << (s - 2));        // w_4^1
    for (unsigned q = tid; q < (elems >> 2); q += 256) {
      const unsigned t = q & (Tn - 1), pi = q >> log_t;
      const unsigned ia = (pi << (2 + log_t)) + t;
      // (the loads of this step are the fresh values the load phase stored: canonical, or < 2 p after the coset table)
      const In xa = ld_tile(ia).template unsafe_assume<kNttIn, 1>(), xb = ld_tile(ia + Tn).template unsafe_assume<kNttIn, 1>(),
               xc = ld_tile(ia + 2 * Tn).template unsafe_assume<kNttIn, 1>(), xd = ld_tile(ia + 3 * Tn).template unsafe_assume<kNttIn, 1>();
      const auto a1 = xa + xb;
      const auto b1 = xa - xb;
      const auto c1 = xc + xd;
      const auto te = (xc - xd) * w3;
      const auto o0 = a1 + c1;
      const auto o2 = a1 - c1;
      const auto o1 = b1 + te;
      const auto o3 = b1 - te;
      static_assert(decltype(o0)::Bound <= kNttFirstOut && decltype(o1)::Bound <= kNttFirstOut &&
                    decltype(o2)::Bound <= kNttFirstOut && decltype(o3)::Bound <= kNttFirstOut, "first radix-4 step bound");
      st_tile(ia, norm(o0).template as<kNttBound, 1>());
      st_tile(ia + 2 * Tn, norm(o2).template as<kNttBound, 1>());
      st_tile(ia + Tn, norm(o1).template as<kNttBound, 1>());
      st_tile(ia + 3 * Tn, norm(o3).template as<kNttBound, 1>());
    }
    __syncthreads();
    lv = 3;
  }
  for (; lv + 1 <= s; lv += 2) {
    const unsigned half = 1u << (lv - 1);
    const unsigned sh1 = s - lv, sh2 = s - lv - 1;
    for (unsigned q = tid; q < (elems >> 2); q += 256) {
      const unsigned t = q & (Tn - 1), pi = q >> log_t;
      const unsigned k = pi & (half - 1), blk = pi >> (lv - 1);
      const unsigned ia = ((((blk << (lv + 1)) + k)) << log_t) + t;
      const unsigned st = half << log_t;
      const El xa = ld_tile(ia), xb = ld_tile(ia + st), xc = ld_tile(ia + 2 * st), xd = ld_tile(ia + 3 * st);
      const Tw w1 = ld_tw(k << sh1);
      const auto tb = xb * w1;
      const auto td = xd * w1;
      const auto a1 = xa + tb;
      const auto b1 = xa - tb;
      const auto c1 = xc + td;
      const auto d1 = xc - td;
      const auto tc = c1 * ld_tw(k << sh2);
      const auto te = d1 * ld_tw((k + half) << sh2);
      // each output = one tile input + value-independent terms: the true bound grows by at most kNttStepGrowth (see kNttBound)
      const auto o0 = a1 + tc;
      const auto o2 = a1 - tc;
      const auto o1 = b1 + te;
      const auto o3 = b1 - te;
      static_assert(decltype(o0)::Bound - kNttBound <= kNttStepGrowth && decltype(o1)::Bound - kNttBound <= kNttStepGrowth &&
                    decltype(o2)::Bound - kNttBound <= kNttStepGrowth && decltype(o3)::Bound - kNttBound <= kNttStepGrowth,
                    "radix-4 step growth");
      st_tile(ia, norm(o0).template unsafe_assume<kNttBound, 1>());
      st_tile(ia + 2 * st, norm(o2).template unsafe_assume<kNttBound, 1>());
      st_tile(ia + st, norm(o1).template unsafe_assume<kNttBound, 1>());
      st_tile(ia + 3 * st, norm(o3).template unsafe_assume<kNttBound, 1>());
    }
    __syncthreads();
  }
  if (lv == s) {
    // the odd level: half = 2^(s - 1), twiddle w_{2^s}^k = tw[k]
    const unsigned half = 1u << (s - 1);
    for (unsigned q = tid; q < (elems >> 1); q += 256) {
      const unsigned t = q & (Tn - 1), k = q >> log_t;
      const unsigned i0 = (k << log_t) + t, i1 = i0 + (half << log_t);
      const El x = ld_tile(i0);
      const auto ty = ld_tile(i1) * ld_tw(k);
      const auto o0 = x + ty;
      const auto o1 = x - ty;
      static_assert(decltype(o0)::Bound - kNttBound <= kNttOddGrowth && decltype(o1)::Bound - kNttBound <= kNttOddGrowth,
                    "odd level growth");
      st_tile(i0, norm(o0).template unsafe_assume<kNttBound, 1>());
      st_tile(i1, norm(o1).template unsafe_assume<kNttBound, 1>());
    }
    __syncthreads();
  }

  // ---- twiddle + store ----
  for (unsigned idx = tid; idx < elems; idx += 256) {
    unsigned t = idx & (Tn - 1);
    unsigned k = idx >> log_t;
    const El v = ld_tile((k << log_t) + t);
    Fe<P, 64, 1> out;
    size_t g;
    if (!p.last) {
      size_t b = b0 + t;
      size_t e = (b * k) << p.log_a;   // < N
      if (p.tw_full) {
        // (the table holds n^-1 at e == 0 when the step carries the scale, 1 otherwise: skipped then)
        if (e || p.scale) out = canon(v * ld_packed(p.tw_full + (((size_t)k << p.log_b) + b)));
        else out = canon(v);
      } else if (e) {
        const auto w = ld_packed(p.tw_lo + (e & ((1u << p.lb) - 1))) * ld_packed(p.tw_hi + (e >> p.lb));
        out = canon(v * w);
      } else if (p.scale) {
        // inverse transform with the n^-1 factor folded into tw_hi: e == 0 still needs it
        out = canon(v * ld_packed(p.tw_hi));
      } else {
        out = canon(v);
      }
      g = ((((a << s) + k) << p.log_b) + b);
    } else {
      g = (k1_0 + t) + (k2 << p.log_n1) + ((size_t)k << (p.log_n1 + p.log_n2));
      if (p.post_full) {
        const auto w = ld_packed(p.post_full + g);
        if (p.scale) out = canon((v * ld_packed(p.scale)) * w);
        else out = canon(v * w);
      } else if (p.scale && p.post_lo) {
        const auto w = ld_packed(p.post_lo + (g & ((1u << p.plb) - 1))) * ld_packed(p.post_hi + (g >> p.plb));
        out = canon((v * ld_packed(p.scale)) * w);
      } else if (p.scale) {
        out = canon(v * ld_packed(p.scale));
      } else if (p.post_lo) {
        const auto w = ld_packed(p.post_lo + (g & ((1u << p.plb) - 1))) * ld_packed(p.post_hi + (g >> p.plb));
        out = canon(v * w);
      } else {
        out = canon(v);
      }
    }
    F o;
    fe_to_words<P>(out, o.l);
    if (p.dst_piece_stride) g = (g >> p.dst_piece_log) * p.dst_piece_stride + (g & (((size_t)1 << p.dst_piece_log) - 1));
    p.dst[blockIdx.y][g] = o;
  }
}

// in place: arkworks-form table (x R32) -> internal form (x R of fp29.h), both packed and canonical
template <class F>
__global__ void to_internal_kernel(F* t, size_t count) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  using P = typename F::Params;
  F c;
#pragma unroll
  for (int i = 0; i < F::NL; i++) c.l[i] = RR<P>::R_WORDS.v[i];
  t[j] = t[j] * c;      // 32-bit Montgomery product: x R32 * (R mod p) / R32 = x R
}
template <class F>
static void* internal_copy(hipStream_t s, const void* src, size_t count) {
  void* d = nullptr;
  DG_HIP(hipMalloc(&d, count * sizeof(F)));
  DG_HIP(hipMemcpyAsync(d, src, count * sizeof(F), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(to_internal_kernel<F>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (F*)d, count);
  return d;
}

// ---- twiddle tables -------------------------------------------------------------------------
// out[j] = c * base^(j << shift), j < count.  base_log2 != 0: base = TWO_ADIC_ROOT^(2^(S-base_log2))
// (or its inverse) instead of *base_ptr.
template <class F>
__global__ void powers_kernel(F* out, size_t count, unsigned shift, const F* base_ptr,
                              unsigned root_log, int invert, const F* c_ptr) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  F base;
  if (base_ptr) {
    base = *base_ptr;
  } else {
    F w;
#pragma unroll
    for (int i = 0; i < F::NL; i++) w.l[i] = F::Params::TWO_ADIC_ROOT[i];
    for (unsigned i = root_log; i < (unsigned)F::Params::TWO_ADICITY; i++) w = w.sqr();
    // w has order 2^root_log; its inverse is w^(2^root_log - 1)
    if (invert) {
      F acc = F::one(), sq = w;
      for (unsigned i = 0; i < root_log; i++) { acc = acc * sq; sq = sq.sqr(); }
      w = acc;
    }
    base = w;
  }
  F r = base.pow_u64((uint64_t)j << shift);
  if (c_ptr) r = r * *c_ptr;
  out[j] = r;
}

// *out = (2^log_n)^-1
template <class F>
__global__ void n_inv_kernel(F* out, unsigned log_n) {
  // inverse of 2 is (p+1)/2; computed as 2^-1 = (R-form) via Fermat once, then raised
  F two = F::one() + F::one();
  F half = two.inv();
  *out = half.pow_u64(log_n);
}

struct Plan {
  unsigned nsteps;
  unsigned s[3];
};
static Plan make_plan(unsigned log_n) {
  Plan pl{};
  if (log_n <= kTileLog && log_n <= kMaxStepLog) {
    pl.nsteps = 1;
    pl.s[0] = log_n;
  } else if (log_n <= 2 * kMaxStepLog) {
    pl.nsteps = 2;
    pl.s[0] = log_n / 2;
    pl.s[1] = log_n - pl.s[0];
  } else {
    pl.nsteps = 3;
    pl.s[0] = log_n / 3;
    pl.s[1] = (log_n - pl.s[0]) / 2;
    pl.s[2] = log_n - pl.s[0] - pl.s[1];
  }
  return pl;
}

// Inter-step twiddles as ONE table per step, in the step's own data layout: out[(k << log_b) + b] =
// lo[e & mask] * hi[e >> lb], e = (b k) << log_a (internal form in, internal form out: (x R)(y R) / R = x y R; `hi` is
// the n^-1-scaled table where the step carries the scale).  The composed form costs a 227-instruction product per
// element and step in a kernel that is VALU-bound (ntt_step_kernel ran at ~80 % of its instruction-issue bound and
// 7 % of HBM); the table costs 32 bytes of coalesced read.  Sizes: 2^(log_n - log_a) elements -- 32 MB for the first
// step of a 2^20 transform, 512 MB at 2^24.  Not built above kFullTwiddleMaxLog (the composed form serves).
constexpr unsigned kFullTwiddleMaxLog = 24;
// ... and not below kFullTwiddleMinLog: measured (profiles/r4b_ntt_tables_ab.txt) 2^22 alone 0.588 -> 0.536 ms, 2^24
// 2.19 -> 2.00, but the h-polynomial at 2^20 only 0.81 -> 0.78 alone and a 2^20 proof +0.1 ms (the table reads share
// the memory system with the digit sort running underneath).  DG16_NTT_TABLE_MIN_LOG overrides (0 = always, 99 = never).
inline unsigned full_twiddle_min_log() {
  static const unsigned v = [] { const char* e = getenv("DG16_NTT_TABLE_MIN_LOG"); return e ? (unsigned)atoi(e) : 21u; }();
  return v;
}
template <class F>
__global__ void __launch_bounds__(256) full_twiddle_kernel(F* __restrict__ out, size_t count, unsigned log_a, unsigned log_b,
                                                            const F* __restrict__ lo, const F* __restrict__ hi, unsigned lb) {
  using P = typename F::Params;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count) return;
  const size_t k = idx >> log_b, b = idx & (((size_t)1 << log_b) - 1);
  const size_t e = (b * k) << log_a;
  const auto w = fe_from_words<P>(lo[e & (((size_t)1 << lb) - 1)].l) * fe_from_words<P>(hi[e >> lb].l);
  F o;
  fe_to_words<P>(canon(w), o.l);
  out[idx] = o;
}
// out[o] = lo[o & mask] * hi[o >> lb], o < count: a split power table flattened (the w_2m^o shift of the h-polynomial)
template <class F>
__global__ void __launch_bounds__(256) flat_powers_kernel(F* __restrict__ out, size_t count, const F* __restrict__ lo,
                                                           const F* __restrict__ hi, unsigned lb) {
  using P = typename F::Params;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= count) return;
  const auto w = fe_from_words<P>(lo[o & (((size_t)1 << lb) - 1)].l) * fe_from_words<P>(hi[o >> lb].l);
  F r;
  fe_to_words<P>(canon(w), r.l);
  out[o] = r;
}

template <class F>
static const TwiddleSet& get_twiddles(Call& k, int curve, unsigned log_n, int inverse) {
  std::lock_guard<std::mutex> g(k.ctx->mu);
  TwiddleKey key{curve, log_n, inverse};
  auto it = k.ctx->twiddles.find(key);
  if (it != k.ctx->twiddles.end()) return it->second;
  TwiddleSet ts;
  ts.lb = log_n < kLoBits ? log_n : kLoBits;
  ts.sm = log_n < kMaxStepLog ? (log_n ? log_n : 1) : kMaxStepLog;
  size_t nlo = (size_t)1 << ts.lb, nhi = (size_t)1 << (log_n - ts.lb), nsm = (size_t)1 << (ts.sm - 1);
  DG_HIP(hipMalloc(&ts.lo, nlo * sizeof(F)));
  DG_HIP(hipMalloc(&ts.hi, nhi * sizeof(F)));
  DG_HIP(hipMalloc(&ts.small, nsm * sizeof(F)));
  DG_HIP(hipMalloc(&ts.n_inv, sizeof(F)));
  hipStream_t s = k.s();
  hipLaunchKernelGGL(n_inv_kernel<F>, dim3(1), dim3(1), 0, s, (F*)ts.n_inv, log_n);
  hipLaunchKernelGGL(powers_kernel<F>, dim3((unsigned)((nlo + 255) / 256)), dim3(256), 0, s, (F*)ts.lo,
                     nlo, 0u, (const F*)nullptr, log_n, inverse, (const F*)nullptr);
  hipLaunchKernelGGL(powers_kernel<F>, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, s, (F*)ts.hi,
                     nhi, ts.lb, (const F*)nullptr, log_n, inverse, (const F*)nullptr);
  if (inverse) {
    DG_HIP(hipMalloc(&ts.hi_scaled, nhi * sizeof(F)));
    hipLaunchKernelGGL(powers_kernel<F>, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, s,
                       (F*)ts.hi_scaled, nhi, ts.lb, (const F*)nullptr, log_n, inverse,
                       (const F*)ts.n_inv);
  }
  // w_{2^sm}^t = w_N^(t << (log_n - sm)); for log_n == 0 the table is a single 1
  hipLaunchKernelGGL(powers_kernel<F>, dim3((unsigned)((nsm + 255) / 256)), dim3(256), 0, s, (F*)ts.small,
                     nsm, log_n >= ts.sm ? log_n - ts.sm : 0, (const F*)nullptr, log_n ? log_n : 1, inverse,
                     (const F*)nullptr);
  ts.lo_i = internal_copy<F>(s, ts.lo, nlo);
  ts.hi_i = internal_copy<F>(s, ts.hi, nhi);
  if (inverse) ts.hi_scaled_i = internal_copy<F>(s, ts.hi_scaled, nhi);
  ts.small_i = internal_copy<F>(s, ts.small, nsm);
  ts.n_inv_i = internal_copy<F>(s, ts.n_inv, 1);
  {
    const Plan pl = make_plan(log_n);
    unsigned consumed = 0;
    for (unsigned j = 0; j + 1 < pl.nsteps && log_n <= kFullTwiddleMaxLog && log_n >= full_twiddle_min_log(); j++) {
      const size_t cnt = (size_t)1 << (log_n - consumed);
      // an optimisation, not a requirement: when HBM is short (0.5 GB per table at 2^24 next to 126 GB of window
      // tables) the step multiplies by the composed lo x hi twiddles instead
      if (hipMalloc(&ts.full[j], cnt * sizeof(F)) != hipSuccess) {
        (void)hipGetLastError();
        ts.full[j] = nullptr;
        optional_table_skipped("inter-step twiddle table", cnt * sizeof(F));
        consumed += pl.s[j];
        continue;
      }
      hipLaunchKernelGGL(full_twiddle_kernel<F>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, (F*)ts.full[j], cnt,
                         consumed, log_n - consumed - pl.s[j], (const F*)ts.lo_i,
                         (const F*)((inverse && j == 0) ? ts.hi_scaled_i : ts.hi_i), ts.lb);
      consumed += pl.s[j];
    }
  }
  DG_HIP(hipGetLastError());
  DG_HIP(hipStreamSynchronize(s));   // tables are shared by all channels from here on
  return k.ctx->twiddles.emplace(key, ts).first->second;
}

// power tables for an arbitrary base g (coset offset): lo[j] = g^j, hi[j] = g^(j << lb)
template <class F>
static void build_power_tables(Call& k, const F* g_dev, unsigned log_n, F* lo, F* hi, unsigned lb) {
  size_t nlo = (size_t)1 << lb, nhi = (size_t)1 << (log_n - lb);
  hipLaunchKernelGGL(powers_kernel<F>, dim3((unsigned)((nlo + 255) / 256)), dim3(256), 0, k.s(), lo, nlo, 0u,
                     g_dev, 0u, 0, (const F*)nullptr);
  hipLaunchKernelGGL(powers_kernel<F>, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, k.s(), hi, nhi, lb,
                     g_dev, 0u, 0, (const F*)nullptr);
  // the NTT kernels multiply by tables in the internal form (see to_internal_kernel)
  hipLaunchKernelGGL(to_internal_kernel<F>, dim3((unsigned)((nlo + 255) / 256)), dim3(256), 0, k.s(), lo, nlo);
  hipLaunchKernelGGL(to_internal_kernel<F>, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, k.s(), hi, nhi);
  DG_HIP(hipGetLastError());
}

template <class F>
__global__ void inv_one_kernel(F* out, const F* in) { *out = in->inv(); }

// data[i] <- transform(in[i]) for i < nb (in may equal data); tmp[i] are scratch buffers of the same size
// (multi-step plans ping-pong).  The nb transforms share every launch (grid.y = nb): a 2^20 transform alone is
// only 1024 workgroups = 4 per CU.  pre_* / post_*: power tables or null; see StepArgs.
template <class F>
static void ntt_run_batch(Call& k, int curve, unsigned nb, const F* const* in, F* const* data, F* const* tmp,
                          unsigned log_n, int inverse, const F* pre_lo, const F* pre_hi, const F* post_lo,
                          const F* post_hi, unsigned plb, unsigned src_piece_log = 0, size_t src_piece_stride = 0,
                          unsigned dst_piece_log = 0, size_t dst_piece_stride = 0, const F* post_full = nullptr) {
  DG_REQUIRE(log_n <= 3 * kMaxStepLog, DG16_ERR_UNSUPPORTED, "log_n > 27 not supported yet");
  DG_REQUIRE(nb >= 1 && nb <= 3, DG16_ERR_BAD_ARG, "1..3 transforms per batch");
  const TwiddleSet& ts = get_twiddles<F>(k, curve, log_n, inverse);
  Plan pl = make_plan(log_n);
  unsigned consumed = 0;
  for (unsigned j = 0; j < pl.nsteps; j++) {
    StepArgs<F> a{};
    a.log_n = log_n;
    a.s = pl.s[j];
    a.log_a = consumed;
    a.log_b = log_n - consumed - pl.s[j];
    a.last = (j == pl.nsteps - 1);
    a.log_n1 = pl.nsteps >= 2 ? pl.s[0] : 0;
    a.log_n2 = pl.nsteps == 3 ? pl.s[1] : 0;
    a.small = (const F*)ts.small_i;
    a.sm = ts.sm;
    a.tw_lo = (const F*)ts.lo_i;
    // n^-1 rides on the first step's twiddles when there is more than one step
    bool fold_scale = inverse && pl.nsteps > 1 && j == 0;
    a.tw_hi = (const F*)(fold_scale ? ts.hi_scaled_i : ts.hi_i);
    a.lb = ts.lb;
    a.tw_full = (!a.last && j < 2) ? (const F*)ts.full[j] : nullptr;
    a.scale = nullptr;
    if (fold_scale) a.scale = (const F*)ts.n_inv_i;                  // marks "tw_hi carries n^-1"
    if (inverse && pl.nsteps == 1) a.scale = (const F*)ts.n_inv_i;   // explicit multiply at the store
    if (j == 0) { a.pre_lo = pre_lo; a.pre_hi = pre_hi; a.src_piece_log = src_piece_log; a.src_piece_stride = src_piece_stride; }
    if (a.last) { a.post_lo = post_lo; a.post_hi = post_hi; a.post_full = post_full; a.dst_piece_log = dst_piece_log; a.dst_piece_stride = dst_piece_stride; }
    a.plb = plb;
    // ping-pong: first step in -> tmp (same layout), middle step in place on tmp, last step tmp -> data;
    // a single step goes in -> data (one workgroup holds the whole vector in LDS before storing)
    for (unsigned i = 0; i < nb; i++) {
      a.src[i] = j == 0 ? in[i] : tmp[i];
      a.dst[i] = (pl.nsteps == 1 || a.last) ? data[i] : tmp[i];
    }
    unsigned log_tile = log_n < kTileLog ? log_n : kTileLog;
    size_t blocks = (size_t)1 << (log_n - log_tile);
    if (a.last && pl.nsteps > 1) {
      unsigned log_t = log_tile - a.s;
      DG_REQUIRE(a.log_n1 >= log_t, DG16_ERR_UNSUPPORTED, "plan violates N1 >= T");
    } else if (!a.last) {
      unsigned log_t = log_tile - a.s;
      DG_REQUIRE(a.log_b >= log_t, DG16_ERR_UNSUPPORTED, "plan violates B >= T");
    }
    hipLaunchKernelGGL(ntt_step_kernel<F>, dim3((unsigned)blocks, nb), dim3(256), 0, k.s(), a);
    DG_HIP(hipGetLastError());
    consumed += pl.s[j];
  }
}
template <class F>
static void ntt_run(Call& k, int curve, F* data, F* tmp, unsigned log_n, int inverse,
                    const F* pre_lo, const F* pre_hi, const F* post_lo, const F* post_hi, unsigned plb) {
  const F* in[1] = {data};
  F* d[1] = {data};
  F* t[1] = {tmp};
  ntt_run_batch<F>(k, curve, 1, in, d, t, log_n, inverse, pre_lo, pre_hi, post_lo, post_hi, plb);
}

template <class F>
static void ntt_typed(Call& k, int curve, void* data, unsigned log_n, int inverse, const void* coset_host) {
  F* d = (F*)data;
  F* tmp = (F*)ws(k.c, 8, sizeof(F) << log_n);
  const F *pre_lo = nullptr, *pre_hi = nullptr, *post_lo = nullptr, *post_hi = nullptr;
  unsigned plb = log_n < kLoBits ? log_n : kLoBits;
  if (coset_host) {
    F* g = (F*)ws(k.c, 9, 2 * sizeof(F));
    DG_HIP(hipMemcpyAsync(g, coset_host, sizeof(F), hipMemcpyHostToDevice, k.s()));
    F* lo = (F*)ws(k.c, 10, sizeof(F) << plb);
    F* hi = (F*)ws(k.c, 11, sizeof(F) << (log_n - plb));
    if (inverse) {
      hipLaunchKernelGGL(inv_one_kernel<F>, dim3(1), dim3(1), 0, k.s(), g + 1, g);
      build_power_tables<F>(k, g + 1, log_n, lo, hi, plb);
      post_lo = lo; post_hi = hi;
    } else {
      build_power_tables<F>(k, g, log_n, lo, hi, plb);
      pre_lo = lo; pre_hi = hi;
    }
  }
  k.begin_dominant();
  ntt_run<F>(k, curve, d, tmp, log_n, inverse, pre_lo, pre_hi, post_lo, post_hi, plb);
  k.end_dominant();
}

const TwiddleSet& get_twiddles_any(Call& k, int curve, unsigned log_n, int inverse) {
  switch (curve) {
    case 0: return get_twiddles<bn254_fr>(k, curve, log_n, inverse);
    case 1: return get_twiddles<bls12_381_fr>(k, curve, log_n, inverse);
    default: return get_twiddles<bls12_377_fr>(k, curve, log_n, inverse);
  }
}

void ntt_launch(Call& k, int curve, void* data, unsigned log_n, int inverse, const void* coset_host) {
  switch (curve) {
    case 0: ntt_typed<bn254_fr>(k, curve, data, log_n, inverse, coset_host); break;
    case 1: ntt_typed<bls12_381_fr>(k, curve, data, log_n, inverse, coset_host); break;
    default: ntt_typed<bls12_377_fr>(k, curve, data, log_n, inverse, coset_host); break;
  }
}

// ---- h polynomial (ark-circom/src/circom/qap.rs:64-91) ------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) mul_sub_kernel(const F* __restrict__ a, const F* __restrict__ b,
                                                       const F* __restrict__ c, F* __restrict__ out, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = a[i] * b[i] - c[i];
}

// w_2m^o for o < m as ONE table (internal form), cached in the twiddle set of the 2m domain
template <class F>
static const F* flat_shift_table(Call& k, const TwiddleSet& t2, unsigned log_m) {
  std::lock_guard<std::mutex> g(k.ctx->mu);
  TwiddleSet& w = const_cast<TwiddleSet&>(t2);       // (the cache entry; guarded by ctx->mu like its creation)
  if (!w.shift_full && !w.shift_full_tried) {
    w.shift_full_tried = true;      // ONE attempt per cache entry: a failing hipMalloc of up to 0.5 GB under ctx->mu on every
                                    // h-polynomial call is a silent slow path
    const size_t m = (size_t)1 << log_m;
    if (hipMalloc(&w.shift_full, m * sizeof(F)) != hipSuccess) {      // optional table: the split lo x hi form serves
      (void)hipGetLastError();
      w.shift_full = nullptr;
      optional_table_skipped("w_2m shift table", m * sizeof(F));
      return nullptr;
    }
    hipLaunchKernelGGL(flat_powers_kernel<F>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, k.s(), (F*)w.shift_full, m,
                       (const F*)t2.lo_i, (const F*)t2.hi_i, t2.lb);
    DG_HIP(hipGetLastError());
    DG_HIP(hipStreamSynchronize(k.s()));               // shared by all channels from here on
  }
  return (const F*)w.shift_full;
}

template <class F>
static void h_poly_typed(Call& k, int curve, const void* a, const void* b, const void* c, unsigned log_m,
                         void* out) {
  size_t bytes = sizeof(F) << log_m;
  F* v[3] = {(F*)ws(k.c, 12, bytes), (F*)ws(k.c, 13, bytes), (F*)ws(k.c, 14, bytes)};
  F* t0 = (F*)ws(k.c, 8, 3 * bytes);
  F* tmp[3] = {t0, t0 + ((size_t)1 << log_m), t0 + ((size_t)2 << log_m)};
  const F* in[3] = {(const F*)a, (const F*)b, (const F*)c};
  // shift tables: powers of w_{2m} (the forward 2m-domain root), applied on the iNTT's store
  const TwiddleSet& t2 = get_twiddles<F>(k, curve, log_m + 1, 0);
  // lo/hi of the 2m domain cover exponents < 2m; we only need o < m -- flattened once into one table (w_2m^o, o < m)
  const F* shift = log_m <= kFullTwiddleMaxLog && log_m >= full_twiddle_min_log() ? flat_shift_table<F>(k, t2, log_m) : nullptr;
  k.begin_dominant();
  // a, b, c go through every step together; the first iNTT step reads the caller's vectors in place
  ntt_run_batch<F>(k, curve, 3, in, v, tmp, log_m, 1, nullptr, nullptr, (const F*)t2.lo_i, (const F*)t2.hi_i, t2.lb,
                   0, 0, 0, 0, shift);
  ntt_run_batch<F>(k, curve, 3, v, v, tmp, log_m, 0, nullptr, nullptr, nullptr, nullptr, 0);
  size_t n = (size_t)1 << log_m;
  size_t blocks = (n + 255) / 256;
  size_t cap = (size_t)k.ctx->compute_units * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(mul_sub_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, k.s(), v[0], v[1], v[2], (F*)out, n);
  DG_HIP(hipGetLastError());
  k.end_dominant();
}

// ---- sharded h polynomial: one process per GPU, two all-to-alls -----------------------------
// The m-point transforms of witness_map (qap.rs:64-91) split as m = N x M over N ranks.  Rank rho owns the cyclic
// rows a[N j + rho] of the evaluation vectors and ends with h[rho + N j]; the N-point cross-rank parts of the inverse
// and the forward transform meet in ONE kernel between the two exchanges (hdist_cross_kernel), the M-point parts are
// the ordinary NTT above.  The reference's counterpart is d_ifft / d_fft (dist-primitives/src/dfft/mod.rs:17-95:
// local levels, gather to the king, remaining levels, scatter); oracle/pyref/hdist.py restates this variant on
// integers.  Exchange buffers are peer-major: [peer][vector a, b, c][S] with S = M / N, one message per peer.

// w^e from a split power table (arkworks form)
template <class F>
__device__ __forceinline__ F tw_lookup(const F* lo, const F* hi, unsigned lb, size_t e) {
  const size_t l = e & (((size_t)1 << lb) - 1), h = e >> lb;
  if (l == 0) return hi[h];
  if (h == 0) return lo[l];
  return lo[l] * hi[h];
}

// x[q] <- sum_k x[k] root^(k q), N = 2^LOGN values in registers; roots[t] = root^t, t < N / 2
template <int LOGN, class F>
__device__ __forceinline__ void small_dft(F* x, const F* roots) {
  constexpr int N = 1 << LOGN;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int j = 0;
#pragma unroll
    for (int b = 0; b < LOGN; b++) j |= ((i >> b) & 1) << (LOGN - 1 - b);
    if (j > i) { const F t = x[i]; x[i] = x[j]; x[j] = t; }
  }
#pragma unroll
  for (int lv = 1; lv <= LOGN; lv++) {
    constexpr int dummy = 0; (void)dummy;
    const int half = 1 << (lv - 1);
#pragma unroll
    for (int q = 0; q < N / 2; q++) {
      const int kk = q & (half - 1), blk = q >> (lv - 1);
      const int i0 = (blk << lv) + kk, i1 = i0 + half;
      const F y = kk ? x[i1] * roots[kk << (LOGN - lv)] : x[i1];
      const F u = x[i0];
      x[i0] = u + y;
      x[i1] = u - y;
    }
  }
}

struct CrossArgs {
  const void *wi_lo, *wi_hi, *wf_lo, *wf_hi, *g_lo, *g_hi, *n_inv;   // w_m^-e, w_m^e, w_2m^e tables; 1 / N
  unsigned wlb, glb;
  unsigned log_m, rank;
  size_t S;                 // elements per (peer, vector) piece
};

// in / out: [peer][vector][S].  One lane per (vector, j): the N values Z[.][v][j] of coefficient column k2 = rank S + j.
template <class F, int LOGN>
__global__ void __launch_bounds__(256) hdist_cross_kernel(const F* __restrict__ in, F* __restrict__ out, CrossArgs p) {
  constexpr int N = 1 << LOGN;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.S) return;
  const unsigned v = blockIdx.y;
  const size_t M = p.S << LOGN;
  const size_t k2 = (size_t)p.rank * p.S + j;
  const F *wi_lo = (const F*)p.wi_lo, *wi_hi = (const F*)p.wi_hi, *wf_lo = (const F*)p.wf_lo, *wf_hi = (const F*)p.wf_hi;
  const F *g_lo = (const F*)p.g_lo, *g_hi = (const F*)p.g_hi;
  F x[N];
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = in[((size_t)i * 3 + v) * p.S + j];
  // t[i1] = Z[i1] w^(-i1 k2)
  {
    const F base = tw_lookup(wi_lo, wi_hi, p.wlb, k2);
    F pw = base;
#pragma unroll
    for (int i = 1; i < N; i++) {
      x[i] = x[i] * pw;
      if (i + 1 < N) pw = pw * base;
    }
  }
  F roots[N / 2 ? N / 2 : 1];
#pragma unroll
  for (int t = 0; t < N / 2; t++) roots[t] = tw_lookup(wi_lo, wi_hi, p.wlb, M * t);   // w_N^-t
  small_dft<LOGN>(x, roots);
  // coefficient M k1 + k2: scale by 1 / N and shift by g^(M k1 + k2)
  {
    const F gm = tw_lookup(g_lo, g_hi, p.glb, M);
    F pw = tw_lookup(g_lo, g_hi, p.glb, k2) * *(const F*)p.n_inv;
#pragma unroll
    for (int i = 0; i < N; i++) {
      x[i] = x[i] * pw;
      if (i + 1 < N) pw = pw * gm;
    }
  }
#pragma unroll
  for (int t = 0; t < N / 2; t++) roots[t] = tw_lookup(wf_lo, wf_hi, p.wlb, M * t);   // w_N^t
  small_dft<LOGN>(x, roots);
  // U[q] = u[q] w^(q k2)
  {
    const F base = tw_lookup(wf_lo, wf_hi, p.wlb, k2);
    F pw = base;
#pragma unroll
    for (int i = 1; i < N; i++) {
      x[i] = x[i] * pw;
      if (i + 1 < N) pw = pw * base;
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) out[((size_t)i * 3 + v) * p.S + j] = x[i];
}

// ---- one transform over N ranks with ONE all-to-all (dg16_ntt_dist) ------------------------------------------------
// X[M k1 + k2] = sum_i1 w_N^(i1 k1) w^(i1 k2) Y_i1[k2],  Y_i1 = the M-point transform of rank i1's cyclic elements
// x[N j + i1].  After the exchange rank sigma holds Z[i1][j] = Y_i1[sigma S + j]; this kernel applies the twiddle and the
// N-point transform over i1 (in registers) and writes out[k1 S + j] = X[M k1 + sigma S + j] (times 1 / N if inverse --
// the local inverse transform has already scaled by 1 / M).
template <class F, int LOGN>
__global__ void __launch_bounds__(256) ntt_dist_cross_kernel(const F* __restrict__ in, F* __restrict__ out, const F* w_lo,
                                                              const F* w_hi, unsigned wlb, const F* n_inv, unsigned rank,
                                                              size_t S) {
  constexpr int N = 1 << LOGN;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const size_t M = S << LOGN, k2 = (size_t)rank * S + j;
  F x[N];
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = in[(size_t)i * S + j];
  {
    const F base = tw_lookup(w_lo, w_hi, wlb, k2);
    F pw = base;
#pragma unroll
    for (int i = 1; i < N; i++) {
      x[i] = x[i] * pw;
      if (i + 1 < N) pw = pw * base;
    }
  }
  F roots[N / 2 ? N / 2 : 1];
#pragma unroll
  for (int t = 0; t < N / 2; t++) roots[t] = tw_lookup(w_lo, w_hi, wlb, M * t);
  small_dft<LOGN>(x, roots);
#pragma unroll
  for (int i = 0; i < N; i++) out[(size_t)i * S + j] = n_inv ? x[i] * *n_inv : x[i];
}

static unsigned log2_exact(unsigned n) {
  unsigned l = 0;
  while ((1u << l) < n) l++;
  DG_REQUIRE((1u << l) == n, DG16_ERR_BAD_ARG, "rank count must be a power of two");
  return l;
}

// stage 0: rows (3 vectors of M) -> iNTT_M -> send buffer [peer][vector][S]
// stage 1: receive buffer -> cross kernel -> send buffer
// stage 2: receive buffer [peer][vector][S] (= W_v[k2], k2 = peer S + j) -> NTT_M -> a b - c -> out (M elements)
template <class F>
static void h_poly_dist_stage_typed(Call& k, int curve, unsigned log_m, unsigned rank, unsigned n_ranks, int stage,
                                    const void* const* in, void* out) {
  const unsigned log_n = log2_exact(n_ranks);
  DG_REQUIRE(log_n >= 1 && log_n <= 3, DG16_ERR_UNSUPPORTED, "sharded h-polynomial: 2, 4 or 8 ranks");
  DG_REQUIRE(log_m >= 2 * log_n, DG16_ERR_BAD_ARG, "sharded h-polynomial: domain smaller than ranks^2");
  DG_REQUIRE(rank < n_ranks, DG16_ERR_BAD_ARG, "rank out of range");
  const unsigned log_M = log_m - log_n, log_S = log_M - log_n;
  const size_t M = (size_t)1 << log_M, S = (size_t)1 << log_S;
  F* t0 = (F*)ws(k.c, 8, 3 * M * sizeof(F));
  F* tmp[3] = {t0, t0 + M, t0 + 2 * M};
  if (stage == 0) {
    const F* src[3] = {(const F*)in[0], (const F*)in[1], (const F*)in[2]};
    F* dst[3] = {(F*)out, (F*)out + S, (F*)out + 2 * S};
    ntt_run_batch<F>(k, curve, 3, src, dst, tmp, log_M, 1, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, log_S, 3 * S);
  } else if (stage == 1) {
    const TwiddleSet& wi = get_twiddles<F>(k, curve, log_m, 1);
    const TwiddleSet& wf = get_twiddles<F>(k, curve, log_m, 0);
    const TwiddleSet& g = get_twiddles<F>(k, curve, log_m + 1, 0);
    const TwiddleSet& nn = get_twiddles<F>(k, curve, log_n, 1);
    CrossArgs a{wi.lo, wi.hi, wf.lo, wf.hi, g.lo, g.hi, nn.n_inv, wi.lb, g.lb, log_m, rank, S};
    dim3 grid((unsigned)((S + 255) / 256), 3);
    switch (log_n) {
      case 1: hipLaunchKernelGGL((hdist_cross_kernel<F, 1>), grid, dim3(256), 0, k.s(), (const F*)in[0], (F*)out, a); break;
      case 2: hipLaunchKernelGGL((hdist_cross_kernel<F, 2>), grid, dim3(256), 0, k.s(), (const F*)in[0], (F*)out, a); break;
      default: hipLaunchKernelGGL((hdist_cross_kernel<F, 3>), grid, dim3(256), 0, k.s(), (const F*)in[0], (F*)out, a); break;
    }
    DG_HIP(hipGetLastError());
  } else {
    const F* base = (const F*)in[0];
    const F* src[3] = {base, base + S, base + 2 * S};
    F* v0 = (F*)ws(k.c, 12, 3 * M * sizeof(F));
    F* v[3] = {v0, v0 + M, v0 + 2 * M};
    ntt_run_batch<F>(k, curve, 3, src, v, tmp, log_M, 0, nullptr, nullptr, nullptr, nullptr, 0, log_S, 3 * S, 0, 0);
    size_t blocks = (M + 255) / 256;
    size_t cap = (size_t)k.ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(mul_sub_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, k.s(), v[0], v[1], v[2], (F*)out, M);
    DG_HIP(hipGetLastError());
  }
}

// stage 0: in (M cyclic elements) -> M-point (i)NTT -> out (natural order = peer-major pieces of S)
// stage 1: in (received [peer][S]) -> cross kernel -> out[k1 S + j] = X[M k1 + rank S + j]
template <class F>
static void ntt_dist_stage_typed(Call& k, int curve, unsigned log_n_total, unsigned rank, unsigned n_ranks, int inverse,
                                 int stage, const void* in, void* out) {
  const unsigned log_n = log2_exact(n_ranks);
  DG_REQUIRE(log_n >= 1 && log_n <= 3, DG16_ERR_UNSUPPORTED, "sharded NTT: 2, 4 or 8 ranks");
  DG_REQUIRE(log_n_total >= 2 * log_n, DG16_ERR_BAD_ARG, "sharded NTT: domain smaller than ranks^2");
  DG_REQUIRE(rank < n_ranks, DG16_ERR_BAD_ARG, "rank out of range");
  const unsigned log_M = log_n_total - log_n;
  const size_t M = (size_t)1 << log_M, S = M >> log_n;
  if (stage == 0) {
    F* tmp = (F*)ws(k.c, 8, M * sizeof(F));
    const F* src[1] = {(const F*)in};
    F* dst[1] = {(F*)out};
    F* t[1] = {tmp};
    ntt_run_batch<F>(k, curve, 1, src, dst, t, log_M, inverse, nullptr, nullptr, nullptr, nullptr, 0);
  } else {
    const TwiddleSet& w = get_twiddles<F>(k, curve, log_n_total, inverse);
    const F* n_inv = inverse ? (const F*)get_twiddles<F>(k, curve, log_n, 1).n_inv : nullptr;
    dim3 grid((unsigned)((S + 255) / 256));
#define DG_NDX(L)                                                                                                   \
  hipLaunchKernelGGL((ntt_dist_cross_kernel<F, L>), grid, dim3(256), 0, k.s(), (const F*)in, (F*)out, (const F*)w.lo, \
                     (const F*)w.hi, w.lb, n_inv, rank, S)
    switch (log_n) {
      case 1: DG_NDX(1); break;
      case 2: DG_NDX(2); break;
      default: DG_NDX(3); break;
    }
#undef DG_NDX
    DG_HIP(hipGetLastError());
  }
}
void ntt_dist_stage(Call& k, int curve, unsigned log_n_total, unsigned rank, unsigned n_ranks, int inverse, int stage,
                    const void* in, void* out) {
  switch (curve) {
    case 0: ntt_dist_stage_typed<bn254_fr>(k, curve, log_n_total, rank, n_ranks, inverse, stage, in, out); break;
    case 1: ntt_dist_stage_typed<bls12_381_fr>(k, curve, log_n_total, rank, n_ranks, inverse, stage, in, out); break;
    default: ntt_dist_stage_typed<bls12_377_fr>(k, curve, log_n_total, rank, n_ranks, inverse, stage, in, out); break;
  }
}
void ntt_dist_launch(Call& k, int curve, const dg16_comm* comm, const void* in, void* out, unsigned log_n_total,
                     int inverse) {
  const unsigned n = comm->n_ranks(comm->self), rank = comm->rank(comm->self);
  const size_t bytes = (((size_t)1 << log_n_total) / n) * 32;
  void* buf_a = ws(k.c, 26, bytes);
  void* buf_b = ws(k.c, 27, bytes);
  ntt_dist_stage(k, curve, log_n_total, rank, n, inverse, 0, in, buf_a);
  int rc = comm->all_to_all(comm->self, buf_a, buf_b, bytes / n, k.s());
  DG_REQUIRE(rc == DG16_OK, DG16_ERR_NET, "all-to-all of the sharded NTT failed");
  ntt_dist_stage(k, curve, log_n_total, rank, n, inverse, 1, buf_b, out);
}

void h_poly_dist_stage(Call& k, int curve, unsigned log_m, unsigned rank, unsigned n_ranks, int stage,
                       const void* const* in, void* out) {
  switch (curve) {
    case 0: h_poly_dist_stage_typed<bn254_fr>(k, curve, log_m, rank, n_ranks, stage, in, out); break;
    case 1: h_poly_dist_stage_typed<bls12_381_fr>(k, curve, log_m, rank, n_ranks, stage, in, out); break;
    default: h_poly_dist_stage_typed<bls12_377_fr>(k, curve, log_m, rank, n_ranks, stage, in, out); break;
  }
}

// stage 0 -> all-to-all -> stage 1 -> all-to-all -> stage 2, stream-ordered on the Call's stream.
// a, b, c: this rank's cyclic rows (M = 2^log_m / N elements each); out: h[rank + N j], j < M.
void h_poly_dist_launch(Call& k, int curve, const dg16_comm* comm, const void* a, const void* b, const void* c,
                        unsigned log_m, void* out) {
  const unsigned n = comm->n_ranks(comm->self), rank = comm->rank(comm->self);
  const size_t M = ((size_t)1 << log_m) / n, bytes = 3 * M * 32;
  void* buf_a = ws(k.c, 26, bytes);
  void* buf_b = ws(k.c, 27, bytes);
  const void* rows[3] = {a, b, c};
  h_poly_dist_stage(k, curve, log_m, rank, n, 0, rows, buf_a);
  int rc = comm->all_to_all(comm->self, buf_a, buf_b, bytes / n, k.s());
  DG_REQUIRE(rc == DG16_OK, DG16_ERR_NET, "all-to-all of the sharded h-polynomial failed");
  const void* in1[1] = {buf_b};
  h_poly_dist_stage(k, curve, log_m, rank, n, 1, in1, buf_a);
  rc = comm->all_to_all(comm->self, buf_a, buf_b, bytes / n, k.s());
  DG_REQUIRE(rc == DG16_OK, DG16_ERR_NET, "all-to-all of the sharded h-polynomial failed");
  h_poly_dist_stage(k, curve, log_m, rank, n, 2, in1, out);
}

void h_poly_launch(Call& k, int curve, const void* a, const void* b, const void* c, unsigned log_m, void* out) {
  switch (curve) {
    case 0: h_poly_typed<bn254_fr>(k, curve, a, b, c, log_m, out); break;
    case 1: h_poly_typed<bls12_381_fr>(k, curve, a, b, c, log_m, out); break;
    default: h_poly_typed<bls12_377_fr>(k, curve, a, b, c, log_m, out); break;
  }
}

}  // namespace dg16
