// dist-primitives on the GPU, literally: packed secret sharing (secret-sharing/src/pss.rs:13-148),
// d_fft / d_ifft (dist-primitives/src/dfft/mod.rs:17-271), d_msm (dmsm/mod.rs:7-98), d_pp
// (dpp/mod.rs:17-88), deg_red (utils/deg_red.rs:10-28), pack_vec / transpose (utils/pack.rs:4-33)
// and ext_wit::h (groth16/src/ext_wit.rs:16-101), over an MpcNet-shaped transport
// (mpc-net/src/lib.rs:37-156) whose payloads stay in HBM.
//
// "Party" = a host thread (LocalNet, the analogue of LocalTestNet, mpc-net/src/multi.rs:227-329) or
// a process with its own GPU (callbacks backed by RCCL).  The king-side steps the reference does
// with one Vec allocation + two tiny FFTs per packed element (dfft/mod.rs:210-219,233-243) are
// constant n x l matrices here (pack = FFT_share o iFFT_secret etc.), applied by one batched kernel
// with the transposes folded into the indexing -- HBM-bound element-wise passes.
#pragma once
#include <condition_variable>

#include "ctx.h"
#include "types.h"

struct dg16_pss {
  dg16_ctx* ctx;
  int curve;
  unsigned l, t, n;
  void* mats;   // device: pack [n][l] | unpack [l][n] | unpack2 [l][n] | (canonical copies of the three) | v2sum [n] canonical
};

namespace dg16 {

constexpr unsigned kMaxParties = 32;

// ---- PSS matrices ------------------------------------------------------------------------------
template <class Fr>
__device__ Fr root_of_unity_dev(unsigned log_n) {
  Fr w;
#pragma unroll
  for (int i = 0; i < Fr::NL; i++) w.l[i] = Fr::Params::TWO_ADIC_ROOT[i];
  for (unsigned i = log_n; i < (unsigned)Fr::Params::TWO_ADICITY; i++) w = w.sqr();
  return w;
}
template <class Fr>
__device__ Fr generator_dev() {
  Fr g;
#pragma unroll
  for (int i = 0; i < Fr::NL; i++) g.l[i] = Fr::Params::GEN[i];
  return g;
}

// one thread per matrix entry; layout documented at dg16_pss::mats
template <class Fr>
__global__ void pss_setup_kernel(unsigned l, Fr* mats) {
  const unsigned n = 4 * l, s = 2 * l;
  unsigned log_n = 0, log_s = 0;
  while ((1u << log_n) < n) log_n++;
  while ((1u << log_s) < s) log_s++;
  const Fr wn = root_of_unity_dev<Fr>(log_n), ws_ = root_of_unity_dev<Fr>(log_s), g = generator_dev<Fr>();
  const Fr wn_inv = wn.inv(), ws_inv = ws_.inv(), g_inv = g.inv();
  const Fr n_inv = Fr::from_u32(n).inv(), s_inv = Fr::from_u32(s).inv();
  Fr* pack = mats;               // [n][l]
  Fr* unpack = pack + n * l;     // [l][n]
  Fr* unpack2 = unpack + l * n;  // [l][n]
  Fr* canon = unpack2 + l * n;   // canonical copies, same order
  Fr* v2sum = canon + 3 * n * l; // [n] canonical: sum_i unpack2[i][j]
  const unsigned tid = threadIdx.x;
  if (tid < n * l) {
    {  // pack[j][i] = sum_{k<s} wn^(jk) * s^-1 * g^-k * ws^(-ik)      (pss.rs:86-92)
      unsigned j = tid / l, i = tid % l;
      Fr acc = Fr::zero();
      for (unsigned k = 0; k < s; k++)
        acc = acc + wn.pow_u64((uint64_t)j * k) * g_inv.pow_u64(k) * ws_inv.pow_u64((uint64_t)i * k);
      pack[tid] = acc * s_inv;
    }
    {  // unpack[i][j] = sum_{k<s} (g ws^i)^k * n^-1 * wn^(-jk)          (pss.rs:110-127)
      unsigned i = tid / n, j = tid % n;
      Fr x = g * ws_.pow_u64(i);
      Fr acc = Fr::zero();
      for (unsigned k = 0; k < s; k++) acc = acc + x.pow_u64(k) * wn_inv.pow_u64((uint64_t)j * k);
      unpack[tid] = acc * n_inv;
      // unpack2[i][j] = sum_{k<n} (g wn^(2i))^k * n^-1 * wn^(-jk)       (pss.rs:131-148)
      Fr x2 = g * wn.pow_u64(2 * i);
      acc = Fr::zero();
      for (unsigned k = 0; k < n; k++) acc = acc + x2.pow_u64(k) * wn_inv.pow_u64((uint64_t)j * k);
      unpack2[tid] = acc * n_inv;
    }
  }
  __syncthreads();
  if (tid < 3 * n * l) canon[tid] = mats[tid].from_mont();
  if (tid < n) {
    Fr acc = Fr::zero();
    for (unsigned i = 0; i < l; i++) acc = acc + unpack2[i * n + tid];
    v2sum[tid] = acc.from_mont();
  }
}

// out[e*ose + r*osr] = sum_c M[r*cols + c] * in[e*ise + c*isc]   (optionally through a gather map)
template <class Fr>
__global__ void __launch_bounds__(256) matvec_kernel(const Fr* __restrict__ M, unsigned rows, unsigned cols,
                                                      const Fr* __restrict__ in, size_t ise, size_t isc,
                                                      Fr* __restrict__ out, size_t ose, size_t osr, size_t count,
                                                      int bitrev_bits /* -1: none; else in index = bitrev(idx) */) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  Fr v[kMaxParties];
  for (unsigned c = 0; c < cols; c++) {
    size_t idx = e * ise + c * isc;
    if (bitrev_bits >= 0) idx = bitrev_bits ? (size_t)(__brevll(idx) >> (64 - bitrev_bits)) : 0;
    v[c] = in[idx];
  }
  for (unsigned r = 0; r < rows; r++) {
    Fr acc = Fr::zero();
    for (unsigned c = 0; c < cols; c++) acc = acc + M[r * cols + c] * v[c];
    out[e * ose + r * osr] = acc;
  }
}

// ---- local butterflies: fft1_in_place, dfft/mod.rs:98-140 (one launch per level i) -----------------
// pair p = j*2*ps + k, (x, y) = (px[p], px[p+ps] * gen^(2^(i-1) * (k+1)))
template <class Fr>
__global__ void __launch_bounds__(256) fft1_level_kernel(Fr* __restrict__ px, size_t npairs, unsigned log_ps,
                                                          unsigned i_level, unsigned log_m, const Fr* __restrict__ lo,
                                                          const Fr* __restrict__ hi, unsigned lb) {
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= npairs) return;
  const size_t ps = (size_t)1 << log_ps;
  size_t k = q & (ps - 1), j = q >> log_ps;
  size_t p = (j << (log_ps + 1)) + k;
  size_t e = ((k + 1) << (i_level - 1)) & (((size_t)1 << log_m) - 1);
  Fr f = lo[e & ((1u << lb) - 1)] * hi[e >> lb];
  Fr x = px[p], y = px[p + ps] * f;
  px[p] = x + y;
  px[p + ps] = x - y;
}

// ---- king: one level of fft2_in_place, dfft/mod.rs:161-176 -------------------------------------------
template <class Fr>
__global__ void __launch_bounds__(256) fft2_level_kernel(const Fr* __restrict__ s1, Fr* __restrict__ s2, size_t m,
                                                          unsigned i_level, unsigned log_m, const Fr* __restrict__ lo,
                                                          const Fr* __restrict__ hi, unsigned lb) {
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // q = k * 2^(i-1) + j
  if (q >= m / 2) return;
  const size_t half = (size_t)1 << (i_level - 1);
  const size_t ps = m >> i_level;
  size_t j = q & (half - 1), k = q >> (i_level - 1);
  size_t e = ((k + 1) << (i_level - 1)) & (m - 1);
  Fr f = lo[e & ((1u << lb) - 1)] * hi[e >> lb];
  Fr x = s1[(k << i_level) + 2 * j];
  Fr y = s1[(k << i_level) + 2 * j + 1] * f;
  s2[k * half + j] = x + y;
  s2[(k + ps) * half + j] = x - y;
}

// out[(i+1) mod m] = in[i] for i < m (rotate_right(1), dfft/mod.rs:177); out[m..total) = 0 (pad, :225-227)
template <class Fr>
__global__ void __launch_bounds__(256) rotate_pad_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, size_t m,
                                                          size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (i < m) out[i] = in[i == 0 ? m - 1 : i - 1];
  else out[i] = Fr::zero();
}

template <class Fr>
__global__ void __launch_bounds__(256) scale_kernel(Fr* __restrict__ x, const Fr* __restrict__ c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * *c;
}

// h = p*q - w on the odd positions of the unpacked 2m evaluations: the reference's
// `for i in 0..m { s1.swap(i, i*l + t) }` (ext_wit.rs:74-76) leaves s1[i] = original s1[i*l + t]
// for every i when l = 2, t = 1 (i*l+t > i: position i*l+t has not been touched when it is read).
template <class Fr>
__global__ void __launch_bounds__(256) h_odd_kernel(const Fr* __restrict__ p, const Fr* __restrict__ q,
                                                     const Fr* __restrict__ w, Fr* __restrict__ h, size_t m, unsigned l,
                                                     unsigned t) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  size_t s = i * l + t;
  h[i] = p[s] * q[s] - w[s];
}

// batched inverse + exclusive structure for d_pp: numden[i] = num[i] * den[i]^-1 (dpp/mod.rs:58-61)
template <class Fr>
__global__ void __launch_bounds__(256) ratio_kernel(const Fr* __restrict__ numden, Fr* __restrict__ out, size_t half) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < half) out[i] = numden[i] * numden[i + half].inv();
}

// inclusive prefix products (dpp/mod.rs:66-69); three-phase scan with one workgroup per 2048 elements
constexpr unsigned kScanTile = 2048;
template <class Fr>
__global__ void __launch_bounds__(256) prefix_prod_tile_kernel(Fr* __restrict__ x, size_t n, Fr* __restrict__ tile_tot) {
  __shared__ Fr sh[256];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  const unsigned per = kScanTile / 256;
  size_t lo = base + (size_t)threadIdx.x * per;
  Fr acc = Fr::one();
  for (unsigned j = 0; j < per; j++)
    if (lo + j < n) { acc = acc * x[lo + j]; x[lo + j] = acc; }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (unsigned d = 1; d < 256; d <<= 1) {
    Fr v = threadIdx.x >= d ? sh[threadIdx.x - d] : Fr::one();
    __syncthreads();
    sh[threadIdx.x] = sh[threadIdx.x] * v;
    __syncthreads();
  }
  Fr pre = threadIdx.x ? sh[threadIdx.x - 1] : Fr::one();
  for (unsigned j = 0; j < per; j++)
    if (lo + j < n) x[lo + j] = x[lo + j] * pre;
  if (threadIdx.x == 255) tile_tot[blockIdx.x] = sh[255];
}
template <class Fr>
__global__ void prefix_prod_tops_kernel(Fr* tile_tot, size_t ntiles) {   // serial over tiles (ntiles = n/2048)
  Fr acc = Fr::one();
  for (size_t i = 0; i < ntiles; i++) { Fr t = tile_tot[i]; tile_tot[i] = acc; acc = acc * t; }
}
template <class Fr>
__global__ void __launch_bounds__(256) prefix_prod_fix_kernel(Fr* __restrict__ x, size_t n, const Fr* __restrict__ tile_tot) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * tile_tot[i / kScanTile];
}

// ---- "in the exponent": out[e][r] = sum_c M[r][c] * P[e][c], M canonical (dmsm/mod.rs:7-68) -----------
template <class F, class Fr>
__global__ void __launch_bounds__(64) matvec_points_kernel(const Fr* __restrict__ Mc, unsigned rows, unsigned cols,
                                                            const Affine<F>* __restrict__ in, Affine<F>* __restrict__ out,
                                                            size_t count) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= count * rows) return;
  size_t e = gid / rows;
  unsigned r = (unsigned)(gid % rows);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (unsigned c = 0; c < cols; c++) {
    XYZZ<F> p = XYZZ<F>::from_affine(in[e * cols + c]);
    acc = acc.add(scalar_mul<F, Fr::NL>(p, Mc[r * cols + c].l));
  }
  out[gid] = acc.to_affine();
}

template <class F>
__global__ void affine_to_jacobian_kernel(const Affine<F>* in, Jacobian<F>* out) {
  *out = XYZZ<F>::from_affine(*in).to_jacobian();
}

template <class T>
static inline unsigned nblk(T n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace dg16
