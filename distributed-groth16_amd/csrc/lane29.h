// Limb-per-lane arithmetic for the LATENCY chains of the 9-limb fields (Horner tail, scalar multiples, the late steps of
// the bucket trees, the king's combinations, the proof assembly) -- round 6.
//
// The wave-cooperative chains of msm_impl.h (dbl_wave29 / add_wave29) run one base-field product per LANE: a level of
// a group operation is then never shorter than one 162-mad product on one lane (227 dependent VALU issues, 0.45 us) plus
// ~36 DPP moves and the selects that route nine-register values between slots (0.79 us as built).  Here a field element
// is ONE register: lane i of a row of 16 lanes holds limb i (W = 29, nine limbs; lanes 9..15 hold zero), and a
// Montgomery product is column-parallel --
//     A   column l of a b on lane l:            9 row broadcasts of a_i, 9 row shifts of b, 10 v_mad_u64_u32
//     B1  T mod R as loose limbs:               every 64-bit column cut into bits 0-28 / 29-57 / 58-63, the upper pieces
//                                               added one and two lanes up (v_add_u32_dpp row_shr): NO carry ripple
//     B2  m = T p' mod R (p' = -1/p mod R):     9 broadcasts, 9 mads against pre-shifted constant registers
//     B3  loose limbs of m                      (any representative of m mod R serves)
//     B4  T + m p:                              9 broadcasts, 10 mads
//     B5  the carry out of the (zero mod R) low half: the three-piece carry plus e = (L_8 + 2) >> 29 in {0..3}
//     B6  limbs 0..6 = columns 9..15 rotated down (row_ror:7), limbs 7, 8 from column 16 and the pieces that left the row
// -- ~100 VALU issues instead of 227 + routing, and a field addition / subtraction is 1-2 instructions + a 3-instruction
// carry pass instead of 9-27.  The four ROWS of a wave run the (up to four) independent products of a level; between
// levels the products change rows with v_permlane16_swap / v_permlane32_swap (gfx950: one instruction exchanges the odd
// rows of one register with the even rows of another; broadcasting all four rows' results to every row is 7 issues).
//
// The algorithm is stated executably, with its bounds, in tools/lane29_model.py (tests/test_lane29_model.py checks it
// against big integers at the extreme limb values).  Bounds: a product takes limbs <= LOOSE = 2^30 + 2^10 (a dual
// product a b + c d needs b, d <= TIGHT = 2^29 + 2^4), values < 2^261; it returns tight limbs and a value
// < T / R + 2.2 p.  Subtraction adds k p in a "spread" form whose limbs dominate the subtrahend's.
#pragma once
#include "fp29.h"
#include "ec29.h"

namespace dg16 {
namespace lane29 {

constexpr int W = 29, N = 9;
constexpr uint32_t MASK = (1u << W) - 1;

// p' = -p^-1 mod 2^261 as nine 29-bit limbs (Montgomery reduction of 1: x with 1 + x p = 0 mod R)
template <class P>
constexpr LimbArr<N> pprime_limbs() {
  static_assert(RR<P>::N == N && RR<P>::W == W, "nine limbs of 29 bits");
  uint64_t acc[2 * N + 1] = {};
  acc[0] = 1;
  LimbArr<N> x{};
  for (int k = 0; k < N; k++) {
    const uint32_t xk = (uint32_t)((acc[k] & MASK) * (uint64_t)RR<P>::INV) & MASK;
    x.v[k] = xk;
    uint64_t carry = 0;
    for (int j = 0; j < N; j++) {
      const uint64_t t = acc[k + j] + (uint64_t)xk * RR<P>::PL.v[j] + carry;
      acc[k + j] = t & MASK;
      carry = t >> W;
    }
    for (int j = k + N; carry && j <= 2 * N; j++) {
      const uint64_t t = acc[j] + carry;
      acc[j] = t & MASK;
      carry = t >> W;
    }
  }
  return x;
}

// k p in normalised limbs (k p < 2^261)
template <class P>
constexpr LimbArr<N> kp_norm(int k) {
  LimbArr<N> r{};
  uint64_t carry = 0;
  for (int i = 0; i < N; i++) {
    const uint64_t v = (uint64_t)RR<P>::PL.v[i] * (uint32_t)k + carry;
    r.v[i] = (uint32_t)(v & MASK);
    carry = v >> W;
  }
  r.v[N - 1] += (uint32_t)(carry << W);
  return r;
}

// multiples of p the chains subtract with (value bound of the subtrahend + 1) and the spread every one of them uses:
// 4 x 2^29 moved from each limb into the one below, so that every limb below the top is >= 2^31 - 4
constexpr int kSpread = 4;
constexpr int kSubKs[4] = {4, 6, 9, 13};
constexpr int kZeroMultiples = 7;       // is_zero compares against 0, p, ..., 6 p (its argument is < 6.5 p)

enum Row : int {
  ROW_PS = 0,                  // 9 rows: PS[i][l] = p_(l - i)
  ROW_PP = 9,                  // 9 rows: PP[i][l] = p'_(l - i), l < 9
  ROW_P16 = 18,                // p_8 on lane 0
  ROW_SUB = 19,                // 4 rows: spread k p
  ROW_JP = 23,                 // 7 rows: j p, normalised
  ROW_ONE = 30,                // R mod p
  ROW_COUNT = 31
};
struct TabData {
  uint32_t t[ROW_COUNT][16];
};
template <class P>
constexpr TabData make_tab() {
  TabData d{};
  const LimbArr<N> pp = pprime_limbs<P>();
  for (int i = 0; i < N; i++)
    for (int l = 0; l < 16; l++) {
      d.t[ROW_PS + i][l] = (l - i >= 0 && l - i < N) ? RR<P>::PL.v[l - i] : 0u;
      d.t[ROW_PP + i][l] = (l - i >= 0 && l - i < N && l < N) ? pp.v[l - i] : 0u;
    }
  d.t[ROW_P16][0] = RR<P>::PL.v[N - 1];
  for (int s = 0; s < 4; s++) {
    const LimbArr<N> kp = rr::kp_limbs<P>(kSubKs[s], kSpread);
    for (int l = 0; l < N; l++) d.t[ROW_SUB + s][l] = kp.v[l];
  }
  for (int j = 0; j < kZeroMultiples; j++) {
    const LimbArr<N> jp = kp_norm<P>(j);
    for (int l = 0; l < N; l++) d.t[ROW_JP + j][l] = jp.v[l];
  }
  for (int l = 0; l < N; l++) d.t[ROW_ONE][l] = RR<P>::ONE.v[l];
  return d;
}
template <class P>
struct Tab {
  static constexpr TabData v = make_tab<P>();
};

#if defined(__HIPCC__)
// ---- row primitives: opaque asm (a DPP read needs two wait states after the VALU write of its source, which hipcc
// cannot see inside an asm; through the builtins its DPP combiner folds moves into consumers, one of which is broken on
// gfx950: DESIGN.md section 7.3) ----------------------------------------------------------------------------------------
#define DG_DPP_FULL "row_mask:0xf bank_mask:0xf"
#define DG_DPP_ZERO "row_mask:0xf bank_mask:0xf bound_ctrl:0"
// all nine limbs of a row's element, each to every lane of its row
__device__ __forceinline__ void bcast9(uint32_t (&o)[N], uint32_t v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32_dpp %0, %9 row_newbcast:0 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %1, %9 row_newbcast:1 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %2, %9 row_newbcast:2 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %3, %9 row_newbcast:3 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %4, %9 row_newbcast:4 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %5, %9 row_newbcast:5 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %6, %9 row_newbcast:6 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %7, %9 row_newbcast:7 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %8, %9 row_newbcast:8 " DG_DPP_FULL
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8])
      : "v"(v));
}
// o[i][l] = v[l - i] (zero below the row), i = 1..8; hi[l] = v[l + 8] (lane 0: limb 8)
__device__ __forceinline__ void shifts9(uint32_t (&o)[N], uint32_t& hi, uint32_t v) {
  o[0] = v;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32_dpp %0, %9 row_shr:1 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %1, %9 row_shr:2 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %2, %9 row_shr:3 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %3, %9 row_shr:4 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %4, %9 row_shr:5 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %5, %9 row_shr:6 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %6, %9 row_shr:7 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %7, %9 row_shr:8 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %8, %9 row_shl:8 " DG_DPP_ZERO
      : "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(hi)
      : "v"(v));
}
// a + shr1(b)
__device__ __forceinline__ uint32_t add_shr1(uint32_t a, uint32_t b) {
  uint32_t r;
  asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 row_shr:1 " DG_DPP_ZERO : "=&v"(r) : "v"(b), "v"(a));
  return r;
}
// p0 + shr1(p1) + shr2(p2)
__device__ __forceinline__ uint32_t add_pieces(uint32_t p0, uint32_t p1, uint32_t p2) {
  uint32_t r;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %1, %2 row_shr:1 " DG_DPP_ZERO "\n\t"
      "v_add_u32_dpp %0, %3, %0 row_shr:2 " DG_DPP_ZERO
      : "=&v"(r)
      : "v"(p1), "v"(p0), "v"(p2));
  return r;
}
// a 64-bit column -> its three pieces summed into the lanes they belong to (loose limbs < 2^30 + 2^7)
__device__ __forceinline__ uint32_t three_piece(uint64_t c, uint32_t* p1_out = nullptr, uint32_t* p2_out = nullptr) {
  const uint32_t p0 = (uint32_t)c & MASK;
  const uint32_t p1 = (uint32_t)(c >> W) & MASK;
  const uint32_t p2 = (uint32_t)(c >> (2 * W));
  if (p1_out) *p1_out = p1;
  if (p2_out) *p2_out = p2;
  return add_pieces(p0, p1, p2);
}
// one parallel carry pass: limbs <= 2^29 + (largest limb >> 29) afterwards; the value (< 2^261) is unchanged
__device__ __forceinline__ uint32_t renorm(uint32_t v) { return add_shr1(v & MASK, v >> W); }

// per-lane constant registers of one field (built once per kernel)
template <class P>
struct K {
  uint32_t ps[N], pp[N], p16, sub[4], lane8, jpv[kZeroMultiples];
  unsigned l16, row;
  __device__ __forceinline__ void init() {
    const unsigned lane = __lane_id();
    l16 = lane & 15u;
    row = lane >> 4;
#pragma unroll
    for (int i = 0; i < N; i++) {
      ps[i] = Tab<P>::v.t[ROW_PS + i][l16];
      pp[i] = Tab<P>::v.t[ROW_PP + i][l16];
    }
    p16 = Tab<P>::v.t[ROW_P16][l16];
#pragma unroll
    for (int s = 0; s < 4; s++) sub[s] = Tab<P>::v.t[ROW_SUB + s][l16];
#pragma unroll
    for (int j = 0; j < kZeroMultiples; j++) jpv[j] = Tab<P>::v.t[ROW_JP + j][l16];
    lane8 = l16 == 8 ? 0xFFFFFFFFu : 0u;
  }
  __device__ __forceinline__ uint32_t jp(int j) const { return jpv[j]; }
  __device__ __forceinline__ uint32_t one() const { return Tab<P>::v.t[ROW_ONE][l16]; }
};

// (a b) / R mod p.  Limbs of a, b <= LOOSE, values < 2^261; result tight (limbs <= 2^29 + 2), value < a b / R + 2.2 p
template <class P>
__device__ __forceinline__ uint32_t mul(const K<P>& k, uint32_t a, uint32_t b) {
  uint32_t ab[N], bs[N], b16;
  bcast9(ab, a);
  shifts9(bs, b16, b);
  uint64_t main = 0;
#pragma unroll
  for (int i = 0; i < N; i++) main += (uint64_t)ab[i] * bs[i];
  uint64_t hi = (uint64_t)ab[N - 1] * b16;
  // B1, B2
  const uint32_t t = three_piece(main);
  uint32_t tb[N];
  bcast9(tb, t);
  uint64_t mc = 0;
#pragma unroll
  for (int i = 0; i < N; i++) mc += (uint64_t)tb[i] * k.pp[i];
  // B3, B4
  const uint32_t m = three_piece(mc);
  uint32_t mb[N];
  bcast9(mb, m);
#pragma unroll
  for (int i = 0; i < N; i++) main += (uint64_t)mb[i] * k.ps[i];
  hi += (uint64_t)mb[N - 1] * k.p16;
  // B5
  uint32_t p1, p2;
  uint32_t L = three_piece(main, &p1, &p2);
  const uint32_t e = ((L + 2u) >> W) & k.lane8;
  L = add_shr1(L, e);
  // B6
  const uint32_t h0 = (uint32_t)hi & MASK, h1 = (uint32_t)(hi >> W) & MASK;
  uint32_t top, rot;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32_dpp %0, %2 row_shr:7 " DG_DPP_ZERO "\n\t"
      "v_add_u32_dpp %0, %3, %0 row_shr:8 " DG_DPP_ZERO "\n\t"
      "v_add_u32_dpp %0, %4, %0 row_shl:8 " DG_DPP_ZERO "\n\t"
      "v_add_u32_dpp %0, %5, %0 row_shl:7 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %1, %6 row_ror:7 " DG_DPP_FULL
      : "=&v"(top), "=&v"(rot)
      : "v"(h0), "v"(h1), "v"(p1), "v"(p2), "v"(L));
  const uint32_t res = k.l16 < 7u ? rot : top;
  return renorm(res);
}

// ---- additions: limb-wise, then one carry pass ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return renorm(a + b); }
__device__ __forceinline__ uint32_t dbl(uint32_t a) { return renorm(a << 1); }
__device__ __forceinline__ uint32_t tpl(uint32_t a) { return renorm(a + (a << 1)); }
// a - b + k p, SUB = index into kSubKs (k > value bound of b in p; limbs of b < 2^31 - 4, of a <= 2^30)
template <int SUB, class P>
__device__ __forceinline__ uint32_t sub(const K<P>& k, uint32_t a, uint32_t b) { return renorm(a + (k.sub[SUB] - b)); }

// ---- rows ---------------------------------------------------------------------------------------------------------------
// r holds one value per row: every row gets all four
__device__ __forceinline__ void rows_to_all(uint32_t r, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3) {
  // odd rows of the first operand <-> even rows of the second: [A B C D], [A B C D] -> [A A C C], [B B D D]
  const auto s = __builtin_amdgcn_permlane16_swap(r, r, false, false);
  const auto e = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false);   // [A A C C] x2 -> [A A A A], [C C C C]
  const auto o = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false);
  r0 = e[0];
  r2 = e[1];
  r1 = o[0];
  r3 = o[1];
}
// element-wise select by row
__device__ __forceinline__ uint32_t by_row(unsigned row, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
  return row == 0 ? v0 : row == 1 ? v1 : row == 2 ? v2 : v3;
}

// carry passes until every limb is below 2^29 (the form Fe<P, B, 1> holds); usually one or two
__device__ __forceinline__ uint32_t full_norm(uint32_t v) {
  while (__builtin_amdgcn_ballot_w64(v > MASK)) v = renorm(v);
  return v;
}
// is the value (tight limbs, < 6.5 p) zero mod p?  Uniform across the wave when the operand is (every row holds it).
template <class P>
__device__ __forceinline__ bool is_zero(const K<P>& k, uint32_t v) {
  v = full_norm(v);
  bool z = false;
#pragma unroll
  for (int j = 0; j < kZeroMultiples; j++) z = z || (__builtin_amdgcn_ballot_w64(v != k.jp(j)) & 0xFFFFull) == 0;
  return z;
}

// ---- XYZZ points, one register per coordinate, the same in every row -------------------------------------------------
struct Pt {
  uint32_t x, y, zz, zzz;
  bool inf;               // uniform
};

// Fe (nine registers, the same in every lane) <-> lane form
template <class P, int B>
__device__ __forceinline__ uint32_t to_lane(const K<P>& k, const Fe<P, B, 1>& f) {
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < N; i++) v = k.l16 == (unsigned)i ? f.l[i] : v;
  return v;
}
template <class P, int B>
__device__ __forceinline__ Fe<P, B, 1> from_lane(uint32_t v) {
  uint32_t o[N];
  bcast9(o, v);
  Fe<P, B, 1> f;
#pragma unroll
  for (int i = 0; i < N; i++) f.l[i] = o[i];
  return f;
}

// 2 p                                                                                         (dbl-2008-s-1, a = 0)
// value bounds (in p): coordinates in < 7, out < 2.5 + 13 (x3, y3), < 2.5 (zz3, zzz3); tools/lane29_model.py
template <class P>
__device__ __forceinline__ Pt dbl_pt(const K<P>& k, const Pt& p) {
  if (p.inf) return p;
  const unsigned row = k.row;
  const uint32_t u = p.y << 1;                                   // loose
  // level 1: rows 0, 2: v = u^2 | rows 1, 3: xx = x^2
  const uint32_t a1 = (row & 1u) ? p.x : u;
  const uint32_t r1 = mul(k, a1, a1);
  const auto s1 = __builtin_amdgcn_permlane16_swap(r1, r1, false, false);
  const uint32_t v = s1[0], xx = s1[1];
  const uint32_t m = tpl(xx);
  // level 2: w = u v | s = x v | m^2 | zz3 = v zz
  const uint32_t a2 = by_row(row, u, p.x, m, p.zz);
  const uint32_t b2 = row == 2 ? m : v;
  const uint32_t r2 = mul(k, a2, b2);
  uint32_t w, s, mm, zz3;
  rows_to_all(r2, w, s, mm, zz3);
  const uint32_t x3 = sub<1>(k, mm, s << 1);                     // 2 s < 5 p
  const uint32_t sx = sub<3>(k, s, x3);                          // x3 < 2.5 p + 6 p
  // level 3: m (s - x3) | w y | zzz3 = w zzz
  const uint32_t a3 = row == 0 ? m : w;
  const uint32_t b3 = by_row(row, sx, p.y, p.zzz, p.zzz);
  const uint32_t r3 = mul(k, a3, b3);
  uint32_t t0, t1, zzz3, unused;
  rows_to_all(r3, t0, t1, zzz3, unused);
  return {x3, sub<0>(k, t0, t1), zz3, zzz3, false};
}

// p + o, complete                                                                                       (add-2008-s)
template <class P>
__device__ __forceinline__ Pt add_pt(const K<P>& k, const Pt& p, const Pt& o) {
  if (o.inf) return p;
  if (p.inf) return o;
  const unsigned row = k.row;
  // level 1: u1 = x1 zz2 | u2 = x2 zz1 | s1 = y1 zzz2 | s2 = y2 zzz1
  const uint32_t r1 = mul(k, by_row(row, p.x, o.x, p.y, o.y), by_row(row, o.zz, p.zz, o.zzz, p.zzz));
  uint32_t u1, u2, s1, s2;
  rows_to_all(r1, u1, u2, s1, s2);
  const uint32_t pd = sub<0>(k, u2, u1), rd = sub<0>(k, s2, s1);          // < 2.5 p + 4 p
  if (is_zero(k, pd)) {
    if (is_zero(k, rd)) return dbl_pt(k, p);
    return {k.one(), k.one(), 0u, 0u, true};
  }
  // level 2: pp = p^2 | rr = r^2 | zz1 zz2 | zzz1 zzz2
  const uint32_t r2 = mul(k, by_row(row, pd, rd, p.zz, p.zzz), by_row(row, pd, rd, o.zz, o.zzz));
  uint32_t pp, rr, zzp, zzzp;
  rows_to_all(r2, pp, rr, zzp, zzzp);
  // level 3: ppp = p pp | q = u1 pp | zz3 = (zz1 zz2) pp
  const uint32_t r3 = mul(k, by_row(row, pd, u1, zzp, zzp), pp);
  uint32_t ppp, q, zz3, unused;
  rows_to_all(r3, ppp, q, zz3, unused);
  const uint32_t x3 = sub<2>(k, rr, ppp + (q << 1));                      // ppp + 2 q < 7.5 p
  const uint32_t qx = sub<3>(k, q, x3);                                   // x3 < 2.5 p + 9 p
  // level 4: r (q - x3) | s1 ppp | zzz3 = (zzz1 zzz2) ppp
  const uint32_t r4 = mul(k, by_row(row, rd, s1, zzzp, zzzp), row == 0 ? qx : ppp);
  uint32_t t0, t1, zzz3;
  rows_to_all(r4, t0, t1, zzz3, unused);
  return {x3, sub<0>(k, t0, t1), zz3, zzz3, false};
}

// XYZZ29 (nine registers per coordinate, uniform across the wave) <-> Pt
template <class F>
__device__ __forceinline__ Pt to_pt(const K<typename FieldOf<F>::Params>& k, const XYZZ29<F>& p) {
  return {to_lane(k, p.x), to_lane(k, p.y), to_lane(k, p.zz), to_lane(k, p.zzz), p.is_inf()};
}
template <class F>
__device__ __forceinline__ XYZZ29<F> from_pt(const K<typename FieldOf<F>::Params>& k, const Pt& p) {
  using P = typename FieldOf<F>::Params;
  constexpr int BS = XYZZ29<F>::BS;
  if (p.inf) return XYZZ29<F>::inf();
  // x3 < 11.25 p goes under the storage bound through a product with R mod p; y3 < 6.7 p, zz3, zzz3 < 3.3 p fit
  const uint32_t one = k.one();
  static_assert(BS >= 448, "y3 < 6.7 p is stored as it is");
  return {from_lane<P, BS>(full_norm(mul(k, p.x, one))), from_lane<P, BS>(full_norm(p.y)),
          from_lane<P, BS>(full_norm(p.zz)), from_lane<P, BS>(full_norm(p.zzz))};
}
// which coordinate fields run their chains in this form: nine-limb base fields (BN254 Fq)
template <class F>
constexpr bool enabled() {
#ifdef DG16_NO_LANE_CHAINS        // (A/B switch: the one-product-per-lane chains of msm_impl.h everywhere)
  return false;
#else
  return !FieldOf<F>::EXT && RR<typename FieldOf<F>::Params>::N == N && RR<typename FieldOf<F>::Params>::W == W;
#endif
}

// ---- lane form in memory: the words of an XYZZ29<F> (nine per coordinate), read and written by the lanes that own the
// limbs.  What the chains store for THEMSELVES may hold tight limbs and coordinates up to ~12 p (raw); what other code
// reads as an XYZZ29 must come from from_pt (normalised limbs, below the storage bound).  Identity: ZZ all zero. --------
template <class F>
__device__ __forceinline__ Pt load_pt(const K<typename FieldOf<F>::Params>& k, const XYZZ29<F>* src) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src);
  const unsigned i = k.l16 < (unsigned)N ? k.l16 : 0u;
  const bool on = k.l16 < (unsigned)N;
  Pt p;
  p.x = on ? w[i] : 0u;
  p.y = on ? w[N + i] : 0u;
  p.zz = on ? w[2 * N + i] : 0u;
  p.zzz = on ? w[3 * N + i] : 0u;
  p.inf = __builtin_amdgcn_ballot_w64(p.zz != 0u) == 0;
  return p;
}
template <class F>
__device__ __forceinline__ void store_pt_raw(const K<typename FieldOf<F>::Params>& k, XYZZ29<F>* dst, const Pt& p) {
  uint32_t* w = reinterpret_cast<uint32_t*>(dst);
  if (k.row == 0 && k.l16 < (unsigned)N) {
    w[k.l16] = p.x;
    w[N + k.l16] = p.y;
    w[2 * N + k.l16] = p.inf ? 0u : p.zz;
    w[3 * N + k.l16] = p.inf ? 0u : p.zzz;
  }
}
// the same, as a proper XYZZ29 (normalised limbs, coordinates below the storage bound: x through a product with R mod p)
template <class F>
__device__ __forceinline__ void store_pt(const K<typename FieldOf<F>::Params>& k, XYZZ29<F>* dst, const Pt& p) {
  Pt q = p;
  if (!p.inf) {
    q.x = full_norm(mul(k, p.x, k.one()));
    q.y = full_norm(p.y);          // y3 < 6.7 p (dbl_pt / add_pt), below the storage bound of 7 p
    q.zz = full_norm(p.zz);
    q.zzz = full_norm(p.zzz);
  }
  store_pt_raw<F>(k, dst, q);
}
// -p (y < 6.7 p -> 9 p - y)
template <class P>
__device__ __forceinline__ Pt neg_pt(const K<P>& k, const Pt& p) {
  return {p.x, sub<2>(k, 0u, p.y), p.zz, p.zzz, p.inf};
}
#endif  // __HIPCC__

}  // namespace lane29
}  // namespace dg16
