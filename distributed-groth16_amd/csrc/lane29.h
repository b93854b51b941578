// Limb-per-lane arithmetic for the LATENCY chains (Horner tail, scalar multiples, small bucket reductions, the top level of
// large ones, the king's combinations, inversions) of all six groups -- round 6.  The nine-limb form is described here;
// the fourteen-limb one (two registers per element) and the quadratic extensions further down.
// EVERY function below assumes all 64 lanes of the wave active and its operands uniform across the four rows (unless a
// product has just put four different results on them): the row primitives are opaque asm that reads other lanes.
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
#include <type_traits>
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

// p - 2 as 32-bit words (the Fermat exponent of inv below)
template <class P>
constexpr LimbArr<P::NL> p_minus_2_words() {
  LimbArr<P::NL> r{};
  uint64_t borrow = 2;
  for (int i = 0; i < P::NL; i++) {
    const uint64_t d = (uint64_t)P::P[i] - borrow;
    r.v[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
  return r;
}
template <class P>
struct PM2 {
  static constexpr LimbArr<P::NL> v = p_minus_2_words<P>();
};

// Multiples of p the group law subtracts with, per class of field (tools/lane_bounds.py walks dbl_pt / add_pt with these
// and checks that the bounds of a chain's running point close below R): SUB[i] = the K of sub<i> (K > value bound of the
// subtrahend), NEG = the multiple a quadratic-extension product negates b.c1 with, ZERO = how many multiples of p a zero
// test compares against (its argument is below ZERO p).  Every constant is used in a "spread" form: 4 x 2^29 moved from
// each limb into the one below, so that every limb below the top is >= 2^31 - 4 and dominates the subtrahend's.
constexpr int kSpread = 4;
template <bool EXT> struct LawK;
template <> struct LawK<false> {
  static constexpr int SUB[4] = {4, 6, 9, 13};
  static constexpr int NEG = 0;
  static constexpr int ZERO = 7;
};
template <> struct LawK<true> {
  static constexpr int SUB[4] = {4, 8, 10, 14};
  static constexpr int NEG = 19;
  static constexpr int ZERO = 8;
};
constexpr int kMaxZero = 8;

enum Row : int {
  ROW_PS = 0,                  // 9 rows: PS[i][l] = p_(l - i)
  ROW_PP = 9,                  // 9 rows: PP[i][l] = p'_(l - i), l < 9
  ROW_P16 = 18,                // p_8 on lane 0
  ROW_SUB = 19,                // 4 rows: spread K p
  ROW_NEG = 23,                // spread NEG p
  ROW_JP = 24,                 // kMaxZero rows: j p, normalised
  ROW_ONE = 32,                // R mod p
  ROW_RMP = 33,                // 2^261 - p (adding it and dropping the carry subtracts p)
  ROW_COUNT = 34
};
struct TabData {
  uint32_t t[ROW_COUNT][16];
};
template <class P, bool EXT>
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
    const LimbArr<N> kp = rr::kp_limbs<P>(LawK<EXT>::SUB[s], kSpread);
    for (int l = 0; l < N; l++) d.t[ROW_SUB + s][l] = kp.v[l];
  }
  if (EXT) {
    const LimbArr<N> kp = rr::kp_limbs<P>(LawK<EXT>::NEG, kSpread);
    for (int l = 0; l < N; l++) d.t[ROW_NEG][l] = kp.v[l];
  }
  for (int j = 0; j < LawK<EXT>::ZERO; j++) {
    const LimbArr<N> jp = kp_norm<P>(j);
    for (int l = 0; l < N; l++) d.t[ROW_JP + j][l] = jp.v[l];
  }
  for (int l = 0; l < N; l++) d.t[ROW_ONE][l] = RR<P>::ONE.v[l];
  {
    // 2^261 - p = (all-ones) - p + 1
    uint64_t carry = 1;
    for (int l = 0; l < N; l++) {
      const uint64_t v = (uint64_t)(MASK - RR<P>::PL.v[l]) + carry;
      d.t[ROW_RMP][l] = (uint32_t)(v & MASK);
      carry = v >> W;
    }
  }
  return d;
}
template <class P, bool EXT>
struct Tab {
  static constexpr TabData v = make_tab<P, EXT>();
};

// ---- fourteen limbs of 28 bits (BLS12-377 / BLS12-381 Fq): an element is TWO registers on one row, `lo` = limbs 0..6 and
// `hi` = limbs 7..13, both on lanes 0..6, so that nothing crosses a row: a product is three column sets C0 = lo lo
// (columns 0..12), C1 = lo hi + hi lo (7..19), C2 = hi hi (14..26) on lanes 0..12 and the Montgomery reduction runs over
// the two digits of base B = 2^196 (m_j = digit_j p'0 mod B; T += m_j p B^j; carry the digit out), each step
// column-parallel on seven lanes: ~190 issues against 392 mads + ~140 on one lane (tools/lane29_model.py: mont14).
namespace l14 {
constexpr int W = 28, N = 14, H = 7;
constexpr uint32_t MASK = (1u << W) - 1;
template <class P>
constexpr LimbArr<H> pprime0_limbs() {       // -p^-1 mod 2^196
  static_assert(RR<P>::N == N && RR<P>::W == W, "fourteen limbs of 28 bits");
  uint64_t acc[2 * H + 2] = {};
  acc[0] = 1;
  LimbArr<H> x{};
  for (int k = 0; k < H; k++) {
    const uint32_t xk = (uint32_t)((acc[k] & MASK) * (uint64_t)RR<P>::INV) & MASK;
    x.v[k] = xk;
    uint64_t carry = 0;
    for (int j = 0; j < H; j++) {            // only the low seven limbs of p matter mod 2^196
      if (k + j >= H) break;
      const uint64_t t = acc[k + j] + (uint64_t)xk * RR<P>::PL.v[j] + carry;
      acc[k + j] = t & MASK;
      carry = t >> W;
    }
  }
  return x;
}
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
enum Row : int {
  ROW_PL = 0,                  // 7 rows: PL[i][l] = p_(l - i), limbs 0..6
  ROW_PH = 7,                  // 7 rows: PH[i][l] = p_(7 + l - i)
  ROW_PP = 14,                 // 7 rows: p'0_(l - i), l < 7
  ROW_SUB = 21,                // 4 x (lo, hi): spread K p
  ROW_NEG = 29,                // (lo, hi): spread NEG BETA p, the multiple a quadratic-extension product negates BETA b.c1 with
  ROW_JP = 31,                 // kMaxZero x (lo, hi): j p, normalised
  ROW_ONE = 47,                // (lo, hi): R mod p
  ROW_COUNT = 49
};
// the spread of the negation constant dominates BETA x a tight limb: BETA + 1 units of 2^28
template <class P> constexpr int neg_spread() { return Fq2Beta<P>::value + 1 < kSpread ? kSpread : Fq2Beta<P>::value + 1; }
struct TabData {
  uint32_t t[ROW_COUNT][16];
};
template <class P, bool EXT>
constexpr TabData make_tab() {
  TabData d{};
  const LimbArr<H> pp = pprime0_limbs<P>();
  for (int i = 0; i < H; i++)
    for (int l = 0; l < 16; l++) {
      const bool in = l - i >= 0 && l - i < H;
      d.t[ROW_PL + i][l] = in ? RR<P>::PL.v[l - i] : 0u;
      d.t[ROW_PH + i][l] = in ? RR<P>::PL.v[H + l - i] : 0u;
      d.t[ROW_PP + i][l] = (in && l < H) ? pp.v[l - i] : 0u;
    }
  for (int s = 0; s < 4; s++) {
    const LimbArr<N> kp = rr::kp_limbs<P>(LawK<EXT>::SUB[s], kSpread);
    for (int l = 0; l < H; l++) {
      d.t[ROW_SUB + 2 * s][l] = kp.v[l];
      d.t[ROW_SUB + 2 * s + 1][l] = kp.v[H + l];
    }
  }
  if (EXT) {
    const LimbArr<N> kp = rr::kp_limbs<P>(LawK<EXT>::NEG * Fq2Beta<P>::value, neg_spread<P>());
    for (int l = 0; l < H; l++) {
      d.t[ROW_NEG][l] = kp.v[l];
      d.t[ROW_NEG + 1][l] = kp.v[H + l];
    }
  }
  for (int j = 0; j < LawK<EXT>::ZERO; j++) {
    const LimbArr<N> jp = kp_norm<P>(j);
    for (int l = 0; l < H; l++) {
      d.t[ROW_JP + 2 * j][l] = jp.v[l];
      d.t[ROW_JP + 2 * j + 1][l] = jp.v[H + l];
    }
  }
  for (int l = 0; l < H; l++) {
    d.t[ROW_ONE][l] = RR<P>::ONE.v[l];
    d.t[ROW_ONE + 1][l] = RR<P>::ONE.v[H + l];
  }
  return d;
}
template <class P, bool EXT>
struct Tab {
  static constexpr TabData v = make_tab<P, EXT>();
};
}  // namespace l14

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

// per-lane constant registers of one field (built once per kernel); EXT: the constants of the quadratic-extension law
template <class P, bool EXT>
struct K {
  using L = LawK<EXT>;
  uint32_t ps[N], pp[N], p16, sub[4], neg, lane8, jpv[L::ZERO];
  unsigned l16, row;
  __device__ __forceinline__ void init() {
    const unsigned lane = __lane_id();
    l16 = lane & 15u;
    row = lane >> 4;
#pragma unroll
    for (int i = 0; i < N; i++) {
      ps[i] = Tab<P, EXT>::v.t[ROW_PS + i][l16];
      pp[i] = Tab<P, EXT>::v.t[ROW_PP + i][l16];
    }
    p16 = Tab<P, EXT>::v.t[ROW_P16][l16];
#pragma unroll
    for (int s = 0; s < 4; s++) sub[s] = Tab<P, EXT>::v.t[ROW_SUB + s][l16];
    neg = Tab<P, EXT>::v.t[ROW_NEG][l16];
#pragma unroll
    for (int j = 0; j < L::ZERO; j++) jpv[j] = Tab<P, EXT>::v.t[ROW_JP + j][l16];
    lane8 = l16 == 8 ? 0xFFFFFFFFu : 0u;
  }
  __device__ __forceinline__ uint32_t one() const { return Tab<P, EXT>::v.t[ROW_ONE][l16]; }
  __device__ __forceinline__ uint32_t r_minus_p() const { return Tab<P, EXT>::v.t[ROW_RMP][l16]; }
};

// columns of a b (+ c d): lane l owns column l; column 16 on lane 0 of `hi`
struct Cols {
  uint64_t main, hi;
};
__device__ __forceinline__ void cols_mad(Cols& c, const uint32_t (&ab)[N], const uint32_t (&bs)[N], uint32_t b16) {
#pragma unroll
  for (int i = 0; i < N; i++) c.main += (uint64_t)ab[i] * bs[i];
  c.hi += (uint64_t)ab[N - 1] * b16;
}
// T -> T / R mod p (B1 .. B6 of the header): tight limbs, value < T / R + 2.01 p
template <class KT>
__device__ __forceinline__ uint32_t reduce(const KT& k, Cols c) {
  const uint32_t t = three_piece(c.main);
  uint32_t tb[N];
  bcast9(tb, t);
  uint64_t mc = 0;
#pragma unroll
  for (int i = 0; i < N; i++) mc += (uint64_t)tb[i] * k.pp[i];
  const uint32_t m = three_piece(mc);
  uint32_t mb[N];
  bcast9(mb, m);
#pragma unroll
  for (int i = 0; i < N; i++) c.main += (uint64_t)mb[i] * k.ps[i];
  c.hi += (uint64_t)mb[N - 1] * k.p16;
  uint32_t p1, p2;
  uint32_t L = three_piece(c.main, &p1, &p2);
  const uint32_t e = ((L + 2u) >> W) & k.lane8;
  L = add_shr1(L, e);
  const uint32_t h0 = (uint32_t)c.hi & MASK, h1 = (uint32_t)(c.hi >> W) & MASK;
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
// (a b) / R mod p.  Limbs of a, b <= LOOSE, values < 2^261; result tight (limbs <= 2^29 + 2), value < a b / R + 2.01 p
template <class KT>
__device__ __forceinline__ uint32_t mul(const KT& k, uint32_t a, uint32_t b) {
  uint32_t ab[N], bs[N], b16;
  bcast9(ab, a);
  shifts9(bs, b16, b);
  Cols c = {0, 0};
  cols_mad(c, ab, bs, b16);
  return reduce(k, c);
}
// carry passes until every limb is below 2^29 (the form Fe<P, B, 1> holds); usually one or two
__device__ __forceinline__ uint32_t full_norm(uint32_t v) {
  while (__builtin_amdgcn_ballot_w64(v > MASK)) v = renorm(v);
  return v;
}
// v (normalised limbs, every row the same) -> v - p when v >= p: the limb-wise comparison as two 9-bit integers (bit l =
// limb l differs upwards / downwards: the highest differing limb decides)
template <class KT>
__device__ __forceinline__ uint32_t cond_sub_p(const KT& k, uint32_t v) {
  const uint32_t pl = k.ps[0];                       // PS[0][l] = p_l
  const unsigned gt = (unsigned)__builtin_amdgcn_ballot_w64(v > pl) & 0x1FFu;
  const unsigned lt = (unsigned)__builtin_amdgcn_ballot_w64(v < pl) & 0x1FFu;
  if (gt < lt) return v;
  uint32_t r = full_norm(v + k.r_minus_p());        // + (2^261 - p); the carry out of limb 8 lands on lane 9 and is dropped
  return k.l16 < (unsigned)N ? r : 0u;
}

// ---- rows ---------------------------------------------------------------------------------------------------------------
// r holds one value per row: every row gets all four
__device__ __forceinline__ void rows_to_all32(uint32_t r, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3) {
  // odd rows of the first operand <-> even rows of the second: [A B C D], [A B C D] -> [A A C C], [B B D D]
  const auto s = __builtin_amdgcn_permlane16_swap(r, r, false, false);
  const auto e = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false);   // [A A C C] x2 -> [A A A A], [C C C C]
  const auto o = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false);
  r0 = e[0];
  r2 = e[1];
  r1 = o[0];
  r3 = o[1];
}

// ---- field policies: what the group law below is written against ----------------------------------------------------
// E: an element (registers, the same in every row unless a product has just put four different ones on the four rows);
// every E handed around has TIGHT limbs (<= 2^29 + 16).
template <class P_>
struct Fq9 {
  using P = P_;
  static constexpr bool EXT = false;
  static constexpr int WORDS = N;
  using E = uint32_t;
  using KT = K<P, false>;
  static __device__ __forceinline__ E zero() { return 0u; }
  static __device__ __forceinline__ E one(const KT& k) { return k.one(); }
  static __device__ __forceinline__ E mul(const KT& k, E a, E b) { return lane29::mul(k, a, b); }
  static __device__ __forceinline__ E dbl(const KT&, E a) { return renorm(a << 1); }
  static __device__ __forceinline__ E tpl(const KT&, E a) { return renorm(a + (a << 1)); }
  static __device__ __forceinline__ E dbl_raw(E a) { return a << 1; }
  static __device__ __forceinline__ E add_raw(E a, E b) { return a + b; }
  // a - b + K p (limbs of b < 2^31 - 4, of a <= 2^30)
  template <int I>
  static __device__ __forceinline__ E sub(const KT& k, E a, E b) { return renorm(a + (k.sub[I] - b)); }
  static __device__ __forceinline__ E sel(bool c, E a, E b) { return c ? a : b; }
  static __device__ __forceinline__ bool is_zero(const KT& k, E v) {
    v = full_norm(v);
    bool z = false;
#pragma unroll
    for (int j = 0; j < KT::L::ZERO; j++) z = z || (__builtin_amdgcn_ballot_w64(v != k.jpv[j]) & 0xFFFFull) == 0;
    return z;
  }
  static __device__ __forceinline__ void swap16(E r, E& even, E& odd) {
    const auto s = __builtin_amdgcn_permlane16_swap(r, r, false, false);
    even = s[0];
    odd = s[1];
  }
  static __device__ __forceinline__ void rows_to_all(E r, E& r0, E& r1, E& r2, E& r3) { rows_to_all32(r, r0, r1, r2, r3); }
  static __device__ __forceinline__ bool any_nonzero(E v) { return __builtin_amdgcn_ballot_w64(v != 0u) != 0; }
  static __device__ __forceinline__ E load(const KT& k, const uint32_t* w) { return k.l16 < (unsigned)N ? w[k.l16] : 0u; }
  static __device__ __forceinline__ void store(const KT& k, uint32_t* w, E v) {
    if (k.row == 0 && k.l16 < (unsigned)N) w[k.l16] = v;
  }
  // normalised limbs and a value below the storage bound BS / 64 p (through a product with R mod p: < 2.1 p; then at
  // most one subtraction of p when the bound is that tight)
  template <int BS>
  static __device__ __forceinline__ E exit_norm(const KT& k, E v, bool reduce_it) {
    if (!reduce_it) return full_norm(v);
    v = full_norm(lane29::mul(k, v, k.one()));
    if constexpr (BS < 192) v = cond_sub_p(k, v);
    return v;
  }
  template <int B>
  static __device__ __forceinline__ E from_regs(const KT& k, const Fe<P, B, 1>& f) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < N; i++) v = k.l16 == (unsigned)i ? f.l[i] : v;
    return v;
  }
  template <int B>
  static __device__ __forceinline__ Fe<P, B, 1> to_regs(E v) {
    uint32_t o[N];
    bcast9(o, v);
    Fe<P, B, 1> f;
#pragma unroll
    for (int i = 0; i < N; i++) f.l[i] = o[i];
    return f;
  }
};
struct E2 {
  uint32_t c0, c1;
};
// quadratic extension u^2 = -1 over a nine-limb field (BN254 Fq2): both components of a product on the SAME row --
// c0 = a0 b0 + a1 (NEG p - b1), c1 = a0 b1 + a1 b0 as two dual-column accumulations over shared broadcasts of a, one
// reduction each: ~235 issues against 2 x 3 base products on three lanes + joins in the form it replaces
template <class P_>
struct Fq9x2 {
  using P = P_;
  static_assert(Fq2Beta<P>::value == 1, "u^2 = -1");
  static constexpr bool EXT = true;
  static constexpr int WORDS = 2 * N;
  using E = E2;
  using KT = K<P, true>;
  using B = Fq9<P>;
  static __device__ __forceinline__ E zero() { return {0u, 0u}; }
  static __device__ __forceinline__ E one(const KT& k) { return {k.one(), 0u}; }
  static __device__ __forceinline__ E mul(const KT& k, E a, E b) {
    const uint32_t nb1 = renorm(k.neg - b.c1);
    uint32_t a0[N], a1[N], b0s[N], b1s[N], n1s[N], b0h, b1h, n1h;
    bcast9(a0, a.c0);
    bcast9(a1, a.c1);
    shifts9(b0s, b0h, b.c0);
    shifts9(b1s, b1h, b.c1);
    shifts9(n1s, n1h, nb1);
    Cols x = {0, 0}, y = {0, 0};
    cols_mad(x, a0, b0s, b0h);
    cols_mad(x, a1, n1s, n1h);
    cols_mad(y, a0, b1s, b1h);
    cols_mad(y, a1, b0s, b0h);
    return {reduce(k, x), reduce(k, y)};
  }
  static __device__ __forceinline__ E dbl(const KT&, E a) { return {renorm(a.c0 << 1), renorm(a.c1 << 1)}; }
  static __device__ __forceinline__ E tpl(const KT&, E a) { return {renorm(a.c0 + (a.c0 << 1)), renorm(a.c1 + (a.c1 << 1))}; }
  static __device__ __forceinline__ E dbl_raw(E a) { return {a.c0 << 1, a.c1 << 1}; }
  static __device__ __forceinline__ E add_raw(E a, E b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  template <int I>
  static __device__ __forceinline__ E sub(const KT& k, E a, E b) {
    return {renorm(a.c0 + (k.sub[I] - b.c0)), renorm(a.c1 + (k.sub[I] - b.c1))};
  }
  static __device__ __forceinline__ E sel(bool c, E a, E b) { return {c ? a.c0 : b.c0, c ? a.c1 : b.c1}; }
  static __device__ __forceinline__ bool is_zero1(const KT& k, uint32_t v) {
    v = full_norm(v);
    bool z = false;
#pragma unroll
    for (int j = 0; j < KT::L::ZERO; j++) z = z || (__builtin_amdgcn_ballot_w64(v != k.jpv[j]) & 0xFFFFull) == 0;
    return z;
  }
  static __device__ __forceinline__ bool is_zero(const KT& k, E v) { return is_zero1(k, v.c0) && is_zero1(k, v.c1); }
  static __device__ __forceinline__ void swap16(E r, E& even, E& odd) {
    const auto s = __builtin_amdgcn_permlane16_swap(r.c0, r.c0, false, false);
    const auto t = __builtin_amdgcn_permlane16_swap(r.c1, r.c1, false, false);
    even = {s[0], t[0]};
    odd = {s[1], t[1]};
  }
  static __device__ __forceinline__ void rows_to_all(E r, E& r0, E& r1, E& r2, E& r3) {
    rows_to_all32(r.c0, r0.c0, r1.c0, r2.c0, r3.c0);
    rows_to_all32(r.c1, r0.c1, r1.c1, r2.c1, r3.c1);
  }
  static __device__ __forceinline__ bool any_nonzero(E v) { return __builtin_amdgcn_ballot_w64((v.c0 | v.c1) != 0u) != 0; }
  static __device__ __forceinline__ E load(const KT& k, const uint32_t* w) {
    const bool on = k.l16 < (unsigned)N;
    const unsigned i = on ? k.l16 : 0u;
    return {on ? w[i] : 0u, on ? w[N + i] : 0u};
  }
  static __device__ __forceinline__ void store(const KT& k, uint32_t* w, E v) {
    if (k.row == 0 && k.l16 < (unsigned)N) {
      w[k.l16] = v.c0;
      w[N + k.l16] = v.c1;
    }
  }
  template <int BS>
  static __device__ __forceinline__ E exit_norm(const KT& k, E v, bool reduce_it) {
    if (!reduce_it) return {full_norm(v.c0), full_norm(v.c1)};
    E r = {full_norm(lane29::mul(k, v.c0, k.one())), full_norm(lane29::mul(k, v.c1, k.one()))};
    if constexpr (BS < 192) r = {cond_sub_p(k, r.c0), cond_sub_p(k, r.c1)};
    return r;
  }
  template <int BB>
  static __device__ __forceinline__ E from_regs(const KT& k, const Fe2<P, BB, 1>& f) {
    uint32_t v0 = 0, v1 = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      v0 = k.l16 == (unsigned)i ? f.c0.l[i] : v0;
      v1 = k.l16 == (unsigned)i ? f.c1.l[i] : v1;
    }
    return {v0, v1};
  }
  template <int BB>
  static __device__ __forceinline__ Fe2<P, BB, 1> to_regs(E v) {
    uint32_t o0[N], o1[N];
    bcast9(o0, v.c0);
    bcast9(o1, v.c1);
    Fe2<P, BB, 1> f;
#pragma unroll
    for (int i = 0; i < N; i++) {
      f.c0.l[i] = o0[i];
      f.c1.l[i] = o1[i];
    }
    return f;
  }
};
// ---- fourteen limbs: the same interface on two registers ---------------------------------------------------------------
namespace l14 {
__device__ __forceinline__ void bcast7(uint32_t (&o)[H], uint32_t v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32_dpp %0, %7 row_newbcast:0 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %1, %7 row_newbcast:1 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %2, %7 row_newbcast:2 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %3, %7 row_newbcast:3 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %4, %7 row_newbcast:4 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %5, %7 row_newbcast:5 " DG_DPP_FULL "\n\t"
      "v_mov_b32_dpp %6, %7 row_newbcast:6 " DG_DPP_FULL
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6])
      : "v"(v));
}
__device__ __forceinline__ void shifts7(uint32_t (&o)[H], uint32_t v) {
  o[0] = v;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32_dpp %0, %6 row_shr:1 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %1, %6 row_shr:2 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %2, %6 row_shr:3 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %3, %6 row_shr:4 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %4, %6 row_shr:5 " DG_DPP_ZERO "\n\t"
      "v_mov_b32_dpp %5, %6 row_shr:6 " DG_DPP_ZERO
      : "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6])
      : "v"(v));
}
__device__ __forceinline__ uint32_t shl7(uint32_t v) {
  uint32_t r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:7 " DG_DPP_ZERO : "=&v"(r) : "v"(v));
  return r;
}
// a + shl7(b)
__device__ __forceinline__ uint32_t add_shl7(uint32_t a, uint32_t b) {
  uint32_t r;
  asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 row_shl:7 " DG_DPP_ZERO : "=&v"(r) : "v"(b), "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t three_piece(uint64_t c) {
  const uint32_t p0 = (uint32_t)c & MASK;
  const uint32_t p1 = (uint32_t)(c >> W) & MASK;
  const uint32_t p2 = (uint32_t)(c >> (2 * W));
  return add_pieces(p0, p1, p2);
}
struct E14 {
  uint32_t lo, hi;
};
template <class P, bool EXT>
struct K14 {
  using L = LawK<EXT>;
  using T = Tab<P, EXT>;
  uint32_t pl[H], ph[H], pp[H], sublo[4], subhi[4], neglo, neghi, jplo[L::ZERO], jphi[L::ZERO], lane6;
  unsigned l16, row;
  __device__ __forceinline__ void init() {
    const unsigned lane = __lane_id();
    l16 = lane & 15u;
    row = lane >> 4;
#pragma unroll
    for (int i = 0; i < H; i++) {
      pl[i] = T::v.t[ROW_PL + i][l16];
      ph[i] = T::v.t[ROW_PH + i][l16];
      pp[i] = T::v.t[ROW_PP + i][l16];
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
      sublo[s] = T::v.t[ROW_SUB + 2 * s][l16];
      subhi[s] = T::v.t[ROW_SUB + 2 * s + 1][l16];
    }
#pragma unroll
    for (int j = 0; j < L::ZERO; j++) {
      jplo[j] = T::v.t[ROW_JP + 2 * j][l16];
      jphi[j] = T::v.t[ROW_JP + 2 * j + 1][l16];
    }
    neglo = T::v.t[ROW_NEG][l16];
    neghi = T::v.t[ROW_NEG + 1][l16];
    lane6 = l16 == 6 ? 0xFFFFFFFFu : 0u;
  }
  __device__ __forceinline__ E14 one() const { return {T::v.t[ROW_ONE][l16], T::v.t[ROW_ONE + 1][l16]}; }
};
// one carry pass over both registers: the carry out of limb 6 (lane 7 of lo) goes to lane 0 of hi
__device__ __forceinline__ E14 renorm(E14 v, unsigned l16) {
  const uint32_t lo2 = add_shr1(v.lo & MASK, v.lo >> W);
  const uint32_t hi2 = add_shl7(add_shr1(v.hi & MASK, v.hi >> W), lo2);
  return {l16 < (unsigned)H ? lo2 : 0u, hi2};
}
__device__ __forceinline__ E14 full_norm(E14 v, unsigned l16) {
  while (__builtin_amdgcn_ballot_w64((v.lo | v.hi) > MASK)) v = renorm(v, l16);
  return v;
}
struct Cols14 {
  uint64_t c0, c1, c2;
};
// one digit of the reduction: cl's low seven columns are cleared (mod B) and carried into cm
template <class KT>
__device__ __forceinline__ void digit(const KT& k, uint64_t cl, uint64_t& cm) {
  const uint32_t t = three_piece(cl);
  uint32_t tb[H];
  bcast7(tb, t);
  uint64_t mc = 0;
#pragma unroll
  for (int i = 0; i < H; i++) mc += (uint64_t)tb[i] * k.pp[i];
  const uint32_t m = three_piece(mc);
  uint32_t mb[H];
  bcast7(mb, m);
#pragma unroll
  for (int i = 0; i < H; i++) {
    cl += (uint64_t)mb[i] * k.pl[i];
    cm += (uint64_t)mb[i] * k.ph[i];
  }
  uint32_t L = three_piece(cl);
  const uint32_t e = ((L + 2u) >> W) & k.lane6;
  L = add_shr1(L, e);
  cm += (uint64_t)shl7(L);
}
template <class KT>
__device__ __forceinline__ E14 reduce(const KT& k, Cols14 c) {
  digit(k, c.c0, c.c1);
  digit(k, c.c1, c.c2);
  const uint32_t res = three_piece(c.c2);
  const E14 r = {k.l16 < (unsigned)H ? res : 0u, shl7(res)};
  return renorm(r, k.l16);
}
__device__ __forceinline__ void cols_mad(Cols14& c, const uint32_t (&al)[H], const uint32_t (&ah)[H], const uint32_t (&bl)[H],
                                         const uint32_t (&bh)[H]) {
#pragma unroll
  for (int i = 0; i < H; i++) {
    c.c0 += (uint64_t)al[i] * bl[i];
    c.c1 += (uint64_t)al[i] * bh[i];
    c.c1 += (uint64_t)ah[i] * bl[i];
    c.c2 += (uint64_t)ah[i] * bh[i];
  }
}
template <class KT>
__device__ __forceinline__ E14 mul(const KT& k, E14 a, E14 b) {
  uint32_t al[H], ah[H], bl[H], bh[H];
  bcast7(al, a.lo);
  bcast7(ah, a.hi);
  shifts7(bl, b.lo);
  shifts7(bh, b.hi);
  Cols14 c = {0, 0, 0};
  cols_mad(c, al, ah, bl, bh);
  return reduce(k, c);
}
}  // namespace l14
template <class P_>
struct Fq14 {
  using P = P_;
  static constexpr bool EXT = false;
  static constexpr int WORDS = l14::N;
  using E = l14::E14;
  using KT = l14::K14<P, false>;
  static constexpr int H = l14::H;
  static __device__ __forceinline__ E zero() { return {0u, 0u}; }
  static __device__ __forceinline__ E one(const KT& k) { return k.one(); }
  static __device__ __forceinline__ E mul(const KT& k, E a, E b) { return l14::mul(k, a, b); }
  static __device__ __forceinline__ E dbl(const KT& k, E a) { return l14::renorm({a.lo << 1, a.hi << 1}, k.l16); }
  static __device__ __forceinline__ E tpl(const KT& k, E a) { return l14::renorm({a.lo + (a.lo << 1), a.hi + (a.hi << 1)}, k.l16); }
  static __device__ __forceinline__ E dbl_raw(E a) { return {a.lo << 1, a.hi << 1}; }
  static __device__ __forceinline__ E add_raw(E a, E b) { return {a.lo + b.lo, a.hi + b.hi}; }
  template <int I>
  static __device__ __forceinline__ E sub(const KT& k, E a, E b) {
    return l14::renorm({a.lo + (k.sublo[I] - b.lo), a.hi + (k.subhi[I] - b.hi)}, k.l16);
  }
  static __device__ __forceinline__ E sel(bool c, E a, E b) { return {c ? a.lo : b.lo, c ? a.hi : b.hi}; }
  static __device__ __forceinline__ bool is_zero(const KT& k, E v) {
    v = l14::full_norm(v, k.l16);
    bool z = false;
#pragma unroll
    for (int j = 0; j < KT::L::ZERO; j++)
      z = z || (__builtin_amdgcn_ballot_w64(v.lo != k.jplo[j] || v.hi != k.jphi[j]) & 0xFFFFull) == 0;
    return z;
  }
  static __device__ __forceinline__ void swap16(E r, E& even, E& odd) {
    const auto s = __builtin_amdgcn_permlane16_swap(r.lo, r.lo, false, false);
    const auto t = __builtin_amdgcn_permlane16_swap(r.hi, r.hi, false, false);
    even = {s[0], t[0]};
    odd = {s[1], t[1]};
  }
  static __device__ __forceinline__ void rows_to_all(E r, E& r0, E& r1, E& r2, E& r3) {
    rows_to_all32(r.lo, r0.lo, r1.lo, r2.lo, r3.lo);
    rows_to_all32(r.hi, r0.hi, r1.hi, r2.hi, r3.hi);
  }
  static __device__ __forceinline__ bool any_nonzero(E v) { return __builtin_amdgcn_ballot_w64((v.lo | v.hi) != 0u) != 0; }
  static __device__ __forceinline__ E load(const KT& k, const uint32_t* w) {
    const bool on = k.l16 < (unsigned)H;
    const unsigned i = on ? k.l16 : 0u;
    return {on ? w[i] : 0u, on ? w[H + i] : 0u};
  }
  static __device__ __forceinline__ void store(const KT& k, uint32_t* w, E v) {
    if (k.row == 0 && k.l16 < (unsigned)H) {
      w[k.l16] = v.lo;
      w[H + k.l16] = v.hi;
    }
  }
  template <int BS>
  static __device__ __forceinline__ E exit_norm(const KT& k, E v, bool reduce_it) {
    static_assert(BS >= 192, "a storage bound of ~2 p would need the conditional subtraction");
    if (reduce_it) v = l14::mul(k, v, k.one());
    return l14::full_norm(v, k.l16);
  }
  template <int BB>
  static __device__ __forceinline__ E from_regs(const KT& k, const Fe<P, BB, 1>& f) {
    uint32_t v0 = 0, v1 = 0;
#pragma unroll
    for (int i = 0; i < H; i++) {
      v0 = k.l16 == (unsigned)i ? f.l[i] : v0;
      v1 = k.l16 == (unsigned)i ? f.l[H + i] : v1;
    }
    return {v0, v1};
  }
  template <int BB>
  static __device__ __forceinline__ Fe<P, BB, 1> to_regs(E v) {
    uint32_t o0[H], o1[H];
    l14::bcast7(o0, v.lo);
    l14::bcast7(o1, v.hi);
    Fe<P, BB, 1> f;
#pragma unroll
    for (int i = 0; i < H; i++) {
      f.l[i] = o0[i];
      f.l[H + i] = o1[i];
    }
    return f;
  }
};
struct E14x2 {
  l14::E14 c0, c1;
};
// quadratic extension u^2 = -BETA over a fourteen-limb field (G2 of BLS12-381: BETA = 1, of BLS12-377: BETA = 5), both
// components of a product on the same row like Fq9x2: c0 = a0 b0 + a1 (NEG BETA p - BETA b1), c1 = a0 b1 + a1 b0
template <class P_>
struct Fq14x2 {
  using P = P_;
  static constexpr bool EXT = true;
  static constexpr int BETA = Fq2Beta<P>::value;
  static constexpr int WORDS = 2 * l14::N;
  static constexpr int H = l14::H;
  using B1 = l14::E14;
  using E = E14x2;
  using KT = l14::K14<P, true>;
  static __device__ __forceinline__ E zero() { return {{0u, 0u}, {0u, 0u}}; }
  static __device__ __forceinline__ E one(const KT& k) { return {k.one(), {0u, 0u}}; }
  static __device__ __forceinline__ B1 rn(const KT& k, B1 v) { return l14::renorm(v, k.l16); }
  static __device__ __forceinline__ E mul(const KT& k, E a, E b) {
    const B1 nb1 = rn(k, {k.neglo - b.c1.lo * (uint32_t)BETA, k.neghi - b.c1.hi * (uint32_t)BETA});
    uint32_t a0l[H], a0h[H], a1l[H], a1h[H], b0l[H], b0h[H], b1l[H], b1h[H], n1l[H], n1h[H];
    l14::bcast7(a0l, a.c0.lo);
    l14::bcast7(a0h, a.c0.hi);
    l14::bcast7(a1l, a.c1.lo);
    l14::bcast7(a1h, a.c1.hi);
    l14::shifts7(b0l, b.c0.lo);
    l14::shifts7(b0h, b.c0.hi);
    l14::shifts7(b1l, b.c1.lo);
    l14::shifts7(b1h, b.c1.hi);
    l14::shifts7(n1l, nb1.lo);
    l14::shifts7(n1h, nb1.hi);
    l14::Cols14 x = {0, 0, 0}, y = {0, 0, 0};
    l14::cols_mad(x, a0l, a0h, b0l, b0h);
    l14::cols_mad(x, a1l, a1h, n1l, n1h);
    l14::cols_mad(y, a0l, a0h, b1l, b1h);
    l14::cols_mad(y, a1l, a1h, b0l, b0h);
    return {l14::reduce(k, x), l14::reduce(k, y)};
  }
  static __device__ __forceinline__ E dbl(const KT& k, E a) {
    return {rn(k, {a.c0.lo << 1, a.c0.hi << 1}), rn(k, {a.c1.lo << 1, a.c1.hi << 1})};
  }
  static __device__ __forceinline__ E tpl(const KT& k, E a) {
    return {rn(k, {a.c0.lo * 3u, a.c0.hi * 3u}), rn(k, {a.c1.lo * 3u, a.c1.hi * 3u})};
  }
  static __device__ __forceinline__ E dbl_raw(E a) { return {{a.c0.lo << 1, a.c0.hi << 1}, {a.c1.lo << 1, a.c1.hi << 1}}; }
  static __device__ __forceinline__ E add_raw(E a, E b) {
    return {{a.c0.lo + b.c0.lo, a.c0.hi + b.c0.hi}, {a.c1.lo + b.c1.lo, a.c1.hi + b.c1.hi}};
  }
  template <int I>
  static __device__ __forceinline__ E sub(const KT& k, E a, E b) {
    return {rn(k, {a.c0.lo + (k.sublo[I] - b.c0.lo), a.c0.hi + (k.subhi[I] - b.c0.hi)}),
            rn(k, {a.c1.lo + (k.sublo[I] - b.c1.lo), a.c1.hi + (k.subhi[I] - b.c1.hi)})};
  }
  static __device__ __forceinline__ B1 sel1(bool c, B1 a, B1 b) { return {c ? a.lo : b.lo, c ? a.hi : b.hi}; }
  static __device__ __forceinline__ E sel(bool c, E a, E b) { return {sel1(c, a.c0, b.c0), sel1(c, a.c1, b.c1)}; }
  static __device__ __forceinline__ bool is_zero1(const KT& k, B1 v) {
    v = l14::full_norm(v, k.l16);
    bool z = false;
#pragma unroll
    for (int j = 0; j < KT::L::ZERO; j++)
      z = z || (__builtin_amdgcn_ballot_w64(v.lo != k.jplo[j] || v.hi != k.jphi[j]) & 0xFFFFull) == 0;
    return z;
  }
  static __device__ __forceinline__ bool is_zero(const KT& k, E v) { return is_zero1(k, v.c0) && is_zero1(k, v.c1); }
  static __device__ __forceinline__ void swap16(E r, E& even, E& odd) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(r.c0.lo, r.c0.lo, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(r.c0.hi, r.c0.hi, false, false);
    const auto s2 = __builtin_amdgcn_permlane16_swap(r.c1.lo, r.c1.lo, false, false);
    const auto s3 = __builtin_amdgcn_permlane16_swap(r.c1.hi, r.c1.hi, false, false);
    even = {{s0[0], s1[0]}, {s2[0], s3[0]}};
    odd = {{s0[1], s1[1]}, {s2[1], s3[1]}};
  }
  static __device__ __forceinline__ void rows_to_all(E r, E& r0, E& r1, E& r2, E& r3) {
    rows_to_all32(r.c0.lo, r0.c0.lo, r1.c0.lo, r2.c0.lo, r3.c0.lo);
    rows_to_all32(r.c0.hi, r0.c0.hi, r1.c0.hi, r2.c0.hi, r3.c0.hi);
    rows_to_all32(r.c1.lo, r0.c1.lo, r1.c1.lo, r2.c1.lo, r3.c1.lo);
    rows_to_all32(r.c1.hi, r0.c1.hi, r1.c1.hi, r2.c1.hi, r3.c1.hi);
  }
  static __device__ __forceinline__ bool any_nonzero(E v) {
    return __builtin_amdgcn_ballot_w64((v.c0.lo | v.c0.hi | v.c1.lo | v.c1.hi) != 0u) != 0;
  }
  static __device__ __forceinline__ E load(const KT& k, const uint32_t* w) {
    const bool on = k.l16 < (unsigned)H;
    const unsigned i = on ? k.l16 : 0u;
    return {{on ? w[i] : 0u, on ? w[H + i] : 0u}, {on ? w[2 * H + i] : 0u, on ? w[3 * H + i] : 0u}};
  }
  static __device__ __forceinline__ void store(const KT& k, uint32_t* w, E v) {
    if (k.row == 0 && k.l16 < (unsigned)H) {
      w[k.l16] = v.c0.lo;
      w[H + k.l16] = v.c0.hi;
      w[2 * H + k.l16] = v.c1.lo;
      w[3 * H + k.l16] = v.c1.hi;
    }
  }
  template <int BS>
  static __device__ __forceinline__ E exit_norm(const KT& k, E v, bool reduce_it) {
    static_assert(BS >= 192, "a storage bound of ~2 p would need the conditional subtraction");
    if (reduce_it) v = {l14::mul(k, v.c0, k.one()), l14::mul(k, v.c1, k.one())};
    return {l14::full_norm(v.c0, k.l16), l14::full_norm(v.c1, k.l16)};
  }
  template <int BB>
  static __device__ __forceinline__ E from_regs(const KT& k, const Fe2<P, BB, 1>& f) {
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
    for (int i = 0; i < H; i++) {
      v0 = k.l16 == (unsigned)i ? f.c0.l[i] : v0;
      v1 = k.l16 == (unsigned)i ? f.c0.l[H + i] : v1;
      v2 = k.l16 == (unsigned)i ? f.c1.l[i] : v2;
      v3 = k.l16 == (unsigned)i ? f.c1.l[H + i] : v3;
    }
    return {{v0, v1}, {v2, v3}};
  }
  template <int BB>
  static __device__ __forceinline__ Fe2<P, BB, 1> to_regs(E v) {
    uint32_t o0[H], o1[H], o2[H], o3[H];
    l14::bcast7(o0, v.c0.lo);
    l14::bcast7(o1, v.c0.hi);
    l14::bcast7(o2, v.c1.lo);
    l14::bcast7(o3, v.c1.hi);
    Fe2<P, BB, 1> f;
#pragma unroll
    for (int i = 0; i < H; i++) {
      f.c0.l[i] = o0[i];
      f.c0.l[H + i] = o1[i];
      f.c1.l[i] = o2[i];
      f.c1.l[H + i] = o3[i];
    }
    return f;
  }
};
template <class F> struct PolicyOf;
template <class P> struct PolicyOf<Fp<P>> { using type = std::conditional_t<RR<P>::N == N, Fq9<P>, Fq14<P>>; };
template <class P> struct PolicyOf<Fp2<Fp<P>>> { using type = std::conditional_t<RR<P>::N == N, Fq9x2<P>, Fq14x2<P>>; };
template <class F> using Ops = typename PolicyOf<F>::type;

// which coordinate fields run their chains in this form: nine-limb fields (BN254: Fq, and Fq2 with u^2 = -1)
template <class F>
constexpr bool enabled() {
#ifdef DG16_NO_LANE_CHAINS        // (A/B switch: the one-product-per-lane chains of msm_impl.h everywhere)
  return false;
#else
  using P = typename FieldOf<F>::Params;
  if constexpr (RR<P>::N == l14::N && RR<P>::W == l14::W) return true;                    // BLS12 G1 and G2
  else if constexpr (RR<P>::N != N || RR<P>::W != W) return false;
  else if constexpr (FieldOf<F>::EXT) return Fq2Beta<P>::value == 1;                      // BN254 G2
  else return true;                                                                        // BN254 G1
#endif
}

// ---- XYZZ points: coordinates as elements, the same in every row -----------------------------------------------------
template <class FO>
struct Pt {
  typename FO::E x, y, zz, zzz;
  bool inf;               // uniform
};
template <class FO>
__device__ __forceinline__ Pt<FO> inf_pt(const typename FO::KT& k) {
  return {FO::one(k), FO::one(k), FO::zero(), FO::zero(), true};
}
template <class FO>
__device__ __forceinline__ typename FO::E by_row(unsigned row, typename FO::E v0, typename FO::E v1, typename FO::E v2,
                                                 typename FO::E v3) {
  return FO::sel(row == 0, v0, FO::sel(row == 1, v1, FO::sel(row == 2, v2, v3)));
}

// 2 p                                                                                         (dbl-2008-s-1, a = 0)
// (value bounds of every temporary: tools/lane_bounds.py)
template <class FO>
__device__ __forceinline__ Pt<FO> dbl_pt(const typename FO::KT& k, const Pt<FO>& p) {
  using E = typename FO::E;
  if (p.inf) return p;
  const unsigned row = k.row;
  const E u = FO::dbl(k, p.y);
  // level 1: rows 0, 2: v = u^2 | rows 1, 3: xx = x^2
  const E a1 = FO::sel((row & 1u) != 0, p.x, u);
  const E r1 = FO::mul(k, a1, a1);
  E v, xx;
  FO::swap16(r1, v, xx);
  const E m = FO::tpl(k, xx);
  // level 2: w = u v | s = x v | m^2 | zz3 = zz v
  const E a2 = by_row<FO>(row, u, p.x, m, p.zz);
  const E b2 = FO::sel(row == 2, m, v);
  const E r2 = FO::mul(k, a2, b2);
  E w, s, mm, zz3;
  FO::rows_to_all(r2, w, s, mm, zz3);
  const E x3 = FO::template sub<1>(k, mm, FO::dbl_raw(s));
  const E sx = FO::template sub<3>(k, s, x3);
  // level 3: (s - x3) m | w y | zzz3 = zzz w
  const E a3 = by_row<FO>(row, sx, w, p.zzz, p.zzz);
  const E b3 = by_row<FO>(row, m, p.y, w, w);
  const E r3 = FO::mul(k, a3, b3);
  E t0, t1, zzz3, unused;
  FO::rows_to_all(r3, t0, t1, zzz3, unused);
  return {x3, FO::template sub<0>(k, t0, t1), zz3, zzz3, false};
}

// p + o, complete                                                                                       (add-2008-s)
template <class FO>
__device__ __forceinline__ Pt<FO> add_pt(const typename FO::KT& k, const Pt<FO>& p, const Pt<FO>& o) {
  using E = typename FO::E;
  if (o.inf) return p;
  if (p.inf) return o;
  const unsigned row = k.row;
  // level 1: u1 = x1 zz2 | u2 = x2 zz1 | s1 = y1 zzz2 | s2 = y2 zzz1
  const E r1 = FO::mul(k, by_row<FO>(row, p.x, o.x, p.y, o.y), by_row<FO>(row, o.zz, p.zz, o.zzz, p.zzz));
  E u1, u2, s1, s2;
  FO::rows_to_all(r1, u1, u2, s1, s2);
  const E pd = FO::template sub<0>(k, u2, u1), rd = FO::template sub<0>(k, s2, s1);
  if (FO::is_zero(k, pd)) {
    if (FO::is_zero(k, rd)) return dbl_pt<FO>(k, p);
    return inf_pt<FO>(k);
  }
  // level 2: pp = p^2 | rr = r^2 | zz1 zz2 | zzz1 zzz2
  const E r2 = FO::mul(k, by_row<FO>(row, pd, rd, p.zz, p.zzz), by_row<FO>(row, pd, rd, o.zz, o.zzz));
  E pp, rr, zzp, zzzp;
  FO::rows_to_all(r2, pp, rr, zzp, zzzp);
  // level 3: ppp = p pp | q = u1 pp | zz3 = (zz1 zz2) pp
  const E r3 = FO::mul(k, by_row<FO>(row, pd, u1, zzp, zzp), pp);
  E ppp, q, zz3, unused;
  FO::rows_to_all(r3, ppp, q, zz3, unused);
  const E x3 = FO::template sub<2>(k, rr, FO::add_raw(ppp, FO::dbl_raw(q)));
  const E qx = FO::template sub<3>(k, q, x3);
  // level 4: (q - x3) r | s1 ppp | zzz3 = (zzz1 zzz2) ppp
  const E r4 = FO::mul(k, by_row<FO>(row, qx, s1, zzzp, zzzp), FO::sel(row == 0, rd, ppp));
  E t0, t1, zzz3;
  FO::rows_to_all(r4, t0, t1, zzz3, unused);
  return {x3, FO::template sub<0>(k, t0, t1), zz3, zzz3, false};
}
// -p (y -> K2 p - y)
template <class FO>
__device__ __forceinline__ Pt<FO> neg_pt(const typename FO::KT& k, const Pt<FO>& p) {
  return {p.x, FO::template sub<2>(k, FO::zero(), p.y), p.zz, p.zzz, p.inf};
}

// XYZZ29 (registers, uniform across the wave) <-> Pt
template <class F>
__device__ __forceinline__ Pt<Ops<F>> to_pt(const typename Ops<F>::KT& k, const XYZZ29<F>& p) {
  using FO = Ops<F>;
  constexpr int BS = XYZZ29<F>::BS;
  return {FO::template from_regs<BS>(k, p.x), FO::template from_regs<BS>(k, p.y), FO::template from_regs<BS>(k, p.zz),
          FO::template from_regs<BS>(k, p.zzz), p.is_inf()};
}
// coordinates under the storage bound with normalised limbs: x always through a product with R mod p (x3 < 13.2 p); y, zz,
// zzz as they are when the bound is 7 p (y3 < 7 p, zz3, zzz3 < 3.3 p: tools/lane_bounds.py), reduced when it is ~2 p
template <class F>
__device__ __forceinline__ Pt<Ops<F>> exit_pt(const typename Ops<F>::KT& k, const Pt<Ops<F>>& p) {
  using FO = Ops<F>;
  constexpr int BS = XYZZ29<F>::BS;
  constexpr bool TIGHT = BS < 448;
  static_assert(BS >= 448 || BS >= 130, "storage bound");
  if (p.inf) return p;
  return {FO::template exit_norm<BS>(k, p.x, true), FO::template exit_norm<BS>(k, p.y, TIGHT),
          FO::template exit_norm<BS>(k, p.zz, TIGHT), FO::template exit_norm<BS>(k, p.zzz, TIGHT), false};
}
template <class F>
__device__ __forceinline__ XYZZ29<F> from_pt(const typename Ops<F>::KT& k, const Pt<Ops<F>>& p_) {
  using FO = Ops<F>;
  constexpr int BS = XYZZ29<F>::BS;
  if (p_.inf) return XYZZ29<F>::inf();
  const Pt<FO> p = exit_pt<F>(k, p_);
  return {FO::template to_regs<BS>(p.x), FO::template to_regs<BS>(p.y), FO::template to_regs<BS>(p.zz),
          FO::template to_regs<BS>(p.zzz)};
}

// ---- lane form in memory: the words of an XYZZ29<F> (WORDS per coordinate), read and written by the lanes that own the
// limbs.  What the chains store for THEMSELVES may hold tight limbs and coordinates up to ~13 p (raw); what other code
// reads as an XYZZ29 goes through store_pt (normalised limbs, below the storage bound).  Identity: ZZ all zero. ----------
template <class F>
__device__ __forceinline__ Pt<Ops<F>> load_pt(const typename Ops<F>::KT& k, const XYZZ29<F>* src) {
  using FO = Ops<F>;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src);
  Pt<FO> p;
  p.x = FO::load(k, w);
  p.y = FO::load(k, w + FO::WORDS);
  p.zz = FO::load(k, w + 2 * FO::WORDS);
  p.zzz = FO::load(k, w + 3 * FO::WORDS);
  p.inf = !FO::any_nonzero(p.zz);
  return p;
}
// the same in two halves, for a loop that fetches one entry ahead: the identity flag needs the data (a ballot), so taking
// it at once would put the load's latency back on the chain
template <class F>
__device__ __forceinline__ Pt<Ops<F>> load_pt_words(const typename Ops<F>::KT& k, const XYZZ29<F>* src) {
  using FO = Ops<F>;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src);
  return {FO::load(k, w), FO::load(k, w + FO::WORDS), FO::load(k, w + 2 * FO::WORDS), FO::load(k, w + 3 * FO::WORDS), false};
}
template <class FO>
__device__ __forceinline__ Pt<FO> with_inf_flag(Pt<FO> p) {
  p.inf = !FO::any_nonzero(p.zz);
  return p;
}
template <class F>
__device__ __forceinline__ void store_pt_raw(const typename Ops<F>::KT& k, XYZZ29<F>* dst, const Pt<Ops<F>>& p) {
  using FO = Ops<F>;
  uint32_t* w = reinterpret_cast<uint32_t*>(dst);
  FO::store(k, w, p.x);
  FO::store(k, w + FO::WORDS, p.y);
  FO::store(k, w + 2 * FO::WORDS, p.inf ? FO::zero() : p.zz);
  FO::store(k, w + 3 * FO::WORDS, p.inf ? FO::zero() : p.zzz);
}
template <class F>
__device__ __forceinline__ void store_pt(const typename Ops<F>::KT& k, XYZZ29<F>* dst, const Pt<Ops<F>>& p) {
  store_pt_raw<F>(k, dst, exit_pt<F>(k, p));
}

// ---- inversion and the affine form of a chain's result -----------------------------------------------------------------
// x^(p - 2) with 4-bit windows: 15 products for the table, then 4 squarings + at most one product per nibble -- ~1.25
// products per bit at 0.24 us (nine limbs) / 0.46 us (fourteen) instead of a lone lane's 32-bit-limb Fermat chain (~1 ms for
// a 377-bit field: what d_msm paid TWICE per round, at the end of every party's MSM and of the king's combination:
// profiles/r6bb_timeline_dmsm.md).  E: an element of a BASE field; mul / sel / one as the policy's.
template <class P, class E, class Mul, class Sel>
__device__ __forceinline__ E pow_p_minus_2(E x, E one, Mul mul, Sel sel) {
  E tab[16];
  tab[0] = one;
  tab[1] = x;
#pragma unroll
  for (int j = 2; j < 16; j++) tab[j] = mul(tab[j - 1], x);
  E acc = one;
  bool started = false;
#pragma unroll 1
  for (int i = 8 * P::NL - 1; i >= 0; i--) {
    const unsigned nib = (PM2<P>::v.v[i >> 3] >> (4 * (i & 7))) & 15u;
    if (started) {
      acc = mul(acc, acc);
      acc = mul(acc, acc);
      acc = mul(acc, acc);
      acc = mul(acc, acc);
    }
    if (nib == 0) continue;
    E t = tab[1];
#pragma unroll
    for (int j = 2; j < 16; j++) t = sel(nib == (unsigned)j, tab[j], t);
    acc = started ? mul(acc, t) : t;
    started = true;
  }
  return acc;
}
template <class P>
__device__ __forceinline__ typename Fq9<P>::E inv(const typename Fq9<P>::KT& k, Fq9<P>, uint32_t x) {
  return pow_p_minus_2<P, uint32_t>(
      x, k.one(), [&](uint32_t a, uint32_t b) { return lane29::mul(k, a, b); },
      [](bool c, uint32_t a, uint32_t b) { return c ? a : b; });
}
template <class P>
__device__ __forceinline__ typename Fq14<P>::E inv(const typename Fq14<P>::KT& k, Fq14<P>, l14::E14 x) {
  return pow_p_minus_2<P, l14::E14>(
      x, k.one(), [&](l14::E14 a, l14::E14 b) { return l14::mul(k, a, b); },
      [](bool c, l14::E14 a, l14::E14 b) { return l14::E14{c ? a.lo : b.lo, c ? a.hi : b.hi}; });
}
// 1 / (a + b u) = (a - b u) / (a^2 + b^2) over u^2 = -1
template <class P>
__device__ __forceinline__ E2 inv(const typename Fq9x2<P>::KT& k, Fq9x2<P>, E2 x) {
  const uint32_t n = renorm(lane29::mul(k, x.c0, x.c0) + lane29::mul(k, x.c1, x.c1));
  const uint32_t ni = pow_p_minus_2<P, uint32_t>(
      n, k.one(), [&](uint32_t a, uint32_t b) { return lane29::mul(k, a, b); },
      [](bool c, uint32_t a, uint32_t b) { return c ? a : b; });
  const uint32_t c1 = lane29::mul(k, x.c1, ni);                         // < 2.1 p: sub<0> (K = 4) serves
  return {lane29::mul(k, x.c0, ni), renorm(k.sub[0] - c1)};
}
// 1 / (a + b u) = (a - b u) / (a^2 + BETA b^2) over u^2 = -BETA
template <class P>
__device__ __forceinline__ E14x2 inv(const typename Fq14x2<P>::KT& k, Fq14x2<P>, E14x2 x) {
  constexpr uint32_t BETA = (uint32_t)Fq2Beta<P>::value;
  const l14::E14 aa = l14::mul(k, x.c0, x.c0), bb = l14::mul(k, x.c1, x.c1);
  const l14::E14 n = l14::renorm({aa.lo + bb.lo * BETA, aa.hi + bb.hi * BETA}, k.l16);
  const l14::E14 ni = pow_p_minus_2<P, l14::E14>(
      n, k.one(), [&](l14::E14 a, l14::E14 b) { return l14::mul(k, a, b); },
      [](bool c, l14::E14 a, l14::E14 b) { return l14::E14{c ? a.lo : b.lo, c ? a.hi : b.hi}; });
  const l14::E14 c1 = l14::mul(k, x.c1, ni);
  return {l14::mul(k, x.c0, ni), l14::renorm({k.sublo[0] - c1.lo, k.subhi[0] - c1.hi}, k.l16)};
}
// (X / ZZ, Y / ZZZ) in the arkworks form of the C ABI; the identity is (0, 0)
template <class F>
__device__ __forceinline__ Affine<F> to_affine(const typename Ops<F>::KT& k, const Pt<Ops<F>>& p) {
  using FO = Ops<F>;
  constexpr int BS = XYZZ29<F>::BS;
  if (p.inf) return Affine<F>::inf();
  const auto ti = inv(k, FO{}, FO::mul(k, p.zz, p.zzz));
  const auto xa = FO::mul(k, p.x, FO::mul(k, ti, p.zzz));
  const auto ya = FO::mul(k, p.y, FO::mul(k, ti, p.zz));
  return {FieldOf<F>::to32(FO::template to_regs<BS>(FO::template exit_norm<BS>(k, xa, true))),
          FieldOf<F>::to32(FO::template to_regs<BS>(FO::template exit_norm<BS>(k, ya, true)))};
}
#endif  // __HIPCC__

}  // namespace lane29
}  // namespace dg16
