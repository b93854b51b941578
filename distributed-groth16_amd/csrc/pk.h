// Resident proving key record shared by the prover translation units.
#pragma once
#include "ctx.h"

namespace dg16 {

struct PkDev {
  int curve = 0;
  size_t num_vars = 0, num_inputs = 0, m = 0;
  unsigned shard = 0, nshards = 1;   // this key holds slice `shard` of every MSM range (multi-GPU)
  bool h_cyclic = false;             // h bases are h_query[shard + nshards * j] (DG16_F_H_CYCLIC): h_lo = 0, h_hi = m / nshards
  size_t ab_lo = 0, ab_hi = 0, l_lo = 0, l_hi = 0, h_lo = 0, h_hi = 0;
  // all device pointers
  void* a_q = nullptr;    // a_query[1..][ab_lo..ab_hi) ++ [delta_g1, 0, 0]           (G1)
  void* b1_q = nullptr;   // b_g1_query[1..][ab_lo..ab_hi) ++ [0, delta_g1, 0]        (G1)
  void* b2_q = nullptr;   // b_g2_query[1..][ab_lo..ab_hi) ++ [0, delta_g2, 0]        (G2)
  void* l_q = nullptr;    // l_query, index-aligned with the slice of w[1..] ++ [0, 0, delta_g1]   (G1)
  void* h_q = nullptr;    // h_query[h_lo..h_hi)                           (G1)
  void* fixed = nullptr;  // alpha_g1, a_query[0], beta_g1, b_g1_query[0] (G1 affine) | beta_g2, b_g2_query[0] (G2 affine)
  // The five base arrays above are TABLES of window multiples, T[w*n + i] = 2^(c*w) * P_i (HBM is 288 GB:
  // 13 rows x 64 B x 2^20 = 0.8 GB per G1 query), so a proof's MSMs have one bucket set and no Horner tail.
  // Row layouts: a_q / b1_q / b2_q / l_q = slice ++ three delta slots (A: [delta_g1, 0, 0]; B1: [0, delta_g1, 0];
  // B: [0, delta_g2, 0]; L: [0, 0, delta_g1]) so the FOUR MSMs share ONE scalar vector w[1..] ++ [r, s, -rs] and one
  // digit sort; l_q holds the identity where the slice position is a public input (l_lo, l_hi: the l_query range).
  unsigned c_ab = 0, c_l = 0, c_h = 0;   // window bits the tables were built for
  unsigned stride = 1;                   // the tables keep every stride-th window's row (HBM budget; 1 = all rows)
  size_t table_bytes = 0;                // HBM held by the five tables
};

}  // namespace dg16

struct dg16_pk {
  dg16_ctx* ctx;
  dg16::PkDev d;
};

namespace dg16 {
inline void pk_free(PkDev& d) {
  for (void* p : {d.a_q, d.b1_q, d.b2_q, d.l_q, d.h_q, d.fixed})
    if (p) hipFree(p);
  d = PkDev{};
}
}  // namespace dg16
