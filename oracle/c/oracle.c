/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the Groth16 hot path (MSM + NTT + h-poly).
 *
 * This file is the checker and the `cpu_baseline` ("port") leg of bench.py.  It is never linked
 * into, loaded by, or called from the product library (distributed-groth16_amd/csrc).
 *
 * Parity status: the reference (/root/reference, Rust on arkworks 0.4) cannot be built here and
 * holds NO golden MSM/NTT output vectors; the arithmetic lives in un-vendored
 * crates (ark-ff/ark-ec/ark-poly 0.4.x).  This restatement is pinned by
 *   - the in-tree known answers of SURVEY.md section 0 (tests/test_oracle_kats.py),
 *   - an independent pure-Python big-int restatement (oracle/pyref) that reproduces the
 *     reference's own relational tests (d_fft == domain.fft, d_msm == msm, mpc proof == proof),
 *   - definitional checks (naive double-and-add MSM, O(n^2) DFT, known-trapdoor Groth16).
 *   - the reference's one complete numeric vector: the snarkjs proof triple of fixtures/million is
 *     accepted by the pairing verifier of oracle/pyref (whose field / curve arithmetic this file is
 *     cross-checked against point by point), and so are the proofs this oracle's prover produces.
 * For numeric MSM/NTT outputs taken alone: "parity unpinned by golden vectors" (none exist upstream).
 *
 * Build: make -C oracle/c   ->  oracle/c/liboracle.so   (gcc -O3 -march=native -fopenmp)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#include "consts_gen.h"

/* ---- field instances ---------------------------------------------------------------------- */
#define FP bn254_fq_
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bn254_fr_
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bls12_381_fq_
#define NL 6
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bls12_381_fr_
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bls12_377_fq_
#define NL 6
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bls12_377_fr_
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL

#define F2 bn254_fq2_
#define FB bn254_fq_
#define F2_BETA 1
#include "fp2_tmpl.h"
#undef F2
#undef FB
#undef F2_BETA
#define F2 bls12_381_fq2_
#define FB bls12_381_fq_
#define F2_BETA 1
#include "fp2_tmpl.h"
#undef F2
#undef FB
#undef F2_BETA
#define F2 bls12_377_fq2_
#define FB bls12_377_fq_
#define F2_BETA 5
#include "fp2_tmpl.h"
#undef F2
#undef FB
#undef F2_BETA

/* ---- group instances ---------------------------------------------------------------------- */
#define EC bn254_g1_
#define BF bn254_fq_
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC bn254_g2_
#define BF bn254_fq2_
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC bls12_381_g1_
#define BF bls12_381_fq_
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC bls12_381_g2_
#define BF bls12_381_fq2_
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC bls12_377_g1_
#define BF bls12_377_fq_
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC bls12_377_g2_
#define BF bls12_377_fq2_
#include "ec_tmpl.h"
#undef EC
#undef BF

/* ---- NTT instances (scalar fields) ---------------------------------------------------------- */
#define FR bn254_fr_
#include "ntt_tmpl.h"
#undef FR
#define FR bls12_381_fr_
#include "ntt_tmpl.h"
#undef FR
#define FR bls12_377_fr_
#include "ntt_tmpl.h"
#undef FR

static int bn254_fr_two_adicity(void) { return BN254_FR_TWO_ADICITY; }
static int bls12_381_fr_two_adicity(void) { return BLS12_381_FR_TWO_ADICITY; }
static int bls12_377_fr_two_adicity(void) { return BLS12_377_FR_TWO_ADICITY; }

enum { ORC_BN254 = 0, ORC_BLS12_381 = 1, ORC_BLS12_377 = 2 };
enum { ORC_OK = 0, ORC_BAD_ARG = 1 };

static unsigned scalar_bits(int curve) {
    return curve == ORC_BN254 ? 254 : curve == ORC_BLS12_381 ? 255 : 253;
}

int orc_num_threads(void) { return omp_get_max_threads(); }

/* SplitMix64, counter mode: the documented input generator of this repo */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* ---- dispatch helpers ------------------------------------------------------------------------ */
#define FIELD_DISPATCH(fid, CALL)                                 \
    switch (fid) {                                                \
    case 0: { CALL(bn254_fq_) } break;                            \
    case 1: { CALL(bls12_381_fq_) } break;                        \
    case 2: { CALL(bls12_377_fq_) } break;                        \
    case 16: { CALL(bn254_fr_) } break;                           \
    case 17: { CALL(bls12_381_fr_) } break;                       \
    case 18: { CALL(bls12_377_fr_) } break;                       \
    default: return ORC_BAD_ARG;                                  \
    }

/* field ids: curve (0..2) for Fq, 16 + curve for Fr.  ops: 0 add 1 sub 2 mul 3 sqr(a) 4 inv(a)
 * 5 to_mont(a) 6 from_mont(a) 7 neg(a).  All operands/results are n elements of limbs*8 bytes. */
int orc_field_op(int fid, int op, const void *a, const void *b, void *out, size_t n) {
#define DO_FIELD_OP(P)                                                              \
    const P##t *x = (const P##t *)a, *y = (const P##t *)b;                          \
    P##t *o = (P##t *)out;                                                          \
    for (size_t i = 0; i < n; i++) {                                                \
        switch (op) {                                                               \
        case 0: P##add(&o[i], &x[i], &y[i]); break;                                 \
        case 1: P##sub(&o[i], &x[i], &y[i]); break;                                 \
        case 2: P##mul(&o[i], &x[i], &y[i]); break;                                 \
        case 3: P##sqr(&o[i], &x[i]); break;                                        \
        case 4: P##inv(&o[i], &x[i]); break;                                        \
        case 5: P##to_mont(&o[i], &x[i]); break;                                    \
        case 6: P##from_mont(&o[i], &x[i]); break;                                  \
        case 7: P##neg(&o[i], &x[i]); break;                                        \
        default: return ORC_BAD_ARG;                                                \
        }                                                                           \
    }
    FIELD_DISPATCH(fid, DO_FIELD_OP)
    return ORC_OK;
}

/* uniform-ish field elements: 4 (or 6) SplitMix64 words reduced by Montgomery-multiplying with R2
 * (i.e. value = words * R mod p in Montgomery form == words mod p as an integer).  Output is in
 * Montgomery form if mont != 0, else canonical. */
int orc_rand_field(int fid, uint64_t seed, size_t n, int mont, void *out) {
#define DO_RAND(P)                                                                  \
    P##t *o = (P##t *)out;                                                          \
    const int nl = (int)(sizeof(P##t) / 8);                                         \
    for (size_t i = 0; i < n; i++) {                                                \
        P##t w, m;                                                                  \
        for (int k = 0; k < nl; k++) w.l[k] = splitmix64(seed * 0x100000001B3ULL + i * 8 + k); \
        w.l[nl - 1] >>= 3; /* < 2^(64nl-3) < a small multiple of p; one mul reduces */ \
        P##to_mont(&m, &w);        /* m = w mod p, Montgomery form */               \
        if (mont) o[i] = m; else P##from_mont(&o[i], &m);                           \
    }
    FIELD_DISPATCH(fid, DO_RAND)
    return ORC_OK;
}

/* ---- groups ---------------------------------------------------------------------------------- */
#define GROUP_DISPATCH(curve, group, CALL)                                          \
    switch ((curve) * 2 + ((group) - 1)) {                                          \
    case 0: { CALL(bn254_g1_, bn254_fq_, 0) } break;                                \
    case 1: { CALL(bn254_g2_, bn254_fq2_, 1) } break;                               \
    case 2: { CALL(bls12_381_g1_, bls12_381_fq_, 0) } break;                        \
    case 3: { CALL(bls12_381_g2_, bls12_381_fq2_, 1) } break;                       \
    case 4: { CALL(bls12_377_g1_, bls12_377_fq_, 0) } break;                        \
    case 5: { CALL(bls12_377_g2_, bls12_377_fq2_, 1) } break;                       \
    default: return ORC_BAD_ARG;                                                    \
    }

static void load_gen_bn254_g1(bn254_g1_aff_t *g) { memcpy(&g->x, bn254_g1_GX, 32); memcpy(&g->y, bn254_g1_GY, 32); }
static void load_gen_bls12_381_g1(bls12_381_g1_aff_t *g) { memcpy(&g->x, bls12_381_g1_GX, 48); memcpy(&g->y, bls12_381_g1_GY, 48); }
static void load_gen_bls12_377_g1(bls12_377_g1_aff_t *g) { memcpy(&g->x, bls12_377_g1_GX, 48); memcpy(&g->y, bls12_377_g1_GY, 48); }
static void load_gen_bn254_g2(bn254_g2_aff_t *g) {
    memcpy(&g->x.c0, bn254_g2_GX_C0, 32); memcpy(&g->x.c1, bn254_g2_GX_C1, 32);
    memcpy(&g->y.c0, bn254_g2_GY_C0, 32); memcpy(&g->y.c1, bn254_g2_GY_C1, 32);
}
static void load_gen_bls12_381_g2(bls12_381_g2_aff_t *g) {
    memcpy(&g->x.c0, bls12_381_g2_GX_C0, 48); memcpy(&g->x.c1, bls12_381_g2_GX_C1, 48);
    memcpy(&g->y.c0, bls12_381_g2_GY_C0, 48); memcpy(&g->y.c1, bls12_381_g2_GY_C1, 48);
}
static void load_gen_bls12_377_g2(bls12_377_g2_aff_t *g) {
    memcpy(&g->x.c0, bls12_377_g2_GX_C0, 48); memcpy(&g->x.c1, bls12_377_g2_GX_C1, 48);
    memcpy(&g->y.c0, bls12_377_g2_GY_C0, 48); memcpy(&g->y.c1, bls12_377_g2_GY_C1, 48);
}
#define load_gen_bls12_377_g2_ load_gen_bls12_377_g2
#define load_gen_bn254_g1_ load_gen_bn254_g1
#define load_gen_bn254_g2_ load_gen_bn254_g2
#define load_gen_bls12_381_g1_ load_gen_bls12_381_g1
#define load_gen_bls12_381_g2_ load_gen_bls12_381_g2
#define load_gen_bls12_377_g1_ load_gen_bls12_377_g1

size_t orc_affine_bytes(int curve, int group) {
    size_t fq = curve == ORC_BN254 ? 32 : 48;
    return 2 * fq * (group == 2 ? 2 : 1);
}

/* generator (affine, Montgomery) */
int orc_generator(int curve, int group, void *out) {
#define DO_GEN(E, B, is2) load_gen_##E((E##aff_t *)out);
    GROUP_DISPATCH(curve, group, DO_GEN)
    return ORC_OK;
}

/* out = k * p ; k = 4 x u64 canonical integer; points affine Montgomery, (0,0) = identity */
int orc_point_mul(int curve, int group, const void *p, const uint64_t *k, void *out) {
#define DO_PMUL(E, B, is2)                                                          \
    E##jac_t j, r;                                                                  \
    E##jac_from_aff(&j, (const E##aff_t *)p);                                       \
    E##jac_mul(&r, &j, k, 4);                                                       \
    E##jac_to_aff((E##aff_t *)out, &r);
    GROUP_DISPATCH(curve, group, DO_PMUL)
    return ORC_OK;
}

int orc_point_add(int curve, int group, const void *p, const void *q, void *out) {
#define DO_PADD(E, B, is2)                                                          \
    E##jac_t a, b, r;                                                               \
    E##jac_from_aff(&a, (const E##aff_t *)p);                                       \
    E##jac_from_aff(&b, (const E##aff_t *)q);                                       \
    E##jac_add(&r, &a, &b);                                                         \
    E##jac_to_aff((E##aff_t *)out, &r);
    GROUP_DISPATCH(curve, group, DO_PADD)
    return ORC_OK;
}

/* Jacobian (x, y, z Montgomery) -> affine; lets tests normalise what the GPU returns */
int orc_jac_to_affine(int curve, int group, const void *jac, void *out) {
#define DO_J2A(E, B, is2) E##jac_to_aff((E##aff_t *)out, (const E##jac_t *)jac);
    GROUP_DISPATCH(curve, group, DO_J2A)
    return ORC_OK;
}

int orc_on_curve(int curve, int group, const void *p) {
    switch (curve * 2 + group - 1) {
    case 0: return bn254_g1_aff_on_curve((const bn254_g1_aff_t *)p, (const bn254_fq_t *)bn254_g1_B);
    case 1: { bn254_fq2_t b; memcpy(&b.c0, bn254_g2_B_C0, 32); memcpy(&b.c1, bn254_g2_B_C1, 32);
              return bn254_g2_aff_on_curve((const bn254_g2_aff_t *)p, &b); }
    case 2: return bls12_381_g1_aff_on_curve((const bls12_381_g1_aff_t *)p, (const bls12_381_fq_t *)bls12_381_g1_B);
    case 3: { bls12_381_fq2_t b; memcpy(&b.c0, bls12_381_g2_B_C0, 48); memcpy(&b.c1, bls12_381_g2_B_C1, 48);
              return bls12_381_g2_aff_on_curve((const bls12_381_g2_aff_t *)p, &b); }
    case 4: return bls12_377_g1_aff_on_curve((const bls12_377_g1_aff_t *)p, (const bls12_377_fq_t *)bls12_377_g1_B);
    case 5: { bls12_377_fq2_t b; memcpy(&b.c0, bls12_377_g2_B_C0, 48); memcpy(&b.c1, bls12_377_g2_B_C1, 48);
              return bls12_377_g2_aff_on_curve((const bls12_377_g2_aff_t *)p, &b); }
    }
    return -1;
}

/* Synthetic bases: P_i = (k0 + i*k1) * G with k0, k1 = SplitMix64-derived 128-bit integers, built
 * by one scalar multiplication per 1024-point chunk plus repeated addition of D = k1*G, then one
 * Montgomery batch inversion per chunk.  Distinct, in the prime-order subgroup, cheap. */
int orc_gen_points(int curve, int group, uint64_t seed, size_t n, void *out, int threads) {
    uint64_t k0[4] = {splitmix64(seed ^ 0xA5A5), splitmix64(seed ^ 0x5A5A), 0, 0};
    uint64_t k1[4] = {splitmix64(seed ^ 0x1234) | 1, splitmix64(seed ^ 0x4321), 0, 0};
#define DO_GENPTS(E, B, is2)                                                        \
    E##aff_t g; load_gen_##E(&g);                                                   \
    E##jac_t gj, dj; E##aff_t d;                                                    \
    E##jac_from_aff(&gj, &g);                                                       \
    E##jac_mul(&dj, &gj, k1, 4);                                                    \
    E##jac_to_aff(&d, &dj);                                                         \
    E##aff_t *o = (E##aff_t *)out;                                                  \
    const size_t CH = 1024;                                                         \
    size_t nch = (n + CH - 1) / CH;                                                 \
    _Pragma("omp parallel for schedule(dynamic, 1) num_threads(threads)")           \
    for (size_t ch = 0; ch < nch; ch++) {                                           \
        size_t lo = ch * CH, hi = lo + CH < n ? lo + CH : n;                        \
        /* k = k0 + lo*k1 (fits 4 limbs: 128-bit * 64-bit + 128-bit) */            \
        uint64_t k[4] = {0, 0, 0, 0};                                               \
        unsigned __int128 acc = (unsigned __int128)k1[0] * lo + k0[0];              \
        k[0] = (uint64_t)acc;                                                       \
        acc = (unsigned __int128)k1[1] * lo + k0[1] + (uint64_t)(acc >> 64);        \
        k[1] = (uint64_t)acc; k[2] = (uint64_t)(acc >> 64);                         \
        E##jac_t *tmp = (E##jac_t *)malloc(sizeof(E##jac_t) * (hi - lo));           \
        B##t *pref = (B##t *)malloc(sizeof(B##t) * (hi - lo));                      \
        E##jac_t cur;                                                               \
        E##jac_mul(&cur, &gj, k, 4);                                                \
        for (size_t i = lo; i < hi; i++) {                                          \
            tmp[i - lo] = cur;                                                      \
            E##jac_add_aff(&cur, &cur, &d, 0);                                      \
        }                                                                           \
        /* batch inversion of the z's (none is zero: k0 + i*k1 < r, nonzero) */     \
        B##t run; B##set_one(&run);                                                 \
        for (size_t i = 0; i < hi - lo; i++) { pref[i] = run; B##mul(&run, &run, &tmp[i].z); } \
        B##t inv; B##inv(&inv, &run);                                               \
        for (size_t i = hi - lo; i-- > 0;) {                                        \
            B##t zi, zi2, zi3;                                                      \
            B##mul(&zi, &inv, &pref[i]);                                            \
            B##mul(&inv, &inv, &tmp[i].z);                                          \
            B##sqr(&zi2, &zi); B##mul(&zi3, &zi2, &zi);                             \
            B##mul(&o[lo + i].x, &tmp[i].x, &zi2);                                  \
            B##mul(&o[lo + i].y, &tmp[i].y, &zi3);                                  \
        }                                                                           \
        free(tmp); free(pref);                                                      \
    }
    GROUP_DISPATCH(curve, group, DO_GENPTS)
    return ORC_OK;
}

/* MSM: bases n x affine (Montgomery, (0,0) = identity); scalars n x 32 bytes, Montgomery form if
 * scalars_mont else canonical; out = affine sum.  algo 0 = Pippenger (arkworks-structured),
 * 1 = naive double-and-add (definition). */
int orc_msm(int curve, int group, const void *bases, const void *scalars, size_t n,
            int scalars_mont, int algo, int threads, void *out) {
    uint64_t *sc = (uint64_t *)malloc(32 * (n ? n : 1));
    if (scalars_mont) {
        int rc = orc_field_op(16 + curve, 6, scalars, NULL, sc, n);
        if (rc) { free(sc); return rc; }
    } else {
        memcpy(sc, scalars, 32 * n);
    }
    if (threads <= 0) threads = omp_get_max_threads();
#define DO_MSM(E, B, is2)                                                           \
    E##jac_t r;                                                                     \
    if (algo == 0) E##msm(&r, (const E##aff_t *)bases, sc, n, scalar_bits(curve), threads); \
    else E##msm_naive(&r, (const E##aff_t *)bases, sc, n);                          \
    E##jac_to_aff((E##aff_t *)out, &r);
    switch (curve * 2 + group - 1) {
    case 0: { DO_MSM(bn254_g1_, bn254_fq_, 0) } break;
    case 1: { DO_MSM(bn254_g2_, bn254_fq2_, 1) } break;
    case 2: { DO_MSM(bls12_381_g1_, bls12_381_fq_, 0) } break;
    case 3: { DO_MSM(bls12_381_g2_, bls12_381_fq2_, 1) } break;
    case 4: { DO_MSM(bls12_377_g1_, bls12_377_fq_, 0) } break;
    case 5: { DO_MSM(bls12_377_g2_, bls12_377_fq2_, 1) } break;
    default: free(sc); return ORC_BAD_ARG;
    }
    free(sc);
    return ORC_OK;
}

/* ---- NTT / h-poly ------------------------------------------------------------------------------ */
#define FR_DISPATCH(curve, CALL)                                  \
    switch (curve) {                                              \
    case 0: { CALL(bn254_fr_) } break;                            \
    case 1: { CALL(bls12_381_fr_) } break;                        \
    case 2: { CALL(bls12_377_fr_) } break;                        \
    default: return ORC_BAD_ARG;                                  \
    }

/* Radix2EvaluationDomain::{fft,ifft}_in_place semantics on 2^log_n Montgomery-form elements,
 * natural order in and out; coset = pointer to the domain offset (Montgomery) or NULL. */
int orc_ntt(int curve, void *data, unsigned log_n, int inverse, const void *coset, int threads) {
    if (threads <= 0) threads = omp_get_max_threads();
#define DO_NTT(P) return P##domain_transform((P##t *)data, log_n, inverse, (const P##t *)coset, threads);
    FR_DISPATCH(curve, DO_NTT)
    return ORC_OK;
}

/* CircomReduction::witness_map_from_matrices from the a, b, c evaluation vectors on
 * (/root/reference/ark-circom/src/circom/qap.rs:64-91).  a, b, c are clobbered; out may alias a. */
int orc_h_poly(int curve, void *a, void *b, void *c, unsigned log_m, void *out, int threads) {
    if (threads <= 0) threads = omp_get_max_threads();
#define DO_H(P) return P##witness_map((P##t *)a, (P##t *)b, (P##t *)c, log_m, (P##t *)out, threads);
    FR_DISPATCH(curve, DO_H)
    return ORC_OK;
}

/* qap::qap (/root/reference/groth16/src/qap.rs:44-91; the same loops open
 * /root/reference/ark-circom/src/circom/qap.rs:34-62): a[i] = <A_i, w>, b[i] = <B_i, w> over CSR rows
 * (Montgomery coefficients, Montgomery assignment), a[nc + j] = w[j] for the instance variables,
 * c = a o b on the constraint rows, zero padding up to m.  rayon-parallel in the reference: OpenMP here. */
int orc_qap(int curve, size_t nc, size_t ni, size_t m, const uint32_t *a_ptr, const uint32_t *a_col,
            const void *a_val, const uint32_t *b_ptr, const uint32_t *b_col, const void *b_val, const void *w,
            void *a, void *b, void *c, int threads) {
    if (threads <= 0) threads = omp_get_max_threads();
    if (nc + ni > m) return ORC_BAD_ARG;
#define DO_QAP(P)                                                                     \
    const P##t *av = (const P##t *)a_val, *bv = (const P##t *)b_val, *wv = (const P##t *)w; \
    P##t *ao = (P##t *)a, *bo = (P##t *)b, *co = (P##t *)c;                            \
    _Pragma("omp parallel for schedule(static) num_threads(threads)")                 \
    for (size_t i = 0; i < m; i++) {                                                  \
        P##t x, y, t;                                                                 \
        P##set_zero(&x); P##set_zero(&y);                                             \
        if (i < nc) {                                                                 \
            for (uint32_t j = a_ptr[i]; j < a_ptr[i + 1]; j++) { P##mul(&t, &av[j], &wv[a_col[j]]); P##add(&x, &x, &t); } \
            for (uint32_t j = b_ptr[i]; j < b_ptr[i + 1]; j++) { P##mul(&t, &bv[j], &wv[b_col[j]]); P##add(&y, &y, &t); } \
            P##mul(&co[i], &x, &y);                                                   \
        } else {                                                                      \
            if (i < nc + ni) x = wv[i - nc];                                          \
            P##set_zero(&co[i]);                                                      \
        }                                                                             \
        ao[i] = x; bo[i] = y;                                                         \
    }                                                                                 \
    return ORC_OK;
    FR_DISPATCH(curve, DO_QAP)
    return ORC_OK;
}

/* root of unity of order 2^log_n (Montgomery) -- FftField::get_root_of_unity */
int orc_root_of_unity(int curve, unsigned log_n, void *out) {
#define DO_ROOT(P) return P##root_of_unity((P##t *)out, log_n);
    FR_DISPATCH(curve, DO_ROOT)
    return ORC_OK;
}
