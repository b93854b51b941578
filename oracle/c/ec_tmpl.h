/* TEST INFRASTRUCTURE ONLY (CPU oracle) -- short-Weierstrass (a = 0) group template, plain C.
 *
 * Include with:  #define EC bn254_g1_    (prefix of the generated functions)
 *                #define BF bn254_fq_    (base field prefix: an fp_tmpl or fp2_tmpl instance)
 *
 * Restates the group arithmetic and `VariableBaseMSM::msm` the reference obtains from ark-ec 0.4
 * (call site /root/reference/dist-primitives/src/dmsm/mod.rs:82; not vendored).
 * The MSM is structured like arkworks' `msm_bigint_wnaf` as published: signed c-bit digits with
 * c = 3 for n < 32 else ln(n)+2, one bucket pass per window, windows parallel (OpenMP here,
 * rayon there), Horner combination of the window sums.  Memory layout of an affine point is
 * x || y in Montgomery form, identity = (0, 0) (as /root/reference/ark-circom/src/zkey.rs:353-361).
 */
#include <stdlib.h>

#define EC_CAT_(a, b) a##b
#define EC_CAT(a, b) EC_CAT_(a, b)
#define ECN(name) EC_CAT(EC, name)
#define BFN(name) EC_CAT(BF, name)
#define BT EC_CAT(BF, t)
#define AFF EC_CAT(EC, aff_t)
#define JAC EC_CAT(EC, jac_t)

typedef struct { BT x, y; } AFF;
typedef struct { BT x, y, z; } JAC;

static inline int ECN(aff_is_inf)(const AFF *p) { return BFN(is_zero)(&p->x) && BFN(is_zero)(&p->y); }
static inline int ECN(jac_is_inf)(const JAC *p) { return BFN(is_zero)(&p->z); }
static inline void ECN(jac_set_inf)(JAC *p) { BFN(set_one)(&p->x); BFN(set_one)(&p->y); BFN(set_zero)(&p->z); }
static inline void ECN(jac_from_aff)(JAC *r, const AFF *p) {
    if (ECN(aff_is_inf)(p)) { ECN(jac_set_inf)(r); return; }
    r->x = p->x; r->y = p->y; BFN(set_one)(&r->z);
}
static inline void ECN(jac_to_aff)(AFF *r, const JAC *p) {
    if (ECN(jac_is_inf)(p)) { BFN(set_zero)(&r->x); BFN(set_zero)(&r->y); return; }
    BT zi, zi2, zi3;
    BFN(inv)(&zi, &p->z);
    BFN(sqr)(&zi2, &zi);
    BFN(mul)(&zi3, &zi2, &zi);
    BFN(mul)(&r->x, &p->x, &zi2);
    BFN(mul)(&r->y, &p->y, &zi3);
}
/* dbl-2009-l */
static inline void ECN(jac_dbl)(JAC *r, const JAC *p) {
    if (ECN(jac_is_inf)(p)) { *r = *p; return; }
    BT a, b, c, d, e, f, t;
    BFN(sqr)(&a, &p->x);
    BFN(sqr)(&b, &p->y);
    BFN(sqr)(&c, &b);
    BFN(add)(&t, &p->x, &b);
    BFN(sqr)(&t, &t);
    BFN(sub)(&t, &t, &a);
    BFN(sub)(&t, &t, &c);
    BFN(dbl)(&d, &t);
    BFN(dbl)(&e, &a);
    BFN(add)(&e, &e, &a);
    BFN(sqr)(&f, &e);
    BT z3;
    BFN(mul)(&z3, &p->y, &p->z);
    BFN(dbl)(&z3, &z3);
    BFN(dbl)(&t, &d);
    BFN(sub)(&r->x, &f, &t);
    BFN(sub)(&t, &d, &r->x);
    BFN(mul)(&t, &e, &t);
    BFN(dbl)(&c, &c); BFN(dbl)(&c, &c); BFN(dbl)(&c, &c);
    BFN(sub)(&r->y, &t, &c);
    r->z = z3;
}
/* add-2007-bl, complete via explicit special cases */
static inline void ECN(jac_add)(JAC *r, const JAC *p, const JAC *q) {
    if (ECN(jac_is_inf)(p)) { *r = *q; return; }
    if (ECN(jac_is_inf)(q)) { *r = *p; return; }
    BT z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t;
    BFN(sqr)(&z1z1, &p->z);
    BFN(sqr)(&z2z2, &q->z);
    BFN(mul)(&u1, &p->x, &z2z2);
    BFN(mul)(&u2, &q->x, &z1z1);
    BFN(mul)(&s1, &p->y, &q->z); BFN(mul)(&s1, &s1, &z2z2);
    BFN(mul)(&s2, &q->y, &p->z); BFN(mul)(&s2, &s2, &z1z1);
    if (BFN(eq)(&u1, &u2)) {
        if (BFN(eq)(&s1, &s2)) { ECN(jac_dbl)(r, p); return; }
        ECN(jac_set_inf)(r); return;
    }
    BFN(sub)(&h, &u2, &u1);
    BFN(sub)(&rr, &s2, &s1);
    BFN(sqr)(&hh, &h);
    BFN(mul)(&hhh, &h, &hh);
    BFN(mul)(&v, &u1, &hh);
    BT x3, y3, z3;
    BFN(sqr)(&x3, &rr);
    BFN(sub)(&x3, &x3, &hhh);
    BFN(dbl)(&t, &v);
    BFN(sub)(&x3, &x3, &t);
    BFN(sub)(&t, &v, &x3);
    BFN(mul)(&y3, &rr, &t);
    BFN(mul)(&t, &s1, &hhh);
    BFN(sub)(&y3, &y3, &t);
    BFN(mul)(&z3, &p->z, &q->z);
    BFN(mul)(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
/* mixed addition (q affine); neg != 0 adds -q */
static inline void ECN(jac_add_aff)(JAC *r, const JAC *p, const AFF *q, int neg) {
    if (ECN(aff_is_inf)(q)) { *r = *p; return; }
    AFF qq = *q;
    if (neg) BFN(neg)(&qq.y, &q->y);
    if (ECN(jac_is_inf)(p)) { ECN(jac_from_aff)(r, &qq); return; }
    BT z1z1, u2, s2, h, rr, hh, hhh, v, t;
    BFN(sqr)(&z1z1, &p->z);
    BFN(mul)(&u2, &qq.x, &z1z1);
    BFN(mul)(&s2, &qq.y, &p->z); BFN(mul)(&s2, &s2, &z1z1);
    if (BFN(eq)(&p->x, &u2)) {
        if (BFN(eq)(&p->y, &s2)) { ECN(jac_dbl)(r, p); return; }
        ECN(jac_set_inf)(r); return;
    }
    BFN(sub)(&h, &u2, &p->x);
    BFN(sub)(&rr, &s2, &p->y);
    BFN(sqr)(&hh, &h);
    BFN(mul)(&hhh, &h, &hh);
    BFN(mul)(&v, &p->x, &hh);
    BT x3, y3, z3;
    BFN(sqr)(&x3, &rr);
    BFN(sub)(&x3, &x3, &hhh);
    BFN(dbl)(&t, &v);
    BFN(sub)(&x3, &x3, &t);
    BFN(sub)(&t, &v, &x3);
    BFN(mul)(&y3, &rr, &t);
    BFN(mul)(&t, &p->y, &hhh);
    BFN(sub)(&y3, &y3, &t);
    BFN(mul)(&z3, &p->z, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
/* k * p, k = little-endian u64 limbs (plain integer, not reduced) */
static inline void ECN(jac_mul)(JAC *r, const JAC *p, const uint64_t *k, int klimbs) {
    JAC acc;
    ECN(jac_set_inf)(&acc);
    for (int i = klimbs * 64 - 1; i >= 0; i--) {
        ECN(jac_dbl)(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) ECN(jac_add)(&acc, &acc, p);
    }
    *r = acc;
}
static inline int ECN(aff_on_curve)(const AFF *p, const BT *b) {
    if (ECN(aff_is_inf)(p)) return 1;
    BT l, r;
    BFN(sqr)(&l, &p->y);
    BFN(sqr)(&r, &p->x);
    BFN(mul)(&r, &r, &p->x);
    BFN(add)(&r, &r, b);
    return BFN(eq)(&l, &r);
}

/* ---- MSM -------------------------------------------------------------------------------- */
static inline unsigned ECN(msm_window)(size_t n) {
    if (n < 32) return 3;
    unsigned lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;     /* floor(log2 n) ... */
    if (((size_t)1 << lg) < n) lg++;               /* ... ark_std::log2 is the ceiling */
    return lg * 69 / 100 + 2;
}
/* scalars: n x 4 u64 limbs, canonical integers < 2^scalar_bits.  out: Jacobian. */
static void ECN(msm)(JAC *out, const AFF *bases, const uint64_t *scalars, size_t n,
                     unsigned scalar_bits, int threads) {
    unsigned c = ECN(msm_window)(n);
    unsigned nwin = (scalar_bits + c - 1) / c;
    /* signed digits (make_digits): digit in [-2^(c-1), 2^(c-1)) except the last window */
    int32_t *digits = (int32_t *)malloc(sizeof(int32_t) * n * nwin);
    for (size_t i = 0; i < n; i++) {
        const uint64_t *s = scalars + 4 * i;
        uint64_t carry = 0;
        for (unsigned w = 0; w < nwin; w++) {
            unsigned bit = w * c;
            unsigned limb = bit / 64, off = bit % 64;
            uint64_t v = 0;
            if (limb < 4) {
                v = s[limb] >> off;
                if (off + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
            }
            v &= (((uint64_t)1 << c) - 1);
            int64_t d = (int64_t)(v + carry);
            carry = ((uint64_t)d + ((uint64_t)1 << (c - 1))) >> c;
            d -= (int64_t)(carry << c);
            if (w == nwin - 1) d += (int64_t)(carry << c);
            digits[i * nwin + w] = (int32_t)d;
        }
    }
    JAC *wsum = (JAC *)malloc(sizeof(JAC) * nwin);
    size_t nb = (size_t)1 << c;   /* last window may hold up to 2^c (arkworks sizes all alike) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (unsigned w = 0; w < nwin; w++) {
        JAC *buckets = (JAC *)malloc(sizeof(JAC) * nb);
        for (size_t b = 0; b < nb; b++) ECN(jac_set_inf)(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int32_t d = digits[i * nwin + w];
            if (d > 0) ECN(jac_add_aff)(&buckets[d - 1], &buckets[d - 1], &bases[i], 0);
            else if (d < 0) ECN(jac_add_aff)(&buckets[-d - 1], &buckets[-d - 1], &bases[i], 1);
        }
        JAC run, acc;
        ECN(jac_set_inf)(&run);
        ECN(jac_set_inf)(&acc);
        for (size_t b = nb; b-- > 0;) {
            ECN(jac_add)(&run, &run, &buckets[b]);
            ECN(jac_add)(&acc, &acc, &run);
        }
        wsum[w] = acc;
        free(buckets);
    }
    JAC total;
    ECN(jac_set_inf)(&total);
    for (unsigned w = nwin; w-- > 1;) {
        ECN(jac_add)(&total, &total, &wsum[w]);
        for (unsigned k = 0; k < c; k++) ECN(jac_dbl)(&total, &total);
    }
    ECN(jac_add)(&total, &total, &wsum[0]);
    *out = total;
    free(wsum);
    free(digits);
}
/* definitional MSM: sum of double-and-add products (slow; cross-check for `msm`) */
static void ECN(msm_naive)(JAC *out, const AFF *bases, const uint64_t *scalars, size_t n) {
    JAC total;
    ECN(jac_set_inf)(&total);
    for (size_t i = 0; i < n; i++) {
        JAC p, t;
        ECN(jac_from_aff)(&p, &bases[i]);
        ECN(jac_mul)(&t, &p, scalars + 4 * i, 4);
        ECN(jac_add)(&total, &total, &t);
    }
    *out = total;
}

#undef ECN
#undef BFN
#undef BT
#undef AFF
#undef JAC
#undef EC_CAT
#undef EC_CAT_
