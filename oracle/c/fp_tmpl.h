/* TEST INFRASTRUCTURE ONLY (CPU oracle) -- prime-field template, plain C.
 *
 * Include with:   #define FP  bn254_fq_      (function / constant prefix)
 *                 #define NL  4              (64-bit limbs)
 * Expects consts_gen.h to have defined  <FP>P, <FP>R, <FP>R2 (uint64_t[NL]) and <FP>INV.
 *
 * Restates the Montgomery prime field the reference gets from ark-ff 0.4 (not vendored;
 * SURVEY.md 8(c)): little-endian u64 limbs, R = 2^(64*NL) (pinned by
 * /root/reference/ark-circom/src/zkey.rs:417-427), elements held in Montgomery form.
 * Algorithm: textbook CIOS Montgomery multiplication on unsigned __int128.
 */
#include <stdint.h>
#include <string.h>

#define FP_CAT_(a, b) a##b
#define FP_CAT(a, b) FP_CAT_(a, b)
#define FN(name) FP_CAT(FP, name)
#define FT FP_CAT(FP, t)

typedef struct { uint64_t l[NL]; } FT;
typedef unsigned __int128 u128_t_;

static inline int FN(is_zero)(const FT *a) {
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc |= a->l[i];
    return acc == 0;
}
static inline int FN(eq)(const FT *a, const FT *b) {
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc |= a->l[i] ^ b->l[i];
    return acc == 0;
}
static inline void FN(set_zero)(FT *a) { memset(a, 0, sizeof(FT)); }
static inline void FN(set_one)(FT *a) { memcpy(a->l, FN(R), sizeof(FT)); }

/* returns 1 if a >= p */
static inline int FN(geq_p)(const uint64_t *a) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > FN(P)[i]) return 1;
        if (a[i] < FN(P)[i]) return 0;
    }
    return 1;
}
static inline void FN(sub_p)(uint64_t *a) {
    uint64_t borrow = 0;
    for (int i = 0; i < NL; i++) {
        u128_t_ d = (u128_t_)a[i] - FN(P)[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
static inline void FN(add)(FT *r, const FT *a, const FT *b) {
    uint64_t carry = 0;
    for (int i = 0; i < NL; i++) {
        u128_t_ s = (u128_t_)a->l[i] + b->l[i] + carry;
        r->l[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    /* p < 2^(64*NL - 1) for every field here, so no carry-out */
    if (FN(geq_p)(r->l)) FN(sub_p)(r->l);
}
static inline void FN(sub)(FT *r, const FT *a, const FT *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < NL; i++) {
        u128_t_ d = (u128_t_)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < NL; i++) {
            u128_t_ s = (u128_t_)r->l[i] + FN(P)[i] + carry;
            r->l[i] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
}
static inline void FN(neg)(FT *r, const FT *a) {
    if (FN(is_zero)(a)) { *r = *a; return; }
    uint64_t borrow = 0;
    for (int i = 0; i < NL; i++) {
        u128_t_ d = (u128_t_)FN(P)[i] - a->l[i] - borrow;
        r->l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
static inline void FN(dbl)(FT *r, const FT *a) { FN(add)(r, a, a); }

static inline void FN(mul)(FT *r, const FT *a, const FT *b) {
    uint64_t t[NL + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < NL; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < NL; j++) {
            u128_t_ acc = (u128_t_)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        u128_t_ acc = (u128_t_)t[NL] + carry;
        t[NL] = (uint64_t)acc;
        t[NL + 1] = (uint64_t)(acc >> 64);
        uint64_t m = t[0] * FN(INV);
        acc = (u128_t_)m * FN(P)[0] + t[0];
        carry = (uint64_t)(acc >> 64);
        for (int j = 1; j < NL; j++) {
            acc = (u128_t_)m * FN(P)[j] + t[j] + carry;
            t[j - 1] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        acc = (u128_t_)t[NL] + carry;
        t[NL - 1] = (uint64_t)acc;
        t[NL] = t[NL + 1] + (uint64_t)(acc >> 64);
    }
    if (t[NL] || FN(geq_p)(t)) FN(sub_p)(t);
    memcpy(r->l, t, sizeof(FT));
}
static inline void FN(sqr)(FT *r, const FT *a) { FN(mul)(r, a, a); }

/* canonical integer (little-endian u64 limbs) -> Montgomery form; input must be < p */
static inline void FN(to_mont)(FT *r, const FT *a) {
    FT r2;
    memcpy(r2.l, FN(R2), sizeof(FT));
    FN(mul)(r, a, &r2);
}
/* Montgomery form -> canonical integer */
static inline void FN(from_mont)(FT *r, const FT *a) {
    FT one;
    memset(&one, 0, sizeof one);
    one.l[0] = 1;
    FN(mul)(r, a, &one);
}
static inline void FN(from_u64)(FT *r, uint64_t v) {
    FT t;
    memset(&t, 0, sizeof t);
    t.l[0] = v;
    FN(to_mont)(r, &t);
}
/* r = a^e, e given as little-endian u64 limbs */
static inline void FN(pow)(FT *r, const FT *a, const uint64_t *e, int elimbs) {
    FT acc, base = *a;
    FN(set_one)(&acc);
    int started = 0;
    for (int i = elimbs * 64 - 1; i >= 0; i--) {
        if (started) FN(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            FN(mul)(&acc, &acc, &base);
            started = 1;
        }
    }
    *r = acc;
}
/* Fermat inversion: a^(p-2); inv(0) = 0 */
static inline void FN(inv)(FT *r, const FT *a) {
    uint64_t e[NL];
    memcpy(e, FN(P), sizeof e);
    /* p - 2: p is odd and its low limb is >= 3 for every field here */
    e[0] -= 2;
    FN(pow)(r, a, e, NL);
}

#undef FN
#undef FT
#undef FP_CAT
#undef FP_CAT_
