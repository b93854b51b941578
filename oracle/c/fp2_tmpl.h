/* TEST INFRASTRUCTURE ONLY (CPU oracle) -- quadratic extension Fq[u]/(u^2 + F2_BETA) template, plain C.
 * (BN254 and BLS12-381 use the non-residue -1, BLS12-377 uses -5; ark-bn254 / ark-bls12-381 / ark-bls12-377
 * Fq2Config::NONRESIDUE.)
 *
 * Include with:  #define F2 bn254_fq2_   #define FB bn254_fq_   #define F2_BETA 1   (u^2 = -F2_BETA, 1 or 5)
 */
#define F2_CAT_(a, b) a##b
#define F2_CAT(a, b) F2_CAT_(a, b)
#define F2N(name) F2_CAT(F2, name)
#define FBN(name) F2_CAT(FB, name)
#define F2T F2_CAT(F2, t)
#define FBT F2_CAT(FB, t)

typedef struct { FBT c0, c1; } F2T;

static inline int F2N(is_zero)(const F2T *a) { return FBN(is_zero)(&a->c0) && FBN(is_zero)(&a->c1); }
static inline int F2N(eq)(const F2T *a, const F2T *b) { return FBN(eq)(&a->c0, &b->c0) && FBN(eq)(&a->c1, &b->c1); }
static inline void F2N(set_zero)(F2T *a) { FBN(set_zero)(&a->c0); FBN(set_zero)(&a->c1); }
static inline void F2N(set_one)(F2T *a) { FBN(set_one)(&a->c0); FBN(set_zero)(&a->c1); }
static inline void F2N(add)(F2T *r, const F2T *a, const F2T *b) { FBN(add)(&r->c0, &a->c0, &b->c0); FBN(add)(&r->c1, &a->c1, &b->c1); }
static inline void F2N(sub)(F2T *r, const F2T *a, const F2T *b) { FBN(sub)(&r->c0, &a->c0, &b->c0); FBN(sub)(&r->c1, &a->c1, &b->c1); }
static inline void F2N(neg)(F2T *r, const F2T *a) { FBN(neg)(&r->c0, &a->c0); FBN(neg)(&r->c1, &a->c1); }
static inline void F2N(dbl)(F2T *r, const F2T *a) { F2N(add)(r, a, a); }
/* r = F2_BETA * a in the base field (small multiple by repeated addition) */
static inline void F2N(beta_mul)(FBT *r, const FBT *a) {
    FBT acc = *a;
    for (int k = 1; k < F2_BETA; k++) FBN(add)(&acc, &acc, a);
    *r = acc;
}
static inline void F2N(mul)(F2T *r, const F2T *a, const F2T *b) {
    /* (a0 + a1 u)(b0 + b1 u) = (a0 b0 - beta a1 b1) + (a0 b1 + a1 b0) u */
    FBT t0, t1, t2, t3;
    FBN(mul)(&t0, &a->c0, &b->c0);
    FBN(mul)(&t1, &a->c1, &b->c1);
    F2N(beta_mul)(&t1, &t1);
    FBN(mul)(&t2, &a->c0, &b->c1);
    FBN(mul)(&t3, &a->c1, &b->c0);
    FBN(sub)(&r->c0, &t0, &t1);
    FBN(add)(&r->c1, &t2, &t3);
}
static inline void F2N(sqr)(F2T *r, const F2T *a) { F2N(mul)(r, a, a); }
static inline void F2N(inv)(F2T *r, const F2T *a) {
    /* 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 + beta a1^2) */
    FBT n, t, ni;
    FBN(sqr)(&n, &a->c0);
    FBN(sqr)(&t, &a->c1);
    F2N(beta_mul)(&t, &t);
    FBN(add)(&n, &n, &t);
    FBN(inv)(&ni, &n);
    FBN(mul)(&r->c0, &a->c0, &ni);
    FBN(mul)(&t, &a->c1, &ni);
    FBN(neg)(&r->c1, &t);
}

#undef F2N
#undef FBN
#undef F2T
#undef FBT
#undef F2_CAT
#undef F2_CAT_
