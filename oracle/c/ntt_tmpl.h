/* TEST INFRASTRUCTURE ONLY (CPU oracle) -- radix-2 NTT template over a scalar field, plain C.
 *
 * Include with:  #define FR bn254_fr_   (an fp_tmpl instance with <FR>TWO_ADIC_ROOT, <FR>GEN)
 *
 * Restates ark-poly 0.4 `Radix2EvaluationDomain::{fft_in_place, ifft_in_place}` (not vendored;
 * reference call sites /root/reference/ark-circom/src/circom/qap.rs:64-85,
 * dist-primitives/src/dfft/mod.rs:40,78-81): natural order in and out, omega =
 * TWO_ADIC_ROOT^(2^(s - log_n)), ifft scales by n^-1, a coset domain multiplies coefficient i by
 * offset^i before the forward transform / by offset^-i after the inverse one.
 * Structure (in-place bit-reversal + iterative DIT, butterflies parallel) is the textbook one.
 */
#define NT_CAT_(a, b) a##b
#define NT_CAT(a, b) NT_CAT_(a, b)
#define FRN(name) NT_CAT(FR, name)
#define FRT NT_CAT(FR, t)

/* defined per instance in oracle.c from the generated <FIELD>_TWO_ADICITY macro */
static int FRN(two_adicity)(void);

static int FRN(root_of_unity)(FRT *out, unsigned log_n) {
    int s = FRN(two_adicity)();
    if ((int)log_n > s) return 1;
    FRT w;
    memcpy(w.l, FRN(TWO_ADIC_ROOT), sizeof w);
    for (int i = 0; i < s - (int)log_n; i++) FRN(sqr)(&w, &w);
    *out = w;
    return 0;
}

static void FRN(bitrev)(FRT *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (j > i) { FRT t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* plain cyclic NTT with root w (order 2^log_n): out[j] = sum_i a[i] w^(ij) */
static void FRN(ntt_core)(FRT *a, unsigned log_n, const FRT *w, int threads) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    FRN(bitrev)(a, log_n);
    FRT *tw = (FRT *)malloc(sizeof(FRT) * (n / 2));
    FRN(set_one)(&tw[0]);
    for (size_t k = 1; k < n / 2; k++) FRN(mul)(&tw[k], &tw[k - 1], w);
    for (unsigned lv = 1; lv <= log_n; lv++) {
        size_t half = (size_t)1 << (lv - 1);
        size_t step = n >> lv;          /* twiddle stride */
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 4096)
        for (size_t t = 0; t < n / 2; t++) {
            size_t k = t & (half - 1);
            size_t start = (t >> (lv - 1)) << lv;
            FRT *x = &a[start + k], *y = &a[start + k + half], yy, xx;
            FRN(mul)(&yy, y, &tw[k * step]);
            xx = *x;
            FRN(add)(x, &xx, &yy);
            FRN(sub)(y, &xx, &yy);
        }
    }
    free(tw);
}

/* a[i] *= c * g^i */
static void FRN(distribute_powers)(FRT *a, size_t n, const FRT *g, const FRT *c, int threads) {
    const size_t CH = 4096;
    size_t nch = (n + CH - 1) / CH;
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 8192)
    for (size_t ch = 0; ch < nch; ch++) {
        size_t lo = ch * CH, hi = lo + CH < n ? lo + CH : n;
        uint64_t e[1] = {lo};
        FRT p;
        FRN(pow)(&p, g, e, 1);
        FRN(mul)(&p, &p, c);
        for (size_t i = lo; i < hi; i++) {
            FRN(mul)(&a[i], &a[i], &p);
            FRN(mul)(&p, &p, g);
        }
    }
}

static int FRN(domain_transform)(FRT *a, unsigned log_n, int inverse, const FRT *coset, int threads) {
    FRT w, one;
    if (FRN(root_of_unity)(&w, log_n)) return 1;
    FRN(set_one)(&one);
    size_t n = (size_t)1 << log_n;
    if (!inverse) {
        if (coset && !FRN(eq)(coset, &one)) FRN(distribute_powers)(a, n, coset, &one, threads);
        FRN(ntt_core)(a, log_n, &w, threads);
    } else {
        FRT wi, ninv, nn, gi;
        FRN(inv)(&wi, &w);
        FRN(ntt_core)(a, log_n, &wi, threads);
        FRN(from_u64)(&nn, (uint64_t)n);
        FRN(inv)(&ninv, &nn);
        if (coset && !FRN(eq)(coset, &one)) FRN(inv)(&gi, coset); else gi = one;
        FRN(distribute_powers)(a, n, &gi, &ninv, threads);
    }
    return 0;
}

/* h = fft(shift(ifft a)) * fft(shift(ifft b)) - fft(shift(ifft c)), shift = multiply coefficient
 * i by w_{2m}^i  (ark-circom/src/circom/qap.rs:64-91) */
static int FRN(witness_map)(FRT *a, FRT *b, FRT *c, unsigned log_m, FRT *out, int threads) {
    FRT root, one;
    if (FRN(root_of_unity)(&root, log_m + 1)) return 1;
    FRN(set_one)(&one);
    size_t m = (size_t)1 << log_m;
    FRT *v[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
        if (FRN(domain_transform)(v[k], log_m, 1, NULL, threads)) return 1;
        FRN(distribute_powers)(v[k], m, &root, &one, threads);
        if (FRN(domain_transform)(v[k], log_m, 0, NULL, threads)) return 1;
    }
#pragma omp parallel for schedule(static) num_threads(threads) if (m >= 4096)
    for (size_t i = 0; i < m; i++) {
        FRT t;
        FRN(mul)(&t, &a[i], &b[i]);
        FRN(sub)(&out[i], &t, &c[i]);
    }
    return 0;
}

#undef FRN
#undef FRT
#undef NT_CAT
#undef NT_CAT_
