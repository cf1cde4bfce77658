/* tools/sanitize/oracle_driver.c -- every entry point of oracle/csmp_oracle.h on small seeded problems, compiled TOGETHER with
 * oracle/csmp_oracle.c under gcc's AddressSanitizer (leak detection on) + UndefinedBehaviorSanitizer by tools/sanitize_cpu.sh.
 * Output buffers are heap blocks of exactly the sizes the header documents, so one element too far is a report.  The checks are
 * sanity only (status codes, sorted distinct in-range indices, the residual the solution leaves): WHAT the oracle computes is pinned
 * by tests/test_oracle.py, which the script also runs against the sanitized build. */
#include "../../oracle/csmp_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int fails = 0;
#define EXPECT(cond)                                                                 \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
            ++fails;                                                                 \
        }                                                                            \
    } while (0)

static uint64_t sd = 0x9E3779B97F4A7C15ull;
static double unif(void) {
    sd = sd * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(sd >> 11) / 9007199254740992.0;
}
static double gauss(void) {
    const double u = unif() + 1e-300, v = unif();
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

/* column-major M x N with leading dimension ld >= M, unit-norm columns (src/util.jl:21-27), as f64 or f32; the padding rows hold
 * NaN: the oracle must never read them */
static void *make_dictionary(int dtype, int64_t M, int64_t N, int64_t ld) {
    const size_t es = dtype == CSO_F32 ? 4 : 8;
    char *A = (char *)malloc((size_t)ld * (size_t)N * es);
    for (int64_t j = 0; j < N; ++j) {
        double *col = (double *)malloc((size_t)M * 8), n2 = 0.0;
        for (int64_t i = 0; i < M; ++i) {
            col[i] = gauss();
            n2 += col[i] * col[i];
        }
        for (int64_t i = 0; i < ld; ++i) {
            const double v = i < M ? col[i] / sqrt(n2) : NAN;
            if (dtype == CSO_F32)
                ((float *)A)[j * ld + i] = (float)v;
            else
                ((double *)A)[j * ld + i] = v;
        }
        free(col);
    }
    return A;
}
static double at(const void *A, int dtype, int64_t ld, int64_t i, int64_t j) {
    return dtype == CSO_F32 ? (double)((const float *)A)[j * ld + i] : ((const double *)A)[j * ld + i];
}
/* b = A x0 with a planted k0-sparse +-1 x0 (src/util.jl:13-19) */
static double *make_signal(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, int64_t k0) {
    double *b = (double *)calloc((size_t)M, 8);
    for (int64_t t = 0; t < k0; ++t) {
        const int64_t j = (int64_t)(unif() * (double)N) % N;
        const double s = unif() < 0.5 ? -1.0 : 1.0;
        for (int64_t i = 0; i < M; ++i) b[i] += s * at(A, dtype, ld, i, j);
    }
    return b;
}
static void check_support(const char *what, const int64_t *idx, const double *val, int64_t nnz, int64_t cap, int64_t N, int sorted) {
    EXPECT(nnz >= 0 && nnz <= cap);
    for (int64_t t = 0; t < nnz && t < cap; ++t) {
        if (!(idx[t] >= 0 && idx[t] < N) || !isfinite(val[t]) || (sorted && t > 0 && !(idx[t] > idx[t - 1]))) {
            fprintf(stderr, "FAILED %s: entry %lld = (%lld, %g)\n", what, (long long)t, (long long)idx[t], val[t]);
            ++fails;
            return;
        }
    }
}
static double residual_norm(const void *A, int dtype, int64_t M, int64_t ld, const int64_t *idx, const double *val, int64_t nnz, const double *b) {
    double *r = (double *)malloc((size_t)M * 8), n2 = 0.0;
    cso_residual(A, dtype, M, ld, idx, val, nnz, b, r);
    for (int64_t i = 0; i < M; ++i) n2 += r[i] * r[i];
    free(r);
    return sqrt(n2);
}

static void family(int dtype, int64_t M, int64_t N, int64_t ld, int64_t k, int nthreads) {
    void *A = make_dictionary(dtype, M, N, ld);
    double *b = make_signal(A, dtype, M, N, ld, k);
    double bn = 0.0;
    for (int64_t i = 0; i < M; ++i) bn += b[i] * b[i];
    bn = sqrt(bn);
    const int64_t kk = k < 1 ? 1 : k;
    int64_t *idx = (int64_t *)malloc((size_t)kk * 8), *ord = (int64_t *)malloc((size_t)kk * 8), nnz = -1, iters = -1;
    double *val = (double *)malloc((size_t)kk * 8);
    /* omp: src/matchingpursuit.jl:62-91 */
    EXPECT(cso_omp(A, dtype, M, N, ld, b, k, 1e-12, idx, val, &nnz, ord, nthreads) == CSO_OK);
    check_support("omp", idx, val, nnz, k, N, 1);
    if (k > 0 && bn > 0) EXPECT(residual_norm(A, dtype, M, ld, idx, val, nnz, b) <= bn * (1 + 1e-12));
    EXPECT(cso_omp(A, dtype, M, N, ld, b, k, -1.0, idx, val, &nnz, NULL, nthreads) == CSO_EINVAL);
    EXPECT(cso_omp(A, dtype, M, N, ld, b, k, 1e300, idx, val, &nnz, NULL, nthreads) == CSO_OK && nnz == (k > 0 ? 1 : 0));  /* update!, THEN the eps test (:78-79): one atom */
    /* gomp with a remainder step (:134-137): l = 3 */
    if (k >= 3) {
        EXPECT(cso_gomp(A, dtype, M, N, ld, b, 3, k, 1e-12, idx, val, &nnz, ord, nthreads) == CSO_OK);
        check_support("gomp", idx, val, nnz, k, N, 1);
        EXPECT(cso_gomp(A, dtype, M, N, ld, b, 3, k, -1.0, idx, val, &nnz, NULL, nthreads) == CSO_EINVAL);
    }
    /* fr (src/forward.jl:44-114) */
    EXPECT(cso_fr(A, dtype, M, N, ld, b, k, 0.0, 0.0, idx, val, &nnz, ord, nthreads) == CSO_OK);
    check_support("fr", idx, val, nnz, k, N, 1);
    /* mp: idx/val sized min(k + nnz0, N); cold and warm (src/matchingpursuit.jl:26-40) */
    {
        const int64_t k2 = 2 * k + 1, cap = k2 < N ? k2 : N;
        int64_t *mi = (int64_t *)malloc((size_t)cap * 8), mn = -1;
        double *mv = (double *)malloc((size_t)cap * 8);
        EXPECT(cso_mp(A, dtype, M, N, ld, b, k2, NULL, NULL, 0, mi, mv, &mn, nthreads) == CSO_OK);
        check_support("mp", mi, mv, mn, cap, N, 1);
        if (mn > 0 && mn + 2 <= N) {
            const int64_t cap2 = (2 + mn) < N ? (2 + mn) : N;
            int64_t *wi = (int64_t *)malloc((size_t)cap2 * 8), wn = -1;
            double *wv = (double *)malloc((size_t)cap2 * 8);
            EXPECT(cso_mp(A, dtype, M, N, ld, b, 2, mi, mv, mn, wi, wv, &wn, nthreads) == CSO_OK);
            check_support("mp warm", wi, wv, wn, cap2, N, 1);
            free(wi);
            free(wv);
        }
        free(mi);
        free(mv);
    }
    /* sp / ompr: 2k <= M (src/twostage.jl:55) */
    if (k >= 1) {
        const int rc = cso_sp(A, dtype, M, N, ld, b, k, 1e-12, -1, idx, val, &nnz, &iters, nthreads);
        EXPECT(rc == (2 * k <= M ? CSO_OK : CSO_ERANGE));
        if (rc == CSO_OK) check_support("sp", idx, val, nnz, k, N, 1);
        if (2 * k <= M && k <= N) {
            EXPECT(cso_ompr(A, dtype, M, N, ld, b, k, 1e-12, -1, idx, val, &nnz, &iters, nthreads) == CSO_OK);
            check_support("ompr", idx, val, nnz, k, N, 1);
        }
    }
    /* srr: idx/val sized k + l; the three initialisations (src/twostage.jl:3-33) */
    if (k >= 1 && k + 2 <= M && k + 2 <= N) {
        const int64_t l = 2;
        int64_t *si = (int64_t *)malloc((size_t)(k + l) * 8), sn = -1, *init = (int64_t *)malloc((size_t)k * 8);
        double *sv = (double *)malloc((size_t)(k + l) * 8);
        for (int init_kind = 1; init_kind <= 2; ++init_kind) {
            EXPECT(cso_srr(A, dtype, M, N, ld, b, k, 1e-12, -1, init_kind, l, si, sv, &sn, &iters, nthreads) == CSO_OK);
            check_support("srr", si, sv, sn, k + l, N, 1);
        }
        for (int64_t t = 0; t < k; ++t) init[t] = (t * 7 + 3) % N;
        for (int64_t t = 0; t < k; ++t)  /* distinct atoms */
            for (int64_t u = 0; u < t; ++u)
                if (init[u] == init[t]) init[t] = (init[t] + 1) % N, u = -1;
        EXPECT(cso_srr_from(A, dtype, M, N, ld, b, k, 1e-12, 3, init, l, si, sv, &sn, &iters, nthreads) == CSO_OK);
        check_support("srr_from", si, sv, sn, k + l, N, 1);
        free(si);
        free(sv);
        free(init);
    }
    /* rmp / foba: idx/val sized min(M, N) + 1 (src/stepwise.jl:5-56) */
    {
        const int64_t cap = (M < N ? M : N) + 1;
        int64_t *ri = (int64_t *)malloc((size_t)cap * 8), rn = -1;
        double *rv = (double *)malloc((size_t)cap * 8);
        EXPECT(cso_rmp_delta(A, dtype, M, N, ld, b, 1e-3, 4, ri, rv, &rn, nthreads) == CSO_OK);
        check_support("rmp_delta", ri, rv, rn, cap, N, 1);
        if (k >= 1) {
            EXPECT(cso_rmp_k(A, dtype, M, N, ld, b, k, ri, rv, &rn, nthreads) == CSO_OK);
            check_support("rmp_k", ri, rv, rn, cap, N, 1);
        }
        EXPECT(cso_foba(A, dtype, M, N, ld, b, 1e-3, ri, rv, &rn, nthreads) == CSO_OK);
        check_support("foba", ri, rv, rn, cap, N, 1);
        free(ri);
        free(rv);
    }
    /* primitives */
    {
        double *out = (double *)malloc((size_t)N * 8), *coef = (double *)malloc((size_t)kk * 8);
        const int64_t am = cso_sweep_abs(A, dtype, M, N, ld, b, out, nthreads);
        EXPECT(am >= 0 && am < N);
        for (int64_t j = 0; j < N; ++j) EXPECT(out[j] <= out[am] && (out[j] < out[am] || j >= am));  /* first maximal index */
        const int64_t kt = k < N ? k : N;
        int64_t *top = (int64_t *)malloc((size_t)(kt > 0 ? kt : 1) * 8);
        cso_topk_desc(out, N, kt, top);
        for (int64_t t = 1; t < kt; ++t) EXPECT(out[top[t - 1]] > out[top[t]] || (out[top[t - 1]] == out[top[t]] && top[t - 1] < top[t]));
        if (kt > 0 && kt <= M) {
            EXPECT(cso_lstsq_cols(A, dtype, M, ld, top, kt, b, coef) == CSO_OK);
            for (int64_t t = 0; t < kt; ++t) EXPECT(isfinite(coef[t]));
        }
        free(top);
        free(out);
        free(coef);
    }
    free(idx);
    free(ord);
    free(val);
    free(b);
    free(A);
}

static void backward(int dtype, int64_t M, int64_t N, int64_t ld, int nthreads) { /* br / lace need N <= M (src/backward.jl:27-35) */
    void *A = make_dictionary(dtype, M, N, ld);
    double *b = make_signal(A, dtype, M, N, ld, 3);
    int64_t *idx = (int64_t *)malloc((size_t)(N + 1) * 8), nnz = -1;
    double *val = (double *)malloc((size_t)(N + 1) * 8);
    for (int lace = 0; lace <= 1; ++lace) {
        EXPECT(cso_br(A, dtype, M, N, ld, b, 1e-6, 1e300, 3, lace, idx, val, &nnz, nthreads) == CSO_OK);
        check_support(lace ? "lace" : "br", idx, val, nnz, N + 1, N, 1);
    }
    free(idx);
    free(val);
    free(b);
    free(A);
}

int main(void) {
    const int64_t shapes[][4] = {/* M, N, ld, k */ {32, 48, 32, 3}, {32, 64, 35, 3}, {24, 40, 24, 7}, {16, 16, 19, 8}, {9, 30, 9, 5}, {12, 5, 12, 4}, {8, 20, 8, 0}, {6, 9, 6, 1}};
    for (size_t s = 0; s < sizeof shapes / sizeof shapes[0]; ++s)
        for (int dtype = 0; dtype <= 1; ++dtype)
            for (int nthreads = 1; nthreads <= 3; nthreads += 2) family(dtype, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], nthreads);
    backward(CSO_F64, 20, 12, 20, 1);
    backward(CSO_F32, 20, 12, 23, 2);
    printf("oracle_driver: %s (%d failed expectation(s))\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
