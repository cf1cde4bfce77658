/*
 * csmp_oracle.c -- CPU restatement (Float64) of CompressedSensing.jl's mp / omp / gomp / sp.
 *
 * TEST INFRASTRUCTURE ONLY -- see csmp_oracle.h for the rules and the parity-pin status
 * ("parity unpinned" for step order / stagnation / eps-stop / tie-breaks: the reference's
 * tests do not cover them and the reference cannot be run here).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * The third-party UpdatableQRFactorizations.jl (v1.0.0, not vendored) is replaced by an
 * append-style Householder QR (hqr_*): the least-squares solution on a full-column-rank
 * support is unique, so any backward-stable QR returns the same coefficients to ~cond*eps;
 * the reference's "insert at sorted position" (src/util.jl:122-123) only permutes them, which
 * we reproduce by reporting coefficients in sorted-index order.
 */
#include "csmp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- element access */
static inline double a_at(const void *A, int dtype, int64_t ld, int64_t row, int64_t col) {
    return dtype == CSO_F32 ? (double)((const float *)A)[col * ld + row]
                            : ((const double *)A)[col * ld + row];
}

static void col_to_f64(const void *A, int dtype, int64_t M, int64_t ld, int64_t col, double *out) {
    if (dtype == CSO_F32) {
        const float *p = (const float *)A + col * ld;
        for (int64_t i = 0; i < M; ++i) out[i] = (double)p[i];
    } else {
        memcpy(out, (const double *)A + col * ld, (size_t)M * sizeof(double));
    }
}

static double col_dot(const void *A, int dtype, int64_t M, int64_t ld, int64_t col, const double *r) {
    double acc = 0.0;
    if (dtype == CSO_F32) {
        const float *p = (const float *)A + col * ld;
#pragma omp simd reduction(+ : acc)
        for (int64_t i = 0; i < M; ++i) acc += (double)p[i] * r[i];
    } else {
        const double *p = (const double *)A + col * ld;
#pragma omp simd reduction(+ : acc)
        for (int64_t i = 0; i < M; ++i) acc += p[i] * r[i];
    }
    return acc;
}

static double nrm2(const double *x, int64_t n) { /* norm(): 2-norm */
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += x[i] * x[i];
    return sqrt(s);
}

/* ---------------------------------------------------------------- sparse vector x
 * SparseVector{Float64,Int64}: sorted nzind + aligned nzval (src/matchingpursuit.jl:76). */
typedef struct {
    int64_t *idx;
    double *val;
    int64_t nnz, cap;
} spvec_t;

static int sp_init(spvec_t *x, int64_t cap) {
    x->idx = (int64_t *)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(int64_t));
    x->val = (double *)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(double));
    x->nnz = 0;
    x->cap = cap;
    return (x->idx && x->val) ? 0 : -1;
}
static void sp_free(spvec_t *x) {
    free(x->idx);
    free(x->val);
}
/* position of i in nzind, or -(insertion point)-1 */
static int64_t sp_find(const spvec_t *x, int64_t i) {
    int64_t lo = 0, hi = x->nnz;
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (x->idx[mid] < i)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < x->nnz && x->idx[lo] == i) return lo;
    return -lo - 1;
}
/* x[i] = v on a SparseVector: overwrite if stored, else insert keeping nzind sorted */
static int64_t sp_set(spvec_t *x, int64_t i, double v) {
    int64_t p = sp_find(x, i);
    if (p >= 0) {
        x->val[p] = v;
        return p;
    }
    p = -p - 1;
    if (x->nnz == x->cap) {
        int64_t nc = x->cap * 2 + 4;
        x->idx = (int64_t *)realloc(x->idx, (size_t)nc * sizeof(int64_t));
        x->val = (double *)realloc(x->val, (size_t)nc * sizeof(double));
        x->cap = nc;
    }
    memmove(x->idx + p + 1, x->idx + p, (size_t)(x->nnz - p) * sizeof(int64_t));
    memmove(x->val + p + 1, x->val + p, (size_t)(x->nnz - p) * sizeof(double));
    x->idx[p] = i;
    x->val[p] = v;
    x->nnz++;
    return p;
}

/* ---------------------------------------------------------------- residual!
 * src/matchingpursuit.jl:158-161: copyto!(r, b); mul!(r, A, x, -1, 1) */
void cso_residual(const void *A, int dtype, int64_t M, int64_t ld, const int64_t *idx,
                  const double *val, int64_t nnz, const double *b, double *r) {
    memcpy(r, b, (size_t)M * sizeof(double));
    for (int64_t t = 0; t < nnz; ++t) {
        const double xv = val[t];
        const int64_t c = idx[t];
        if (dtype == CSO_F32) {
            const float *p = (const float *)A + c * ld;
            for (int64_t i = 0; i < M; ++i) r[i] -= (double)p[i] * xv;
        } else {
            const double *p = (const double *)A + c * ld;
            for (int64_t i = 0; i < M; ++i) r[i] -= p[i] * xv;
        }
    }
}

/* ---------------------------------------------------------------- argmaxinner!
 * src/matchingpursuit.jl:181-185: mul!(Ar, A', r); Ar = abs(Ar); argmax(Ar)
 * Julia's argmax returns the FIRST maximal index. */
int64_t cso_sweep_abs(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *r,
                      double *out, int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t j = 0; j < N; ++j) out[j] = fabs(col_dot(A, dtype, M, ld, j, r));
    int64_t best = 0;
    for (int64_t j = 1; j < N; ++j)
        if (out[j] > out[best]) best = j;
    return best;
}

/* signed variant for mp's coefficient dot(A[:,i], r): src/matchingpursuit.jl:29 */
static double signed_dot(const void *A, int dtype, int64_t M, int64_t ld, int64_t col, const double *r) {
    return col_dot(A, dtype, M, ld, col, r);
}

/* ---------------------------------------------------------------- partialsortperm(v,1:k,rev=true)
 * src/matchingpursuit.jl:192.  Julia's Perm ordering: descending by value, ties broken by
 * ascending index. */
typedef struct {
    double v;
    int64_t i;
} vi_t;
static int cmp_desc(const void *pa, const void *pb) {
    const vi_t *a = (const vi_t *)pa, *b = (const vi_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->i > b->i) - (a->i < b->i);
}
static int cmp_asc(const void *pa, const void *pb) {
    const vi_t *a = (const vi_t *)pa, *b = (const vi_t *)pb;
    if (a->v < b->v) return -1;
    if (a->v > b->v) return 1;
    return (a->i > b->i) - (a->i < b->i);
}
void cso_topk_desc(const double *v, int64_t n, int64_t k, int64_t *out) {
    vi_t *t = (vi_t *)malloc((size_t)n * sizeof(vi_t));
    for (int64_t i = 0; i < n; ++i) {
        t[i].v = v[i];
        t[i].i = i;
    }
    qsort(t, (size_t)n, sizeof(vi_t), cmp_desc);
    if (k > n) k = n;
    for (int64_t i = 0; i < k; ++i) out[i] = t[i].i;
    free(t);
}

/* ---------------------------------------------------------------- append-style Householder QR
 * Stands in for UpdatableQR(T, n, k) + add_column! + ldiv! (call sites:
 * src/matchingpursuit.jl:58,112,175; src/util.jl:123) and for qr!/ldiv! in factorize!/solve!
 * (src/matchingpursuit.jl:219-227; src/twostage.jl:104-107). */
typedef struct {
    int64_t M, cap, j;
    double *V;    /* M x cap: reflector i lives in V[i*M + i .. i*M + M) */
    double *beta; /* cap */
    double *R;    /* cap x cap, column-major, upper triangular */
    double *Rt;   /* the same entries row-major (Rt[i*cap + t] = R[t*cap + i]): the back substitution walks ROWS of R, and a stride of
                   * `cap` doubles made a 4096-atom solve spend two minutes in cache misses.  Same numbers, same order of operations. */
    double *w;    /* M scratch */
    /* Q'b of the LAST right-hand side hqr_solve saw, with the number of reflectors applied to it: ldiv! after every append applies
     * H_0 .. H_{j-1} to the same b again -- the first j-1 of them reproduce what the previous call computed, bit for bit (the same
     * operations on the same numbers in the same order), so only the new reflectors are applied.  A time saver of the test
     * infrastructure, not a change of arithmetic: a 4096-atom solve at M = 4096 spends a third of its time there. */
    double *yb;
    const double *yb_src;
    int64_t yb_j;
} hqr_t;

static int hqr_init(hqr_t *F, int64_t M, int64_t cap) {
    if (cap > M) cap = M;
    if (cap < 1) cap = 1;
    F->M = M;
    F->cap = cap;
    F->j = 0;
    F->V = (double *)calloc((size_t)M * (size_t)cap, sizeof(double));
    F->beta = (double *)calloc((size_t)cap, sizeof(double));
    F->R = (double *)calloc((size_t)cap * (size_t)cap, sizeof(double));
    F->Rt = (double *)calloc((size_t)cap * (size_t)cap, sizeof(double));
    F->w = (double *)calloc((size_t)M, sizeof(double));
    F->yb = (double *)calloc((size_t)M, sizeof(double));
    F->yb_src = NULL;
    F->yb_j = 0;
    return (F->V && F->beta && F->R && F->Rt && F->w && F->yb) ? 0 : -1;
}
static void hqr_free(hqr_t *F) {
    free(F->V);
    free(F->beta);
    free(F->R);
    free(F->Rt);
    free(F->w);
    free(F->yb);
}

static void hqr_apply_qt(const hqr_t *F, double *y) { /* y <- Q' y */
    const int64_t M = F->M;
    for (int64_t i = 0; i < F->j; ++i) {
        const double *v = F->V + i * M;
        double s = 0.0;
        for (int64_t t = i; t < M; ++t) s += v[t] * y[t];
        s *= F->beta[i];
        for (int64_t t = i; t < M; ++t) y[t] -= s * v[t];
    }
}

/* add_column!(F, a): append a (length M) as the last column */
static int hqr_append(hqr_t *F, const double *a) {
    const int64_t M = F->M, j = F->j;
    if (j >= F->cap) return -1;
    double *w = F->w;
    memcpy(w, a, (size_t)M * sizeof(double));
    hqr_apply_qt(F, w);
    double nrm = 0.0;
    for (int64_t t = j; t < M; ++t) nrm += w[t] * w[t];
    nrm = sqrt(nrm);
    const double alpha = (w[j] > 0.0) ? -nrm : nrm;
    double *v = F->V + j * M;
    for (int64_t t = 0; t < j; ++t) v[t] = 0.0;
    for (int64_t t = j; t < M; ++t) v[t] = w[t];
    v[j] -= alpha;
    double vtv = 0.0;
    for (int64_t t = j; t < M; ++t) vtv += v[t] * v[t];
    F->beta[j] = (vtv > 0.0) ? 2.0 / vtv : 0.0;
    for (int64_t t = 0; t < j; ++t) F->R[j * F->cap + t] = w[t];
    F->R[j * F->cap + j] = alpha;
    for (int64_t t = 0; t <= j; ++t) F->Rt[t * F->cap + j] = F->R[j * F->cap + t];
    F->j = j + 1;
    return 0;
}

/* ldiv!(F, b): coefficient vector (insertion order) minimising ||A_S c - b|| */
static void hqr_solve(hqr_t *F, const double *b, double *c) {
    const int64_t M = F->M, j = F->j;
    double *y = F->yb;
    if (F->yb_src != b || F->yb_j > j) { /* another right-hand side (or a factorisation started over) */
        memcpy(y, b, (size_t)M * sizeof(double));
        F->yb_src = b;
        F->yb_j = 0;
    }
    for (int64_t i = F->yb_j; i < j; ++i) { /* the reflectors this right-hand side has not seen yet (hqr_apply_qt's loop body) */
        const double *v = F->V + i * M;
        double s = 0.0;
        for (int64_t t = i; t < M; ++t) s += v[t] * y[t];
        s *= F->beta[i];
        for (int64_t t = i; t < M; ++t) y[t] -= s * v[t];
    }
    F->yb_j = j;
    for (int64_t i = j - 1; i >= 0; --i) {
        double s = y[i];
        const double *row = F->Rt + i * F->cap;
        for (int64_t t = i + 1; t < j; ++t) s -= row[t] * c[t];
        c[i] = s / row[i];
    }
}

int cso_lstsq_cols(const void *A, int dtype, int64_t M, int64_t ld, const int64_t *cols, int64_t j,
                   const double *b, double *coef) {
    hqr_t F;
    if (hqr_init(&F, M, j) != 0) return CSO_ENOMEM;
    double *a = (double *)malloc((size_t)M * sizeof(double));
    for (int64_t t = 0; t < j; ++t) {
        col_to_f64(A, dtype, M, ld, cols[t], a);
        hqr_append(&F, a);
    }
    hqr_solve(&F, b, coef);
    free(a);
    hqr_free(&F);
    return CSO_OK;
}

/* ---------------------------------------------------------------- active set with updatable QR
 * x (sorted) + the QR in insertion order + the map between the two.
 * addindex!: src/util.jl:118-126;  ldiv!!: src/matchingpursuit.jl:170-176. */
typedef struct {
    spvec_t x;
    hqr_t F;
    int64_t *order; /* order[t] = atom appended t-th */
    int64_t norder;
    double *acol, *coef;
} active_t;

static int act_init(active_t *S, int64_t M, int64_t cap) {
    if (cap > M) cap = M;
    if (cap < 1) cap = 1;
    memset(S, 0, sizeof(*S));
    if (sp_init(&S->x, cap) != 0) return -1;
    if (hqr_init(&S->F, M, cap) != 0) return -1;
    S->order = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    S->acol = (double *)malloc((size_t)M * sizeof(double));
    S->coef = (double *)malloc((size_t)cap * sizeof(double));
    S->norder = 0;
    return (S->order && S->acol && S->coef) ? 0 : -1;
}
static void act_free(active_t *S) {
    sp_free(&S->x);
    hqr_free(&S->F);
    free(S->order);
    free(S->acol);
    free(S->coef);
}
/* addindex!(x, AiQR, a, i): only if i is not already in the support (src/util.jl:119) */
static int act_add(active_t *S, const void *A, int dtype, int64_t M, int64_t ld, int64_t i) {
    if (sp_find(&S->x, i) >= 0) return 0;
    if (S->F.j >= S->F.cap) return 0; /* capacity = M: a thin QR cannot take more columns */
    sp_set(&S->x, i, NAN);            /* x[i] = NaN placeholder (src/util.jl:120) */
    col_to_f64(A, dtype, M, ld, i, S->acol);
    hqr_append(&S->F, S->acol);
    S->order[S->norder++] = i;
    return 1;
}
/* ldiv!!(x.nzval, AiQR, b, r): x.nzval = AiQR \ b, in sorted-index order */
static void act_solve(active_t *S, const double *b) {
    hqr_solve(&S->F, b, S->coef);
    for (int64_t t = 0; t < S->norder; ++t) {
        int64_t p = sp_find(&S->x, S->order[t]);
        S->x.val[p] = S->coef[t];
    }
}

static void emit(const spvec_t *x, int64_t *idx, double *val, int64_t *nnz) {
    for (int64_t t = 0; t < x->nnz; ++t) {
        idx[t] = x->idx[t];
        val[t] = x->val[t];
    }
    *nnz = x->nnz;
}

/* ---------------------------------------------------------------- mp
 * src/matchingpursuit.jl:26-40.  update!: residual!; i = argmaxinner!; x[i] += dot(A[:,i], r) */
int cso_mp(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b,
           int64_t k, const int64_t *idx0, const double *val0, int64_t nnz0,
           int64_t *idx, double *val, int64_t *nnz, int nthreads) {
    spvec_t x;
    if (sp_init(&x, k + nnz0) != 0) return CSO_ENOMEM;
    for (int64_t t = 0; t < nnz0; ++t) sp_set(&x, idx0[t], val0[t]);
    double *r = (double *)malloc((size_t)M * sizeof(double));
    double *Ar = (double *)malloc((size_t)N * sizeof(double));
    for (int64_t it = 0; it < k; ++it) {
        cso_residual(A, dtype, M, ld, x.idx, x.val, x.nnz, b, r);     /* :27 */
        const int64_t i = cso_sweep_abs(A, dtype, M, N, ld, r, Ar, nthreads); /* :28 */
        const double d = signed_dot(A, dtype, M, ld, i, r);           /* :29 */
        const int64_t p = sp_find(&x, i);
        if (p >= 0)
            x.val[p] += d;
        else if (d != 0.0) /* SparseVector setindex! does not store a structural zero */
            sp_set(&x, i, d);
    }
    emit(&x, idx, val, nnz);
    free(r);
    free(Ar);
    sp_free(&x);
    return CSO_OK;
}

/* ---------------------------------------------------------------- omp
 * update!(P::OMP, x): src/matchingpursuit.jl:62-70;  driver omp(A,b,eps,k): :73-82 */
int cso_omp(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b,
            int64_t k, double eps, int64_t *idx, double *val, int64_t *nnz, int64_t *order,
            int nthreads) {
    if (!(eps >= 0.0)) return CSO_EINVAL; /* :74 */
    active_t S;
    if (act_init(&S, M, k) != 0) return CSO_ENOMEM; /* OMP(A,b,k): UpdatableQR(T,n,k) :58 */
    double *r = (double *)malloc((size_t)M * sizeof(double));
    double *Ar = (double *)malloc((size_t)N * sizeof(double));
    for (int64_t it = 0; it < k; ++it) { /* :77 */
        int progressed = 0;
        if (S.x.nnz < M) {                                                       /* :63 */
            cso_residual(A, dtype, M, ld, S.x.idx, S.x.val, S.x.nnz, b, r);      /* :64 */
            const int64_t i = cso_sweep_abs(A, dtype, M, N, ld, r, Ar, nthreads); /* :65 */
            if (sp_find(&S.x, i) < 0) {                                          /* :66 */
                progressed = act_add(&S, A, dtype, M, ld, i);                    /* :67 */
                act_solve(&S, b);                                                /* :68 */
            }
        }
        cso_residual(A, dtype, M, ld, S.x.idx, S.x.val, S.x.nnz, b, r);
        if (!(nrm2(r, M) >= eps)) break; /* :79  norm(residual!) >= eps || break */
        /* a no-op update! (support full, or the arg-max already selected: :63,:66) leaves x
         * unchanged, so every remaining iteration repeats it verbatim: stop here. */
        if (!progressed) break;
    }
    emit(&S.x, idx, val, nnz);
    if (order)
        for (int64_t t = 0; t < S.norder; ++t) order[t] = S.order[t];
    free(r);
    free(Ar);
    act_free(&S);
    return CSO_OK;
}

/* ---------------------------------------------------------------- gomp
 * update!(P::GOMP, x, l): src/matchingpursuit.jl:116-123;  driver: :126-139 */
static void gomp_update(active_t *S, const void *A, int dtype, int64_t M, int64_t N, int64_t ld,
                        const double *b, int64_t l, double *r, double *Ar, int64_t *top,
                        int nthreads) {
    if (!(S->x.nnz < M)) return;                                       /* :117 */
    cso_residual(A, dtype, M, ld, S->x.idx, S->x.val, S->x.nnz, b, r); /* :118 */
    cso_sweep_abs(A, dtype, M, N, ld, r, Ar, nthreads);                /* :190-191 */
    if (l > N) l = N;
    cso_topk_desc(Ar, N, l, top);                                      /* :192 */
    for (int64_t t = 0; t < l; ++t) act_add(S, A, dtype, M, ld, top[t]); /* :120, util.jl:129-134 */
    act_solve(S, b);                                                   /* :121 */
}

int cso_gomp(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b,
             int64_t l, int64_t k, double eps, int64_t *idx, double *val, int64_t *nnz,
             int64_t *order, int nthreads) {
    if (!(eps >= 0.0)) return CSO_EINVAL; /* :127 */
    if (l < 1) return CSO_EINVAL;
    active_t S;
    /* GOMP(A,b,l): QR capacity defaults to M (:108,:128); k only bounds what is ever added */
    if (act_init(&S, M, (k + l < M) ? k + l : M) != 0) return CSO_ENOMEM;
    double *r = (double *)malloc((size_t)M * sizeof(double));
    double *Ar = (double *)malloc((size_t)N * sizeof(double));
    int64_t *top = (int64_t *)malloc((size_t)(l > 0 ? l : 1) * sizeof(int64_t));
    for (int64_t it = 0; it < k / l; ++it) { /* :130 */
        gomp_update(&S, A, dtype, M, N, ld, b, l, r, Ar, top, nthreads);
        cso_residual(A, dtype, M, ld, S.x.idx, S.x.val, S.x.nnz, b, r);
        if (!(nrm2(r, M) >= eps)) break; /* :132 */
    }
    const int64_t rem = k % l; /* :134 */
    if (rem > 0) gomp_update(&S, A, dtype, M, N, ld, b, rem, r, Ar, top, nthreads); /* :135-137 */
    emit(&S.x, idx, val, nnz);
    if (order)
        for (int64_t t = 0; t < S.norder; ++t) order[t] = S.order[t];
    free(r);
    free(Ar);
    free(top);
    act_free(&S);
    return CSO_OK;
}

/* ---------------------------------------------------------------- sp
 * SP ctor: src/twostage.jl:54-61;  sp_acquisition!: :67-72;  update!: :75-83;  sp: :87-101;
 * solve!: :104-107 -> factorize! (src/matchingpursuit.jl:219-227): dense QR of A[:, nzind]. */
static void sp_solve(spvec_t *x, const void *A, int dtype, int64_t M, int64_t ld, const double *b) {
    cso_lstsq_cols(A, dtype, M, ld, x->idx, x->nnz, b, x->val);
}
static void sp_acquire(spvec_t *x, const void *A, int dtype, int64_t M, int64_t N, int64_t ld,
                       const double *b, int64_t k, double *r, double *Ar, int64_t *top,
                       int nthreads) {
    cso_residual(A, dtype, M, ld, x->idx, x->val, x->nnz, b, r); /* :68 */
    cso_sweep_abs(A, dtype, M, N, ld, r, Ar, nthreads);
    cso_topk_desc(Ar, N, k, top);                                /* :69 */
    for (int64_t t = 0; t < k; ++t) sp_set(x, top[t], NAN);      /* :70  @. x[i] = NaN */
    sp_solve(x, A, dtype, M, ld, b);                             /* :71 */
}

int cso_sp(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
           double delta, int64_t maxiter, int64_t *idx, double *val, int64_t *nnz, int64_t *iters,
           int nthreads) {
    if (2 * k > M) return CSO_ERANGE; /* :55 */
    if (k > N) return CSO_ERANGE;
    if (maxiter < 0) maxiter = 16 * k; /* :87 default */
    spvec_t x;
    if (sp_init(&x, 2 * k) != 0) return CSO_ENOMEM;
    double *r = (double *)malloc((size_t)M * sizeof(double));
    double *Ar = (double *)malloc((size_t)N * sizeof(double));
    int64_t *top = (int64_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(int64_t));
    vi_t *small = (vi_t *)malloc((size_t)(2 * k > 0 ? 2 * k : 1) * sizeof(vi_t));
    sp_acquire(&x, A, dtype, M, N, ld, b, k, r, Ar, top, nthreads); /* :90 */
    cso_residual(A, dtype, M, ld, x.idx, x.val, x.nnz, b, r);
    double resnorm = nrm2(r, M); /* :91 */
    int64_t it = 0;
    for (; it < maxiter;) { /* :92 */
        const double oldnorm = resnorm;
        /* update!(P::SP, x): :75-83 (nnz(x) == k holds by construction) */
        sp_acquire(&x, A, dtype, M, N, ld, b, k, r, Ar, top, nthreads); /* :77 */
        const int64_t drop = x.nnz - k;
        if (drop > 0) { /* :78-81: remove the (nnz-k) smallest |coef|, ties by position */
            for (int64_t t = 0; t < x.nnz; ++t) {
                small[t].v = fabs(x.val[t]);
                small[t].i = t;
            }
            qsort(small, (size_t)x.nnz, sizeof(vi_t), cmp_asc);
            char *kill = (char *)calloc((size_t)x.nnz, 1);
            for (int64_t t = 0; t < drop; ++t) kill[small[t].i] = 1;
            int64_t w = 0;
            for (int64_t t = 0; t < x.nnz; ++t)
                if (!kill[t]) {
                    x.idx[w] = x.idx[t];
                    x.val[w] = x.val[t];
                    ++w;
                }
            x.nnz = w;
            free(kill);
        }
        sp_solve(&x, A, dtype, M, ld, b); /* :82 */
        ++it;
        cso_residual(A, dtype, M, ld, x.idx, x.val, x.nnz, b, r);
        resnorm = nrm2(r, M);                              /* :95 */
        if (resnorm <= delta || oldnorm <= resnorm) break; /* :96 */
    }
    emit(&x, idx, val, nnz);
    if (iters) *iters = it;
    free(r);
    free(Ar);
    free(top);
    free(small);
    sp_free(&x);
    return CSO_OK;
}

/* ---------------------------------------------------------------- ompr (OMP with replacement)
 * OMPR ctor: src/twostage.jl:124-131;  update!(P::OMPR, x, eta=1): :134-180;  ompr: :184-202.
 * The support is first filled by oblivious_acquisition! (src/matchingpursuit.jl:207-216: the k atoms
 * best correlated with b, QR columns added in sorted order, one LS solve).  add_column! followed by
 * remove_column! on the updatable QR (:171-175) is restated as a fresh least-squares solve on the new
 * support: the LS solution on a full-rank support is unique. */
int cso_ompr(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
             double delta, int64_t maxiter, int64_t *idx, double *val, int64_t *nnz, int64_t *iters,
             int nthreads) {
    if (k < 1 || k > N || k > M) return CSO_ERANGE;
    if (maxiter < 0) maxiter = M; /* :185 maxiter = size(A, 1) */
    spvec_t x;
    if (sp_init(&x, k + 1) != 0) return CSO_ENOMEM;
    double *r = (double *)malloc((size_t)M * sizeof(double));
    double *Ar = (double *)malloc((size_t)N * sizeof(double));
    double *cs = (double *)malloc((size_t)N * sizeof(double));
    int64_t *top = (int64_t *)malloc((size_t)k * sizeof(int64_t));
    /* oblivious_acquisition!(P, x, k): :190 */
    memcpy(r, b, (size_t)M * sizeof(double)); /* residual of the empty x */
    cso_sweep_abs(A, dtype, M, N, ld, r, Ar, nthreads);
    cso_topk_desc(Ar, N, k, top);
    for (int64_t t = 0; t < k; ++t) sp_set(&x, top[t], NAN);
    cso_lstsq_cols(A, dtype, M, ld, x.idx, x.nnz, b, x.val);
    cso_residual(A, dtype, M, ld, x.idx, x.val, x.nnz, b, r);
    double resnorm = nrm2(r, M); /* :192 */
    int64_t it = 0;
    while (it < maxiter) { /* :193 */
        const double oldnorm = resnorm;
        /* update!(P, x): residual (r is current); Ar = x + eta * A'r with eta = 1 (:136-138) */
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
        for (int64_t j = 0; j < N; ++j) cs[j] = col_dot(A, dtype, M, ld, j, r);
        for (int64_t t = 0; t < x.nnz; ++t) cs[x.idx[t]] += x.val[t];
        /* arg-max of |Ar| over j not in the support, strict '>' from m = 0 (:139-151) */
        double m = 0.0;
        int64_t best = -1;
        for (int64_t j = 0; j < N; ++j) {
            if (sp_find(&x, j) >= 0) continue;
            const double f = fabs(cs[j]);
            if (f > m) {
                m = f;
                best = j;
            }
        }
        ++it;
        if (best >= 0) { /* :153-155: i == 0 -> return x */
            const int64_t qr_i = sp_set(&x, best, NAN);                 /* :158-159 */
            for (int64_t t = 0; t < x.nnz; ++t) x.val[t] = cs[x.idx[t]]; /* :165 */
            int64_t jmin = 0;                                           /* :167 argmin(abs, nzval): first minimum */
            for (int64_t t = 1; t < x.nnz; ++t)
                if (fabs(x.val[t]) < fabs(x.val[jmin])) jmin = t;
            memmove(x.idx + jmin, x.idx + jmin + 1, (size_t)(x.nnz - jmin - 1) * sizeof(int64_t)); /* :168-169 */
            memmove(x.val + jmin, x.val + jmin + 1, (size_t)(x.nnz - jmin - 1) * sizeof(double));
            x.nnz--;
            (void)qr_i; /* :171-175 QR add/remove == re-solve on the new support */
            cso_lstsq_cols(A, dtype, M, ld, x.idx, x.nnz, b, x.val); /* :178 */
        }
        cso_residual(A, dtype, M, ld, x.idx, x.val, x.nnz, b, r);
        resnorm = nrm2(r, M);                              /* :196 */
        if (resnorm <= delta || oldnorm <= resnorm) break; /* :197 */
    }
    emit(&x, idx, val, nnz);
    if (iters) *iters = it;
    free(r);
    free(Ar);
    free(cs);
    free(top);
    sp_free(&x);
    return CSO_OK;
}

/* ---------------------------------------------------------------- forward regression (OLS)
 * src/forward.jl:44-73 (fr / forward_step!), :75-82 (forward_δ!), :99-114 (ols_rescaling!).
 *
 * forward_step!: guard nnz(x) < n (:58); r = b - A x, stop unless norm(r) > max_ε (:59-61);
 * δ²_j = <a_j, r>² / rescaling_j with rescaling_j = |a_j|² - |Q_S' a_j|² (:104-112) and
 * δ²[x.nzind] = 0 (:80); (max, i) = findmax(δ²) = first maximum (:63); the atom is added only if
 * min_δ² < max (:64), otherwise the step fails and fr stops (:52).
 *
 * The reference recomputes Q'A (an M x M x N product) at every step.  |Q_S' a_j|² = Σ_i (q_i' a_j)²
 * is a sum over the columns of any orthonormal basis of span(A_S), so the restatement evaluates
 * the SAME sum progressively: one new term (q_new' a_j)² per step, q_new being the explicit last
 * column of this file's Householder Q.  (oracle_np.fr recomputes it from scratch with a fresh
 * LAPACK QR at every step; tests/test_oracle.py diffs the two.)
 *
 * Not defined by the reference's tests ("parity unpinned"): NaN scores (0/0 on an atom that lies
 * in span(A_S) with zero correlation) -- Julia's findmax would return the NaN; here, as on the
 * GPU, a NaN never wins a comparison. */
static void hqr_last_q(const hqr_t *F, double *q) { /* q = Q e_{j-1} = H_0 ... H_{j-1} e_{j-1} */
    const int64_t M = F->M, j = F->j;
    memset(q, 0, (size_t)M * sizeof(double));
    q[j - 1] = 1.0;
    for (int64_t i = j - 1; i >= 0; --i) {
        const double *v = F->V + i * M;
        double s = 0.0;
        for (int64_t t = i; t < M; ++t) s += v[t] * q[t];
        s *= F->beta[i];
        for (int64_t t = i; t < M; ++t) q[t] -= s * v[t];
    }
}

int cso_fr(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
           double max_eps, double min_delta, int64_t *idx, double *val, int64_t *nnz, int64_t *order,
           int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
    active_t S;
    if (act_init(&S, M, k) != 0) return CSO_ENOMEM;
    double *r = (double *)malloc((size_t)M * sizeof(double));
    double *q = (double *)malloc((size_t)M * sizeof(double));
    double *resc = (double *)malloc((size_t)N * sizeof(double)); /* P.rescaling */
    double *d2 = (double *)malloc((size_t)N * sizeof(double));   /* P.δ² */
    char *insupp = (char *)calloc((size_t)N, 1);
    const double min_d2 = min_delta * min_delta; /* :64 */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t j = 0; j < N; ++j) { /* sum!(abs2, rescaling', A)  :108 */
        double s = 0.0;
        for (int64_t i = 0; i < M; ++i) {
            const double a = a_at(A, dtype, ld, i, j);
            s += a * a;
        }
        resc[j] = s;
    }
    for (int64_t it = 0; it < k; ++it) { /* :51 */
        if (!(S.x.nnz < M)) break;       /* :58 */
        cso_residual(A, dtype, M, ld, S.x.idx, S.x.val, S.x.nnz, b, r); /* :59 */
        if (!(nrm2(r, M) > max_eps)) break;                             /* :60-61 */
        const int have_q = it > 0; /* every successful step appended exactly one column */
        if (have_q) hqr_last_q(&S.F, q);
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t j = 0; j < N; ++j) {
            if (have_q) {
                const double g = col_dot(A, dtype, M, ld, j, q);
                resc[j] -= g * g; /* :109-113, one more row of Q'A */
            }
            const double c = col_dot(A, dtype, M, ld, j, r); /* :77 */
            d2[j] = insupp[j] ? 0.0 : c * c / resc[j];       /* :79-80 */
        }
        int64_t best = 0; /* findmax: first maximum (:63); NaN never wins, see header */
        double bv = -1.0;
        for (int64_t j = 0; j < N; ++j)
            if (d2[j] > bv) {
                bv = d2[j];
                best = j;
            }
        if (!(min_d2 < bv)) break; /* :64,:69-71: the solve leaves x as it is */
        if (!act_add(&S, A, dtype, M, ld, best)) break;
        insupp[best] = 1;
        act_solve(&S, b); /* :67 */
    }
    emit(&S.x, idx, val, nnz);
    if (order)
        for (int64_t t = 0; t < S.norder; ++t) order[t] = S.order[t];
    free(r);
    free(q);
    free(resc);
    free(d2);
    free(insupp);
    act_free(&S);
    return CSO_OK;
}

/* ---------------------------------------------------------------- stepwise regression with replacement
 * srr(A,b,k,δ; maxiter=4k, initialization, l): src/twostage.jl:3-33, x starting empty.
 *   initialization 1: oblivious_acquisition! (src/matchingpursuit.jl:207-216): the k atoms best
 *                     correlated with b;   2: k forward-regression steps update!(P::FR, x) (:12-15)
 *   each iteration: l forward steps forward_step!(P,x,0,0) (src/forward.jl:56-73), then
 *   backward_step!(P,x,Inf,Inf) (src/backward.jl:51-67) until nnz(x) == k: the atom with the
 *   smallest δ²_i = x_i² / γ_i, γ = diag((R'R)^-1) (backward_δ!/get_gamma :70-83) leaves; first
 *   minimum in nzind order.  Stops when norm(r) <= δ or the residual did not decrease (:28-30).
 * The factorisation is rebuilt from scratch (in nzind order) whenever the support changes, and the
 * forward rescaling |a_j|² - |Q'a_j|² is recomputed from that fresh factorisation for every atom:
 * O(M N k) per step, small problems only -- deliberately none of the product's incremental tricks. */
typedef struct {
    const void *A;
    int dtype;
    int64_t M, N, ld;
    const double *b;
    int64_t *S; /* sorted support */
    int64_t n;
    double *coef, *r, *norm2;
    hqr_t F;
    int nthreads;
    double last_max_d2;
} srr_t;

static void srr_refit(srr_t *P) { /* QR of A[:, S] in nzind order; coef = AiQR \ b; r = b - A x */
    P->F.j = 0;
    P->F.yb_src = NULL; /* (the factorisation starts over: the cached Q'b belongs to the old reflectors) */
    P->F.yb_j = 0;
    double *a = (double *)malloc((size_t)P->M * sizeof(double));
    for (int64_t t = 0; t < P->n; ++t) {
        col_to_f64(P->A, P->dtype, P->M, P->ld, P->S[t], a);
        hqr_append(&P->F, a);
    }
    free(a);
    if (P->n > 0) hqr_solve(&P->F, P->b, P->coef);
    cso_residual(P->A, P->dtype, P->M, P->ld, P->S, P->coef, P->n, P->b, P->r);
}
static void srr_insert(srr_t *P, int64_t i) {
    int64_t p = P->n;
    while (p > 0 && P->S[p - 1] > i) {
        P->S[p] = P->S[p - 1];
        --p;
    }
    P->S[p] = i;
    P->n += 1;
}
static int srr_in(const srr_t *P, int64_t i) {
    for (int64_t t = 0; t < P->n; ++t)
        if (P->S[t] == i) return 1;
    return 0;
}
/* forward_step!(P, x, max_eps, min_delta) with the rescaling from scratch; returns 1 if an atom was added */
static int srr_forward(srr_t *P, double max_eps, double min_d2, int guarded) {
    if (!(P->n < P->M)) return 0;
    if (guarded && !(nrm2(P->r, P->M) > max_eps)) return 0;
    const int64_t M = P->M, N = P->N, n = P->n;
    double *d2 = (double *)malloc((size_t)N * sizeof(double));
#pragma omp parallel num_threads(P->nthreads)
    {
        double *w = (double *)malloc((size_t)M * sizeof(double));
#pragma omp for schedule(static)
        for (int64_t j = 0; j < N; ++j) {
            col_to_f64(P->A, P->dtype, M, P->ld, j, w);
            double c = 0.0;
            for (int64_t t = 0; t < M; ++t) c += w[t] * P->r[t];
            hqr_apply_qt(&P->F, w); /* first n entries = Q_S' a_j */
            double resc = P->norm2[j];
            for (int64_t t = 0; t < n; ++t) resc -= w[t] * w[t];
            d2[j] = c * c / resc;
        }
        free(w);
    }
    for (int64_t t = 0; t < n; ++t) d2[P->S[t]] = 0.0;
    int64_t best = 0;
    double bv = -1.0;
    for (int64_t j = 0; j < N; ++j)
        if (d2[j] > bv) {
            bv = d2[j];
            best = j;
        }
    free(d2);
    P->last_max_d2 = bv; /* maximum(P.δ²) after the step (foba, src/stepwise.jl:52) */
    if (guarded && !(min_d2 < bv)) return 0;
    if (srr_in(P, best)) return 0; /* addindex! is a no-op for an atom already in the support */
    srr_insert(P, best);
    srr_refit(P);
    return 1;
}
/* backward_step!(P, x, max_eps, max_delta): drops the atom of least δ² = x_i²/γ_i if
 * sqrt(min + |r|²) < max_eps and min < max_delta² (src/backward.jl:58); returns 1 if one was dropped */
static int srr_backward_sel(srr_t *P, double max_eps, double max_d2, int lace);
static int srr_backward_thr(srr_t *P, double max_eps, double max_d2) { return srr_backward_sel(P, max_eps, max_d2, 0); }
/* lace != 0: the atom of least |x_i| is the candidate (LACE, src/backward.jl:247-270) and its δ² decides */
static int srr_backward_sel(srr_t *P, double max_eps, double max_d2, int lace) {
    const int64_t n = P->n;
    if (!(n > 0)) return 0;
    const double normr = nrm2(P->r, P->M);
    double bkey = INFINITY;
    double *y = (double *)malloc((size_t)n * sizeof(double));
    int64_t best = -1;
    double bv = INFINITY;
    for (int64_t p = 0; p < n; ++p) { /* γ_p = |R^-T e_p|² */
        for (int64_t t = 0; t < n; ++t) y[t] = 0.0;
        y[p] = 1.0 / P->F.R[p * P->F.cap + p];
        double g = y[p] * y[p];
        for (int64_t i = p + 1; i < n; ++i) {
            double s = 0.0;
            for (int64_t t = p; t < i; ++t) s += P->F.R[i * P->F.cap + t] * y[t];
            y[i] = -s / P->F.R[i * P->F.cap + i];
            g += y[i] * y[i];
        }
        const double d = P->coef[p] * P->coef[p] / g;
        const double key = lace ? fabs(P->coef[p]) : d;
        if (key < bkey) { /* findmin / argmin(abs, ·): first minimum */
            bkey = key;
            bv = d;
            best = p;
        }
    }
    free(y);
    if (best < 0) return 0; /* all NaN */
    if (!(sqrt(bv + normr * normr) < max_eps && bv < max_d2)) return 0; /* :58 */
    for (int64_t t = best; t + 1 < n; ++t) P->S[t] = P->S[t + 1];
    P->n -= 1;
    srr_refit(P);
    return 1;
}

static int srr_backward(srr_t *P) { return srr_backward_thr(P, INFINITY, INFINITY); }

static int srr_core(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
                    double delta, int64_t maxiter, int initialization, const int64_t *init, int64_t l, int64_t *idx,
                    double *val, int64_t *nnz, int64_t *iters, int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (nthreads > 16) nthreads = 16;
#else
    nthreads = 1;
#endif
    if (k < 1 || k > N || k + l > M || l < 1) return CSO_ERANGE;
    if (initialization != 1 && initialization != 2 && !(initialization == 3 && init)) return CSO_EINVAL;
    if (maxiter < 0) maxiter = 4 * k; /* :5 */
    srr_t P;
    memset(&P, 0, sizeof P);
    P.A = A; P.dtype = dtype; P.M = M; P.N = N; P.ld = ld; P.b = b; P.nthreads = nthreads;
    P.S = (int64_t *)malloc((size_t)(k + l + 1) * sizeof(int64_t));
    P.coef = (double *)calloc((size_t)(k + l + 1), sizeof(double));
    P.r = (double *)malloc((size_t)M * sizeof(double));
    P.norm2 = (double *)malloc((size_t)N * sizeof(double));
    if (hqr_init(&P.F, M, k + l) != 0) return CSO_ENOMEM;
    for (int64_t j = 0; j < N; ++j) {
        double s = 0.0;
        for (int64_t i = 0; i < M; ++i) {
            const double a = a_at(A, dtype, ld, i, j);
            s += a * a;
        }
        P.norm2[j] = s;
    }
    memcpy(P.r, b, (size_t)M * sizeof(double));
    if (initialization == 1) { /* oblivious_acquisition!(P, x, k) */
        double *Ar = (double *)malloc((size_t)N * sizeof(double));
        int64_t *top = (int64_t *)malloc((size_t)k * sizeof(int64_t));
        cso_sweep_abs(A, dtype, M, N, ld, P.r, Ar, nthreads);
        cso_topk_desc(Ar, N, k, top);
        for (int64_t t = 0; t < k; ++t) srr_insert(&P, top[t]);
        srr_refit(&P);
        free(Ar);
        free(top);
    } else if (initialization == 3) { /* random_acquisition!(P, x, k), src/matchingpursuit.jl:195-204: the k indices are the
                                       * caller's draw (the reference takes them from Julia's RNG), sorted, appended in order */
        for (int64_t t = 0; t < k; ++t) srr_insert(&P, init[t]);
        srr_refit(&P);
    } else { /* k times update!(P::FR, x): src/forward.jl:88-95 (no residual / decrease guards) */
        for (int64_t t = 0; t < k; ++t) srr_forward(&P, 0.0, 0.0, 0);
    }
    double resnorm = nrm2(P.r, M); /* :18 */
    int64_t it = 0;
    while (it < maxiter) { /* :19 */
        const double oldnorm = resnorm;
        for (int64_t s = 0; s < l; ++s)
            if (!srr_forward(&P, 0.0, 0.0, 1)) break; /* :21-23 */
        while (P.n > k)
            if (!srr_backward(&P)) break; /* :24-26 */
        resnorm = nrm2(P.r, M);
        ++it;
        if (resnorm <= delta || oldnorm <= resnorm) break; /* :28-30 */
    }
    for (int64_t t = 0; t < P.n; ++t) {
        idx[t] = P.S[t];
        val[t] = P.coef[t];
    }
    *nnz = P.n;
    if (iters) *iters = it;
    free(P.S);
    free(P.coef);
    free(P.r);
    free(P.norm2);
    hqr_free(&P.F);
    return CSO_OK;
}
int cso_srr(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
            double delta, int64_t maxiter, int initialization, int64_t l, int64_t *idx, double *val,
            int64_t *nnz, int64_t *iters, int nthreads) {
    if (initialization == 3) return CSO_EINVAL; /* the draw is the caller's: cso_srr_from */
    return srr_core(A, dtype, M, N, ld, b, k, delta, maxiter, initialization, NULL, l, idx, val, nnz, iters, nthreads);
}
/* srr with initialization = 3: `init` holds the k distinct atoms random_acquisition! would have drawn (any order) */
int cso_srr_from(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
                 double delta, int64_t maxiter, const int64_t *init, int64_t l, int64_t *idx, double *val,
                 int64_t *nnz, int64_t *iters, int nthreads) {
    if (!init || k < 1) return CSO_EINVAL;
    int64_t *srt = (int64_t *)malloc((size_t)k * sizeof(int64_t));
    memcpy(srt, init, (size_t)k * sizeof(int64_t));
    for (int64_t a = 1; a < k; ++a) { /* sort!(ind) */
        const int64_t v = srt[a];
        int64_t c = a - 1;
        for (; c >= 0 && srt[c] > v; --c) srt[c + 1] = srt[c];
        srt[c + 1] = v;
    }
    int bad = srt[0] < 0 || srt[k - 1] >= N;
    for (int64_t a = 1; a < k; ++a) bad |= srt[a] == srt[a - 1];
    const int rc = bad ? CSO_EINVAL
                       : srr_core(A, dtype, M, N, ld, b, k, delta, maxiter, 3, srt, l, idx, val, nnz, iters, nthreads);
    free(srt);
    return rc;
}


/* ---------------------------------------------------------------- relevance matching pursuit, FoBa
 * src/stepwise.jl: rmp(A,b,δ,maxiter) :5-26, rmp(A,b,k) :32-43, foba(A,b,δ) :47-56 (x starting empty).
 * All three are loops over forward_step! / backward_step! of the StepwiseRegression object. */
static int stepwise_init(srr_t *P, const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b,
                         int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (nthreads > 16) nthreads = 16;
#else
    nthreads = 1;
#endif
    memset(P, 0, sizeof *P);
    P->A = A; P->dtype = dtype; P->M = M; P->N = N; P->ld = ld; P->b = b; P->nthreads = nthreads;
    const int64_t cap = (M < N ? M : N) + 1;
    P->S = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    P->coef = (double *)calloc((size_t)cap, sizeof(double));
    P->r = (double *)malloc((size_t)M * sizeof(double));
    P->norm2 = (double *)malloc((size_t)N * sizeof(double));
    if (hqr_init(&P->F, M, cap) != 0) return CSO_ENOMEM;
    for (int64_t j = 0; j < N; ++j) {
        double s = 0.0;
        for (int64_t i = 0; i < M; ++i) {
            const double a = a_at(A, dtype, ld, i, j);
            s += a * a;
        }
        P->norm2[j] = s;
    }
    memcpy(P->r, b, (size_t)M * sizeof(double));
    return CSO_OK;
}
static void stepwise_finish(srr_t *P, int64_t *idx, double *val, int64_t *nnz) {
    for (int64_t t = 0; t < P->n; ++t) {
        idx[t] = P->S[t];
        val[t] = P->coef[t];
    }
    *nnz = P->n;
    free(P->S);
    free(P->coef);
    free(P->r);
    free(P->norm2);
    hqr_free(&P->F);
}
/* !(xt ≈ x): isapprox on the dense images, rtol = sqrt(eps) (Julia default) */
static int x_changed(const srr_t *P, const int64_t *S0, const double *c0, int64_t n0) {
    double d2 = 0.0, na = 0.0, nb = 0.0;
    int64_t i = 0, j = 0;
    while (i < P->n || j < n0) {
        double a = 0.0, b = 0.0;
        if (j >= n0 || (i < P->n && P->S[i] < S0[j])) a = P->coef[i++];
        else if (i >= P->n || S0[j] < P->S[i]) b = c0[j++];
        else { a = P->coef[i++]; b = c0[j++]; }
        d2 += (a - b) * (a - b); na += a * a; nb += b * b;
    }
    const double mx = sqrt(na > nb ? na : nb);
    return !(sqrt(d2) <= 1.4901161193847656e-08 * mx);
}

int cso_rmp_delta(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, double delta,
                  int64_t maxiter, int64_t *idx, double *val, int64_t *nnz, int nthreads) {
    srr_t P;
    if (stepwise_init(&P, A, dtype, M, N, ld, b, nthreads) != 0) return CSO_ENOMEM;
    const int64_t cap = (M < N ? M : N) + 1;
    int64_t *S0 = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    double *c0 = (double *)malloc((size_t)cap * sizeof(double));
    int64_t n0 = 0;
    const double d2 = delta * delta;
    for (int64_t it = 0; it < maxiter; ++it) { /* :10 */
        for (int64_t s = 0; s < M; ++s)        /* :12-14 */
            if (!srr_forward(&P, 0.0, d2, 1)) break;
        if (!x_changed(&P, S0, c0, n0)) break; /* :15 */
        n0 = P.n;
        memcpy(S0, P.S, (size_t)n0 * sizeof(int64_t));
        memcpy(c0, P.coef, (size_t)n0 * sizeof(double));
        for (int64_t s = P.n; s >= 1; --s) /* :18-20 */
            if (!srr_backward_thr(&P, INFINITY, d2)) break;
        if (!x_changed(&P, S0, c0, n0)) break; /* :21 */
        n0 = P.n;
        memcpy(S0, P.S, (size_t)n0 * sizeof(int64_t));
        memcpy(c0, P.coef, (size_t)n0 * sizeof(double));
    }
    free(S0);
    free(c0);
    stepwise_finish(&P, idx, val, nnz);
    return CSO_OK;
}

int cso_rmp_k(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, int64_t k,
              int64_t *idx, double *val, int64_t *nnz, int nthreads) {
    srr_t P;
    if (stepwise_init(&P, A, dtype, M, N, ld, b, nthreads) != 0) return CSO_ENOMEM;
    for (int64_t s = 0; s < M; ++s) /* :36-38 */
        if (!srr_forward(&P, 0.0, 0.0, 1)) break;
    for (int64_t s = P.n; s >= k + 1; --s) /* :39-41 */
        if (!srr_backward_thr(&P, INFINITY, INFINITY)) break;
    stepwise_finish(&P, idx, val, nnz);
    return CSO_OK;
}

int cso_foba(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, double delta,
             int64_t *idx, double *val, int64_t *nnz, int nthreads) {
    srr_t P;
    if (stepwise_init(&P, A, dtype, M, N, ld, b, nthreads) != 0) return CSO_ENOMEM;
    const double d2 = delta * delta;
    for (int64_t s = 0; s < M; ++s) { /* :50 */
        if (!srr_forward(&P, 0.0, d2, 1)) break;              /* :51 */
        const double half = sqrt(P.last_max_d2) / 2.0;        /* :52-53 */
        while (srr_backward_thr(&P, INFINITY, half * half)) { /* :53 */
        }
    }
    stepwise_finish(&P, idx, val, nnz);
    return CSO_OK;
}


/* ---------------------------------------------------------------- backward regression, LACE
 * br(A,b,max_eps,max_delta,k) src/backward.jl:27-35 (fbr :154-162 is the same algorithm on the normal
 * equations) and lace(A,b,eps,delta,k) :233-242: start from the least-squares solution on ALL N <= M
 * columns, then backward steps until k atoms are left or a threshold stops them. */
int cso_br(const void *A, int dtype, int64_t M, int64_t N, int64_t ld, const double *b, double max_eps,
           double max_delta, int64_t k, int lace, int64_t *idx, double *val, int64_t *nnz, int nthreads) {
    if (N > M) return CSO_ERANGE;
    srr_t P;
    if (stepwise_init(&P, A, dtype, M, N, ld, b, nthreads) != 0) return CSO_ENOMEM;
    for (int64_t j = 0; j < N; ++j) P.S[j] = j;
    P.n = N;
    srr_refit(&P);
    for (int64_t s = N; s >= k + 1; --s)
        if (!srr_backward_sel(&P, max_eps, max_delta * max_delta, lace)) break;
    stepwise_finish(&P, idx, val, nnz);
    return CSO_OK;
}
