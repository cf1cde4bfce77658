// host/steps_twostage.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// the step-level functors of the two-stage solvers -- P = SP(A, b, k) / OMPR(A, b, k), sp_acquisition!(P, x) /
// oblivious_acquisition!(P, x, k), update!(P, x) -- over the job objects the drivers csmp_sp and csmp_ompr run (SpJob,
// host/gomp_sp.hpp; OmprJob, host/twostage.hpp): a host that steps gets the very phases the driver strings together.
// As with the other functors (host/steps_sharding.hpp) the solver owns x: the host reads it back with csmp_solver_state.
// ------------------------------------------------------------------------------------------ SP  (src/twostage.jl:42-83)
// SP(A, b, k): :54-61.  x0 (optional): the x the host will pass to update! -- any k-sparse vector: the reference recomputes the
// residual from x at the top of every acquisition (:68), so its VALUES matter, not only its support.
static int sp_functor_begin(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, const int64_t* idx0, const double* val0, int64_t nnz0) {
    if (2 * k > ctx->M) return fail(ctx, CSMP_ERANGE, "2k > length(b) is invalid for Subspace Pursuit");  // :55
    if (k > ctx->N) return fail(ctx, CSMP_ERANGE, "sp: k > number of atoms");
    if (nnz0 < 0 || nnz0 > 2 * k || (nnz0 > 0 && (!idx0 || !val0))) return fail(ctx, CSMP_EINVAL, "SP: bad initial x");
    SpJob& j = ctx->spjob;
    ctx->gate = nullptr;
    j.c = ctx;
    j.k = k;
    j.delta = 0.0;
    j.it = 0;
    j.maxiter = 0;
    j.rc = CSMP_OK;
    j.phase = SpJob::IDLE;
    if (!j.ev) HIPCHECK(hipEventCreateWithFlags(&j.ev, hipEventDisableTiming));
    CHECK(solver_ensure(ctx, (int)(2 * k), (int)(2 * k)));
    ctx->s.begun = false;
    j.screened = screened_on(ctx) && k <= 4096;
    if (j.screened) CHECK(screened_ensure(ctx));
    CHECK(upload_b(ctx, b, b_dtype));
    std::vector<std::pair<int64_t, double>> x0((size_t)nnz0);
    for (int64_t t = 0; t < nnz0; ++t) x0[(size_t)t] = {idx0[t], val0[t]};
    std::sort(x0.begin(), x0.end());
    j.xi.clear();
    j.xv.clear();
    for (auto& e : x0) {
        if (e.first < 0 || e.first >= ctx->N || (!j.xi.empty() && j.xi.back() == e.first)) return fail(ctx, CSMP_EINVAL, "SP: the indices of x must be distinct atoms");
        j.xi.push_back(e.first);
        j.xv.push_back(e.second);
    }
    if (nnz0 > 0) CHECK(upload_support(ctx, j.xi.data(), j.xv.data(), nnz0));  // r = b - A x (residual!, src/matchingpursuit.jl:158-161)
    j.resnorm = -1.0;
    return CSMP_OK;
}
// sp_acquisition!(P, x, kk) (:67-72): the kk atoms best correlated with the residual of x join it, least squares on the union
static int sp_functor_acquire(csmp_ctx* ctx, int64_t kk) {
    SpJob& j = ctx->spjob;
    const int64_t k = j.k;
    if (kk < 1 || (int64_t)j.xi.size() + kk > 2 * k || kk > ctx->N)
        return fail(ctx, CSMP_ERANGE, "sp_acquisition!: nnz(x) + k exceeds the 2k columns SP(A, b, k) holds");
    j.k = kk;  // (the selection count of this acquisition; restored below)
    j.oldnorm = -1.0;
    j.phase = SpJob::SELECT;
    int rc = sp_job_select(j, j.screened);
    while (rc == CSMP_OK && j.phase == SpJob::SELECT) {  // (an uncertified screened selection comes back as a second SELECT, exact)
        rc = sp_job_wait(j);
        if (rc == CSMP_OK) rc = sp_job_selected(j, SpJob::LS_FIRST);
    }
    if (rc == CSMP_OK) rc = sp_job_wait(j);
    if (rc == CSMP_OK) rc = sp_job_ls_done(j);
    j.k = k;
    j.phase = SpJob::IDLE;
    return rc;
}
// update!(P::SP, x) (:75-83): acquisition of k atoms, the nnz - k smallest coefficients leave, least squares on the k kept
static int sp_functor_update(csmp_ctx* ctx) {
    SpJob& j = ctx->spjob;
    if ((int64_t)j.xi.size() != j.k)  // :76 throws this string
        return fail(ctx, CSMP_ESTATE, "nnz(x) = " + std::to_string(j.xi.size()) + " \xe2\x89\xa0 " + std::to_string(j.k) + " = k");
    j.maxiter = j.it + 1;  // (the job stops after this update!)
    j.oldnorm = -1.0;      // (no "the prune handed the old support back" shortcut: the device residual must end as b - A x of the x returned)
    j.phase = SpJob::SELECT;
    int rc = sp_job_select(j, j.screened);
    while (rc == CSMP_OK && j.phase == SpJob::SELECT) {
        rc = sp_job_wait(j);
        if (rc == CSMP_OK) rc = sp_job_selected(j, SpJob::LS_UNION);
    }
    while (rc == CSMP_OK && j.phase != SpJob::DONE) {
        rc = sp_job_wait(j);
        if (rc == CSMP_OK) rc = sp_job_advance(j);
    }
    j.phase = SpJob::IDLE;
    return rc;
}

// ------------------------------------------------------------------------------------------ OMPR  (src/twostage.jl:110-180)
static int ompr_functor_begin(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, int64_t nnz0) {
    if (k > ctx->N || k > ctx->M) return fail(ctx, CSMP_ERANGE, "ompr: k exceeds size(A)");
    // (an x that is not empty has no factorisation to stand on: OMPR(A, b, k) starts from an EMPTY UpdatableQR, :124-129, and the
    // reference's own driver zeroes a short x before the acquisition, :187-190)
    if (nnz0 != 0) return fail(ctx, CSMP_EINVAL, "OMPR: x starts empty; fill it with oblivious_acquisition!(P, x, k)");
    return ompr_job(ctx).begin(ctx, b, b_dtype, k);
}
static int ompr_functor_acquire(csmp_ctx* ctx, int64_t kk) {
    OmprJob& j = ompr_job(ctx);
    if (!j.xi.empty()) return fail(ctx, CSMP_ESTATE, "oblivious_acquisition!(P::OMPR, x, k): x is not empty");
    if (kk != j.k) return fail(ctx, CSMP_ERANGE, "oblivious_acquisition!(P::OMPR, x, k): k must be the k of OMPR(A, b, k) (update! accepts nothing else, src/twostage.jl:135)");
    return j.acquire();
}
static int ompr_functor_update(csmp_ctx* ctx) {
    OmprJob& j = ompr_job(ctx);
    if ((int64_t)j.xi.size() != j.k)  // :135
        return fail(ctx, CSMP_ESTATE, "nnz(x) = " + std::to_string(j.xi.size()) + " \xe2\x89\xa0 " + std::to_string(j.k) + " = k");
    return j.update();
}

// current x, ||b - A x||, of an SP / OMPR functor (csmp_solver_state)
static int twostage_functor_state(csmp_ctx* ctx, int64_t* idx, double* val, int64_t* nnz, double* resnorm, int64_t* order, int* stop) {
    const bool issp = ctx->s.algo == CSMP_ALGO_SP;
    const std::vector<int64_t>& xi = issp ? ctx->spjob.xi : ompr_job(ctx).xi;
    const std::vector<double>& xv = issp ? ctx->spjob.xv : ompr_job(ctx).xv;
    for (size_t t = 0; t < xi.size(); ++t) {
        if (idx) idx[t] = xi[t];
        if (val) val[t] = xv[t];
        if (order) order[t] = xi[t];  // (these solvers keep no selection order: the sorted support)
    }
    if (nnz) *nnz = (int64_t)xi.size();
    if (stop) *stop = 0;
    if (resnorm) CHECK(residual_norm(ctx, resnorm));
    return CSMP_OK;
}

// sp_acquisition!(P, x, k) for the SP functor (src/twostage.jl:67-72); oblivious_acquisition!(P, x, k) for OMPR, OMP and GOMP
// (src/matchingpursuit.jl:207-216: residual of x, the k best atoms join -- those already there are skipped, src/util.jl:128-134 --,
// one least-squares solve)
extern "C" int csmp_solver_acquire(csmp_ctx* ctx, int64_t k) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_acquire: no solver begun");
    if (k < 1) return fail(ctx, CSMP_EINVAL, "solver_acquire: k < 1");
    HIPCHECK(hipSetDevice(ctx->dev));
    switch (ctx->s.algo) {
        case CSMP_ALGO_SP: return sp_functor_acquire(ctx, k);
        case CSMP_ALGO_OMPR: return ompr_functor_acquire(ctx, k);
        case CSMP_ALGO_OMP:
        case CSMP_ALGO_GOMP: return gomp_update(ctx, k, 0.0, 0, STOP_FULL, false);
        default: return fail(ctx, CSMP_EINVAL, "solver_acquire: MP and FR have no acquisition step");
    }
}
