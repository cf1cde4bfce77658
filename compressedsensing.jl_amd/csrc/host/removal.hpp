// host/removal.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// column removal (down-dates) and the explicit-inverse mode.
// ------------------------------------------------------------------------------------------ column removal
static int del_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.kcap > kTMaxCols) return fail(ctx, CSMP_ERANGE, "column removal supports at most 4095 columns");
    if (s.R2) return CSMP_OK;
    // all of the group or none of it: R2 alone would tell the next call that the group exists
    auto all = [&]() -> int {
        CHECK(dmalloc(ctx, &s.Gdel, (size_t)2 * s.kcap + 2));
        CHECK(dmalloc(ctx, &s.qdrop, s.Mpad));
        CHECK(dmalloc(ctx, &s.bwd, s.kcap));
        CHECK(dmalloc(ctx, &s.bwd_coef, s.kcap));
        CHECK(dmalloc(ctx, &s.bwd_info, 2));
        CHECK(dmalloc(ctx, &s.delmeta, 4));
        CHECK(dmalloc(ctx, &s.qsave, s.Mpad));
        CHECK(dmalloc(ctx, &s.delpos, 1));
        CHECK(dmalloc(ctx, &s.R2, (size_t)s.kcap * s.kcap));
        return CSMP_OK;
    };
    const int rc = all();
    if (rc != CSMP_OK) {
        dfree(s.R2); dfree(s.Gdel); dfree(s.qdrop); dfree(s.bwd); dfree(s.bwd_coef); dfree(s.bwd_info); dfree(s.delmeta); dfree(s.qsave); dfree(s.delpos);
    }
    return rc;
}

// remove_column!(AiQR, *delpos) -- the insertion position is read from device memory (-1: nothing happens)
static int launch_delete(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    const int threads = std::min(1024, ((s.kcap + 1 + 63) / 64) * 64);
    hipLaunchKernelGGL(k_qrdel_r, dim3(1), dim3(threads), 0, ctx->stream, (const double*)s.R, s.R2, s.kcap, s.z, s.sel, s.st,
                       (const int*)s.delpos, s.Gdel, s.scal, s.delmeta);
    HIPCHECK(hipGetLastError());
    std::swap(s.R, s.R2);
    hipLaunchKernelGGL(k_qrdel_q, dim3(s.G), dim3(64), 0, ctx->stream, s.Q, s.ldq, (const double*)s.Gdel, (const double*)s.scal,
                       (const int*)s.delmeta, s.r, s.qdrop, s.qsave);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

static int launch_delete_atom(csmp_ctx* ctx, int atom) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_find_pos, dim3(1), dim3(256), 0, ctx->stream, (const int*)s.sel, (const DevState*)s.st, atom, s.delpos);
    HIPCHECK(hipGetLastError());
    return launch_delete(ctx);
}

// ---- explicit-inverse mode (csmp_tinv.hpp): T = R^-1 kept next to R by the two-stage solvers
static int tinv_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    CHECK(del_ensure(ctx));
    if (s.T) return CSMP_OK;
    const size_t nch = (size_t)(s.kcap + kTChunk - 1) / kTChunk;
    auto all = [&]() -> int {  // (all of the group or none of it, as del_ensure)
        CHECK(dmalloc(ctx, &s.T2, (size_t)s.kcap * s.kcap));
        CHECK(dmalloc(ctx, &s.tpd, nch * s.kcap));
        CHECK(dmalloc(ctx, &s.tpn, nch * s.kcap));
        CHECK(dmalloc(ctx, &s.tmeta, 2));
        CHECK(dmalloc(ctx, &s.T, (size_t)s.kcap * s.kcap));
        // the strictly lower triangles stay ZERO for the life of the buffers (every writer writes entries on and above the diagonal
        // only): the removal's rotation chains read whole rows without a bounds test (k_tdel_apply)
        HIPCHECK(hipMemsetAsync(s.T, 0, (size_t)s.kcap * s.kcap * sizeof(double), ctx->stream));
        HIPCHECK(hipMemsetAsync(s.T2, 0, (size_t)s.kcap * s.kcap * sizeof(double), ctx->stream));
        return CSMP_OK;
    };
    const int rc = all();
    if (rc != CSMP_OK) {
        dfree(s.T); dfree(s.T2); dfree(s.tpd); dfree(s.tpn); dfree(s.tmeta);
    }
    return rc;
}
// T = R^-1 for the columns factorised so far
static int launch_tinv_build(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.kcap <= 257)
        hipLaunchKernelGGL((k_tinv_build<4, 4>), dim3(s.kcap), dim3(64), 0, ctx->stream, (const double*)s.R, s.kcap,
                           (const DevState*)s.st, s.T, s.tmeta);
    else if (s.kcap <= 1025)
        hipLaunchKernelGGL((k_tinv_build<16, 2>), dim3(s.kcap), dim3(64), 0, ctx->stream, (const double*)s.R, s.kcap,
                           (const DevState*)s.st, s.T, s.tmeta);
    else if (s.kcap <= 2049)
        hipLaunchKernelGGL((k_tinv_build_big<32>), dim3(s.kcap), dim3(64), 0, ctx->stream, (const double*)s.R, s.kcap,
                           (const DevState*)s.st, s.T, s.tmeta);
    else  // up to 4096 columns: 64 entries per lane
        hipLaunchKernelGGL((k_tinv_build_big<64>), dim3(s.kcap), dim3(64), 0, ctx->stream, (const double*)s.R, s.kcap,
                           (const DevState*)s.st, s.T, s.tmeta);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}
static int launch_tinv_mv(csmp_ctx* ctx, int mode) {
    Solver& s = ctx->s;
    const dim3 grid((s.kcap + 63) / 64, (s.kcap + kTChunk - 1) / kTChunk);
    hipLaunchKernelGGL(k_tinv_matvec, grid, dim3(64), 0, ctx->stream, (const double*)s.T, s.kcap, (const DevState*)s.st,
                       (const int*)s.tmeta, (const double*)s.z, (const double*)s.R, mode, s.tpd, s.tpn);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_tinv_fin, dim3(1), dim3(256), 0, ctx->stream, s.T, s.kcap, (const DevState*)s.st, s.tmeta,
                       (const double*)s.R, mode, (const double*)s.tpd, (const double*)s.tpn, s.bwd_coef, s.bwd);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}
// after launch_append: the column the append may have added enters T (no-op if it added none)
static int launch_tinv_append(csmp_ctx* ctx) { return launch_tinv_mv(ctx, 1); }
// x = T z (insertion order, s.bwd_coef) and the backward scores x^2 / gamma (s.bwd)
static int launch_tinv_solve(csmp_ctx* ctx) { return launch_tinv_mv(ctx, 0); }
// remove_column!(AiQR, *delpos) with the rotations taken from T.  keep_R == false (the two-stage solvers ompr / srr / rmp / foba / br,
// which take every coefficient from T z): R is NOT down-dated -- nothing of theirs reads an old column of R again (an append
// writes the new column and k_tinv_fin reads only that one), and walking 64 columns of R per workgroup with one column per lane
// was a third of the kernel.  The step-level csmp_solver_remove keeps R: its functor goes on with plain appends and back-solves.
static int launch_delete_t(csmp_ctx* ctx, bool keep_R = false) {
    Solver& s = ctx->s;
    if (s.kcap <= 1023) {
        const int threads = std::min(1024, ((s.kcap + 1 + 63) / 64) * 64);
        hipLaunchKernelGGL(k_tdel_prep<1>, dim3(1), dim3(threads), 0, ctx->stream, (const double*)s.T, s.kcap, (const double*)s.z, s.sel,
                           s.st, (const int*)s.delpos, s.Gdel, s.scal, s.delmeta, s.tmeta);
    } else {  // four columns per thread: supports up to 4096
        hipLaunchKernelGGL(k_tdel_prep<4>, dim3(1), dim3(1024), 0, ctx->stream, (const double*)s.T, s.kcap, (const double*)s.z, s.sel,
                           s.st, (const int*)s.delpos, s.Gdel, s.scal, s.delmeta, s.tmeta);
    }
    HIPCHECK(hipGetLastError());
    const int NB = (s.kcap + 63) / 64;
#define TDEL_APPLY(PART)                                                                                                          \
    hipLaunchKernelGGL(k_tdel_apply<PART>, dim3(s.G + 2 * NB + 1 + (s.kcap + 3) / 4), dim3(64), 0, ctx->stream, s.Q, s.ldq, s.G, (const double*)s.T, s.T2, \
                       (const double*)s.R, s.R2, s.kcap, NB, s.z, (const double*)s.Gdel, (const double*)s.scal,                   \
                       (const int*)s.delmeta, s.r, s.qdrop, s.qsave, keep_R ? 0 : 1)
    if (ctx->tune_diag_split) {
        TDEL_APPLY(1);
        TDEL_APPLY(2);
        TDEL_APPLY(3);
    } else {
        TDEL_APPLY(0);
    }
#undef TDEL_APPLY
    HIPCHECK(hipGetLastError());
    std::swap(s.T, s.T2);
    if (keep_R) std::swap(s.R, s.R2);
    return CSMP_OK;
}
static int launch_delete_atom_t(csmp_ctx* ctx, int atom, bool keep_R = false) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_find_pos, dim3(1), dim3(256), 0, ctx->stream, (const int*)s.sel, (const DevState*)s.st, atom, s.delpos);
    HIPCHECK(hipGetLastError());
    return launch_delete_t(ctx, keep_R);
}
// fetch_sorted in explicit-inverse mode: coefficients from T z, emitted in index order
// (resnorm != NULL: the residual norm travels in the same synchronisation)
static int fetch_sorted_t(csmp_ctx* ctx, std::vector<int64_t>& idx, std::vector<double>& val, double* resnorm = nullptr) {
    Solver& s = ctx->s;
    CHECK(launch_tinv_solve(ctx));
    hipLaunchKernelGGL(k_emit_sorted, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.bwd_coef, (const int*)s.sel,
                       (const DevState*)s.st, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap,
                       resnorm ? (const double*)s.r : (const double*)nullptr, (int)ctx->M, s.scal + 1);  // (||r||^2 in the same launch)
    HIPCHECK(hipGetLastError());
    idx.assign((size_t)s.outcap, 0);
    val.assign((size_t)s.outcap, 0.0);
    std::vector<int64_t> hi((size_t)s.outcap);
    std::vector<double> hv((size_t)s.outcap);
    int64_t hn = 0;
    double n2 = 0.0;
    PinFetch f(ctx);
    CHECK(f.begin((size_t)s.outcap * 16 + 64));
    CHECK(f.add(hi.data(), s.out_idx, (size_t)s.outcap * 8));
    CHECK(f.add(hv.data(), s.out_val, (size_t)s.outcap * 8));
    CHECK(f.add(&hn, s.out_nnz, 8));
    if (resnorm) CHECK(f.add(&n2, s.scal + 1, 8));
    CHECK(f.wait());
    idx.assign(hi.begin(), hi.begin() + hn);
    val.assign(hv.begin(), hv.begin() + hn);
    if (resnorm) *resnorm = std::sqrt(n2);
    return CSMP_OK;
}

static int ls_on_columns(csmp_ctx* ctx, const std::vector<int>& cols);  // (host/gomp_sp.hpp)
// dropindex!(x, AiQR, i) on the step-level solver (src/util.jl:137-161): atom leaves the support
extern "C" int csmp_solver_remove(csmp_ctx* ctx, int64_t atom) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_remove: no solver begun");
    if (ctx->s.algo == CSMP_ALGO_MP) return fail(ctx, CSMP_EINVAL, "solver_remove: MP keeps no factorisation");
    if (ctx->s.algo == CSMP_ALGO_FR) return fail(ctx, CSMP_EINVAL, "solver_remove: use csmp_srr / the backward step for FR");
    if (ctx->s.algo == CSMP_ALGO_SP || ctx->s.algo == CSMP_ALGO_OMPR) return fail(ctx, CSMP_EINVAL, "solver_remove: SP and OMPR choose the atoms that leave themselves (update!)");
    HIPCHECK(hipSetDevice(ctx->dev));
    if (ctx->s.kcap > kDelMaxCols) {
        // k_qrdel_r walks R with one thread per column in one workgroup (1023 columns).  Beyond that the down-date takes its
        // rotations from the explicit inverse T = R^-1, as the two-stage solvers do (csmp_tinv.hpp: up to 4095 columns); the
        // functor's appends do not maintain T, so it is rebuilt from R here (one launch, O(j^3 / 64) per lane: a step primitive)
        if (ctx->s.kcap > kTMaxCols) {
            // Beyond the 4095 columns the rotation kernels scan in one workgroup (GOMP's default capacity size(A,1) at M = 8192):
            // the atom leaves the list and the factorisation is rebuilt from the columns that stay, in their insertion order
            // (panel appends, O(M j^2)) -- dropindex! has no cap in the reference (src/util.jl:137-161), so neither has this.
            Solver& s = ctx->s;
            DevState hs;
            HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            std::vector<int> cols((size_t)std::max(hs.nsel, 0));
            if (hs.nsel > 0) HIPCHECK(hipMemcpy(cols.data(), s.sel, (size_t)hs.nsel * sizeof(int), hipMemcpyDeviceToHost));
            auto it = std::find(cols.begin(), cols.end(), (int)atom);
            if (it == cols.end()) return CSMP_OK;  // (absent: a no-op, as below)
            cols.erase(it);
            CHECK(ls_on_columns(ctx, cols));
            s.jh = (int)cols.size();
            return CSMP_OK;
        }
        CHECK(tinv_ensure(ctx));
        CHECK(launch_tinv_build(ctx));
        return launch_delete_atom_t(ctx, (int)atom, /*keep_R*/ true);
    }
    CHECK(del_ensure(ctx));
    return launch_delete_atom(ctx, (int)atom);
}

extern "C" int csmp_fr_scores(csmp_ctx* ctx, double* delta2) {
    if (!ctx || !delta2) return CSMP_EINVAL;
    if (!ctx->s.dvec) return fail(ctx, CSMP_ESTATE, "fr_scores: no forward-regression step has run");
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(hipMemcpyAsync(delta2, ctx->s.dvec, (size_t)ctx->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return CSMP_OK;
}

extern "C" int csmp_solver_state(csmp_ctx* ctx, int64_t* idx, double* val, int64_t* nnz, double* resnorm, int64_t* order,
                                 int* stop) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_state: no solver begun");
    HIPCHECK(hipSetDevice(ctx->dev));
    Solver& s = ctx->s;
    if (s.algo == CSMP_ALGO_SP || s.algo == CSMP_ALGO_OMPR) return twostage_functor_state(ctx, idx, val, nnz, resnorm, order, stop);
    if (resnorm) {
        hipLaunchKernelGGL(k_norm2, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.r, (int)ctx->M, s.scal);
        HIPCHECK(hipGetLastError());
        double n2 = 0.0;
        HIPCHECK(hipMemcpyAsync(&n2, s.scal, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        *resnorm = std::sqrt(n2);
    }
    {
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (s.algo != CSMP_ALGO_MP) s.jh = std::min(s.kcap, hs.nsel);  // the host's support bound snaps to the true count
        if (stop) *stop = hs.done & (STOP_EPS | STOP_STAG | STOP_FULL);
    }
    if (s.algo == CSMP_ALGO_MP) return mp_collect(ctx, nullptr, nullptr, 0, idx, val, nnz);
    CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));
    return download_result(ctx, s.outcap, idx, val, nnz, order);
}
