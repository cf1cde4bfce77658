// host/steps_sharding.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// step-level API (the Update functors), csmp_clone, column-sharded OMP, signal-sharding helpers.
// ------------------------------------------------------------------------------------------ step-level API
// (the two-stage functors SP and OMPR: host/steps_twostage.hpp)
static int sp_functor_begin(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, const int64_t* idx0, const double* val0, int64_t nnz0);
static int sp_functor_update(csmp_ctx* ctx);
static int ompr_functor_begin(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, int64_t nnz0);
static int ompr_functor_update(csmp_ctx* ctx);
static int twostage_functor_state(csmp_ctx* ctx, int64_t* idx, double* val, int64_t* nnz, double* resnorm, int64_t* order, int* stop);

extern "C" int csmp_solver_begin(csmp_ctx* ctx, int algo, const void* b, int b_dtype, int64_t kcap, const int64_t* idx0,
                                 const double* val0, int64_t nnz0) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || kcap < 1) return fail(ctx, CSMP_EINVAL, "solver_begin: bad arguments");
    if (algo != CSMP_ALGO_MP && algo != CSMP_ALGO_OMP && algo != CSMP_ALGO_GOMP && algo != CSMP_ALGO_FR && algo != CSMP_ALGO_SP && algo != CSMP_ALGO_OMPR)
        return fail(ctx, CSMP_EINVAL, "solver_begin: unknown algo");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    if (algo == CSMP_ALGO_SP || algo == CSMP_ALGO_OMPR) {  // kcap is the k of SP(A, b, k) / OMPR(A, b, k)
        ctx->s.begun = false;
        CHECK(algo == CSMP_ALGO_SP ? sp_functor_begin(ctx, b, b_dtype, kcap, idx0, val0, nnz0) : ompr_functor_begin(ctx, b, b_dtype, kcap, nnz0));
        ctx->s.algo = algo;
        ctx->s.begun = true;
        return CSMP_OK;
    }
    const int kc = algo == CSMP_ALGO_MP ? (int)kcap : (int)std::min<int64_t>(kcap, ctx->M);
    CHECK(solver_ensure(ctx, kc, kc, algo != CSMP_ALGO_MP));
    if (algo == CSMP_ALGO_FR) CHECK(fr_ensure(ctx));
    CHECK(upload_b(ctx, b, b_dtype));
    if (nnz0 > 0) {
        if (algo != CSMP_ALGO_MP) return fail(ctx, CSMP_EINVAL, "warm start is only defined for MP (src/matchingpursuit.jl:34)");
        CHECK(upload_support(ctx, idx0, val0, nnz0));
    }
    ctx->s.algo = algo;
    ctx->s.begun = true;
    return CSMP_OK;
}

static int gomp_update(csmp_ctx* ctx, int64_t l, double eps, int check_eps, int skipmask, bool block);

extern "C" int csmp_solver_step(csmp_ctx* ctx, int64_t l) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_step: no solver begun");
    HIPCHECK(hipSetDevice(ctx->dev));
    int rc = CSMP_OK;
    switch (ctx->s.algo) {
        case CSMP_ALGO_MP: return mp_step(ctx);
        case CSMP_ALGO_SP: return sp_functor_update(ctx);
        case CSMP_ALGO_OMPR: return ompr_functor_update(ctx);
        case CSMP_ALGO_OMP: {
            // update!(P::OMP, x) alone: no eps logic (that belongs to the omp driver)
            CHECK(launch_sweep(ctx, ctx->s.r, 0.0, 0, STOP_FULL));
            rc = launch_append(ctx, 1, 0, STOP_FULL);
            break;
        }
        case CSMP_ALGO_FR: {
            // update!(P::FR, x): nnz < n guard, acquisition_index! = argmax δ², addindex!, solve (src/forward.jl:88-95)
            const int skip = STOP_FULL | STOP_STAG;  // (a step that found no finite score would otherwise downdate rho2 twice)
            CHECK(launch_fr_sweep(ctx, ctx->s.jh == 0, -HUGE_VAL, skip));
            rc = launch_append(ctx, 3, 0, skip, false, -1.0, ctx->s.fr_grid);
            break;
        }
        default: rc = gomp_update(ctx, l, 0.0, 0, STOP_FULL, false);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------ shared dictionary
// A second context on the same GPU that BORROWS the resident dictionary of `src` (no copy): the independent
// P objects of the reference -- P1 = OMP(A, b1); P2 = OMP(A, b2) share A and nothing else
// (src/matchingpursuit.jl:44-60).  `src` must outlive the clone and keep its dictionary.
// Every per-context option (csmp_set_option) of src -> dst: what a clone starts from, and what the internal twins of the batch
// drivers are refreshed with on every call.  The resident Gram matrix is NOT inherited (8 N^2 bytes per context: a clone that wants
// it asks for it).
static void copy_options(csmp_ctx* dst, const csmp_ctx* src) {
    dst->pipeline = src->pipeline;
    dst->opt_in_flight = src->opt_in_flight;
    dst->opt_batch_cert = src->opt_batch_cert;
    dst->opt_batch_window = src->opt_batch_window;
    dst->opt_batch_screen = src->opt_batch_screen;
    dst->opt_screened = src->opt_screened;
    dst->tune_sweep_grid = src->tune_sweep_grid;  // (measurement overrides, csmp_tune: the twins of the batch drivers sweep like their parent)
    dst->tune_sweep_U = src->tune_sweep_U;
    dst->tune_sweep_dyn = src->tune_sweep_dyn;
    dst->tick_sweep_first = src->tick_sweep_first;
    dst->claim_pools = src->claim_pools;
    dst->tune_pipelines = src->tune_pipelines;
    dst->tune_pair_lds_kib = src->tune_pair_lds_kib;
    dst->tune_pair_split = src->tune_pair_split;
    dst->tune_sweep_lds_kib = src->tune_sweep_lds_kib;
    dst->sweep_lds_req = src->sweep_lds_req;
    dst->tune_sweep_short = src->tune_sweep_short;
    dst->tune_screen_static = src->tune_screen_static;
    dst->tune_phase_rows = src->tune_phase_rows;
    dst->tick_nblk = src->tick_nblk;
    dst->tune_swap_refuse = src->tune_swap_refuse;
    dst->tune_rebuild_direct = src->tune_rebuild_direct;
}
extern "C" int csmp_clone(csmp_ctx* src, csmp_ctx** out) {
    if (!src || !out) return CSMP_EINVAL;
    *out = nullptr;
    if (!src->dA) return fail(src, CSMP_ESTATE, "clone: no dictionary set (csmp_set_dictionary)");
    csmp_ctx* c = nullptr;
    const int rc = csmp_create(&c, src->dev);
    if (rc != CSMP_OK) {
        src->err = g_create_err;
        return rc;
    }
    c->dA = src->dA;
    c->ownA = false;
    c->share = src->share;  // (null for a borrowed device pointer: the caller keeps that alive)
    if (c->share) c->share->refs += 1;
    c->streamed = src->streamed;
    copy_options(c, src);
    c->dtype = src->dtype;
    c->M = src->M;
    c->N = src->N;
    c->ld = src->ld;
    c->Mv = src->Mv;
    c->col_offset = src->col_offset;
    const int rc2 = configure_sweep(c);
    if (rc2 != CSMP_OK) {
        src->err = c->err;
        csmp_destroy(c);
        return rc2;
    }
    *out = c;
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ column-sharded OMP
// See csmp_shard.hpp.  The ctx holds columns [col_offset, col_offset + N) of the global dictionary; a solve is
// csmp_solver_begin(CSMP_ALGO_OMP) on every rank, then per step csmp_shard_sweep -> the caller's all_gather of
// one record per rank -> csmp_shard_append, and csmp_solver_state at the end (identical on every rank).
extern "C" int csmp_shard_config(csmp_ctx* ctx, int64_t col_offset) {
    if (!ctx) return CSMP_EINVAL;
    if (col_offset < 0 || col_offset + ctx->N > 0x7fffffff) return fail(ctx, CSMP_ERANGE, "shard_config: global column indices must fit 31 bits");
    ctx->col_offset = col_offset;
    return CSMP_OK;
}

extern "C" int64_t csmp_shard_record_bytes(const csmp_ctx* ctx) {
    if (!ctx || !ctx->dA) return 0;
    return (int64_t)shard_record_bytes(ctx->Mv, ctx->dtype == CSMP_F32 ? 4 : 8);
}

static int shard_ready(csmp_ctx* ctx, const char* who) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun || ctx->s.algo != CSMP_ALGO_OMP)
        return fail(ctx, CSMP_ESTATE, (std::string(who) + ": begin the solve with csmp_solver_begin(CSMP_ALGO_OMP)").c_str());
    return CSMP_OK;
}

// steps 1-2: argmaxinner!(P) over the local columns (+ the driver's residual test of the previous iteration,
// src/matchingpursuit.jl:79, when check_eps != 0) and the rank's record, written to DEVICE memory at rec_dev
extern "C" int csmp_shard_sweep(csmp_ctx* ctx, double eps, int check_eps, void* rec_dev) {
    CHECK(shard_ready(ctx, "shard_sweep"));
    if (!rec_dev || !(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "shard_sweep: rec_dev == NULL or eps < 0");
    HIPCHECK(hipSetDevice(ctx->dev));
    Solver& s = ctx->s;
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL;
    CHECK(launch_sweep(ctx, s.r, eps, check_eps, skip));
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_shard_pack<float>, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.pval, (const int*)s.pidx,
                           ctx->sweep_grid, (const double*)s.cvec, (const float*)ctx->dA, ctx->ld, ctx->Mv, ctx->col_offset,
                           (const DevState*)s.st, skip, (char*)rec_dev);
    else
        hipLaunchKernelGGL(k_shard_pack<double>, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.pval, (const int*)s.pidx,
                           ctx->sweep_grid, (const double*)s.cvec, (const double*)ctx->dA, ctx->ld, ctx->Mv, ctx->col_offset,
                           (const DevState*)s.st, skip, (char*)rec_dev);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// steps 4-5: global arg-max over the nrec gathered records (DEVICE memory, csmp_shard_record_bytes apart),
// then update!(P::OMP, x)'s guards, add_column! and the residual update on the winning column
extern "C" int csmp_shard_append(csmp_ctx* ctx, const void* recs_dev, int nrec) {
    CHECK(shard_ready(ctx, "shard_append"));
    if (!recs_dev || nrec < 1) return fail(ctx, CSMP_EINVAL, "shard_append: recs_dev == NULL or nrec < 1");
    HIPCHECK(hipSetDevice(ctx->dev));
    Solver& s = ctx->s;
    const size_t es = ctx->dtype == CSMP_F32 ? 4 : 8;
    if (!s.extcol) HIPCHECK(hipMalloc(&s.extcol, (size_t)ctx->Mv * es));
    const int64_t rb = (int64_t)shard_record_bytes(ctx->Mv, es);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_shard_pick<float>, dim3(1), dim3(256), 0, ctx->stream, (const char*)recs_dev, nrec, rb, ctx->Mv,
                           (float*)s.extcol, s.cands, s.ncands, s.st);
    else
        hipLaunchKernelGGL(k_shard_pick<double>, dim3(1), dim3(256), 0, ctx->stream, (const char*)recs_dev, nrec, rb, ctx->Mv,
                           (double*)s.extcol, s.cands, s.ncands, s.st);
    HIPCHECK(hipGetLastError());
    return launch_append(ctx, 4, 0, STOP_EPS | STOP_STAG | STOP_FULL, false, 0.0, 0, s.extcol);
}

// ------------------------------------------------------------------------------------------ signal sharding helpers
// The data path of the signal-sharded batch has ONE exchange: every rank's results.  These
// three host-side helpers fix its layout so that any host language can run it over its own collective
// (torch.distributed / RCCL here, MPI.jl from Julia): contiguous blocks of signals per rank, and per signal one row
// of 2k + 1 Float64 = [idx_0 .. idx_{k-1} | val_0 .. val_{k-1} | nnz] (indices are exact in Float64 below 2^53).
// (csmp_shard_range, csmp_pack_results, csmp_unpack_results: host/hostonly.hpp -- no context, no HIP)
