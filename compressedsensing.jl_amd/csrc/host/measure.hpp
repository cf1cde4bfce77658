// host/measure.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// measurement entry points.
// ------------------------------------------------------------------------------------------ measurement
extern "C" int csmp_profile_enable(csmp_ctx* ctx, int on) {
    if (!ctx) return CSMP_EINVAL;
    ctx->prof = on != 0;
    ctx->prof_every = on > 1 ? on : 1;  // on = n > 1: time every n-th sweep launch
    ctx->prof_count = 0;
    return CSMP_OK;
}

extern "C" int csmp_profile_read(csmp_ctx* ctx, int64_t* sweep_launches, double* sweep_ms, int reset) {
    if (!ctx) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(sync_all(ctx));
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        ctx->prof_ms += ms;
        ctx->prof_n += 1;
    }
    ctx->ev_used = 0;
    if (sweep_launches) *sweep_launches = ctx->prof_n;
    if (sweep_ms) *sweep_ms = ctx->prof_ms;
    if (reset) {
        ctx->prof_n = 0;
        ctx->prof_ms = 0.0;
    }
    return CSMP_OK;
}

// what an event pair reads with NOTHING between its two records (the marker packets themselves): csmp_profile_read's sums carry one
// of these per timed launch, and a caller that wants launch durations subtracts it
extern "C" int csmp_profile_overhead(csmp_ctx* ctx, int reps, double* avg_ms) {
    if (!ctx || reps < 1 || !avg_ms) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    std::vector<hipEvent_t> ev((size_t)reps * 2);
    for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
    for (auto& e : ev) HIPCHECK(hipEventRecord(e, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (int i = 0; i < reps; ++i) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, ev[(size_t)i * 2], ev[(size_t)i * 2 + 1]));
        sum += ms;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    *avg_ms = sum / reps;
    return CSMP_OK;
}

extern "C" int csmp_bench_sweep(csmp_ctx* ctx, int variant, int reps, double* avg_ms) {
    if (!ctx || reps < 1) return CSMP_EINVAL;
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (variant != 0) return fail(ctx, CSMP_ESTATE, "bench_sweep: variant 0 (the product kernel) is the only one");
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(solver_ensure(ctx, 1, 1, false));
    ctx->s.begun = false;
    std::vector<double> r((size_t)ctx->M);
    uint64_t sd = 0x9E3779B97F4A7C15ull;
    for (auto& v : r) {
        sd = sd * 6364136223846793005ull + 1442695040888963407ull;
        v = ((double)(sd >> 11) / 9007199254740992.0) - 0.5;
    }
    CHECK(upload_b(ctx, r.data(), CSMP_F64));
    const bool was = ctx->prof;
    ctx->prof = false;
    for (int i = 0; i < 3; ++i) CHECK(launch_sweep(ctx, ctx->s.r, 0.0, 0, 0));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; ++i) CHECK(launch_sweep(ctx, ctx->s.r, 0.0, 0, 0));
    HIPCHECK(hipEventRecord(e1, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    float ms0 = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms0, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ctx->prof = was;
    if (avg_ms) *avg_ms = (double)ms0 / reps;
    return CSMP_OK;
}

extern "C" int csmp_sweep_config(const csmp_ctx* ctx, int* unit_loads, int* phases, int* workgroups, int* tick_workgroups, int64_t* lds_bytes, int* dynamic) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->dA) return CSMP_ESTATE;
    if (unit_loads) *unit_loads = ctx->sweep_U;
    if (phases) *phases = ctx->sweep_ph ? (ctx->Mv + ctx->sweep_KP - 1) / ctx->sweep_KP : 1;
    if (workgroups) *workgroups = ctx->sweep_grid;
    if (tick_workgroups) *tick_workgroups = ctx->tick_nblk > 0 ? ctx->tick_nblk : ctx->tick_grid;
    if (lds_bytes) *lds_bytes = (int64_t)ctx->sweep_lds;
    if (dynamic) *dynamic = ctx->sweep_dyn ? 1 : 0;
    return CSMP_OK;
}

extern "C" int csmp_tune(csmp_ctx* ctx, int key, int64_t value) {
    if (!ctx) return CSMP_EINVAL;
    if (value < 0 || value > (1 << 20)) return fail(ctx, CSMP_EINVAL, "csmp_tune: value out of range");
    switch (key) {
        case CSMP_TUNE_SWEEP_GRID: ctx->tune_sweep_grid = (int)value; break;
        case CSMP_TUNE_SWEEP_UNIT:
            if (value != 0 && value != 4 && value != 8 && value != 16) return fail(ctx, CSMP_EINVAL, "csmp_tune: unit loads must be 0, 4, 8 or 16");
            ctx->tune_sweep_U = (int)value;
            break;
        case CSMP_TUNE_TICK_GRID: ctx->tick_nblk = (int)value; break;  // (configure_sweep below: the dynamic sweep's grid limit)
        case CSMP_TUNE_SWEEP_DYN: ctx->tune_sweep_dyn = value ? 1 : 0; break;
        case CSMP_TUNE_CLAIM_POOLS:
            if (value < 1 || value > 4096) return fail(ctx, CSMP_EINVAL, "csmp_tune: claim pools must be 1..4096");
            ctx->claim_pools = (int)value;
            return CSMP_OK;
        case CSMP_TUNE_PIPELINES: ctx->tune_pipelines = value == 1 ? 1 : 0; return CSMP_OK;
        case CSMP_TUNE_TICK_ORDER: ctx->tick_sweep_first = value != 0; return CSMP_OK;
        case CSMP_TUNE_REBUILD_DIRECT: ctx->tune_rebuild_direct = value ? 1 : 0; return CSMP_OK;
        case CSMP_TUNE_SWAP_REFUSE: ctx->tune_swap_refuse = value ? 1 : 0; return CSMP_OK;
        case CSMP_TUNE_DIAG_SPLIT: ctx->tune_diag_split = value ? 1 : 0; return CSMP_OK;
        case CSMP_TUNE_BATCH_BUDGET_MIB: ctx->tune_batch_budget_mib = value; return CSMP_OK;
        default: return fail(ctx, CSMP_EINVAL, "csmp_tune: unknown key");
    }
    if (ctx->dA) {
        HIPCHECK(hipSetDevice(ctx->dev));
        HIPCHECK(sync_all(ctx));
        CHECK(configure_sweep(ctx));
    }
    return CSMP_OK;
}
