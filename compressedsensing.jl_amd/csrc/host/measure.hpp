// host/measure.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// measurement entry points.
// ------------------------------------------------------------------------------------------ measurement
extern "C" int csmp_profile_enable(csmp_ctx* ctx, int on) {
    if (!ctx) return CSMP_EINVAL;
    ctx->prof = on != 0;
    ctx->prof_every = on > 1 ? on : 1;  // on = n > 1: time every n-th sweep launch
    ctx->prof_count = 0;
    ctx->prof_first = ctx->prof_last = -1;
    if (ctx->prof) {
        HIPCHECK(hipSetDevice(ctx->dev));
        if (!ctx->prof_ref) HIPCHECK(hipEventCreate(&ctx->prof_ref));
        HIPCHECK(hipEventRecord(ctx->prof_ref, ctx->stream));
    }
    return CSMP_OK;
}

// The timed launches as ONE window.  csmp_omp_batch runs two pipelines side by side from two signals on (host/forward.hpp, host/omp.hpp): their
// sweep launches overlap, so a launch's own duration says nothing about the bandwidth -- the bytes of BOTH streams' launches move
// during it.  This call reports, over this context and its twin: the launches from the first to the last sampled one on each stream
// (all of them, sampled or not), the time from the earliest of their start events to the latest of their end events (one clock:
// HIP events on the two streams, measured from the event csmp_profile_enable recorded), and the mean duration of a sampled launch.
// Launches outside a stream's first .. last sampled one are not counted although part of their traffic may fall into the window:
// bytes / window never overstates.  Call it BEFORE csmp_profile_read (which consumes the events).
extern "C" int csmp_profile_window(csmp_ctx* ctx, int64_t* launches, double* window_ms, double* mean_launch_ms, int* streams) {
    if (!ctx) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(sync_all(ctx));
    csmp_ctx* cc[2] = {ctx, ctx->twins[0]};
    int64_t n = 0, pairs = 0;
    double t0 = 0.0, t1 = 0.0, dur = 0.0;
    int ns = 0;
    for (csmp_ctx* c : cc) {
        if (!c || c->ev_used < 2 || c->prof_first < 0 || !ctx->prof_ref) continue;
        if (c != ctx) HIPCHECK(hipStreamSynchronize(c->stream));
        float s = 0.f, e = 0.f;
        HIPCHECK(hipEventElapsedTime(&s, ctx->prof_ref, c->ev[0]));
        HIPCHECK(hipEventElapsedTime(&e, ctx->prof_ref, c->ev[c->ev_used - 1]));
        if (ns == 0 || s < t0) t0 = s;
        if (ns == 0 || e > t1) t1 = e;
        n += c->prof_last - c->prof_first + 1;
        for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
            dur += ms;
            pairs += 1;
        }
        ns += 1;
    }
    if (launches) *launches = n;
    if (window_ms) *window_ms = ns ? t1 - t0 : 0.0;
    if (mean_launch_ms) *mean_launch_ms = pairs ? dur / (double)pairs : 0.0;
    if (streams) *streams = ns;
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
    ctx->prof_first = ctx->prof_last = -1;
    if (csmp_ctx* tw = ctx->twins[0]) {  // (the second pipeline of csmp_omp_batch: its sampled launches count like this context's)
        if (tw->ev_used >= 2) HIPCHECK(hipStreamSynchronize(tw->stream));
        for (size_t i = 0; i + 1 < tw->ev_used; i += 2) {
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, tw->ev[i], tw->ev[i + 1]));
            ctx->prof_ms += ms;
            ctx->prof_n += 1;
        }
        tw->ev_used = 0;
        tw->prof_first = tw->prof_last = -1;
    }
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

// what the library holds right now, process-wide (host/track.hpp): bytes and blocks of device memory, bytes of page-locked host memory,
// host ranges registered with the device, events, streams
extern "C" int csmp_live_resources(int64_t* device_bytes, int64_t* device_blocks, int64_t* pinned_bytes, int64_t* registered_ranges, int64_t* events,
                                   int64_t* streams) {
    std::lock_guard<std::mutex> g(csmp_track::mu);
    int64_t db = 0, pb = 0;
    for (const auto& kv : csmp_track::dev) db += (int64_t)kv.second;
    for (const auto& kv : csmp_track::pinned) pb += (int64_t)kv.second;
    if (device_bytes) *device_bytes = db;
    if (device_blocks) *device_blocks = (int64_t)csmp_track::dev.size();
    if (pinned_bytes) *pinned_bytes = pb;
    if (registered_ranges) *registered_ranges = (int64_t)csmp_track::registered.load();
    if (events) *events = (int64_t)csmp_track::events.load();
    if (streams) *streams = (int64_t)csmp_track::streams.load();
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

extern "C" int csmp_sweep_config(const csmp_ctx* ctx, int* unit_loads, int* phases, int* workgroups, int* tick_workgroups, int64_t* lds_bytes, int* dynamic, int* columns_per_unit) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->dA) return CSMP_ESTATE;
    if (unit_loads) *unit_loads = ctx->sweep_U;
    if (phases) *phases = ctx->sweep_ph ? (ctx->Mv + ctx->sweep_KP - 1) / ctx->sweep_KP : 1;
    if (workgroups) *workgroups = ctx->sweep_grid;
    if (tick_workgroups) *tick_workgroups = ctx->tick_nblk > 0 ? ctx->tick_nblk : ctx->tick_grid;
    if (lds_bytes) *lds_bytes = (int64_t)ctx->sweep_lds;
    if (dynamic) *dynamic = ctx->sweep_dyn ? 1 : 0;
    if (columns_per_unit) *columns_per_unit = ctx->short_cpu > 0 ? 8 / ctx->short_nch : 1;
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
        case CSMP_TUNE_PHASE_ROWS: ctx->tune_phase_rows = (int)value; break;
        case CSMP_TUNE_SWEEP_SHORT: ctx->tune_sweep_short = value == 1 ? 1 : 0; break;
        case CSMP_TUNE_SWEEP_LDS_KIB:
            if (value > 159) return fail(ctx, CSMP_EINVAL, "csmp_tune: at most 159 KiB of LDS");
            ctx->tune_sweep_lds_kib = (int)value;
            break;
        case CSMP_TUNE_SWEEP_DYN:
            if (value < 0 || value > 64) return fail(ctx, CSMP_EINVAL, "csmp_tune: sweep_dyn must be 0 (static), 1 (every column claimed) or 2..64 (the last 1 / n of a workgroup's columns claimed)");
            ctx->tune_sweep_dyn = (int)value;
            break;
        case CSMP_TUNE_CLAIM_POOLS:
            if (value < 1 || value > 4096) return fail(ctx, CSMP_EINVAL, "csmp_tune: claim pools must be 1..4096");
            ctx->claim_pools = (int)value;
            return CSMP_OK;
        case CSMP_TUNE_PAIR_LDS_KIB:
            if (value > 159) return fail(ctx, CSMP_EINVAL, "csmp_tune: at most 159 KiB of LDS");
            ctx->tune_pair_lds_kib = (int)value;
            return CSMP_OK;
        case CSMP_TUNE_SCREEN_STATIC: ctx->tune_screen_static = value == 1 ? 1 : 0; return CSMP_OK;
        case CSMP_TUNE_PAIR_SPLIT: ctx->tune_pair_split = value == 1 ? 1 : 0; return CSMP_OK;
        case CSMP_TUNE_FAIL_ALLOC: ctx->tune_fail_alloc = (int)value; return CSMP_OK;
        case CSMP_TUNE_PIPELINES:
            if (value > 2) return fail(ctx, CSMP_EINVAL, "csmp_tune: pipelines must be 0 (automatic), 1 or 2");
            ctx->tune_pipelines = (int)value;
            return CSMP_OK;
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
