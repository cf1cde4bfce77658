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

// experimental column-per-wave variants (f32 dictionary, full chunks only): cpw in {1,2}, U in {4,8,16}
#ifdef CSMP_EXPERIMENTS
template <int U>
static hipError_t sweep_launch_pf(csmp_ctx* ctx, int grid, const double* r) {
    auto kern = k_sweep_pf<float, U, true>;
    Solver& s = ctx->s;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), ctx->sweep_lds, ctx->stream, (const float*)ctx->dA, ctx->ld, ctx->Mv,
                       ctx->N, r, s.cvec, s.pval, s.pidx, s.st, 0.0, 0, 0);
    return hipGetLastError();
}
static hipError_t sweep_launch_cpw(csmp_ctx* ctx, int cpw, int U, int grid, const double* r) {
    const size_t lds = ctx->sweep_lds;
    if (cpw == 3 && U == 16) return sweep_launch_pf<16>(ctx, grid, r);
    if (cpw == 3 && U == 8) return sweep_launch_pf<8>(ctx, grid, r);
    if (cpw == 3 && U == 4) return sweep_launch_pf<4>(ctx, grid, r);
    if (cpw == 3 && U == 2) return sweep_launch_pf<2>(ctx, grid, r);
    if (cpw == 1 && U == 4) return sweep_launch_t<float, double, 4, true, true, 1>(ctx, grid, lds, r, 0.0, 0, 0);
    if (cpw == 1 && U == 8) return sweep_launch_t<float, double, 8, true, true, 1>(ctx, grid, lds, r, 0.0, 0, 0);
    if (cpw == 1 && U == 16) return sweep_launch_t<float, double, 16, true, true, 1>(ctx, grid, lds, r, 0.0, 0, 0);
    if (cpw == 2 && U == 2) return sweep_launch_t<float, double, 2, true, true, 2>(ctx, grid, lds, r, 0.0, 0, 0);
    if (cpw == 2 && U == 4) return sweep_launch_t<float, double, 4, true, true, 2>(ctx, grid, lds, r, 0.0, 0, 0);
    if (cpw == 2 && U == 8) return sweep_launch_t<float, double, 8, true, true, 2>(ctx, grid, lds, r, 0.0, 0, 0);
    return hipErrorInvalidValue;
}

#endif

// variant = U + 8*nt + 16*f32acc + 256*workgroups_per_CU (0 = product configuration)
// variant >= 1<<20: experimental: (variant>>20) = cpw, bits 0-7 = U, bits 8-15 = workgroups per CU
extern "C" int csmp_bench_sweep(csmp_ctx* ctx, int variant, int reps, double* avg_ms) {
    if (!ctx || reps < 1) return CSMP_EINVAL;
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
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
    int U = ctx->sweep_U, grid = ctx->sweep_grid;
    bool nt = ctx->sweep_nt, f32acc = false;
#ifdef CSMP_EXPERIMENTS
    const int cpwx = variant >> 20;
    if (cpwx) {
        if (ctx->dtype != CSMP_F32) return fail(ctx, CSMP_EINVAL, "bench_sweep: experimental variants are f32 only");
        U = variant & 0xff;
        const int per_cu = (variant >> 8) & 0xff;
        const int64_t groups = (ctx->N + 4 * (cpwx == 3 ? 1 : cpwx) - 1) / (4 * (cpwx == 3 ? 1 : cpwx));
        grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->prop.multiProcessorCount * (per_cu ? per_cu : 4), groups));
        if (grid > ctx->prop.multiProcessorCount * 8) grid = ctx->prop.multiProcessorCount * 8;
        if (const char* sn = tune_env("CSMP_SWEEP_NBLK")) grid = std::max(1, atoi(sn));
        if (ctx->Mv % (256 * U)) return fail(ctx, CSMP_EINVAL, "bench_sweep: M must be a multiple of 256*U");
        for (int i = 0; i < 3; ++i) HIPCHECK(sweep_launch_cpw(ctx, cpwx, U, grid, ctx->s.r));
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventCreate(&e1));
        HIPCHECK(hipEventRecord(e0, ctx->stream));
        for (int i = 0; i < reps; ++i) HIPCHECK(sweep_launch_cpw(ctx, cpwx, U, grid, ctx->s.r));
        HIPCHECK(hipEventRecord(e1, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (avg_ms) *avg_ms = (double)ms / reps;
        return CSMP_OK;
    }
#else
    if (variant != 0) return fail(ctx, CSMP_ESTATE, "bench_sweep: experimental variants need a build with -DCSMP_EXPERIMENTS (make experiments)");
#endif
#ifdef CSMP_EXPERIMENTS
    if (variant != 0) {
        U = variant & 7;
        nt = (variant & 8) != 0;
        f32acc = (variant & 16) != 0;
        const int per_cu = (variant >> 8) & 0xff;
        if (per_cu > 0) {
            const int64_t groups = (ctx->N + 15) / 16;
            grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->prop.multiProcessorCount * per_cu, groups));
        }
        if (U != 1 && U != 2 && U != 4) return fail(ctx, CSMP_EINVAL, "bench_sweep: U must be 1, 2 or 4");
    }
#endif
    const bool was = ctx->prof;
    ctx->prof = false;
    if (variant == 0) {
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
#ifdef CSMP_EXPERIMENTS
    for (int i = 0; i < 3; ++i) CHECK(launch_sweep_cfg(ctx, ctx->s.r, 0.0, 0, 0, U, nt, f32acc, grid));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; ++i) CHECK(launch_sweep_cfg(ctx, ctx->s.r, 0.0, 0, 0, U, nt, f32acc, grid));
    HIPCHECK(hipEventRecord(e1, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ctx->prof = was;
    if (avg_ms) *avg_ms = (double)ms / reps;
    return CSMP_OK;
#else
    (void)U; (void)grid; (void)nt; (void)f32acc;
    return CSMP_OK;
#endif
}
