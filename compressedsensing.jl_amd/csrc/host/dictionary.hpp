// host/dictionary.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// sweep configuration and launch; the resident dictionary.
// ------------------------------------------------------------------------------------------ sweep launch
template <typename TA, typename TACC, int U, bool FULL, bool NT, int CPW = kCPW>
static hipError_t sweep_launch_t(csmp_ctx* ctx, int grid, size_t lds, const double* r, double eps, int check_eps, int skipmask) {
    auto kern = k_sweep<TA, TACC, U, FULL, NT, CPW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    Solver& s = ctx->s;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv,
                       ctx->N, r, s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask);
    return hipGetLastError();
}


// product configuration: one column per wave at a time (CPW = 1), U chunks = U KiB in flight per lane-row
template <typename TA>
static hipError_t sweep_product(csmp_ctx* ctx, int U, bool full, int grid, size_t lds, const double* r, double eps,
                                int check_eps, int skipmask) {
    if (!full) return sweep_launch_t<TA, double, 1, false, false, 1>(ctx, grid, lds, r, eps, check_eps, skipmask);
    if (U >= 8) {  // software-pipelined kernel
        Solver& s = ctx->s;
        if (lds > 64 * 1024) {
            hipError_t e = U == 16 ? hipFuncSetAttribute((const void*)k_sweep_pf<TA, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                   : hipFuncSetAttribute((const void*)k_sweep_pf<TA, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        if (U == 16)
            hipLaunchKernelGGL((k_sweep_pf<TA, 16, true>), dim3(grid), dim3(kSweepThreads), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld,
                               ctx->Mv, ctx->N, r, s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask);
        else
            hipLaunchKernelGGL((k_sweep_pf<TA, 8, true>), dim3(grid), dim3(kSweepThreads), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld,
                               ctx->Mv, ctx->N, r, s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask);
        return hipGetLastError();
    }
    switch (U) {
        case 4: return sweep_launch_t<TA, double, 4, true, true, 1>(ctx, grid, lds, r, eps, check_eps, skipmask);
        case 2: return sweep_launch_t<TA, double, 2, true, true, 1>(ctx, grid, lds, r, eps, check_eps, skipmask);
        default: return sweep_launch_t<TA, double, 1, true, true, 1>(ctx, grid, lds, r, eps, check_eps, skipmask);
    }
}

// true when this sweep launch is one of the sampled ones
static bool prof_pick(csmp_ctx* ctx) {
    if (!ctx->prof) return false;
    return (ctx->prof_count++ % ctx->prof_every) == 0;
}

static int prof_mark(csmp_ctx* ctx) {
    if (ctx->ev_used == ctx->ev.size()) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        ctx->ev.push_back(e);
    }
    HIPCHECK(hipEventRecord(ctx->ev[ctx->ev_used++], ctx->stream));
    return CSMP_OK;
}


static int launch_sweep(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask) {
    const bool timed = prof_pick(ctx);
    if (timed) CHECK(prof_mark(ctx));
    hipError_t e = ctx->dtype == CSMP_F32
                       ? sweep_product<float>(ctx, ctx->sweep_U, ctx->sweep_full, ctx->sweep_grid, ctx->sweep_lds, r, eps, check_eps, skipmask)
                       : sweep_product<double>(ctx, ctx->sweep_U, ctx->sweep_full, ctx->sweep_grid, ctx->sweep_lds, r, eps, check_eps, skipmask);
    HIPCHECK(e);
    if (timed) CHECK(prof_mark(ctx));
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ dictionary
static int configure_sweep(csmp_ctx* ctx) {
    const int vec = ctx->dtype == CSMP_F32 ? 4 : 2;
    ctx->sweep_lds = sweep_lds_bytes(ctx->Mv, vec);
    if (ctx->sweep_lds > 160 * 1024 - 512) return fail(ctx, CSMP_ERANGE, "M too large: the residual must fit the 160 KiB LDS");
    // Measured on MI355X at 4096 x 65536 f32 (tools/probe_sweep*.py, profiles/): ONE column per wave
    // at a time, non-temporal loads, software-pipelined across columns (the next column's 16 KiB are
    // requested before the current one is reduced), and only 192 workgroups (3/4 of the CUs):
    // 154.5 us = 6.95 TB/s.  More workgroups, or several columns per wave, mean more concurrent DRAM
    // streams and LESS bandwidth (768 workgroups: 6.6 TB/s; 4 columns per wave: 6.1 TB/s).
    const int rows = kWave * vec;
    const int nchunk = (ctx->Mv + rows - 1) / rows;
    ctx->sweep_full = (ctx->Mv % rows) == 0;
    ctx->sweep_U = 1;
    const int umax = 16;
    if (ctx->sweep_full)
        for (int u : {16, 8, 4, 2})
            if (u <= umax && nchunk % u == 0) {
                ctx->sweep_U = u;
                break;
            }
    ctx->tick_U = ctx->sweep_U;
    if (ctx->sweep_full && ctx->sweep_U == 16) ctx->tick_U = 8;  // (16 | nchunk implies 8 | nchunk)
    ctx->sweep_nt = true;
    int per_cu = ctx->sweep_U == 16 ? 3 : 4;
    per_cu = (int)std::min<size_t>((size_t)per_cu, (160 * 1024) / ctx->sweep_lds);
    if (per_cu < 1) per_cu = 1;
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    int64_t grid = (int64_t)ctx->prop.multiProcessorCount * per_cu;
    if (ctx->sweep_full && ctx->sweep_U == 16) grid = (int64_t)ctx->prop.multiProcessorCount * 3 / 4;  // pipelined kernel
    if (ctx->sweep_full && ctx->sweep_U == 8) grid = (int64_t)ctx->prop.multiProcessorCount;
    ctx->sweep_grid = (int)std::max<int64_t>(1, std::min<int64_t>(grid, groups));
    return CSMP_OK;
}

extern "C" int csmp_set_dictionary(csmp_ctx* ctx, const void* A, int64_t M, int64_t N, int64_t ldA, int dtype, int loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!A || M < 1 || N < 1 || ldA < M) return fail(ctx, CSMP_EDIM, "set_dictionary: need A != NULL, M,N >= 1, ldA >= M");
    if (dtype != CSMP_F32 && dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "set_dictionary: dtype must be CSMP_F32 or CSMP_F64");
    if (M > (int64_t)1 << 30 || N > (int64_t)1 << 31) return fail(ctx, CSMP_ERANGE, "set_dictionary: M or N too large");
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    for (auto& t : ctx->twins) {  // (the twins of the batch drivers hold the previous dictionary)
        if (t) (void)csmp_destroy(t);
        t = nullptr;
    }
    dict_release(ctx);  // (clones that still hold the previous dictionary keep it alive)
    HIPCHECK(sync_all(ctx));
    for (int q = 2; q >= 0; --q) {
        activate_slot(ctx, q);
        solver_free(ctx->s);
    }
    batch_free(ctx->bt, false);
    const size_t es = dtype == CSMP_F32 ? 4 : 8;
    const int vec = 16 / (int)es;
    const bool borrow = loc == CSMP_DEVICE && ((uintptr_t)A % 16 == 0) && (M % vec == 0) && (ldA % vec == 0);
    ctx->dtype = dtype;
    ctx->M = M;
    ctx->N = N;
    ctx->col_offset = 0;
    if (borrow) {
        ctx->dA = const_cast<void*>(A);
        ctx->ld = ldA;
        ctx->Mv = (int)M;
    } else {
        const int64_t ld = ((M + vec - 1) / vec) * vec;
        void* d = nullptr;
        HIPCHECK(hipMalloc(&d, (size_t)ld * (size_t)N * es));
        ctx->dA = d;
        ctx->ownA = true;
        ctx->share = new DictShare{d, 1};
        ctx->ld = ld;
        ctx->Mv = (int)ld;
        if (ld != M) HIPCHECK(hipMemsetAsync(d, 0, (size_t)ld * (size_t)N * es, ctx->stream));
        HIPCHECK(hipMemcpy2DAsync(d, (size_t)ld * es, A, (size_t)ldA * es, (size_t)M * es, (size_t)N,
                                  loc == CSMP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
    }
    return configure_sweep(ctx);
}
