// host/screened.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// single-signal OMP with the screened sweep (csmp_screened.hpp; option CSMP_OPT_SCREENED_SWEEP).

// bf16 image, certificate coefficients, sweep grid: once per dictionary / option value
static int screened_ensure(csmp_ctx* ctx) {
    // the int8 image only where one step resolves every column (see csmp_omp_batch_mfma): a dictionary that is not flat gets the
    // bf16 image instead of a screen that certifies nothing
    CHECK(batch_meta(ctx));
    Batch& b = ctx->bt;
    const bool i8 = ctx->opt_screened == 2 && b.amax_host <= 8.0f * b.arms_host;
    const bool f16 = ctx->opt_screened == 3;
    ctx->scr_image = i8 ? 2 : f16 ? 3 : 1;
    CHECK(i8 ? batch_dict8(ctx) : f16 ? batch_dict16(ctx) : batch_dict(ctx));
    // The certificate's coefficients are recomputed on every call (a few flops; the column-norm reduction behind them is cached per
    // dictionary in bt.anorm_host, which csmp_set_dictionary resets): nothing of a previous dictionary can survive in them.
    if (i8) {  // int8 image: statistical bound only (see csmp_omp_batch_mfma, host/batched.hpp)
        CHECK(batch_colnorm(ctx));
        ctx->scr_cert_abs = 8.0 * (double)b.astep / std::sqrt(12.0);
        ctx->scr_cert_abs2 = 8.0 * (double)b.anorm_host / std::sqrt(12.0);
        ctx->scr_cert_rel = std::ldexp(1.0, -6) + std::ldexp(1.0, -20);
        ctx->scr_kwin = kWinMax;
    } else {
        ctx->scr_cert_abs2 = 0.0;
        // ONE rounded operand here: the image of the dictionary (unit roundoff u: 2^-11 binary16, 2^-8 bf16; + 2^-23 for a Float64
        // dictionary's double rounding); the residual enters the sweep in Float32 (2^-24) and the sums are Float32 FMAs in a fixed
        // order, round to nearest (2^-24 each, fewer than Mk of them on any path to a sum):
        //     |<a,r> - screened| <= (u + 2^-24 + Mk 2^-24)(1 + 2^-10) |a|_2 |r|_2  [+ binary16 entries below the normal range, charged a flush to zero: Mk 2^-27]
        const double u_img = (f16 ? std::ldexp(1.0, -11) : std::ldexp(1.0, -8)) + std::ldexp(1.0, -23);
        if (ctx->opt_batch_cert == 1) {
            CHECK(batch_colnorm(ctx));
            ctx->scr_cert_abs = ((u_img + std::ldexp(1.0, -24) + (double)b.Mk * std::ldexp(1.0, -24)) * (1.0 + std::ldexp(1.0, -10)) +
                                 (f16 ? (double)b.Mk * std::ldexp(1.0, -27) : 0.0)) * (double)b.anorm_host;
            ctx->scr_cert_rel = std::ldexp(1.0, -20);
            ctx->scr_kwin = kWinMax;
        } else {  // statistical: 8 sigma of independent roundings of the image's entries + the coherent term (host/batched.hpp)
            ctx->scr_cert_abs = 8.0 * std::sqrt(2.0 / 3.0) * u_img * 0.5 * (double)b.amax_host;
            ctx->scr_cert_rel = 2.0 * u_img * 1.01 + std::ldexp(1.0, -20);
            ctx->scr_kwin = kWinMax / 2;
        }
    }
    // grid: one workgroup per CU (measured at configs[1], tools/probes/sweep_probe.hip: 192 / 256 / 384 / 512 workgroups 90 / 81 / 89 / 97 us)
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) * kScrCols - 1) / ((kSweepThreads / kWave) * kScrCols);
    const int maxgrid = ctx->prop.multiProcessorCount * 8;
    ctx->scr_grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ctx->prop.multiProcessorCount, maxgrid), groups));
    if (ctx->scr_grid > kScrPartWgs) ctx->scr_grid -= ctx->scr_grid % kScrPartWgs;  // whole ticket partitions
    return CSMP_OK;
}

// NT = 1024 when the solve runs alone (16 waves rescore the window in one round); 256 in the batch forms, where a 1024-thread
// workgroup (126 registers) would not fit on a CU beside another solve's sweep workgroup and would wait for one to finish
template <typename TA, int NT>
static hipError_t pick1_launch_t(csmp_ctx* ctx, int ncand, int skipmask, int mp_select) {
    Solver& s = ctx->s;
    constexpr int U = sizeof(TA) == 4 ? 16 : 8;
    const size_t lds = b_pick_lds_bytes(ctx->Mv, (int)(16 / sizeof(TA)));
    auto kern = k_pick1<TA, U, NT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(NT), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, (const float*)s.scr_val, (const int*)s.scr_idx,
                       ncand, s.st, (const double*)s.r, s.Mpad, s.pval, s.pidx, ctx->scr_cert_abs, ctx->scr_cert_rel, ctx->scr_kwin, skipmask,
                       s.scr_tickets, ctx->scr_grid / kScrPartWgs + 1, ctx->scr_cert_abs2, mp_select);
    return hipGetLastError();
}
template <typename TA>
static hipError_t pick1_launch(csmp_ctx* ctx, int ncand, int skipmask, int mp_select = 0) {
    return ctx->scr_lone ? pick1_launch_t<TA, 1024>(ctx, ncand, skipmask, mp_select) : pick1_launch_t<TA, 256>(ctx, ncand, skipmask, mp_select);
}

template <typename TA, int NT>
static hipError_t pickS_launch_t(csmp_ctx* ctx, int ncand, int S, int skipmask) {
    Solver& s = ctx->s;
    constexpr int U = sizeof(TA) == 4 ? 16 : 8;
    const size_t lds = b_pick_lds_bytes(ctx->Mv, (int)(16 / sizeof(TA)));
    auto kern = k_pickS<TA, U, NT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(NT), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, (const float*)s.scr_val, (const int*)s.scr_idx,
                       ncand, s.st, (const double*)s.r, s.Mpad, S, s.cands, s.cvals, s.ncands, ctx->scr_cert_abs, ctx->scr_cert_rel, ctx->scr_kwin,
                       skipmask, s.scr_tickets, ctx->scr_grid / kScrPartWgs + 1, ctx->scr_cert_abs2);
    return hipGetLastError();
}
template <typename TA>
static hipError_t pickS_launch(csmp_ctx* ctx, int ncand, int S, int skipmask) {  // (workgroup size: as pick1_launch)
    return ctx->scr_lone ? pickS_launch_t<TA, 1024>(ctx, ncand, S, skipmask) : pickS_launch_t<TA, 256>(ctx, ncand, S, skipmask);
}

// the sweep over the bf16 image: candidates of every workgroup into s.scr_val / s.scr_idx
static int launch_sweep_bf16(csmp_ctx* ctx, double eps, int check_eps, int skip, bool wide = false) {
    // bit 1 (csmp_tune(CSMP_TUNE_SCREEN_STATIC, 1), a measurement switch): the column groups dealt out STATICALLY, no ticket counters.
    // Measured: a sweep that has the chip to itself needs the tickets (the 2-GiB image of configs[4]: 315 us against 357 with the
    // static split); with three solves in flight the two forms are within 2 % of each other (11.5-11.9e3 atoms/s either way).
    check_eps = (check_eps ? 1 : 0) | (ctx->tune_screen_static == 1 ? 2 : 0);
    Solver& s = ctx->s;
    Batch& b = ctx->bt;
    const bool timed = prof_pick(ctx);
    if (timed) CHECK(prof_mark(ctx));
    const int lc = wide ? kScrCandK : kScrCand;  // candidates listed per workgroup (wide: Subspace Pursuit's top-k)
    if (ctx->scr_image == 2) {  // the int8 image: chunks of 1024 rows
        const int nchunk = (b.Mk8 + 1023) / 1024;
        const size_t lds = sweep_i8_lds_bytes(b.Mk8, lc);
#define CSMP_SCR8L(U, DD, FULL, LCV)                                                                                                      \
    {                                                                                                                                     \
        if (lds > 48 * 1024)                                                                                                              \
            HIPCHECK(hipFuncSetAttribute((const void*)k_sweep_i8<U, DD, FULL, kScrCols, LCV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_sweep_i8<U, DD, FULL, kScrCols, LCV>), dim3(ctx->scr_grid), dim3(kSweepThreads), lds, ctx->stream,             \
                           (const signed char*)b.A8, b.Mk8, ctx->N, (const double*)s.r, s.Mpad, s.scr_val, s.scr_idx, s.st, eps, check_eps, \
                           skip, s.scr_tickets, b.astep);                                                                                 \
    }
#define CSMP_SCR8(U, DD, FULL)                                  \
    {                                                           \
        if (wide) CSMP_SCR8L(U, DD, FULL, kScrCandK)            \
        else CSMP_SCR8L(U, DD, FULL, kScrCand)                  \
    }
        const bool whole = b.Mk8 % 1024 == 0;
        if (whole && nchunk % 2 == 0) CSMP_SCR8(2, 3, true)
        else if (whole) CSMP_SCR8(1, 4, true)
        else if (nchunk >= 2) CSMP_SCR8(2, 3, false)
        else CSMP_SCR8(1, 4, false)
#undef CSMP_SCR8
#undef CSMP_SCR8L
        HIPCHECK(hipGetLastError());
        if (timed) CHECK(prof_mark(ctx));
        return CSMP_OK;
    }
    const int nchunk = (b.Mk + 511) / 512;
    const size_t lds = sweep_bf16_lds_bytes(b.Mk, lc);
    const bool f16 = ctx->scr_image == 3;  // the binary16 image: the same stream, layout and LDS as bf16
#define CSMP_SCRL(U, DD, FULL, LCV)                                                                                                       \
    {                                                                                                                                     \
        if (f16) {                                                                                                                        \
            if (lds > 48 * 1024)                                                                                                          \
                HIPCHECK(hipFuncSetAttribute((const void*)k_sweep_f16<U, DD, FULL, kScrCols, LCV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_sweep_f16<U, DD, FULL, kScrCols, LCV>), dim3(ctx->scr_grid), dim3(kSweepThreads), lds, ctx->stream,        \
                               (const unsigned short*)b.Ah, b.Mk, ctx->N, (const double*)s.r, s.Mpad, s.scr_val, s.scr_idx, s.st, eps, check_eps, skip, \
                               s.scr_tickets, 1.0f / b.ascale16);                                                                         \
        } else {                                                                                                                          \
            if (lds > 48 * 1024)                                                                                                          \
                HIPCHECK(hipFuncSetAttribute((const void*)k_sweep_bf16<U, DD, FULL, kScrCols, LCV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_sweep_bf16<U, DD, FULL, kScrCols, LCV>), dim3(ctx->scr_grid), dim3(kSweepThreads), lds, ctx->stream,       \
                               (const __bf16*)b.Ab, b.Mk, ctx->N, (const double*)s.r, s.Mpad, s.scr_val, s.scr_idx, s.st, eps, check_eps, skip, \
                               s.scr_tickets);                                                                                            \
        }                                                                                                                                 \
    }
#define CSMP_SCR(U, DD, FULL)                                   \
    {                                                           \
        if (wide) CSMP_SCRL(U, DD, FULL, kScrCandK)             \
        else CSMP_SCRL(U, DD, FULL, kScrCand)                   \
    }
    // items of U chunks of four columns (4 U KiB per wave and load group): U whole chunks per item where the column allows
    const bool whole = b.Mk % 512 == 0;
    if (whole && nchunk % 2 == 0) CSMP_SCR(2, 3, true)
    else if (whole) CSMP_SCR(1, 4, true)
    else if (nchunk >= 2) CSMP_SCR(2, 3, false)
    else CSMP_SCR(1, 4, false)
#undef CSMP_SCR
#undef CSMP_SCRL
    HIPCHECK(hipGetLastError());
    if (timed) CHECK(prof_mark(ctx));
    return CSMP_OK;
}

// the tickets of the sweep that has just run, back to zero (the pick kernels of omp / gomp do it themselves)
static int scr_tickets_reset(csmp_ctx* ctx) {
    HIPCHECK(hipMemsetAsync(ctx->s.scr_tickets, 0, (size_t)(ctx->scr_grid / kScrPartWgs + 1) * kScrTicketStride * sizeof(unsigned), ctx->stream));
    return CSMP_OK;
}

// Subspace Pursuit's selection (argmaxinner!(P, k)) on the screened sweep: leaves the k atoms in s.cands / s.ncands as
// launch_topS does, and s.scr_flag = 1 when the set could not be certified (the caller repeats the acquisition exactly)
static int sp_select_screened(csmp_ctx* ctx, int k) {
    Solver& s = ctx->s;
    CHECK(launch_sweep_bf16(ctx, 0.0, 0, 0, /*wide=*/true));
    CHECK(scr_tickets_reset(ctx));
    const int ncand = ctx->scr_grid * kScrCandK;
    HIPCHECK(hipMemsetAsync(s.cvec, 0, (size_t)ctx->N * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_spk_scatter, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, (const float*)s.scr_val, (const int*)s.scr_idx, ncand, s.cvec,
                       s.scr_cb, s.scr_flag);
    HIPCHECK(hipGetLastError());
    CHECK(launch_topS(ctx, k));  // the k-th largest screened value: s.cvals[k - 1]
    const size_t lds = b_pick_lds_bytes(ctx->Mv, ctx->dtype == CSMP_F32 ? 4 : 2);
    const int grid = std::max(1, std::min(ncand / 4, ctx->prop.multiProcessorCount));
    if (ctx->dtype == CSMP_F32) {
        if (lds > 48 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_spk_rescore<float, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_spk_rescore<float, 16>), dim3(grid), dim3(256), lds, ctx->stream, (const float*)ctx->dA, ctx->ld, ctx->Mv, (const float*)s.scr_val,
                           (const int*)s.scr_idx, ncand, (const DevState*)s.st, (const double*)s.r, s.Mpad, (const double*)s.cvals, (const int*)s.ncands, k,
                           s.cvec, s.scr_cb, ctx->scr_cert_abs, ctx->scr_cert_rel, ctx->scr_cert_abs2);
    } else {
        if (lds > 48 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_spk_rescore<double, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_spk_rescore<double, 8>), dim3(grid), dim3(256), lds, ctx->stream, (const double*)ctx->dA, ctx->ld, ctx->Mv, (const float*)s.scr_val,
                           (const int*)s.scr_idx, ncand, (const DevState*)s.st, (const double*)s.r, s.Mpad, (const double*)s.cvals, (const int*)s.ncands, k,
                           s.cvec, s.scr_cb, ctx->scr_cert_abs, ctx->scr_cert_rel, ctx->scr_cert_abs2);
    }
    HIPCHECK(hipGetLastError());
    CHECK(launch_topS(ctx, k));  // the k largest exact values
    hipLaunchKernelGGL(k_spk_cert, dim3(1), dim3(1), 0, ctx->stream, (const double*)s.cvals, (const int*)s.ncands, k, (const unsigned long long*)s.scr_cb, s.scr_flag);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// update!(P::OMP, x) with the screened sweep: bf16 sweep -> certified pick -> the exact path's append chain
static int omp_step_screened(csmp_ctx* ctx, double eps, int check_eps, bool optimistic) {
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    CHECK(launch_sweep_bf16(ctx, eps, check_eps, skip));
    HIPCHECK(ctx->dtype == CSMP_F32 ? pick1_launch<float>(ctx, ctx->scr_grid * kScrCand, skip) : pick1_launch<double>(ctx, ctx->scr_grid * kScrCand, skip));
    return launch_append(ctx, 1, 0, skip, optimistic, 0.0, /*nblk_sweep: the pick is the only "partial"*/ 1);
}

// update!(P::MP, x) (src/matchingpursuit.jl:26-31) with the screened sweep: the certified pick also does k_select's job
static int mp_step_screened(csmp_ctx* ctx) {
    CHECK(launch_sweep_bf16(ctx, 0.0, 0, 0));
    HIPCHECK(ctx->dtype == CSMP_F32 ? pick1_launch<float>(ctx, ctx->scr_grid * kScrCand, 0, 1) : pick1_launch<double>(ctx, ctx->scr_grid * kScrCand, 0, 1));
    return launch_mp_update(ctx);
}

// ompr's update! with the screened sweep: the image sweep, the certified arg-max (which also does k_select's job) and the exact
// correlations on the support (cols: the support's atoms on the device, n of them) into s.coef
static int ompr_sweep_screened(csmp_ctx* ctx, const int* cols_dev, int n) {
    Solver& s = ctx->s;
    CHECK(launch_sweep_bf16(ctx, 0.0, 0, 0));
    HIPCHECK(ctx->dtype == CSMP_F32 ? pick1_launch<float>(ctx, ctx->scr_grid * kScrCand, 0, 1) : pick1_launch<double>(ctx, ctx->scr_grid * kScrCand, 0, 1));
    const size_t lds = b_pick_lds_bytes(ctx->Mv, ctx->dtype == CSMP_F32 ? 4 : 2);
    const int grid = std::max(1, std::min((n + 3) / 4, ctx->prop.multiProcessorCount));
    if (ctx->dtype == CSMP_F32) {
        if (lds > 48 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_cols_dot<float, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_cols_dot<float, 16>), dim3(grid), dim3(256), lds, ctx->stream, (const float*)ctx->dA, ctx->ld, ctx->Mv, cols_dev, n,
                           (const double*)s.r, s.Mpad, s.coef);
    } else {
        if (lds > 48 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_cols_dot<double, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_cols_dot<double, 8>), dim3(grid), dim3(256), lds, ctx->stream, (const double*)ctx->dA, ctx->ld, ctx->Mv, cols_dev, n,
                           (const double*)s.r, s.Mpad, s.coef);
    }
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// update!(P::GOMP, x, l) with the screened sweep: bf16 sweep -> certified top-l pick -> the exact path's (panel) appends
static int gomp_update_screened(csmp_ctx* ctx, int64_t l, double eps, int check_eps, int skipmask, bool block) {
    l = std::min<int64_t>(l, ctx->N);
    CHECK(launch_sweep_bf16(ctx, eps, check_eps, skipmask));
    HIPCHECK(ctx->dtype == CSMP_F32 ? pickS_launch<float>(ctx, ctx->scr_grid * kScrCand, (int)l, skipmask)
                                    : pickS_launch<double>(ctx, ctx->scr_grid * kScrCand, (int)l, skipmask));
    if (block && l > 1) return launch_block_appends(ctx, (int)l, skipmask);
    for (int64_t w = 0; w < l; ++w) CHECK(launch_append(ctx, 2, (int)w, skipmask));
    return CSMP_OK;
}

// the twin of a batch driver sweeps its parent's image (the twin is destroyed before the parent's image is: csmp_set_dictionary,
// csmp_destroy) and carries its certificate option
static int screened_ensure_pair(csmp_ctx* ctx, csmp_ctx* twin) {
    twin->opt_batch_cert = ctx->opt_batch_cert;
    twin->opt_screened = ctx->opt_screened;
    CHECK(screened_ensure(ctx));
    if (!twin->bt.meta_valid) {
        Batch &tb = twin->bt, &pb = ctx->bt;
        tb.Mk = pb.Mk;
        tb.Npad = pb.Npad;
        tb.n_atiles = pb.n_atiles;
        tb.amax_host = pb.amax_host;
        tb.arms_host = pb.arms_host;
        tb.meta_valid = true;
    }
    if (ctx->bt.a8_valid && !twin->bt.a8_valid) {
        Batch &tb = twin->bt, &pb = ctx->bt;
        tb.A8 = pb.A8;
        tb.Mk8 = pb.Mk8;
        tb.astep = pb.astep;
        tb.a8_valid = tb.a8_borrowed = true;
    }
    if (ctx->bt.ah_valid && !twin->bt.ah_valid) {
        Batch &tb = twin->bt, &pb = ctx->bt;
        tb.Ah = pb.Ah;
        tb.ascale16 = pb.ascale16;
        tb.Mk = pb.Mk;
        tb.Npad = pb.Npad;
        tb.n_atiles = pb.n_atiles;
        tb.amax_host = pb.amax_host;
        tb.ah_valid = tb.ah_borrowed = true;
    }
    if (ctx->bt.ab_valid && !twin->bt.ab_valid) {
        Batch &tb = twin->bt, &pb = ctx->bt;
        tb.Ab = pb.Ab;
        tb.Mk = pb.Mk;
        tb.Npad = pb.Npad;
        tb.n_atiles = pb.n_atiles;
        tb.amax_host = pb.amax_host;
        tb.ab_valid = tb.ab_borrowed = true;
    }
    twin->bt.anorm_host = ctx->bt.anorm_host;
    const int rc = screened_ensure(twin);
    if (rc != CSMP_OK) ctx->err = twin->err;
    return rc;
}

// omp for many signals with the screened sweep: TWO solves in flight, the context's and a twin's, out of phase (the pattern of
// csmp_gomp_batch): one signal's pick and append stages run under the other's sweep.  A signal whose solve was flagged --
// an uncertified pick, or a column that failed the DGKS test of the optimistic append chain -- is solved again by the exact
// path after the one synchronisation.
static int omp_screened_enqueue(csmp_ctx* c, const void* col_dev, int b_dtype, int64_t k, double eps, int64_t* d_idx, double* d_val,
                                int64_t* d_nnz, int* d_flag, hipEvent_t after_first_sweep) {
    int rc = b_dtype == CSMP_F32 ? init_from_device_t<float>(c, (const float*)col_dev) : init_from_device_t<double>(c, (const double*)col_dev);
    for (int64_t t = 0; t < k && rc == CSMP_OK; ++t) {
        rc = omp_step_screened(c, eps, t > 0, true);
        if (t == 0 && rc == CSMP_OK && after_first_sweep && hipEventRecord(after_first_sweep, c->stream) != hipSuccess) return CSMP_EHIP;
    }
    if (rc != CSMP_OK) return rc;
    return launch_finish(c, d_idx, d_val, d_nnz, nullptr, (int)k, d_flag);
}

static int omp_batch_screened(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k, double eps,
                              int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (nsig == 0) return CSMP_OK;
    HIPCHECK(hipSetDevice(ctx->dev));
    // solves in flight: the context's and its twins', each one sweep behind the previous one
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ctx->opt_in_flight, 3), nsig));  // (CSMP_OPT_SOLVES_IN_FLIGHT)
    if (T > 1) CHECK(twins_ensure(ctx, T - 1));
    csmp_ctx* cc[3] = {ctx, T > 1 ? ctx->twins[0] : nullptr, T > 2 ? ctx->twins[1] : nullptr};
    if (T == 1) CHECK(screened_ensure(ctx));
    for (int q = 1; q < T; ++q) CHECK(screened_ensure_pair(ctx, cc[q]));
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    for (int q = 0; q < T; ++q) {
        const int rc = solver_ensure(cc[q], kc, (int)k);
        if (rc != CSMP_OK) {
            if (q) ctx->err = cc[q]->err;
            return rc;
        }
        cc[q]->s.begun = false;
    }
    const size_t es = b_dtype == CSMP_F32 ? 4 : 8;
    void* dB = const_cast<void*>(B);
    DevTmp tB, tIdx, tVal, tNnz, tFlag;
    if (b_loc == CSMP_HOST) {
        HIPCHECK(tB.alloc((size_t)ldB * (size_t)nsig * es));
        dB = tB.p;
        HIPCHECK(hipMemcpy(dB, B, (size_t)ldB * (size_t)nsig * es, hipMemcpyHostToDevice));
    }
    int64_t *d_idx = idx, *d_nnz = nnz;
    double* d_val = val;
    if (out_loc == CSMP_HOST) {
        HIPCHECK(tIdx.alloc((size_t)k * nsig * 8));
        HIPCHECK(tVal.alloc((size_t)k * nsig * 8));
        HIPCHECK(tNnz.alloc((size_t)nsig * 8));
        d_idx = (int64_t*)tIdx.p;
        d_val = (double*)tVal.p;
        d_nnz = (int64_t*)tNnz.p;
    }
    HIPCHECK(tFlag.alloc((size_t)nsig * sizeof(int)));
    int* d_flag = (int*)tFlag.p;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    for (int q = 0; q + 1 < T; ++q)
        if (!cc[q]->ev_twin) HIPCHECK(hipEventCreateWithFlags(&cc[q]->ev_twin, hipEventDisableTiming));
    for (int64_t sgn = 0; sgn < nsig; ++sgn) {
        const int q = (int)(sgn % T);
        csmp_ctx* c = cc[q];
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        if (sgn > 0 && sgn < T) HIPCHECK(hipStreamWaitEvent(c->stream, cc[q - 1]->ev_twin, 0));  // a twin starts one sweep behind: out of phase
        const int rc = omp_screened_enqueue(c, col, b_dtype, k, eps, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, d_flag + sgn,
                                            sgn + 1 < T ? c->ev_twin : nullptr);
        if (rc != CSMP_OK) {
            if (c != ctx) ctx->err = c->err;
            for (int w = 0; w < T; ++w) (void)hipStreamSynchronize(cc[w]->stream);
            return rc;
        }
    }
    for (int w = 1; w < T; ++w) HIPCHECK(hipStreamSynchronize(cc[w]->stream));
    std::vector<int> hf((size_t)nsig);
    HIPCHECK(hipMemcpyAsync(hf.data(), d_flag, (size_t)nsig * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    int rc = CSMP_OK;
    for (int64_t sgn = 0; sgn < nsig && rc == CSMP_OK; ++sgn) {
        ctx->scr_solves += 1;
        if (hf[sgn] & (STOP_REORTH | STOP_UNCERTAIN)) {  // again, by the exact path with the full append chain
            ctx->scr_fallbacks += (hf[sgn] & STOP_UNCERTAIN) ? 1 : 0;
            const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
            rc = b_dtype == CSMP_F32 ? init_from_device_t<float>(ctx, (const float*)col) : init_from_device_t<double>(ctx, (const double*)col);
            for (int64_t t = 0; t < k && rc == CSMP_OK; ++t) rc = omp_step(ctx, eps, t > 0, false);
            if (rc == CSMP_OK) rc = launch_finish(ctx, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, nullptr, (int)k, d_flag + sgn);
            if (rc == CSMP_OK) {
                HIPCHECK(hipMemcpyAsync(&hf[sgn], d_flag + sgn, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHECK(hipStreamSynchronize(ctx->stream));
            }
        }
    }
    if (out_loc == CSMP_HOST) {
        if (rc == CSMP_OK) {
            HIPCHECK(hipMemcpyAsync(idx, d_idx, (size_t)k * nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(val, d_val, (size_t)k * nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(nnz, d_nnz, (size_t)nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHECK(hipStreamSynchronize(ctx->stream));
    }
    return rc;
}

// screened solves made by this context and how many of them were repeated with the exact sweep (failed certificate)
extern "C" int csmp_screened_stats(csmp_ctx* ctx, int64_t* solves, int64_t* fallbacks, int reset) {
    if (!ctx) return CSMP_EINVAL;
    if (solves) *solves = ctx->scr_solves;
    if (fallbacks) *fallbacks = ctx->scr_fallbacks;
    if (reset) ctx->scr_solves = ctx->scr_fallbacks = 0;
    return CSMP_OK;
}
