// host/forward.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// forward regression (OLS) sweeps and drivers; the omp / fr batch drivers; mp.
// ------------------------------------------------------------------------------------------ forward regression (OLS)
// one pass of k_fr_sweep (csmp_forward.hpp): nq = -1 first step (norms), 0 scores only, 1 / 2 directions
struct FrPass {
    int nq = 1;
    const double* q1 = nullptr;  // null with nq >= 1: the last Q column, looked up on the device
    double s1 = -1.0;
    const double* q2 = nullptr;
    double s2 = 1.0;
    int64_t qstride = 0;  // nq == 4: the directions are q1 + d*qstride
    const int* unmark = nullptr;
    int update_only = 0;
};

template <typename TA, int U, bool FULL, int NQ>
static hipError_t fr_sweep_launch_t(csmp_ctx* ctx, const FrPass& ps, int grid, size_t lds, double max_eps, int skipmask) {
    Solver& s = ctx->s;
    auto kern = k_fr_sweep<TA, U, FULL, NQ>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, ctx->N,
                       (const double*)s.r, (const double*)s.Q, s.ldq, ps.q1, ps.s1, ps.q2, ps.s2, ps.unmark, ps.update_only, s.rho2,
                       s.dvec, s.pval, s.pidx, (const int*)s.sel, s.st, max_eps, skipmask);
    return hipGetLastError();
}
template <typename TA, int U, bool FULL>
static hipError_t fr_sweep_launch_nq(csmp_ctx* ctx, const FrPass& ps, int grid, size_t lds, double max_eps, int skipmask) {
    switch (ps.nq) {
        case -1: return fr_sweep_launch_t<TA, U, FULL, -1>(ctx, ps, grid, lds, max_eps, skipmask);
        case 0: return fr_sweep_launch_t<TA, U, FULL, 0>(ctx, ps, grid, lds, max_eps, skipmask);
        case 1: return fr_sweep_launch_t<TA, U, FULL, 1>(ctx, ps, grid, lds, max_eps, skipmask);
        case 2: return fr_sweep_launch_t<TA, U, FULL, 2>(ctx, ps, grid, lds, max_eps, skipmask);
        default: return hipErrorInvalidValue;
    }
}
template <typename TA>
static hipError_t fr_sweep_launch(csmp_ctx* ctx, const FrPass& ps, int U, bool full, int grid, size_t lds, double max_eps, int skipmask) {
    if (!full) return fr_sweep_launch_nq<TA, 4, false>(ctx, ps, grid, lds, max_eps, skipmask);
    if (U == 16) return fr_sweep_launch_nq<TA, 16, true>(ctx, ps, grid, lds, max_eps, skipmask);
    return fr_sweep_launch_nq<TA, 8, true>(ctx, ps, grid, lds, max_eps, skipmask);
}

// block size of the forward-regression sweep: 16 or 8 chunks when they tile M exactly, else the
// predicated 4-chunk kernel
static void fr_config(const csmp_ctx* ctx, int nq, int& U, bool& full, size_t& lds, int& grid) {
    const int vec = ctx->dtype == CSMP_F32 ? 4 : 2;
    const int rows = kWave * vec;
    U = 4;
    full = false;
    if (ctx->Mv % rows == 0) {
        const int nchunk = ctx->Mv / rows;
        // Measured at 4096 x 65536 f32 (profiles/r01_bench_fr_line.json): 8-chunk blocks on one workgroup per CU
        // 168 us, 16-chunk blocks on 3/4 of the CUs (the OMP sweep's optimum) 173 us -- with a second LDS image
        // to read per chunk, the extra waves hide more than the extra DRAM streams cost.
        // With TWO directions the three images leave room for one workgroup per CU only: four waves, and what hides the DRAM latency
        // is the loads each of them keeps in flight -- 16-chunk blocks on 15/16 of the CUs 175 us, 8-chunk blocks on all of them 187
        // (16 on all: 184, on 7/8: 176, on 3/4: 194; profiles/r05_bench_srr_kernel_stats.csv).
        const int umax = ctx->tune_sweep_U == 16 ? 16 : ctx->tune_sweep_U == 8 ? 8 : nq == 2 ? 16 : 8;  // (csmp_tune: measurement override)
        for (int u : {16, 8})
            if (u <= umax && nchunk % u == 0) {
                U = u;
                full = true;
                break;
            }
    }
    lds = fr_sweep_lds_bytes(ctx->Mv, vec, U, nq);
    const int cus = ctx->prop.multiProcessorCount;
    int64_t g = U == 16 ? (int64_t)cus * (nq == 2 ? 15 : 12) / 16 : (int64_t)cus;  // (nq < 2 with 16-chunk blocks: as the OMP sweep, configure_sweep)
    if (ctx->tune_sweep_grid > 0) g = std::min<int64_t>(ctx->tune_sweep_grid, (int64_t)cus * 8);
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    grid = (int)std::max<int64_t>(1, std::min<int64_t>(g, groups));
}

// A column whose LDS images (the residual and up to two directions: 24 M bytes) exceed the LDS takes the tall path: the same pass
// as separate launches (csmp_forward.hpp, k_fr_combine)
static bool fr_tall(const csmp_ctx* ctx, int nq) {
    int U, grid; bool full; size_t lds;
    fr_config(ctx, nq, U, full, lds, grid);
    return lds > 160 * 1024 - 512;
}
static int fr_combine_grid(const csmp_ctx* ctx) {  // (its partials land in pval / pidx: cus * 8 + 8 entries, solver_alloc)
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(2048, ctx->prop.multiProcessorCount * 8 + 8), (ctx->N + 255) / 256));
}

static int fr_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    int U; bool full; size_t lds;
    fr_config(ctx, 1, U, full, lds, s.fr_grid);
    if (fr_tall(ctx, 2)) {  // (some pass of some solver on this dictionary will take the tall path: srr's two directions at the latest)
        if (!s.frg1) CHECK(dmalloc(ctx, &s.frg1, (size_t)ctx->N));
        if (!s.frg2) CHECK(dmalloc(ctx, &s.frg2, (size_t)ctx->N));
        if (!s.frq) CHECK(dmalloc(ctx, &s.frq, (size_t)s.Mpad));
    }
    if (!s.rho2) CHECK(dmalloc(ctx, &s.rho2, (size_t)ctx->N));
    if (!s.dvec) CHECK(dmalloc(ctx, &s.dvec, (size_t)ctx->N));
    return CSMP_OK;
}

// forward_δ! on a tall dictionary: g = A'q per direction and c = A'r by the product sweep, then k_fr_combine
static int launch_fr_pass_tall(csmp_ctx* ctx, const FrPass& ps, double max_eps, int skipmask) {
    Solver& s = ctx->s;
    const int M = (int)ctx->M;
    if (ps.nq >= 1) {
        const double* q1 = ps.q1;
        if (!q1) {
            hipLaunchKernelGGL(k_fr_lastq, dim3((s.Mpad + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.Q, s.ldq, (const DevState*)s.st, s.Mpad, M, s.frq);
            HIPCHECK(hipGetLastError());
            q1 = s.frq;
        }
        CHECK(launch_sweep(ctx, q1, 0.0, 0, skipmask, s.frg1));
    }
    if (ps.nq == 2) CHECK(launch_sweep(ctx, ps.q2, 0.0, 0, skipmask, s.frg2));
    if (ps.nq < 0) {
        const unsigned grid = (unsigned)((ctx->N + 3) / 4);
        if (ctx->dtype == CSMP_F32)
            hipLaunchKernelGGL(k_fr_colnorm2<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, M, ctx->N, s.frg1);
        else
            hipLaunchKernelGGL(k_fr_colnorm2<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, M, ctx->N, s.frg1);
        HIPCHECK(hipGetLastError());
    }
    CHECK(launch_sweep(ctx, s.r, 0.0, 0, skipmask));  // (last: the sweep's prologue leaves ||r||^2 in the control block)
    const dim3 grid((unsigned)fr_combine_grid(ctx));
    s.fr_grid = (int)grid.x;  // (the arg-max partials the append reads are k_fr_combine's)
#define FR_COMBINE(NQ)                                                                                                                      \
    hipLaunchKernelGGL(k_fr_combine<NQ>, grid, dim3(256), 0, ctx->stream, ctx->N, (const double*)s.cvec, (const double*)s.frg1, ps.s1,       \
                       (const double*)s.frg2, ps.s2, ps.unmark, ps.update_only, s.rho2, s.dvec, s.pval, s.pidx, (const int*)s.sel, s.st, max_eps, \
                       skipmask)
    switch (ps.nq) {
        case -1: FR_COMBINE(-1); break;
        case 0: FR_COMBINE(0); break;
        case 1: FR_COMBINE(1); break;
        default: FR_COMBINE(2); break;
    }
#undef FR_COMBINE
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// forward_δ! + the residual-norm guard of forward_step! (src/forward.jl:59-61,75-82)
static int launch_fr_pass(csmp_ctx* ctx, const FrPass& ps, double max_eps, int skipmask) {
    if (fr_tall(ctx, ps.nq)) return launch_fr_pass_tall(ctx, ps, max_eps, skipmask);
    int U, grid; bool full; size_t lds;
    fr_config(ctx, ps.nq, U, full, lds, grid);
    ctx->s.fr_grid = grid;  // (the partials of THIS pass: a dictionary near the LDS limit mixes fused and tall passes)
    const bool timed = !ps.update_only && prof_pick(ctx);
    if (timed) CHECK(prof_mark(ctx));
    hipError_t e = ctx->dtype == CSMP_F32 ? fr_sweep_launch<float>(ctx, ps, U, full, grid, lds, max_eps, skipmask)
                                          : fr_sweep_launch<double>(ctx, ps, U, full, grid, lds, max_eps, skipmask);
    HIPCHECK(e);
    if (timed) CHECK(prof_mark(ctx));
    return CSMP_OK;
}
static int launch_fr_sweep(csmp_ctx* ctx, bool first, double max_eps, int skipmask) {
    FrPass ps;
    ps.nq = first ? -1 : 1;
    return launch_fr_pass(ctx, ps, max_eps, skipmask);
}

// forward_step!(P, x, max_ε, min_δ): src/forward.jl:56-73
static int fr_step(csmp_ctx* ctx, bool first, double max_eps, double min_d2, bool optimistic) {
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    CHECK(launch_fr_sweep(ctx, first, max_eps, skip));
    return launch_append(ctx, 3, 0, skip, optimistic, min_d2, ctx->s.fr_grid);
}

// Forward regression for up to three signals advanced together (the omp_ticks schedule with the OLS sweep):
// at tick n slot n%3 sweeps, slot (n-1)%3 runs its k_qr1 stage (mode 3), slot (n-2)%3 its k_qr2 stage.
template <typename TA, int U, int NQ>
static hipError_t tick_fr_launch_t(csmp_ctx* ctx, const TickFr<TA>& sw, const TickQr1<TA>& q1, const TickQr2& q2, int G, size_t lds,
                                   double min_d2) {
    auto kern = k_tick_fr<TA, U, NQ>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(2 * G + sw.nblk), dim3(kSweepThreads), lds, ctx->stream, sw, q1, q2, G, min_d2);
    return hipGetLastError();
}
// One pipeline's schedule (the TickPipe of host/omp.hpp with the OLS sweep)
struct FrPipe {
    csmp_ctx* ctx = nullptr;
    bool present[3] = {false, false, false};
    int nblk = 0, U = 8;
    size_t lds = 0;
};
static void fr_pipe_begin(FrPipe& fp, csmp_ctx* ctx, const bool present[3], int64_t k) {
    fp.ctx = ctx;
    for (int q = 0; q < 3; ++q) fp.present[q] = present[q];
    activate_slot(ctx, 0);
    int grid; bool full; size_t flds;
    fr_config(ctx, 1, fp.U, full, flds, grid);
    fp.nblk = grid;
    fp.lds = std::max(flds, qr_lds_bytes((int)std::min<int64_t>(k, ctx->s.kcap)));
}
template <typename TA>
static int fr_pipe_launch(FrPipe& fp, int64_t n, int64_t k, double max_eps, double min_d2, bool optimistic) {
    csmp_ctx* ctx = fp.ctx;
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    Solver* sl[3] = {&ctx->s, &ctx->park[1], &ctx->park[2]};  // (slot 0 is the active one: fr_pipe_begin)
    const int G = sl[0]->G;
    const int zs = (int)(n % 3), ys = (int)((n + 2) % 3), xs = (int)((n + 1) % 3);
    const int64_t tz = (n - zs) / 3, ty = (n - 1 - ys) / 3, tx = (n - 2 - xs) / 3;
    const bool az = fp.present[zs] && n >= zs && tz < k;
    const bool ay = fp.present[ys] && n >= 1 + ys && ty < k && (n - 1 - ys) % 3 == 0;
    const bool ax = fp.present[xs] && n >= 2 + xs && tx < k && (n - 2 - xs) % 3 == 0;
    if (!az && !ay && !ax) return CSMP_OK;
    int jh1 = 0;
    if (ay) {
        jh1 = std::min(sl[ys]->jh, sl[ys]->kcap);
        sl[ys]->jh_last = jh1;
        if (sl[ys]->jh < sl[ys]->kcap) sl[ys]->jh += 1;
    }
    const Solver& z = *sl[zs];
    TickFr<TA> sw;
    sw.A = (const TA*)ctx->dA; sw.ld = ctx->ld; sw.Mv = ctx->Mv; sw.N = ctx->N;
    sw.r = z.r; sw.Q = z.Q; sw.ldq = z.ldq; sw.rho2 = z.rho2; sw.dvec = z.dvec; sw.pval = z.pval; sw.pidx = z.pidx;
    sw.sel = z.sel; sw.st = z.st; sw.max_eps = max_eps; sw.skipmask = skip; sw.nblk = fp.nblk; sw.active = az ? 1 : 0;
    auto q1 = tick_qr1_params<TA>(ctx, *sl[ys], skip, fp.nblk, jh1, ay ? 1 : 0);
    q1.mode = 3;
    auto q2 = tick_qr2_params(ctx, *sl[xs], sl[xs]->jh_last, optimistic ? 1 : 0, ax ? 1 : 0);
    auto launch = [&](const TickFr<TA>& s_, const TickQr1<TA>& a_, const TickQr2& b_, int g_, size_t lds_) -> hipError_t {
        if (fp.U == 16)
            return tz == 0 ? tick_fr_launch_t<TA, 16, -1>(ctx, s_, a_, b_, g_, lds_, min_d2) : tick_fr_launch_t<TA, 16, 1>(ctx, s_, a_, b_, g_, lds_, min_d2);
        return tz == 0 ? tick_fr_launch_t<TA, 8, -1>(ctx, s_, a_, b_, g_, lds_, min_d2) : tick_fr_launch_t<TA, 8, 1>(ctx, s_, a_, b_, g_, lds_, min_d2);
    };
    const bool timed = az && ay && ax && prof_pick(ctx);
    if (timed) CHECK(prof_mark(ctx));
    HIPCHECK(launch(sw, q1, q2, G, fp.lds));
    if (timed) CHECK(prof_mark(ctx));
    return CSMP_OK;
}
template <typename TA>
static int fr_ticks(csmp_ctx* ctx, const bool present[3], int64_t k, double max_eps, double min_d2, bool optimistic) {
    FrPipe fp;
    fr_pipe_begin(fp, ctx, present, k);
    for (int64_t n = 0; n < 3 * k + 2; ++n) CHECK(fr_pipe_launch<TA>(fp, n, k, max_eps, min_d2, optimistic));
    return CSMP_OK;
}
// Two such pipelines side by side (the omp_ticks_pair of host/omp.hpp with the OLS sweep): the ticks of both ask for lds_req bytes of
// LDS -- above half a CU's, one workgroup per CU -- so that the two streams' workgroups queue for the CUs
template <typename TA>
static int fr_ticks_pair(csmp_ctx* ca, const bool pa[3], csmp_ctx* cb, const bool pb[3], int64_t k, double max_eps, double min_d2,
                         bool optimistic, size_t lds_req) {
    FrPipe fa, fb;
    fr_pipe_begin(fa, ca, pa, k);
    fr_pipe_begin(fb, cb, pb, k);
    fa.lds = std::max(fa.lds, lds_req);
    fb.lds = std::max(fb.lds, lds_req);
    for (int64_t n = 0; n < 3 * k + 2; ++n) {
        CHECK(fr_pipe_launch<TA>(fa, n, k, max_eps, min_d2, optimistic));
        const int rb = fr_pipe_launch<TA>(fb, n, k, max_eps, min_d2, optimistic);
        if (rb != CSMP_OK) {
            ca->err = cb->err;
            return rb;
        }
    }
    return CSMP_OK;
}
// fr(A, b, max_ε, min_δ, k) = ols = oomp = ormp, x starting empty: src/forward.jl:44-54
extern "C" int csmp_fr(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double max_eps, double min_delta, int64_t* idx,
                       double* val, int64_t* nnz, int64_t* order) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || k < 0) return fail(ctx, CSMP_EINVAL, "fr: b == NULL or k < 0");
    if (max_eps != max_eps || min_delta != min_delta) return fail(ctx, CSMP_EINVAL, "fr: max_eps / min_delta is NaN");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    CHECK(solver_ensure(ctx, kc, (int)std::max<int64_t>(k, 1)));
    CHECK(fr_ensure(ctx));
    ctx->s.begun = false;
    const double min_d2 = min_delta * min_delta;  // :64
    for (int pass = 0; pass < 2; ++pass) {  // optimistic append chain, repeated with re-orthogonalisation if flagged (see csmp_omp)
        const bool optimistic = pass == 0;
        CHECK(upload_b(ctx, b, b_dtype));
        for (int64_t t = 0; t < k; ++t) {
            CHECK(fr_step(ctx, t == 0, max_eps, min_d2, optimistic));
            if ((t + 1) % kPollSteps == 0 && t + 1 < k) {
                bool stopped = false;
                CHECK(solver_poll(ctx, &stopped));
                if (stopped) break;
            }
        }
        CHECK(launch_finish(ctx, ctx->s.out_idx, ctx->s.out_val, ctx->s.out_nnz, ctx->s.out_order, ctx->s.outcap));
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, ctx->s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (!(hs.done & STOP_REORTH)) {
            break;
        }
    }
    CHECK(download_result(ctx, ctx->s.outcap, idx, val, nnz, order));
    return CSMP_OK;
}

// omp (algo = CSMP_ALGO_OMP: p1 = eps) or fr (CSMP_ALGO_FR: p1 = max_eps, p2 = min_delta^2) for every column of B
static int batch_impl(csmp_ctx* ctx, int algo, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                      double eps, double p2, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    const bool isfr = algo == CSMP_ALGO_FR;
    if (!ctx) return CSMP_EINVAL;
    if (!isfr && !(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");
    if (!B || nsig < 0 || k < 1 || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "batch: bad arguments");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    CHECK(solver_ensure(ctx, kc, (int)k));
    if (isfr) CHECK(fr_ensure(ctx));
    ctx->s.begun = false;
    const size_t es = b_dtype == CSMP_F32 ? 4 : 8;
    void* dB = const_cast<void*>(B);
    DevTmp tB, tIdx, tVal, tNnz;  // freed on every return path
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
    int rc = CSMP_OK;
    activate_slot(ctx, 0);
    if (ctx->s.sigcap < nsig) {
        HIPCHECK(sync_all(ctx));
        dfree(ctx->s.sigflags);
        HIPCHECK(hipMalloc((void**)&ctx->s.sigflags, (size_t)nsig * sizeof(int)));
        ctx->s.sigcap = (int)nsig;
    }
    int* const sigflags = ctx->s.sigflags;  // (a pointer VALUE: ctx->s itself is swapped by activate_slot)
    auto solve_one = [&](int64_t sgn, bool optimistic) -> int {
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        int r2 = b_dtype == CSMP_F32 ? init_from_device_t<float>(ctx, (const float*)col)
                                     : init_from_device_t<double>(ctx, (const double*)col);
        for (int64_t t = 0; t < k && r2 == CSMP_OK; ++t)
            r2 = isfr ? fr_step(ctx, t == 0, eps, p2, optimistic) : omp_step(ctx, eps, t > 0, optimistic);
        if (r2 == CSMP_OK) r2 = launch_finish(ctx, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, nullptr, (int)k, sigflags + sgn);
        return r2;
    };
    // optimistic two-kernel append chain for every signal, no host synchronisation.  Signals are
    // taken three at a time through the tick kernel (k_tick): one launch per atom carries the sweep
    // of one signal and the two short append stages of the other two, so the latency-bound chain
    // is hidden underneath the HBM-bound sweep.  Bit-identical to the one-at-a-time path.
    const bool opt = true;
    // (the tick kernel carries the LDS form of the append stages: supports beyond qr_max_cols() go one signal at a time through
    // launch_append, whose spill kernels have no such bound)
    bool pipe = ctx->pipeline && nsig >= 2 && kc <= qr_max_cols();
    if (isfr) {  // the tick kernel exists for the exact-tiling FR sweeps only
        int U, g; bool full; size_t l;
        fr_config(ctx, 1, U, full, l, g);
        pipe = pipe && full && !fr_tall(ctx, 1);
    }
    int64_t sgn = 0;
    // TWO pipelines: the second half of the triples runs on a twin context and stream beside the first (omp_ticks_pair, fr_ticks_pair).
    // From two signals on and dictionaries of 4 MiB on; csmp_tune(CSMP_TUNE_PIPELINES, 1) keeps one, 2 takes two whatever the size.
    constexpr int64_t kPairMinSignals = 2;
    constexpr size_t kPairMinBytes = (size_t)4 << 20;  // two pipelines: 1 MiB -10 %, 8 MiB +35 %, 32 MiB +40 %, 64 MiB ... 1 GiB +5 ... +16 %
    csmp_ctx* tw = nullptr;
    // (forward regression too: its ticks under the same LDS request -- one workgroup per CU -- 6.28e3 -> 6.57e3 atoms/s at the benchmark
    // shape; without the request 6.47e3.  With the sweep body as round 5 left it the same pairing had measured 5.99e3 against 5.96e3.)
    // ... and where a sweep is long enough for its tail to matter: dictionaries of kPairMinBytes and more (measured: tools/probes/
    // pair_sizes.py); csmp_tune(CSMP_TUNE_PIPELINES, 2) takes two pipelines whatever the size
    const size_t dict_bytes = (size_t)ctx->Mv * (size_t)ctx->N * (ctx->dtype == CSMP_F32 ? 4 : 8);
    if (pipe && nsig >= kPairMinSignals && ctx->tune_pipelines != 1 && (ctx->tune_pipelines == 2 || dict_bytes >= kPairMinBytes)) {
        rc = twins_ensure(ctx, 1);
        if (rc == CSMP_OK) {
            tw = ctx->twins[0];
            tw->prof = ctx->prof;  // (csmp_profile_*: the second pipeline's launches are sampled like the first's)
            tw->prof_every = ctx->prof_every;
            for (int q = 0; q < 3 && rc == CSMP_OK; ++q) {
                activate_slot(tw, q);
                rc = solver_ensure(tw, kc, (int)k);
                if (rc == CSMP_OK && isfr) rc = fr_ensure(tw);
            }
            activate_slot(tw, 0);
            if (rc != CSMP_OK) ctx->err = tw->err;
        }
        if (rc == CSMP_OK && !ctx->ev_twin) HIPCHECK(hipEventCreateWithFlags(&ctx->ev_twin, hipEventDisableTiming));
        if (rc == CSMP_OK && !tw->ev_twin) HIPCHECK(hipEventCreateWithFlags(&tw->ev_twin, hipEventDisableTiming));
        if (rc == CSMP_OK) {  // (the twin starts behind everything this context's stream holds: the caller's buffers, the slots' allocation)
            HIPCHECK(hipEventRecord(ctx->ev_twin, ctx->stream));
            HIPCHECK(hipStreamWaitEvent(tw->stream, ctx->ev_twin, 0));
        }
    }
    if (pipe) {
        for (int q = 1; q < 3 && rc == CSMP_OK; ++q) {
            activate_slot(ctx, q);
            rc = solver_ensure(ctx, kc, (int)k);
            if (rc == CSMP_OK && isfr) rc = fr_ensure(ctx);
        }
        activate_slot(ctx, 0);
        auto init_triple = [&](csmp_ctx* c, int64_t first, int64_t end, bool present[3]) -> int {
            int r2 = CSMP_OK;
            for (int q = 0; q < 3 && r2 == CSMP_OK; ++q) {
                present[q] = first + q < end;
                if (!present[q]) continue;
                activate_slot(c, q);
                const char* col = (const char*)dB + (size_t)(first + q) * (size_t)ldB * es;
                r2 = b_dtype == CSMP_F32 ? init_from_device_t<float>(c, (const float*)col) : init_from_device_t<double>(c, (const double*)col);
            }
            if (r2 != CSMP_OK && c != ctx) ctx->err = c->err;
            return r2;
        };
        auto finish_triple = [&](csmp_ctx* c, int64_t first, const bool present[3]) -> int {
            int r2 = CSMP_OK;
            for (int q = 0; q < 3 && r2 == CSMP_OK; ++q) {
                if (!present[q]) continue;
                activate_slot(c, q);
                r2 = launch_finish(c, d_idx + (first + q) * k, d_val + (first + q) * k, d_nnz + first + q, nullptr, (int)k, sigflags + first + q);
            }
            activate_slot(c, 0);
            if (r2 != CSMP_OK && c != ctx) ctx->err = c->err;
            return r2;
        };
        if (tw && rc == CSMP_OK) {
            // rounds of 3 + 3 signals (this context's pipeline + the twin's), then the remainder in rounds of 1 + 1 and a last lone
            // signal: measured on the 1-GiB dictionary, atoms/s of a whole batch -- 3 + 3: 6680, 1 + 1: 6690, 2 + 2: 6486, and the
            // rounds whose pipelines hold different numbers 3 + 2: 6340, 2 + 1: 6332 (one pipeline of three: 6306, a lone signal:
            // 5830; tools/probes/few_signals.sh).  Two streams with a sweep ready each keep the HBM busy; what costs is a round in
            // which one stream's ticks have sweeps the other's have not.
            const size_t fr_pair_lds = (size_t)(ctx->tune_pair_lds_kib > 0 ? ctx->tune_pair_lds_kib : kPairLdsKiB) * 1024;
            struct Round { int64_t fa; int ca; int64_t fb; int cb; };
            std::vector<Round> rounds;
            {
                int64_t at = 0;
                for (; nsig - at >= 6; at += 6) rounds.push_back({at, 3, at + 3, 3});
                for (; nsig - at >= 2; at += 2) rounds.push_back({at, 1, at + 1, 1});
                if (at < nsig) rounds.push_back({at, 1, 0, 0});
            }
            for (size_t j = 0; j < rounds.size() && rc == CSMP_OK; ++j) {
                bool pa[3], pb[3];
                const int64_t fa = rounds[j].fa, fb = rounds[j].fb;
                const bool hasb = rounds[j].cb > 0;
                rc = init_triple(ctx, fa, fa + rounds[j].ca, pa);
                if (rc == CSMP_OK && hasb) rc = init_triple(tw, fb, fb + rounds[j].cb, pb);
                if (rc != CSMP_OK) break;
                if (hasb && isfr)
                    rc = ctx->dtype == CSMP_F32 ? fr_ticks_pair<float>(ctx, pa, tw, pb, k, eps, p2, opt, fr_pair_lds)
                                                : fr_ticks_pair<double>(ctx, pa, tw, pb, k, eps, p2, opt, fr_pair_lds);
                else if (hasb)
                    rc = ctx->dtype == CSMP_F32 ? omp_ticks_pair<float>(ctx, pa, tw, pb, k, eps, opt, kPairTickGrid)
                                                : omp_ticks_pair<double>(ctx, pa, tw, pb, k, eps, opt, kPairTickGrid);
                else if (isfr)
                    rc = ctx->dtype == CSMP_F32 ? fr_ticks<float>(ctx, pa, k, eps, p2, opt) : fr_ticks<double>(ctx, pa, k, eps, p2, opt);
                else
                    rc = ctx->dtype == CSMP_F32 ? omp_ticks<float>(ctx, pa, k, eps, opt) : omp_ticks<double>(ctx, pa, k, eps, opt);
                if (rc == CSMP_OK) rc = finish_triple(ctx, fa, pa);
                if (rc == CSMP_OK && hasb) rc = finish_triple(tw, fb, pb);
            }
            sgn = nsig;
            // this context's stream goes on behind the twin's last launch (a failed enqueue: drain both before anything is released)
            if (rc == CSMP_OK) {
                HIPCHECK(hipEventRecord(tw->ev_twin, tw->stream));
                HIPCHECK(hipStreamWaitEvent(ctx->stream, tw->ev_twin, 0));
            } else {
                (void)hipStreamSynchronize(tw->stream);
                (void)hipStreamSynchronize(ctx->stream);
            }
        }
        for (; sgn < nsig && rc == CSMP_OK; sgn += 3) {
            bool present[3];
            rc = init_triple(ctx, sgn, nsig, present);
            if (rc == CSMP_OK && isfr)
                rc = ctx->dtype == CSMP_F32 ? fr_ticks<float>(ctx, present, k, eps, p2, opt) : fr_ticks<double>(ctx, present, k, eps, p2, opt);
            else if (rc == CSMP_OK)
                rc = ctx->dtype == CSMP_F32 ? omp_ticks<float>(ctx, present, k, eps, opt) : omp_ticks<double>(ctx, present, k, eps, opt);
            if (rc == CSMP_OK) rc = finish_triple(ctx, sgn, present);
        }
        activate_slot(ctx, 0);
    }
    for (; sgn < nsig && rc == CSMP_OK; ++sgn) rc = solve_one(sgn, opt);
    // ... then ONE synchronisation: a signal whose support failed the DGKS test (flagged on the
    // device, nothing committed for the failing column) is solved again with the full chain
    if (rc == CSMP_OK) {
        std::vector<int> hf((size_t)nsig);
        HIPCHECK(hipMemcpyAsync(hf.data(), sigflags, (size_t)nsig * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        for (int64_t sgn = 0; sgn < nsig && rc == CSMP_OK; ++sgn)
            if (hf[sgn] & STOP_REORTH) rc = solve_one(sgn, false);
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

extern "C" int csmp_omp_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                              double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if ((b_loc != CSMP_HOST && b_loc != CSMP_DEVICE) || (out_loc != CSMP_HOST && out_loc != CSMP_DEVICE))
        return fail(ctx, CSMP_EINVAL, "b_loc / out_loc must be CSMP_HOST or CSMP_DEVICE");
    if (screened_on(ctx) && nsig > 0) {
        if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");
        if (!B || !idx || !val || !nnz || k < 1 || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "batch: bad arguments");
        if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
        if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
        return omp_batch_screened(ctx, B, b_dtype, ldB, nsig, b_loc, k, eps, idx, val, nnz, out_loc);
    }
    return batch_impl(ctx, CSMP_ALGO_OMP, B, b_dtype, ldB, nsig, b_loc, k, eps, 0.0, idx, val, nnz, out_loc);
}

// fr(A, B[:,s], max_eps, min_delta, k) for every column of B: the forward-regression sweeps of three signals
// at a time are pipelined against one another's append stages exactly like csmp_omp_batch's
extern "C" int csmp_fr_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                             double max_eps, double min_delta, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if ((b_loc != CSMP_HOST && b_loc != CSMP_DEVICE) || (out_loc != CSMP_HOST && out_loc != CSMP_DEVICE))
        return fail(ctx, CSMP_EINVAL, "b_loc / out_loc must be CSMP_HOST or CSMP_DEVICE");
    if (max_eps != max_eps || min_delta != min_delta) return fail(ctx, CSMP_EINVAL, "fr_batch: max_eps / min_delta is NaN");
    return batch_impl(ctx, CSMP_ALGO_FR, B, b_dtype, ldB, nsig, b_loc, k, max_eps, min_delta * min_delta, idx, val, nnz, out_loc);
}

// warm start: support/values -> device lists, r = b - A x
static int upload_support(csmp_ctx* ctx, const int64_t* idx0, const double* val0, int64_t nnz0) {
    Solver& s = ctx->s;
    std::vector<int> hi((size_t)nnz0);
    for (int64_t t = 0; t < nnz0; ++t) {
        if (idx0[t] < 0 || idx0[t] >= ctx->N) return fail(ctx, CSMP_ERANGE, "warm start: index out of range");
        hi[t] = (int)idx0[t];
    }
    HIPCHECK(hipMemcpyAsync(s.cands, hi.data(), (size_t)nnz0 * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipMemcpyAsync(s.coef, val0, (size_t)nnz0 * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    const int grid = ((int)ctx->M + 255) / 256;
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_residual<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M,
                           (const int*)s.cands, (const double*)s.coef, (const int*)nullptr, (int)nnz0, (const double*)s.b, s.r);
    else
        hipLaunchKernelGGL(k_residual<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M,
                           (const int*)s.cands, (const double*)s.coef, (const int*)nullptr, (int)nnz0, (const double*)s.b, s.r);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// MP bookkeeping on the host side of the boundary: the device returns the k (atom, <a,r>) pairs in
// step order; x[i] += d is replayed in that order (same summation order as src/matchingpursuit.jl:29)
static int mp_collect(csmp_ctx* ctx, const int64_t* idx0, const double* val0, int64_t nnz0, int64_t* idx, double* val,
                      int64_t* nnz) {
    Solver& s = ctx->s;
    DevState hs;
    HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    const int n = hs.nsel;
    std::vector<int> hsel((size_t)std::max(n, 1));
    std::vector<double> hz((size_t)std::max(n, 1));
    HIPCHECK(hipMemcpy(hsel.data(), s.sel, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hz.data(), s.z, (size_t)n * 8, hipMemcpyDeviceToHost));
    std::vector<std::pair<int64_t, double>> x;
    for (int64_t t = 0; t < nnz0; ++t) x.push_back({idx0[t], val0[t]});
    std::sort(x.begin(), x.end(), [](auto& a, auto& c) { return a.first < c.first; });
    for (int t = 0; t < n; ++t) {
        auto it = std::lower_bound(x.begin(), x.end(), (int64_t)hsel[t], [](auto& a, int64_t v) { return a.first < v; });
        if (it != x.end() && it->first == hsel[t])
            it->second += hz[t];
        else if (hz[t] != 0.0)  // SparseVector setindex! does not store a structural zero
            x.insert(it, {(int64_t)hsel[t], hz[t]});
    }
    for (size_t t = 0; t < x.size(); ++t) {
        if (idx) idx[t] = x[t].first;
        if (val) val[t] = x[t].second;
    }
    if (nnz) *nnz = (int64_t)x.size();
    return CSMP_OK;
}

static int mp_step(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.jh >= s.kcap) return fail(ctx, CSMP_ERANGE, "mp: more steps than the capacity this solver was begun with");
    s.jh += 1;  // (MP: steps taken; the log of (atom, coefficient) pairs holds kcap of them)
    CHECK(launch_sweep(ctx, ctx->s.r, 0.0, 0, 0));
    CHECK(launch_select(ctx, 0, 0));
    return launch_mp_update(ctx);
}

extern "C" int csmp_mp(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, const int64_t* idx0, const double* val0,
                       int64_t nnz0, int64_t* idx, double* val, int64_t* nnz) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || k < 0 || nnz0 < 0 || (nnz0 > 0 && (!idx0 || !val0))) return fail(ctx, CSMP_EINVAL, "mp: bad arguments");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(solver_ensure(ctx, (int)std::max<int64_t>(std::max(k, nnz0), 1), 1, false));  // MP keeps no factorisation
    ctx->s.begun = false;
    // CSMP_OPT_SCREENED_SWEEP: the sweeps read the image, every pick is certified (host/screened.hpp); a solve with an uncertified
    // pick is repeated with the exact sweep
    bool screened = screened_on(ctx);
    if (screened) CHECK(screened_ensure(ctx));
    ctx->scr_lone = true;  // (one solve at a time: reset on every way out below)
    struct LoneReset {
        csmp_ctx* c;
        ~LoneReset() { c->scr_lone = false; }
    } lone_reset{ctx};
    for (int attempt = 0; attempt < 2; ++attempt) {
        CHECK(upload_b(ctx, b, b_dtype));
        if (nnz0 > 0) CHECK(upload_support(ctx, idx0, val0, nnz0));
        for (int64_t t = 0; t < k; ++t) {
            if (screened) {
                Solver& s = ctx->s;
                if (s.jh >= s.kcap) return fail(ctx, CSMP_ERANGE, "mp: more steps than the capacity this solver was begun with");
                s.jh += 1;
                CHECK(mp_step_screened(ctx));
            } else {
                CHECK(mp_step(ctx));
            }
        }
        if (!screened) break;
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, ctx->s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        ctx->scr_solves += 1;
        if (hs.uncertain == 0) break;
        ctx->scr_fallbacks += 1;
        screened = false;
    }
    return mp_collect(ctx, idx0, val0, nnz0, idx, val, nnz);
}
