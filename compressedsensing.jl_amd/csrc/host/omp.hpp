// host/omp.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// the tick pipeline (three signals in flight) and the omp driver.
// ------------------------------------------------------------------------------------------ tick kernel (3 signals in flight)
template <typename TA>
static TickSweep<TA> tick_sweep_params(csmp_ctx* ctx, Solver& s, double eps, int check_eps, int skipmask, int nblk, int active) {
    TickSweep<TA> p;
    p.claim = p.claim_next = nullptr;
    p.npools = ctx->claim_pools | (ctx->tune_sweep_dyn << 16);
    if (active && ctx->sweep_dyn) claim_sets(s, p.claim, p.claim_next);
    p.A = (const TA*)ctx->dA; p.ld = ctx->ld; p.Mv = ctx->Mv; p.N = ctx->N;
    p.r = s.r; p.cvec = s.cvec; p.pval = s.pval; p.pidx = s.pidx; p.st = s.st;
    p.eps = eps; p.check_eps = check_eps; p.skipmask = skipmask; p.nblk = nblk; p.active = active; p.KP = ctx->sweep_KP;
    p.pcap = ctx->sweep_pcap;
    return p;
}
template <typename TA>
static TickQr1<TA> tick_qr1_params(csmp_ctx* ctx, const Solver& s, int skipmask, int nblk_sweep, int jh, int active) {
    TickQr1<TA> p;
    p.A = (const TA*)ctx->dA; p.ld = ctx->ld; p.M = (int)ctx->M;
    p.Q = s.Q; p.ldq = s.ldq; p.st = s.st; p.avec = s.avec; p.P1 = s.P1;
    p.G = s.G; p.kcap = s.kcap; p.jpad = qr_jpad(jh); p.mode = 1;
    p.pval = s.pval; p.pidx = s.pidx; p.nblk_sweep = nblk_sweep;
    p.cands = s.cands; p.ncands = s.ncands; p.which = 0; p.sel = s.sel; p.skipmask = skipmask;
    p.r = s.r; p.P1s = s.P1s; p.jh = jh; p.active = active;
    return p;
}
static TickQr2 tick_qr2_params(csmp_ctx* ctx, const Solver& s, int jh, int optimistic, int active) {
    TickQr2 p;
    p.Q = s.Q; p.ldq = s.ldq; p.st = s.st; p.avec = s.avec; p.r = s.r;
    p.P1 = s.P1; p.P1s = s.P1s; p.G = s.G;
    p.W1 = s.W1; p.vvec = s.vvec; p.P2 = s.P2; p.P2s = s.P2s; p.R = s.R; p.z = s.z; p.sel = s.sel;
    p.kcap = s.kcap; p.jpad = qr_jpad(jh); p.force_reorth = 0; p.jh = jh; p.optimistic = optimistic;
    p.active = active;
    return p;
}

template <typename TA, int U, bool PH, bool STEADY = false, bool DYN = false>
static hipError_t tick_launch_t(csmp_ctx* ctx, const TickSweep<TA>& sw, const TickQr1<TA>& q1, const TickQr2& q2, int G, size_t lds) {
    auto kern = k_tick<TA, U, PH, STEADY, DYN>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(2 * G + sw.nblk), dim3(DYN ? kSweepDynThreads : kSweepThreads), lds, ctx->stream, sw, q1, q2, G, ctx->tick_sweep_first ? 1 : 0);
    return hipGetLastError();
}
// steady: all three stages of this tick are live (the launches the bench's roofline is quoted on)
template <typename TA, int U, bool PH, bool DYN = false>
static hipError_t tick_launch_s(csmp_ctx* ctx, const TickSweep<TA>& sw, const TickQr1<TA>& q1, const TickQr2& q2, int G, size_t lds, bool steady) {
    return steady ? tick_launch_t<TA, U, PH, true, DYN>(ctx, sw, q1, q2, G, lds) : tick_launch_t<TA, U, PH, false, DYN>(ctx, sw, q1, q2, G, lds);
}
template <typename TA>
static hipError_t tick_launch(csmp_ctx* ctx, const TickSweep<TA>& sw, const TickQr1<TA>& q1, const TickQr2& q2, int G, size_t lds, bool steady) {
    if (ctx->sweep_ph) return tick_launch_s<TA, 8, true>(ctx, sw, q1, q2, G, lds, steady);
    if (ctx->sweep_dyn) {
        switch (ctx->sweep_U) {
            case 16: return tick_launch_s<TA, 16, false, true>(ctx, sw, q1, q2, G, lds, steady);
            case 8: return tick_launch_s<TA, 8, false, true>(ctx, sw, q1, q2, G, lds, steady);
            default: return tick_launch_s<TA, 4, false, true>(ctx, sw, q1, q2, G, lds, steady);
        }
    }
    switch (ctx->sweep_U) {
        case 16: return tick_launch_s<TA, 16, false>(ctx, sw, q1, q2, G, lds, steady);
        case 8: return tick_launch_s<TA, 8, false>(ctx, sw, q1, q2, G, lds, steady);
        default: return tick_launch_s<TA, 4, false>(ctx, sw, q1, q2, G, lds, steady);
    }
}

// OMP for up to three signals (solver slots 0..2, already initialised with their b) advanced
// together: at tick n slot n%3 sweeps, slot (n-1)%3 runs its k_qr1 stage, slot (n-2)%3 its k_qr2
// stage.  k steps per signal = 3k+2 ticks.  present[q] == false leaves slot q idle.
// One pipeline's schedule: which slots are live at tick n, and the launch of that tick.
struct TickPipe {
    csmp_ctx* ctx = nullptr;
    bool present[3] = {false, false, false};
    int nblk = 0;
    size_t lds = 0;
    size_t lds_sweep = 0;  // > 0: a tick is TWO launches -- the append stages alone (lds), then the sweep alone under this LDS request
};
static void tick_pipe_begin(TickPipe& tp, csmp_ctx* ctx, const bool present[3], int64_t k, int grid_override) {
    tp.ctx = ctx;
    for (int q = 0; q < 3; ++q) tp.present[q] = present[q];
    activate_slot(ctx, 0);
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    // Measured at 4096 x 65536 f32: 8-chunk load blocks on ONE workgroup per CU (the append stages of the other two
    // signals share those CUs) 160.4 us per tick; 16-chunk blocks on 176 workgroups (11/12 of the stand-alone sweep's
    // optimum of 192) 162.6 us.
    const int64_t auto_nblk = grid_override > 0 ? (int64_t)grid_override : (int64_t)ctx->tick_grid;  // (configure_sweep)
    tp.nblk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ctx->tick_nblk > 0 ? ctx->tick_nblk : auto_nblk, groups),
                                                         ctx->prop.multiProcessorCount * 8 + 8));  // (pval / pidx: solver_alloc)
    tp.lds = std::max(ctx->sweep_lds, qr_lds_bytes((int)std::min<int64_t>(k, ctx->s.kcap)));  // (jh never exceeds k here)
}
template <typename TA>
static int tick_pipe_launch(TickPipe& tp, int64_t n, int64_t k, double eps, bool optimistic) {
    csmp_ctx* ctx = tp.ctx;
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    Solver* sl[3] = {&ctx->s, &ctx->park[1], &ctx->park[2]};  // (slot 0 is the active one: tick_pipe_begin)
    const int G = sl[0]->G;
    const int zs = (int)(n % 3), ys = (int)((n + 2) % 3), xs = (int)((n + 1) % 3);  // sweep, qr1, qr2 slots
    const int64_t tz = (n - zs) / 3, ty = (n - 1 - ys) / 3, tx = (n - 2 - xs) / 3;
    const bool az = tp.present[zs] && n >= zs && tz < k;
    const bool ay = tp.present[ys] && n >= 1 + ys && ty < k && (n - 1 - ys) % 3 == 0;
    const bool ax = tp.present[xs] && n >= 2 + xs && tx < k && (n - 2 - xs) % 3 == 0;
    if (!az && !ay && !ax) return CSMP_OK;
    int jh1 = 0;
    if (ay) {
        jh1 = std::min(sl[ys]->jh, sl[ys]->kcap);
        sl[ys]->jh_last = jh1;
        if (sl[ys]->jh < sl[ys]->kcap) sl[ys]->jh += 1;
    }
    const auto sw = tick_sweep_params<TA>(ctx, *sl[zs], eps, tz > 0 ? 1 : 0, skip, tp.nblk, az ? 1 : 0);
    const auto q1 = tick_qr1_params<TA>(ctx, *sl[ys], skip, tp.nblk, jh1, ay ? 1 : 0);
    const auto q2 = tick_qr2_params(ctx, *sl[xs], sl[xs]->jh_last, optimistic ? 1 : 0, ax ? 1 : 0);
    // steady: all three stages live (one pipeline: the launches the roofline is quoted on).  Two pipelines: a tick's sweep is a
    // launch of its own, the same work whatever the other stages do -- every one of them counts, the fill and drain ticks' too
    const bool steady = tp.lds_sweep > 0 ? az : (az && ay && ax);
    const bool timed = steady && prof_pick(ctx);
    if (tp.lds_sweep > 0) {
        // the append stages first, in a launch of their own that asks for what they need (it shares the CUs with the OTHER pipeline's
        // sweep), then the sweep alone with the large LDS request that keeps its workgroups one to a CU (omp_ticks_pair)
        if (ay || ax) {
            auto sw0 = sw;
            sw0.active = 0;
            sw0.nblk = 0;
            HIPCHECK(tick_launch<TA>(ctx, sw0, q1, q2, G, tp.lds, false));
        }
        if (az) {
            auto q10 = q1;
            auto q20 = q2;
            q10.active = 0;
            q20.active = 0;
            if (timed) CHECK(prof_mark(ctx));
            HIPCHECK(tick_launch<TA>(ctx, sw, q10, q20, 0, tp.lds_sweep, steady));
            if (timed) CHECK(prof_mark(ctx));
        }
        return CSMP_OK;
    }
    if (timed) CHECK(prof_mark(ctx));
    HIPCHECK(tick_launch<TA>(ctx, sw, q1, q2, G, tp.lds, steady));
    if (timed) CHECK(prof_mark(ctx));
    return CSMP_OK;
}
template <typename TA>
static int omp_ticks(csmp_ctx* ctx, const bool present[3], int64_t k, double eps, bool optimistic) {
    TickPipe tp;
    tick_pipe_begin(tp, ctx, present, k, 0);
    for (int64_t n = 0; n < 3 * k + 2; ++n) CHECK(tick_pipe_launch<TA>(tp, n, k, eps, optimistic));
    return CSMP_OK;
}
constexpr int kPairLdsKiB = 81;     // dynamic LDS of a tick of two pipelines side by side: more than half a CU's 160 KiB = one workgroup per CU
constexpr int kPairTickGrid = 192;  // sweep workgroups of each of two pipelines side by side (measured: 176 -> 6.41e3, 192 -> 6.46e3, 224 -> 6.45e3, 256 -> 6.41e3 atoms/s)
// TWO pipelines side by side: a second triple of signals on a twin context (its own stream), the launches of the two enqueued
// alternately.  The sweeps of the two then share the HBM, out of step with one another: the last workgroups of one tick, its
// launch boundary and the staging of its residual image fall under the other pipeline's stream instead of leaving the memory
// system idle (6.03e3 -> 6.46e3 atoms/s with 192 sweep workgroups each).
template <typename TA>
static int omp_ticks_pair(csmp_ctx* a, const bool pa[3], csmp_ctx* b, const bool pb[3], int64_t k, double eps, bool optimistic, int grid) {
    TickPipe ta, tb;
    tick_pipe_begin(ta, a, pa, k, grid);
    tick_pipe_begin(tb, b, pb, k, grid);
    // ONE workgroup per CU (an LDS request above half of the 160 KiB): the workgroups of the two pipelines' launches then QUEUE for the
    // CUs instead of all being resident at once, and the dispatcher hands a CU that a workgroup of one tick has left to the next
    // workgroup in line -- of the other pipeline's tick, whose sweep does not depend on this one.  The chip is never waiting for the
    // slowest workgroups of a launch (they finish 139 ... 160 us into a 157-us sweep), for a launch boundary or for a residual image.
    const size_t excl = (size_t)(a->tune_pair_lds_kib > 0 ? a->tune_pair_lds_kib : kPairLdsKiB) * 1024;
    if (a->tune_pair_split == 1) {  // (measurement: the fused tick under the large request)
        ta.lds = std::max(ta.lds, excl);
        tb.lds = std::max(tb.lds, excl);
    } else {
        ta.lds_sweep = std::max(a->sweep_lds, excl);
        tb.lds_sweep = std::max(b->sweep_lds, excl);
        ta.lds = qr_lds_bytes((int)std::min<int64_t>(k, a->s.kcap));  // (the append stages' launch asks for what IT needs)
        tb.lds = qr_lds_bytes((int)std::min<int64_t>(k, b->s.kcap));
    }
    for (int64_t n = 0; n < 3 * k + 2; ++n) {
        CHECK(tick_pipe_launch<TA>(ta, n, k, eps, optimistic));
        const int rc = tick_pipe_launch<TA>(tb, n, k, eps, optimistic);
        if (rc != CSMP_OK) {
            a->err = b->err;
            return rc;
        }
    }
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ drivers
extern "C" int csmp_omp(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double eps, int64_t* idx, double* val,
                        int64_t* nnz, int64_t* order) {
    if (!ctx) return CSMP_EINVAL;
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");  // src/matchingpursuit.jl:74
    if (!b || k < 0) return fail(ctx, CSMP_EINVAL, "omp: b == NULL or k < 0");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));  // UpdatableQR(T, n, k): :58
    CHECK(solver_ensure(ctx, kc, (int)std::max<int64_t>(k, 1)));
    ctx->s.begun = false;
    // CSMP_OPT_SCREENED_SWEEP: first with the bf16-image sweep + certified picks (csmp_screened.hpp); a solve in which a pick
    // could not be certified is repeated with the exact sweep.
    // Within an attempt: optimistic two-kernel append chain first; if any column failed the DGKS test (flagged on the
    // device, nothing committed) the solve is repeated with the second Gram-Schmidt pass enabled
    bool screened = screened_on(ctx);
    if (screened) CHECK(screened_ensure(ctx));
    struct LoneGuard {  // (csmp_omp is one solve at a time)
        csmp_ctx* c;
        explicit LoneGuard(csmp_ctx* x) : c(x) { c->scr_lone = true; }
        ~LoneGuard() { c->scr_lone = false; }
    } lone_guard(ctx);
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool uncertain = false;
        for (int pass = 0; pass < 2; ++pass) {
            const bool optimistic = pass == 0;
            CHECK(upload_b(ctx, b, b_dtype));
            for (int64_t t = 0; t < k; ++t) {
                CHECK(screened ? omp_step_screened(ctx, eps, t > 0, optimistic) : omp_step(ctx, eps, t > 0, optimistic));
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
            uncertain = screened && hs.uncertain > 0;
            if (!(hs.done & STOP_REORTH)) {
                break;
            }
        }
        if (screened) {
            ctx->scr_solves += 1;
            ctx->scr_fallbacks += uncertain ? 1 : 0;
        }
        if (!uncertain) break;
        screened = false;
    }
    CHECK(download_result(ctx, ctx->s.outcap, idx, val, nnz, order));
    return CSMP_OK;
}
