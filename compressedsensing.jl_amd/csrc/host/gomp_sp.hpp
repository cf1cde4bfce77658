// host/gomp_sp.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// multi-column append, top-S, gomp (+ batch form), whole-set least squares, Subspace Pursuit (+ batch form), primitives.
// ------------------------------------------------------------------------------------------ multi-column append
static int block_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.blk_kcap >= s.kcap && s.Apan) return CSMP_OK;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    dfree(s.Apan); dfree(s.Vpan); dfree(s.PB1); dfree(s.W1b); dfree(s.PG); dfree(s.Gsum); dfree(s.pan_atoms);
    constexpr int nent = kPanelMax * kPanelMax + 2 * kPanelMax;
    CHECK(dmalloc(ctx, &s.Apan, (size_t)kPanelMax * s.ldq));
    CHECK(dmalloc(ctx, &s.Vpan, (size_t)kPanelMax * s.ldq));
    CHECK(dmalloc(ctx, &s.PB1, (size_t)s.kcap * kPanelMax * s.G));
    CHECK(dmalloc(ctx, &s.W1b, (size_t)s.kcap * kPanelMax));
    CHECK(dmalloc(ctx, &s.PG, (size_t)nent * s.G));
    CHECK(dmalloc(ctx, &s.Gsum, (size_t)nent));
    CHECK(dmalloc(ctx, &s.pan_atoms, kPanelMax));
    s.blk_kcap = s.kcap;
    return CSMP_OK;
}

// add_column! for up to PB atoms cands[base .. base+want) at once (atoms already in the support are skipped)
template <typename TA, int PB>
static int launch_block_append_t(csmp_ctx* ctx, int base, int want, int skipmask) {
    Solver& s = ctx->s;
    const int jh = std::min(s.jh, s.kcap);
    const size_t l1 = blk1_lds_bytes<PB>(), l2 = blk2_lds_bytes<PB>(), l3 = blk3_lds_bytes<PB>();
    if (l1 > 64 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_blk1<TA, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1));
    if (l2 > 64 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_blk2<PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
    if (l3 > 64 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_blk3<PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
    const int csplit = std::max(1, std::min(std::min(4, ctx->prop.multiProcessorCount / std::max(1, s.G)), (jh + kWave - 1) / kWave));
    hipLaunchKernelGGL((k_blk1<TA, PB>), dim3(s.G, csplit), dim3(kQrThreads), l1, ctx->stream, (const TA*)ctx->dA, ctx->ld, (int)ctx->M,
                       (const double*)s.Q, s.ldq, s.st, (const int*)s.cands, (const int*)s.ncands, base, want, (const int*)s.sel,
                       s.kcap, skipmask, s.Apan, s.PB1, s.G, s.pan_atoms);
    HIPCHECK(hipGetLastError());
    const int n1 = jh * PB;
    if (n1 > 0) {
        hipLaunchKernelGGL(k_red, dim3((n1 + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.PB1, s.W1b, n1, s.G, (int64_t)s.kcap * PB, (const DevState*)s.st, s.R, s.kcap, PB);
        HIPCHECK(hipGetLastError());
    }
    hipLaunchKernelGGL((k_blk2<PB>), dim3(s.G), dim3(kQrThreads), l2, ctx->stream, (const double*)s.Q, s.ldq, (const DevState*)s.st,
                       (const double*)s.Apan, (const double*)s.W1b, (const double*)s.r, s.Vpan, s.PG, s.G);
    HIPCHECK(hipGetLastError());
    constexpr int nent = blk2_nent<PB>();
    hipLaunchKernelGGL(k_red, dim3((nent + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.PG, s.Gsum, nent, s.G, (int64_t)nent, (const DevState*)s.st, (double*)nullptr, 0, PB);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL((k_blk3<PB>), dim3(s.G), dim3(kQrThreads), l3, ctx->stream, s.Q, s.ldq, s.st, (const double*)s.Vpan,
                       (const double*)s.Gsum, (const double*)s.W1b, s.r, s.R, s.z, s.sel, (const int*)s.pan_atoms, s.kcap);
    HIPCHECK(hipGetLastError());
    s.jh = std::min(s.kcap, s.jh + std::min(want, PB));
    return CSMP_OK;
}

// panels of <= 32 atoms over cands[0 .. n)
static int launch_block_appends(csmp_ctx* ctx, int n, int skipmask) {
    CHECK(block_ensure(ctx));
    for (int base = 0; base < n;) {
        const int want = std::min(n - base, kPanelMax);
        int rc;
        if (want <= 4)
            rc = ctx->dtype == CSMP_F32 ? launch_block_append_t<float, 4>(ctx, base, want, skipmask)
                                        : launch_block_append_t<double, 4>(ctx, base, want, skipmask);
        else
            rc = ctx->dtype == CSMP_F32 ? launch_block_append_t<float, kPanelMax>(ctx, base, want, skipmask)
                                        : launch_block_append_t<double, kPanelMax>(ctx, base, want, skipmask);
        CHECK(rc);
        base += want;
    }
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ top-S, GOMP, LS, SP
// cands[0..S) <- the S atoms with the largest |c| (descending, ties by ascending index), on device
static int launch_topS(csmp_ctx* ctx, int S) {
    Solver& s = ctx->s;
    if (S < 1 || S > s.kcap) return fail(ctx, CSMP_ERANGE, "top-S: S out of range");
    if (S <= kTopSmall && (size_t)s.top_nb * S * sizeof(double) <= 48 * 1024) {
        hipLaunchKernelGGL(k_top_local, dim3(s.top_nb), dim3(256), 0, ctx->stream, (const double*)s.cvec, ctx->N, S, s.top_lv, s.top_li);
        HIPCHECK(hipGetLastError());
        const int n = s.top_nb * S;
        hipLaunchKernelGGL(k_top_merge, dim3(1), dim3(256), (size_t)n * sizeof(double), ctx->stream, (const double*)s.top_lv,
                           (const int*)s.top_li, n, S, s.cands, s.cvals, s.ncands);
        HIPCHECK(hipGetLastError());
        return CSMP_OK;
    }
    const int S_eff = (int)std::min<int64_t>(S, ctx->N);
    const int grid = (int)std::min<int64_t>((ctx->N + 255) / 256, (int64_t)ctx->prop.multiProcessorCount * 4);
    hipLaunchKernelGGL(k_rs_init, dim3(1), dim3(256), 0, ctx->stream, s.rs, S_eff);
    // (few, fat workgroups: every workgroup flushes its non-empty bins with global atomics, and pass 0 -- the exponent -- puts
    // all keys into a dozen bins)
    const int hgrid = (int)std::min<int64_t>((ctx->N + 2047) / 2048, (int64_t)ctx->prop.multiProcessorCount);
    for (int pass = 0; pass < 2; ++pass)  // the exponent, the leading 11 mantissa bits
        hipLaunchKernelGGL(k_rs_hist, dim3(hgrid), dim3(256), 0, ctx->stream, (const double*)s.cvec, ctx->N, s.rs, kRsSettle);
    hipLaunchKernelGGL(k_rs_tail, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.cvec, ctx->N, s.rs, kRsSettle);  // (a no-op once settled)
    hipLaunchKernelGGL(k_rs_collect, dim3(grid), dim3(256), 0, ctx->stream, (const double*)s.cvec, ctx->N, s.rs, s.rs_gt, s.rs_eq, kRsEqCap);
    const int pairs = S_eff <= 4096 ? S_eff : 0;  // (value, index) pairs of the final rank sort staged in LDS
    const size_t lds = (size_t)pairs * 12 + 16 + (size_t)kRsEqCap * 12 + 16;
    HIPCHECK(hipFuncSetAttribute((const void*)k_rs_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_rs_finish, dim3((S_eff + 255) / 256), dim3(256), lds, ctx->stream, (const double*)s.cvec, ctx->N, s.rs,
                       (const int*)s.rs_gt, (const int*)s.rs_eq, kRsEqCap, s.rs_work, s.cands, s.cvals, s.ncands, pairs);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// update!(P::GOMP, x, l): src/matchingpursuit.jl:116-123
static int gomp_update(csmp_ctx* ctx, int64_t l, double eps, int check_eps, int skipmask, bool block = false) {
    l = std::min<int64_t>(l, ctx->N);
    CHECK(launch_sweep(ctx, ctx->s.r, eps, check_eps, skipmask));
    CHECK(launch_topS(ctx, (int)l));
    if (block && l > 1) return launch_block_appends(ctx, (int)l, skipmask);  // the l atoms join the QR together
    for (int64_t w = 0; w < l; ++w) CHECK(launch_append(ctx, 2, (int)w, skipmask));
    return CSMP_OK;
}

extern "C" int csmp_gomp(csmp_ctx* ctx, const void* b, int b_dtype, int64_t l, int64_t k, double eps, int64_t* idx,
                         double* val, int64_t* nnz, int64_t* order) {
    if (!ctx) return CSMP_EINVAL;
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");  // src/matchingpursuit.jl:127
    if (!b || k < 0 || l < 1) return fail(ctx, CSMP_EINVAL, "gomp: b == NULL, k < 0 or l < 1");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    // GOMP(A,b,l): QR capacity M (:108,:128); at most k atoms are ever added, and top-l needs l slots
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(k, l), std::max<int64_t>(ctx->M, l)));
    CHECK(solver_ensure(ctx, kc, (int)std::max<int64_t>(k + l, 1)));
    ctx->s.begun = false;
    // first with the multi-column append (the l atoms of a step join the QR in one panel); a panel
    // that fails its DGKS test flags the solve, which is then repeated with the column-wise chain.  With
    // CSMP_OPT_SCREENED_SWEEP the sweeps read the bf16 image and the top-l pick is certified (host/screened.hpp); a solve
    // with an uncertified step is repeated with the exact sweep.
    bool screened = screened_on(ctx) && l <= kTopSmall;
    if (screened) CHECK(screened_ensure(ctx));
    ctx->scr_lone = true;  // (one solve at a time: the pick kernel may take a whole CU)
    struct LoneReset {
        csmp_ctx* c;
        ~LoneReset() { c->scr_lone = false; }
    } lone_reset{ctx};
    bool block = l <= kPanelMax;
    for (int attempt = 0; attempt < 4; ++attempt) {
        CHECK(upload_b(ctx, b, b_dtype));
        const int main_skip = STOP_EPS | STOP_FULL | STOP_REORTH;
        for (int64_t it = 0; it < k / l; ++it) {  // :130-133
            CHECK(screened ? gomp_update_screened(ctx, l, eps, it > 0, main_skip, block) : gomp_update(ctx, l, eps, it > 0, main_skip, block));
            if ((it + 1) % kPollSteps == 0 && it + 1 < k / l) {
                bool stopped = false;
                CHECK(solver_poll(ctx, &stopped));
                if (stopped) break;  // (the remainder step below still runs, as in the reference)
            }
        }
        const int64_t rem = k % l;  // :134
        if (rem > 0)  // :135-137: runs even after an eps-break
            CHECK(screened ? gomp_update_screened(ctx, rem, 0.0, 0, STOP_FULL | STOP_REORTH, block) : gomp_update(ctx, rem, 0.0, 0, STOP_FULL | STOP_REORTH, block));
        CHECK(launch_finish(ctx, ctx->s.out_idx, ctx->s.out_val, ctx->s.out_nnz, ctx->s.out_order, ctx->s.outcap));
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, ctx->s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (screened) {
            ctx->scr_solves += 1;
            if (hs.uncertain > 0) {
                ctx->scr_fallbacks += 1;
                screened = false;
                continue;
            }
        }
        if ((hs.done & STOP_REORTH) && block) {
            block = false;
            continue;
        }
        break;
    }
    CHECK(download_result(ctx, ctx->s.outcap, idx, val, nnz, order));
    return CSMP_OK;
}

// ---- gomp for many signals: TWO solves in flight, one per stream
// One signal's step is a chain: the dictionary sweep (HBM-bound, 0.6 ms at config 5), then top-S and the panel append (eight
// short kernels, ~55 us, a fraction of the chip) -- nothing of the same signal can run beside them.  Another signal's sweep
// can: signals alternate between this context and a twin (a clone on its own stream), everything is enqueued up front, and
// the twin's first sweep is held back until this context's first sweep has finished, so that the two chains run OUT of
// phase: each signal's short stages fall under the other's sweep (in phase they would fall on each other).  Results are
// those of csmp_gomp signal by signal (the same kernels in the same order on each stream).
// the first n twins exist and carry this context's options
static int twins_ensure(csmp_ctx* ctx, int n) {
    for (int t = 0; t < n; ++t) {
        if (!ctx->twins[t]) {
            const int rc = csmp_clone(ctx, &ctx->twins[t]);
            if (rc != CSMP_OK) return rc;
        }
        copy_options(ctx->twins[t], ctx);
    }
    return CSMP_OK;
}

static int gomp_enqueue(csmp_ctx* c, const void* col_dev, int b_dtype, int64_t l, int64_t k, double eps, bool block, int64_t* d_idx,
                        double* d_val, int64_t* d_nnz, int* d_flag, hipEvent_t after_first_sweep, bool screened = false) {
    int rc = b_dtype == CSMP_F32 ? init_from_device_t<float>(c, (const float*)col_dev) : init_from_device_t<double>(c, (const double*)col_dev);
    if (rc != CSMP_OK) return rc;
    const int main_skip = STOP_EPS | STOP_FULL | STOP_REORTH;
    for (int64_t it = 0; it < k / l; ++it) {  // src/matchingpursuit.jl:130-133
        rc = screened ? gomp_update_screened(c, l, eps, it > 0, main_skip, block) : gomp_update(c, l, eps, it > 0, main_skip, block);
        if (rc != CSMP_OK) return rc;
        if (it == 0 && after_first_sweep && hipEventRecord(after_first_sweep, c->stream) != hipSuccess) return CSMP_EHIP;
    }
    const int64_t rem = k % l;
    if (rem > 0) {  // :134-137: runs even after an eps-break
        rc = screened ? gomp_update_screened(c, rem, 0.0, 0, STOP_FULL | STOP_REORTH, block) : gomp_update(c, rem, 0.0, 0, STOP_FULL | STOP_REORTH, block);
        if (rc != CSMP_OK) return rc;
    }
    return launch_finish(c, d_idx, d_val, d_nnz, nullptr, (int)k, d_flag);
}

extern "C" int csmp_gomp_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t l, int64_t k,
                               double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if ((b_loc != CSMP_HOST && b_loc != CSMP_DEVICE) || (out_loc != CSMP_HOST && out_loc != CSMP_DEVICE))
        return fail(ctx, CSMP_EINVAL, "b_loc / out_loc must be CSMP_HOST or CSMP_DEVICE");
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");  // src/matchingpursuit.jl:127
    if (!B || nsig < 0 || k < 1 || l < 1 || l > k || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "gomp_batch: bad arguments (needs 1 <= l <= k)");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (nsig == 0) return CSMP_OK;
    HIPCHECK(hipSetDevice(ctx->dev));
    // solves in flight: two (the context and a twin) on the exact sweep -- two 4-GiB sweeps already share the HBM -- and up to
    // three (CSMP_OPT_SOLVES_IN_FLIGHT) on the screened sweep, whose launches are short enough for their fixed parts to matter
    const bool screened = screened_on(ctx) && l <= kTopSmall;  // sweeps over the image, certified top-l picks (host/screened.hpp)
    int T = (int)std::min<int64_t>(screened ? std::min(ctx->opt_in_flight, 3) : 2, nsig);
    T = std::max(T, 1);
    if (T > 1) CHECK(twins_ensure(ctx, T - 1));
    csmp_ctx* cc[3] = {ctx, T > 1 ? ctx->twins[0] : nullptr, T > 2 ? ctx->twins[1] : nullptr};
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));  // at most k atoms are ever added (GOMP's own capacity is M: :108)
    for (int q = 0; q < T; ++q) {
        const int rc = solver_ensure(cc[q], kc, (int)(k + l));
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
    HIPCHECK(hipStreamSynchronize(ctx->stream));  // (the caller's buffers and our temporaries are ready before either stream starts)
    for (int q = 0; q + 1 < T; ++q)
        if (!cc[q]->ev_twin) HIPCHECK(hipEventCreateWithFlags(&cc[q]->ev_twin, hipEventDisableTiming));
    const bool block = l <= kPanelMax;
    if (screened) {
        if (T == 1) CHECK(screened_ensure(ctx));
        for (int q = 1; q < T; ++q) CHECK(screened_ensure_pair(ctx, cc[q]));
    }
    for (int64_t sgn = 0; sgn < nsig; ++sgn) {
        const int q = (int)(sgn % T);
        csmp_ctx* c = cc[q];
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        if (sgn > 0 && sgn < T) HIPCHECK(hipStreamWaitEvent(c->stream, cc[q - 1]->ev_twin, 0));  // a twin starts one sweep behind: out of phase
        const int rc = gomp_enqueue(c, col, b_dtype, l, k, eps, block, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, d_flag + sgn,
                                    sgn + 1 < T ? c->ev_twin : nullptr, screened);
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
    // a panel that failed its DGKS test flagged the solve (nothing committed): that signal again, column by column; a solve
    // with an uncertified pick (screened sweep): again with the exact sweep
    int rc = CSMP_OK;
    for (int64_t sgn = 0; sgn < nsig && rc == CSMP_OK; ++sgn) {
        if (screened) {
            ctx->scr_solves += 1;
            ctx->scr_fallbacks += (hf[sgn] & STOP_UNCERTAIN) ? 1 : 0;
        }
        bool blk = block;
        while (rc == CSMP_OK && (hf[sgn] & (STOP_REORTH | STOP_UNCERTAIN))) {
            if (hf[sgn] & STOP_REORTH) {
                if (!blk && !(hf[sgn] & STOP_UNCERTAIN)) break;  // (the column-wise chain flags nothing it cannot handle)
                blk = false;
            }
            const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
            rc = gomp_enqueue(ctx, col, b_dtype, l, k, eps, blk, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, d_flag + sgn, nullptr, false);
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

// state reset + r = b for a fresh factorisation on the same b (SP re-factorises from scratch)
static int solver_restart(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_init<double>, dim3(s.Mpad / 256), dim3(256), 0, ctx->stream, (const double*)s.b, (int)ctx->M, s.Mpad, s.bstage, s.r, s.st);
    HIPCHECK(hipGetLastError());
    s.jh = 0;
    return CSMP_OK;
}

// factorize! + ldiv! (src/matchingpursuit.jl:219-227, src/twostage.jl:104-107) on the columns
// `cols` (host list): QR by successive appends, residual r = b - A_S c as a by-product.
static int ls_on_columns(csmp_ctx* ctx, const std::vector<int>& cols) {
    Solver& s = ctx->s;
    s.fac_valid = false;
    if ((int)cols.size() > s.kcap) return fail(ctx, CSMP_ERANGE, "least squares: more columns than the QR capacity");
    CHECK(solver_restart(ctx));
    const int n = (int)cols.size();
    HIPCHECK(hipMemcpyAsync(s.cands, cols.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipMemcpyAsync(s.ncands, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    if (n > 1) {  // panels of 32 columns; verified through the device flag
        CHECK(launch_block_appends(ctx, n, STOP_REORTH));
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (!(hs.done & STOP_REORTH)) return CSMP_OK;
        CHECK(solver_restart(ctx));  // a panel failed its DGKS test: column-wise chain with re-orthogonalisation
    }
    for (int w = 0; w < n; ++w) CHECK(launch_append(ctx, 2, w, 0));
    return CSMP_OK;
}

// ---- whole-set least squares (csmp_gram.hpp): Gram matrix on the matrix cores + blocked Cholesky, no Q
// atom -> position marks of the host-side set algebra (one generation per question; N entries, allocated on first use)
static void marks_begin(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.hstamp.size() != (size_t)ctx->N) {
        s.hstamp.assign((size_t)ctx->N, 0u);
        s.hpos.assign((size_t)ctx->N, 0);
        s.hgen = 0;
    }
    if (++s.hgen == 0) {
        std::fill(s.hstamp.begin(), s.hstamp.end(), 0u);
        s.hgen = 1;
    }
}
static inline void mark_put(csmp_ctx* ctx, int atom, int pos) {
    ctx->s.hstamp[(size_t)atom] = ctx->s.hgen;
    ctx->s.hpos[(size_t)atom] = pos;
}
static inline int mark_get(const csmp_ctx* ctx, int atom) { return ctx->s.hstamp[(size_t)atom] == ctx->s.hgen ? ctx->s.hpos[(size_t)atom] : -1; }

static int gram_split_for(const csmp_ctx* ctx, int np, int jtile0 = 0, int ncols = -1) {
    // pieces of k_gram on or above the diagonal (from column tile jtile0 on: the bordered extension computes the new columns'
    // tiles only); the rows are split so that ONE round of workgroups (two per CU) covers them: a second, partly filled round
    // would cost as much as a full one
    int TJ = np / kGramWgJ;
    if (ncols >= 0) TJ = std::min(TJ, (ncols + kGramWgJ - 1) / kGramWgJ);  // (tiles past the set's own columns are not computed: k_gram's jtile1)
    int pieces = 0;
    for (int J = jtile0; J < TJ; ++J) pieces += (J * kGramWgJ + kGramWgJ - 1) / kGramWgI + 1;
    const int slots = (ctx->dtype == CSMP_F32 ? 3 : 2) * ctx->prop.multiProcessorCount;  // k_gram's workgroups per CU
    int nsplit = std::max(1, slots / std::max(1, pieces));
    nsplit = std::min<int>(nsplit, std::max<int>(1, (int)(ctx->M / 64)));  // at least four 16-row blocks each
    return std::min(nsplit, 32);
}
static int gram_ensure(csmp_ctx* ctx, int np, int nsplit) {
    Solver& s = ctx->s;
    if (s.gram_np >= np && s.gram_split >= nsplit) return CSMP_OK;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    np = std::max(np, s.gram_np);
    nsplit = std::max(nsplit, s.gram_split);
    dfree(s.Gm); dfree(s.Dfac); dfree(s.Gpart); dfree(s.gdiag); dfree(s.rpart); dfree(s.Acomp); dfree(s.Gkeep); dfree(s.gdkeep); dfree(s.kpos); dfree(s.rhs_part); dfree(s.rn2part);
    dfree(s.Gkeep2); dfree(s.gdkeep2); dfree(s.Wb); dfree(s.Gin); dfree(s.Gm2); dfree(s.ytmp);
    s.gram_np = s.gram_split = 0;
    s.keep_valid = false;
    s.fac_valid = false;
    CHECK(dmalloc(ctx, &s.Gkeep, (size_t)np * np));
    CHECK(dmalloc(ctx, &s.gdkeep, (size_t)np));
    CHECK(dmalloc(ctx, &s.Gkeep2, (size_t)np * np));  // the bordered extension assembles the next kept matrix beside the current one
    CHECK(dmalloc(ctx, &s.gdkeep2, (size_t)np));
    CHECK(dmalloc(ctx, &s.Wb, (size_t)np * np));      // W = R_F^-T G_FN bordered by z_F
    CHECK(dmalloc(ctx, &s.Gin, (size_t)np * np));     // G_FN (k_wgemm's input)
    CHECK(dmalloc(ctx, &s.kpos, (size_t)np));
    CHECK(dmalloc(ctx, &s.rn2part, (size_t)(ctx->M + 255) / 256));
    CHECK(dmalloc(ctx, &s.rhs_part, (size_t)np * (size_t)(((ctx->M + 15) / 16 * 16 + 255) / 256)));
    CHECK(dmalloc(ctx, &s.Gm, (size_t)np * np));
    CHECK(dmalloc(ctx, &s.Gm2, (size_t)np * np));  // the extension's Schur complement is factorised here: (R_F^-1)' stays in Gm for its back substitution
    CHECK(dmalloc(ctx, &s.ytmp, (size_t)np));
    CHECK(dmalloc(ctx, &s.Dfac, (size_t)np * kCholNB));  // the factored diagonal blocks (chol_row_body)
    CHECK(dmalloc(ctx, &s.Gpart, (size_t)nsplit * np * np));
    CHECK(dmalloc(ctx, &s.gdiag, (size_t)np));
    CHECK(dmalloc(ctx, &s.rpart, (size_t)((np + kResChunk - 1) / kResChunk) * s.Mpad));
    HIPCHECK(hipMalloc(&s.Acomp, (size_t)np * (size_t)((ctx->M + 15) / 16 * 16) * (ctx->dtype == CSMP_F32 ? 4 : 8)));
    s.gram_np = np;
    s.gram_split = nsplit;
    return CSMP_OK;
}

// factorize! + ldiv! on the columns `cols` taken together: enqueues the Gram matrix, its Cholesky factorisation, the
// export of (R, z, support), the back substitution + sorted emission into the slot's out arrays and the residual
// r = b - A_S x.  No host synchronisation; a set that fails the DGKS test leaves STOP_REORTH in the control block (and
// nothing exported): the caller checks it with the results and falls back to ls_on_columns.
// Bordered extension (csmp_gram.hpp): `cols` holds the set F whose factor the slot still carries (same b) plus new columns.
// Enqueues everything ls_gram_t does, with the Cholesky chain over the NEW columns only.  order = [F in factor order | new, sorted].
template <typename TA>
static int ls_gram_extend_t(csmp_ctx* ctx, const std::vector<int>& order, int nF, const std::vector<int>& posF) {
    Solver& s = ctx->s;
    const int n = (int)order.size(), nN = n - nF, M = (int)ctx->M;
    const int np = ((n + 1 + kGramTile - 1) / kGramTile) * kGramTile;
    const int np2 = ((nN + 1 + kGramTile - 1) / kGramTile) * kGramTile;
    const int ldw = ((nF + 15) / 16) * 16;
    const int nsplit = std::min(gram_split_for(ctx, np, nF / kGramWgJ, n), s.gram_split);  // (the new columns' tiles fill the round)
    // (gram_ensure has run: the caller checked the buffers' sizes before it decided for this path)
    CHECK(solver_restart(ctx));  // r = b, control block reset; R, z, sel of F stay where they are
    void* pcv = nullptr;
    CHECK(pin_get(ctx, 2, (size_t)(2 * n + 2) * 4, &pcv));
    int* pcols = (int*)pcv;
    int* ppos = pcols + n + 1;
    for (int t = 0; t < n; ++t) pcols[t] = order[t];
    pcols[n] = n;
    for (int t = 0; t < nF; ++t) ppos[t] = posF[t];
    hipLaunchKernelGGL(k_put_lists, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const int*)pcols, n, nF, s.cands, s.ncands, s.kpos);
    const int blk = 16;
    const int rps = (((M + nsplit - 1) / nsplit + blk - 1) / blk) * blk;
    const int64_t ldo = ((int64_t)M + 15) / 16 * 16;
    const int nchunk = (int)((ldo + 255) / 256);
    hipLaunchKernelGGL(k_gather_cols<TA>, dim3((unsigned)nchunk, np), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M,
                       (const int*)s.cands, n, (TA*)s.Acomp, ldo, (const double*)s.b, np, s.rhs_part);
    hipLaunchKernelGGL(k_gram<TA>, dim3(np / kGramWgJ, (np + kGramWgI - 1) / kGramWgI, nsplit), dim3(256), 0, ctx->stream, (const TA*)s.Acomp, ldo, np,
                       rps, s.Gpart, nF / kGramWgJ, (n + kGramWgJ - 1) / kGramWgJ);
    HIPCHECK(hipGetLastError());
    const int64_t nel = std::max<int64_t>((int64_t)np * np, (int64_t)ldw * np2);
    hipLaunchKernelGGL(k_ext_reduce, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gpart, nsplit, nF, n, np,
                       (const double*)s.rhs_part, nchunk, (const double*)s.Gkeep, s.keep_np, (const int*)s.kpos, (const double*)s.z, s.Gkeep2, s.gdkeep2,
                       s.Wb, ldw, np2, s.Gin);
    HIPCHECK(hipGetLastError());
    // W = R_F^-T G_FN = Tt G_FN: a tiled product, no dependent chain; Tt = (R_F^-1)' sits in the augmented columns of F's own
    // factorisation (s.Gm: k_schur_reduce below overwrites it AFTER this product, in stream order)
    hipLaunchKernelGGL(k_wgemm, dim3((nF + 31) / 32, (nN + 31) / 32), dim3(256), 0, ctx->stream, (const double*)(s.Gm + (size_t)s.tt_col0 * s.tt_ld), s.tt_ld,
                       (const double*)s.Gin, ldw, nF, nN, s.Wb, ldw);
    HIPCHECK(hipGetLastError());
    // [W z_F]'[W z_F] on the Float64 matrix cores: the columns of Wb are the "dictionary" (ldw rows), a few row slices
    const int nsplit2 = std::max(1, std::min(std::min(nsplit, 8), ldw / 64));
    const int rps2 = (((ldw + nsplit2 - 1) / nsplit2 + blk - 1) / blk) * blk;
    hipLaunchKernelGGL(k_gram<double>, dim3(np2 / kGramWgJ, (np2 + kGramWgI - 1) / kGramWgI, nsplit2), dim3(256), 0, ctx->stream, (const double*)s.Wb,
                       (int64_t)ldw, np2, rps2, s.Gpart, 0);
    // The Schur complement is factorised AUGMENTED by the unit vectors too (room permitting), in a buffer of its own: with
    // (R_C^-1)' beside R_C and (R_F^-1)' still in s.Gm the solution is three products -- x_N = R_C^-1 z_N, y = z_F - W x_N,
    // x_F = R_F^-1 y -- instead of the back substitution's chain over all n columns (four 256-column super-blocks at n = 1024).
    const int npa2c = ((np2 + nN + kGramTile - 1) / kGramTile) * kGramTile;
    const bool aug2 = npa2c <= s.gram_np;
    const int npa2 = aug2 ? npa2c : np2;
    double* G2 = aug2 ? s.Gm2 : s.Gm;
    const int64_t nel2 = (int64_t)np2 * np2 + (int64_t)(npa2 - np2) * npa2;
    hipLaunchKernelGGL(k_schur_reduce, dim3((unsigned)((nel2 + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gkeep2, np, nF, nN, np2,
                       (const double*)s.Gpart, nsplit2, G2, s.gdiag, npa2);
    HIPCHECK(hipGetLastError());
    const int nsteps = (nN + kCholNB - 1) / kCholNB;
    {
        const int left0 = npa2 - kCholNB;
        hipLaunchKernelGGL(k_chol_row, dim3(std::max(1, (left0 + kCholRowCols - 1) / kCholRowCols)), dim3(kCholThreads), 0, ctx->stream, G2, npa2, nN,
                           0, (const double*)s.gdiag, s.st, s.Dfac);
    }
    for (int kb = 0; kb + 1 < nsteps; ++kb) {
        const int left = npa2 - (kb + 1) * kCholNB;
        const int left2 = left - kCholNB;
        const int Tt = (left + kGramTile - 1) / kGramTile;
        const int ntrail = left > kCholNB ? Tt * (Tt + 1) / 2 : 0;
        const int nrow = std::max(1, (left2 + kCholRowCols - 1) / kCholRowCols);
        hipLaunchKernelGGL(k_chol_step, dim3(nrow + ntrail), dim3(kCholThreads), 0, ctx->stream, G2, npa2, nN, kb, (const double*)s.gdiag, s.st,
                           nrow, s.Dfac, np2);
    }
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_gram_export_b, dim3((unsigned)(((int64_t)n * nN + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)G2, npa2, nF, nN,
                       (const int*)s.cands, (const double*)s.Wb, ldw, s.R, s.kcap, s.z, s.sel, s.st, (const double*)s.Dfac);
    HIPCHECK(hipGetLastError());
    s.tt_pending = false;
    std::swap(s.Gkeep, s.Gkeep2);
    std::swap(s.gdkeep, s.gdkeep2);
    s.keep_cols = order;
    s.keep_n = n;
    s.keep_np = np;
    s.keep_valid = true;
    s.jh = std::min(s.kcap, n);
    if (aug2) {
        hipLaunchKernelGGL(k_tt_gemv, dim3((nN + 3) / 4), dim3(256), 0, ctx->stream, (const double*)(G2 + (size_t)np2 * npa2), npa2, (const double*)(s.z + nF),
                           (const DevState*)s.st, nN, s.coef + nF, n);
        hipLaunchKernelGGL(k_wx, dim3((nF + 15) / 16), dim3(256), 0, ctx->stream, (const double*)s.Wb, ldw, nF, nN, (const double*)s.z,
                           (const double*)(s.coef + nF), (const DevState*)s.st, n, s.ytmp);
        hipLaunchKernelGGL(k_tt_gemv, dim3((nF + 3) / 4), dim3(256), 0, ctx->stream, (const double*)(s.Gm + (size_t)s.tt_col0 * s.tt_ld), s.tt_ld,
                           (const double*)s.ytmp, (const DevState*)s.st, nF, s.coef, n);
        const int ne = std::max(n, s.outcap);
        hipLaunchKernelGGL(k_trsv_emit, dim3((ne + 255) / 256), dim3(256), (size_t)(s.kcap + 4) * sizeof(int), ctx->stream,
                           (const double*)s.coef, (const int*)s.sel, (const DevState*)s.st, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap,
                           (int*)nullptr);
        HIPCHECK(hipGetLastError());
    } else {
        CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));
    }
    const int nch = (n + kResChunk - 1) / kResChunk;
    hipLaunchKernelGGL(k_residual_part<TA>, dim3((M + 255) / 256, nch), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M,
                       (const int*)s.cands, (const double*)s.coef, n, s.rpart);
    hipLaunchKernelGGL(k_residual_sum, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.rpart, nch, M, (const double*)s.b,
                       s.r, (const DevState*)s.st, s.rn2part);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

template <typename TA>
static int ls_gram_t(csmp_ctx* ctx, const std::vector<int>& cols) {
    Solver& s = ctx->s;
    const int n = (int)cols.size(), M = (int)ctx->M;
    const int np = ((n + 1 + kGramTile - 1) / kGramTile) * kGramTile;
    const int nsplit = gram_split_for(ctx, np, 0, n);
    // A set that k further columns may extend (Subspace Pursuit: the k atoms of the support before an acquisition) is factorised
    // AUGMENTED by the unit vectors (csmp_gram.hpp, k_gram_reduce): (R^-1)' comes out beside R, for free on this chain.
    const bool aug = n >= 64 && 2 * n <= s.kcap;
    const int npa = aug ? ((np + n + kGramTile - 1) / kGramTile) * kGramTile : np;
    // (a set that may be an extension of the slot's factor splits its rows finer: fewer tiles to spread over the same round)
    const int nsplit_ext = (s.fac_valid && s.fac_cols.size() >= 64 && (int)s.fac_cols.size() < n)
                               ? gram_split_for(ctx, np, (int)s.fac_cols.size() / kGramWgJ, n) : nsplit;
    CHECK(gram_ensure(ctx, npa, std::max(nsplit, nsplit_ext)));  // (a reallocation drops fac_valid and keep_valid)
    const bool can_extend = s.fac_valid;
    s.fac_valid = false;  // whatever happens below rewrites the slot; the caller confirms the new factor once it has seen it succeed
    s.fac_pending.assign(cols.begin(), cols.end());  // the factor order of this solve
    if (can_extend) {
        // The slot still holds the factor of a set F on this very b.  If F lies inside `cols` (and its Gram matrix inside the kept
        // one), only the new columns are factorised (ls_gram_extend_t).
        const int nF = (int)s.fac_cols.size();
        if (nF >= 64 && nF < n && s.keep_valid && n <= s.kcap && s.tt_gen == s.fac_gen &&  // (tt_gen: (R_F^-1)' sits in s.Gm)
            std::is_sorted(cols.begin(), cols.end())) {
            // (set algebra on atom marks: linear in the sets -- these lines sit on the solve's latency chain)
            std::vector<int> posF((size_t)nF);
            bool ok = true;
            marks_begin(ctx);
            for (int t = 0; t < s.keep_n; ++t) mark_put(ctx, s.keep_cols[t], t);
            for (int t = 0; t < nF && ok; ++t) ok = (posF[t] = mark_get(ctx, s.fac_cols[t])) >= 0;  // F inside the kept set
            if (ok) {
                marks_begin(ctx);
                for (int t = 0; t < n; ++t) mark_put(ctx, cols[t], t);
                for (int t = 0; t < nF && ok; ++t) ok = mark_get(ctx, s.fac_cols[t]) >= 0;  // F inside cols
            }
            if (ok) {
                marks_begin(ctx);
                for (int t = 0; t < nF; ++t) mark_put(ctx, s.fac_cols[t], t);
                std::vector<int> order(s.fac_cols);
                order.reserve((size_t)n);
                for (int c : cols)
                    if (mark_get(ctx, c) < 0) order.push_back(c);
                if ((int)order.size() == n) {
                    s.fac_pending = order;
                    return ls_gram_extend_t<TA>(ctx, order, nF, posF);
                }
            }
        }
    }
    CHECK(solver_restart(ctx));
    // the column list (and, for a subset, its positions in the kept set) go up from a page-locked buffer that lives until the
    // next call -- every caller drains the stream before it comes back here
    void* pcv = nullptr;
    CHECK(pin_get(ctx, 2, (size_t)(2 * n + 2) * 4, &pcv));
    int* pcols = (int*)pcv;
    int* ppos = pcols + n + 1;
    for (int t = 0; t < n; ++t) pcols[t] = cols[t];
    pcols[n] = n;
    // A set inside the last computed one: its bordered Gram matrix is a principal submatrix of the kept one -- gathered, not recomputed
    bool subset = s.keep_valid && n <= s.keep_n;
    if (subset) {
        marks_begin(ctx);
        for (int t = 0; t < s.keep_n; ++t) mark_put(ctx, s.keep_cols[t], t);
        for (int t = 0; t < n && subset; ++t) subset = (ppos[t] = mark_get(ctx, cols[t])) >= 0;
    }
    hipLaunchKernelGGL(k_put_lists, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const int*)pcols, n, subset ? n : 0, s.cands, s.ncands, s.kpos);
    if (subset) {
        const int64_t nel = (int64_t)np * np + (int64_t)(npa - np) * npa;
        hipLaunchKernelGGL(k_gram_subset, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gkeep, s.keep_np, s.keep_n,
                           (const double*)s.gdkeep, (const int*)s.kpos, n, np, s.Gm, s.gdiag, npa);
        HIPCHECK(hipGetLastError());
    } else {
        const int blk = 16;
        const int rps = (((M + nsplit - 1) / nsplit + blk - 1) / blk) * blk;
        const int64_t ldo = ((int64_t)M + 15) / 16 * 16;
        const int nchunk = (int)((ldo + 255) / 256);
        hipLaunchKernelGGL(k_gather_cols<TA>, dim3((unsigned)nchunk, np), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M,
                           (const int*)s.cands, n, (TA*)s.Acomp, ldo, (const double*)s.b, np, s.rhs_part);
        hipLaunchKernelGGL(k_gram<TA>, dim3(np / kGramWgJ, (np + kGramWgI - 1) / kGramWgI, nsplit), dim3(256), 0, ctx->stream, (const TA*)s.Acomp, ldo, np,
                           rps, s.Gpart, 0, (n + kGramWgJ - 1) / kGramWgJ);
        HIPCHECK(hipGetLastError());
        const int64_t nel = (int64_t)np * np + (int64_t)(npa - np) * npa;
        hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gpart, nsplit, n, np,
                           s.Gm, s.gdiag, (const double*)s.rhs_part, nchunk, s.Gkeep, s.gdkeep, npa);
        HIPCHECK(hipGetLastError());
        s.keep_cols.assign(cols.begin(), cols.end());
        s.keep_n = n;
        s.keep_np = np;
        s.keep_valid = true;
    }
    // Block rows 0 .. ceil(n / 32) - 1 are all that is needed: the bordered column n is a column of their row panels (or of
    // the last diagonal block when n is not a multiple of 32); the corner b'b - z'z and the identity padding are never read.
    // (augmented: the row panels run on to column npa; trailing tiles whose ROWS lie in the augmented range are skipped)
    const int nsteps = (n + kCholNB - 1) / kCholNB;
    {
        const int left0 = npa - kCholNB;
        hipLaunchKernelGGL(k_chol_row, dim3(std::max(1, (left0 + kCholRowCols - 1) / kCholRowCols)), dim3(kCholThreads), 0, ctx->stream, s.Gm, npa, n,
                           0, (const double*)s.gdiag, s.st, s.Dfac);
    }
    for (int kb = 0; kb + 1 < nsteps; ++kb) {  // one launch per step: trailing update of panel kb + block row kb + 1
        const int left = npa - (kb + 1) * kCholNB;  // columns from the next block row on
        const int left2 = left - kCholNB;           // columns to the right of the next diagonal block
        const int Tt = (left + kGramTile - 1) / kGramTile;
        const int ntrail = left > kCholNB ? Tt * (Tt + 1) / 2 : 0;
        const int nrow = std::max(1, (left2 + kCholRowCols - 1) / kCholRowCols);
        hipLaunchKernelGGL(k_chol_step, dim3(nrow + ntrail), dim3(kCholThreads), 0, ctx->stream, s.Gm, npa, n, kb, (const double*)s.gdiag, s.st,
                           nrow, s.Dfac, np);
    }
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_gram_export, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gm, npa, n,
                       (const int*)s.cands, s.R, s.kcap, s.z, s.sel, s.st, (const double*)s.Dfac);
    HIPCHECK(hipGetLastError());
    s.tt_pending = aug;  // (confirmed with the factor: ls_fetch)
    s.tt_col0 = np;
    s.tt_ld = npa;
    s.jh = std::min(s.kcap, n);
    if (aug) {  // x = R^-1 z through the explicit inverse beside R: one product instead of the back substitution's chain
        hipLaunchKernelGGL(k_tt_gemv, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, (const double*)(s.Gm + (size_t)np * npa), npa, (const double*)s.z,
                           (const DevState*)s.st, n, s.coef, n);
        const int ne = std::max(n, s.outcap);
        hipLaunchKernelGGL(k_trsv_emit, dim3((ne + 255) / 256), dim3(256), (size_t)(s.kcap + 4) * sizeof(int), ctx->stream,
                           (const double*)s.coef, (const int*)s.sel, (const DevState*)s.st, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap,
                           (int*)nullptr);
        HIPCHECK(hipGetLastError());
    } else {
        CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));  // x = R^-1 z (s.coef: the order of cols) + sorted emission
    }
    const int nch = (n + kResChunk - 1) / kResChunk;
    hipLaunchKernelGGL(k_residual_part<TA>, dim3((M + 255) / 256, nch), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M,
                       (const int*)s.cands, (const double*)s.coef, n, s.rpart);
    hipLaunchKernelGGL(k_residual_sum, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.rpart, nch, M, (const double*)s.b,
                       s.r, (const DevState*)s.st, s.rn2part);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}
static bool gram_applicable(const csmp_ctx* ctx, size_t n) {
    // worth it from a few panels on; needs QR capacity for R and distinct columns (the callers guarantee those)
    return n >= 64;
}
static int ls_gram(csmp_ctx* ctx, const std::vector<int>& cols) {
    return ctx->dtype == CSMP_F32 ? ls_gram_t<float>(ctx, cols) : ls_gram_t<double>(ctx, cols);
}

extern "C" int csmp_lstsq(csmp_ctx* ctx, const int64_t* cols, int64_t ncols, const void* b, int b_dtype, double* coef) {
    if (!ctx) return CSMP_EINVAL;
    if (!cols || !b || !coef || ncols < 1) return fail(ctx, CSMP_EINVAL, "lstsq: bad arguments");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (ncols > ctx->M) return fail(ctx, CSMP_ERANGE, "lstsq: more columns than rows");
    std::vector<int> c((size_t)ncols);
    for (int64_t t = 0; t < ncols; ++t) {
        if (cols[t] < 0 || cols[t] >= ctx->N) return fail(ctx, CSMP_ERANGE, "lstsq: column index out of range");
        c[t] = (int)cols[t];
    }
    std::vector<int> srt = c;
    std::sort(srt.begin(), srt.end());
    if (std::adjacent_find(srt.begin(), srt.end()) != srt.end()) return fail(ctx, CSMP_EINVAL, "lstsq: duplicate column");
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(solver_ensure(ctx, (int)ncols, (int)ncols));
    ctx->s.begun = false;
    CHECK(upload_b(ctx, b, b_dtype));
    Solver& s = ctx->s;
    if (gram_applicable(ctx, c.size())) {
        CHECK(ls_gram(ctx, c));
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipMemcpyAsync(coef, s.coef, (size_t)ncols * 8, hipMemcpyDeviceToHost, ctx->stream));  // the order of cols
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (!(hs.done & STOP_REORTH) && hs.nsel == (int)ncols) return CSMP_OK;
    }
    CHECK(ls_on_columns(ctx, c));
    CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));
    HIPCHECK(hipMemcpyAsync(coef, s.coef, (size_t)ncols * 8, hipMemcpyDeviceToHost, ctx->stream));  // insertion order = cols order
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return CSMP_OK;
}

static int residual_norm(csmp_ctx* ctx, double* out) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_norm2, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.r, (int)ctx->M, s.scal);
    HIPCHECK(hipGetLastError());
    double n2 = 0.0;
    PinFetch f(ctx);
    CHECK(f.begin(8));
    CHECK(f.add(&n2, s.scal, 8));
    CHECK(f.wait());
    *out = std::sqrt(n2);
    return CSMP_OK;
}

// current support + coefficients (sorted by index) to the host
static int fetch_sorted(csmp_ctx* ctx, std::vector<int64_t>& idx, std::vector<double>& val) {
    Solver& s = ctx->s;
    CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));
    idx.assign((size_t)s.outcap, 0);
    val.assign((size_t)s.outcap, 0.0);
    int64_t n = 0;
    CHECK(download_result(ctx, s.outcap, idx.data(), val.data(), &n, nullptr));
    idx.resize((size_t)n);
    val.resize((size_t)n);
    return CSMP_OK;
}

// ---- Subspace Pursuit as a RESUMABLE job (src/twostage.jl:54-101).
// One sp is a chain of device phases separated by host decisions on a few hundred numbers: which atoms join (the union with the
// support), which leave (the prune by |coefficient|), whether to go on (the residual norms).  A job ENQUEUES a phase -- kernels,
// the copies of what the host has to look at into the context's page-locked landing area, an event -- and RESUMES once the event
// has fired.  csmp_sp drives one job with blocking waits; csmp_sp_batch drives several, each on a context (stream) of its own,
// from the CALLING thread: a job whose event has not fired yet is skipped and another one advanced -- no threads inside the
// library, nothing of one signal waits on the host work of another.
// sp_acquisition!(P, x, k), first half (src/twostage.jl:67-69): sweep on the current residual + the k best atoms, on their way to
// the host.  CSMP_OPT_SCREENED_SWEEP: the sweep reads the image and the top-k SET is certified (host/screened.hpp).
static int sp_job_select(SpJob& j, bool scr) {
    csmp_ctx* ctx = j.c;
    j.sel_screened = scr;
    Solver& s = ctx->s;
    const int k = (int)j.k;
    void* pv = nullptr;
    CHECK(pin_get(ctx, 1, (size_t)k * 4 + 16, &pv));  // (page-locked: the small copies do not block the host one by one)
    int* top = (int*)pv;
    int* pnt = top + k;
    pnt[1] = 0;
    // Several solves in flight (csmp_sp_batch): the sweeps of different solves run ONE AFTER THE OTHER.  Two HBM-bound sweeps side
    // by side finish no sooner than back to back, and while they share the memory system nothing is left for the others to hide
    // their latency chains under; queued behind one another, every sweep has the chains of the other solves beside it.
    if (ctx->gate) HIPCHECK(hipStreamWaitEvent(ctx->stream, *ctx->gate, 0));
    if (scr) {
        CHECK(sp_select_screened(ctx, k));
        if (ctx->gate) HIPCHECK(hipEventRecord(*ctx->gate, ctx->stream));
    } else {
        CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
        if (ctx->gate) HIPCHECK(hipEventRecord(*ctx->gate, ctx->stream));
        CHECK(launch_topS(ctx, k));
    }
    hipLaunchKernelGGL(k_land_sel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, (const int*)s.cands, (const int*)s.ncands, k,
                       scr ? (const int*)s.scr_flag : (const int*)nullptr, top);  // [atoms | count | flag], written over the host link
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(j.ev, ctx->stream));
    return CSMP_OK;
}

// Least squares on j.cols, first half: the whole-set path enqueued with everything the host needs afterwards ([idx | val | control
// block | shares of ||r||^2] into ONE landing area); small or coherent sets take the append chain, synchronously (rare here).
static int sp_job_ls(SpJob& j, bool want_norm) {
    csmp_ctx* ctx = j.c;
    Solver& s = ctx->s;
    j.want_norm = want_norm;
    j.gram_inflight = gram_applicable(ctx, j.cols.size());
    if (j.gram_inflight) {
        CHECK(ls_gram(ctx, j.cols));
        const size_t n = j.cols.size();
        const size_t nshare = (size_t)(ctx->M + 255) / 256;
        const size_t need = n * 16 + sizeof(DevState) + 16 + nshare * 8;
        void* pv = nullptr;
        CHECK(pin_get(ctx, 1, need, &pv));
        int64_t* pi = (int64_t*)pv;
        double* pvv = (double*)(pi + n);
        DevState* phs = (DevState*)(pvv + n);
        double* pn2 = (double*)((char*)phs + ((sizeof(DevState) + 7) / 8) * 8);
        const int nthr = (int)std::max<size_t>(std::max(n, nshare), sizeof(DevState) / sizeof(int));
        hipLaunchKernelGGL(k_land_ls, dim3((nthr + 255) / 256), dim3(256), 0, ctx->stream, (const int64_t*)s.out_idx, (const double*)s.out_val, (int)n,
                           (const DevState*)s.st, (const double*)s.rn2part, want_norm ? (int)nshare : 0, pi, pvv, (int*)phs, pn2);
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipEventRecord(j.ev, ctx->stream));
    return CSMP_OK;
}
// ... second half: the solution (sorted) into the job, ||r|| when asked; a set the whole-set path refused goes through the chain
static int sp_job_ls_done(SpJob& j) {
    csmp_ctx* ctx = j.c;
    Solver& s = ctx->s;
    if (j.gram_inflight) {
        const size_t n = j.cols.size();
        const size_t nshare = (size_t)(ctx->M + 255) / 256;
        int64_t* pi = (int64_t*)ctx->pin[1];
        double* pvv = (double*)(pi + n);
        DevState* phs = (DevState*)(pvv + n);
        double* pn2 = (double*)((char*)phs + ((sizeof(DevState) + 7) / 8) * 8);
        const DevState hs = *phs;
        if (!(hs.done & STOP_REORTH) && hs.nsel == (int)n) {
            s.fac_cols.swap(s.fac_pending);  // the slot now holds this set's factor (R, z, sel) on the current b: a later superset extends it
            s.fac_valid = true;
            s.fac_gen += 1;
            if (s.tt_pending) s.tt_gen = s.fac_gen;
            j.xi.assign(pi, pi + n);
            j.xv.assign(pvv, pvv + n);
            if (j.want_norm) {
                double n2 = 0.0;
                for (size_t q = 0; q < nshare; ++q) n2 += pn2[q];
                j.resnorm = std::sqrt(n2);
            }
            return CSMP_OK;
        }
    }
    CHECK(ls_on_columns(ctx, j.cols));
    CHECK(fetch_sorted(ctx, j.xi, j.xv));
    if (j.want_norm) CHECK(residual_norm(ctx, &j.resnorm));
    return CSMP_OK;
}

// the selection has arrived: an uncertified screened acquisition is repeated with the exact sweep; else @. x[i] = NaN (:70) --
// the union with the support -- and solve! (:71) goes out
static int sp_job_selected(SpJob& j, SpJob::Phase next) {
    csmp_ctx* ctx = j.c;
    const int k = (int)j.k;
    const int* top = (const int*)ctx->pin[1];
    const int* pnt = top + k;
    if (j.sel_screened) {
        ctx->scr_solves += 1;
        if (pnt[1] != 0) {
            ctx->scr_fallbacks += 1;
            return sp_job_select(j, false);  // (the same phase again, with the exact sweep -- this acquisition only)
        }
    }
    const int nt = *pnt;
    j.prev_xi = j.xi;  // the support this acquisition starts from and its solution: if the prune hands the same set back, that IS the result
    j.prev_xv = j.xv;
    std::vector<int> mine(j.xi.begin(), j.xi.end()), fresh(top, top + nt);  // (the support comes sorted; the k best atoms by value)
    std::sort(fresh.begin(), fresh.end());
    fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
    j.cols.clear();
    std::set_union(mine.begin(), mine.end(), fresh.begin(), fresh.end(), std::back_inserter(j.cols));
    j.phase = next;
    return sp_job_ls(j, next == SpJob::LS_FIRST);
}

static int sp_job_begin(SpJob& j, csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter) {
    j.c = ctx;
    j.k = k;
    j.delta = delta;
    j.maxiter = maxiter < 0 ? 16 * k : maxiter;  // :87
    j.it = 0;
    j.xi.clear();
    j.xv.clear();
    j.rc = CSMP_OK;
    if (!j.ev) HIPCHECK(hipEventCreateWithFlags(&j.ev, hipEventDisableTiming));
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(solver_ensure(ctx, (int)(2 * k), (int)(2 * k)));
    ctx->s.begun = false;
    j.screened = screened_on(ctx) && k <= 4096;
    if (j.screened) CHECK(screened_ensure(ctx));
    CHECK(upload_b(ctx, b, b_dtype));
    j.phase = SpJob::SELECT;
    j.oldnorm = -1.0;  // (marks the first acquisition: :90-91)
    return sp_job_select(j, j.screened);
}

// the pending phase's event has fired: take its results, enqueue the next phase (or finish)
static int sp_job_advance(SpJob& j) {
    switch (j.phase) {
        case SpJob::SELECT:
            return sp_job_selected(j, j.oldnorm < 0.0 ? SpJob::LS_FIRST : SpJob::LS_UNION);
        case SpJob::LS_FIRST:  // :90-91 done; the loop starts (:92)
            CHECK(sp_job_ls_done(j));
            if (j.it >= j.maxiter) {
                j.phase = SpJob::DONE;
                return CSMP_OK;
            }
            j.oldnorm = j.resnorm;
            j.phase = SpJob::SELECT;
            return sp_job_select(j, j.screened);  // update!(P::SP, x): :75-83, acquisition :77
        case SpJob::LS_UNION: {
            CHECK(sp_job_ls_done(j));
            const int64_t drop = (int64_t)j.xi.size() - j.k;
            if (drop > 0) {  // :78-81: delete the (nnz-k) smallest |coef|, ties by position
                std::vector<int> pos(j.xi.size());
                for (size_t t = 0; t < pos.size(); ++t) pos[t] = (int)t;
                std::nth_element(pos.begin(), pos.begin() + drop, pos.end(), [&](int a, int c) {  // (|coef|, position): a strict order
                    const double fa = std::fabs(j.xv[a]), fc = std::fabs(j.xv[c]);
                    return fa < fc || (fa == fc && a < c);
                });
                std::vector<char> kill(j.xi.size(), 0);
                for (int64_t t = 0; t < drop; ++t) kill[pos[t]] = 1;
                std::vector<int64_t> keep;
                for (size_t t = 0; t < j.xi.size(); ++t)
                    if (!kill[t]) keep.push_back(j.xi[t]);
                j.xi.swap(keep);
            }
            if (j.oldnorm >= 0.0 && j.xi == j.prev_xi) {
                // The prune returned the support the iteration started from.  solve! on it (:82) is the very computation that
                // produced x and r before the acquisition -- the reference gets the same numbers again, bit for bit, and stops by
                // oldnorm <= resnorm (:96).  So do we, without running it a second time.
                j.xv = j.prev_xv;
                j.resnorm = j.oldnorm;
                ++j.it;
                j.phase = SpJob::DONE;
                return CSMP_OK;
            }
            j.cols.clear();
            for (auto i : j.xi) j.cols.push_back((int)i);
            j.phase = SpJob::LS_PRUNED;
            return sp_job_ls(j, true);  // :82, :95
        }
        case SpJob::LS_PRUNED:
            CHECK(sp_job_ls_done(j));
            ++j.it;
            if (j.resnorm <= j.delta || j.oldnorm <= j.resnorm || j.it >= j.maxiter) {  // :96, :92
                j.phase = SpJob::DONE;
                return CSMP_OK;
            }
            j.oldnorm = j.resnorm;
            j.phase = SpJob::SELECT;
            return sp_job_select(j, j.screened);
        default:
            return CSMP_OK;
    }
}

// the pending phase of a job, waited for on the calling thread: a phase lasts 0.1-1 ms and the solve resumes on this thread, so poll
// (the wake-up of a blocking wait would sit on the chain five times per solve); something that takes far longer than a phase is
// waited for the ordinary way
static int sp_job_wait(SpJob& j) {
    csmp_ctx* ctx = j.c;
    hipError_t q = hipErrorNotReady;
    for (int spin = 0; spin < 50000 && q == hipErrorNotReady; ++spin) q = hipEventQuery(j.ev);
    if (q == hipErrorNotReady) q = hipEventSynchronize(j.ev);
    HIPCHECK(q);
    return CSMP_OK;
}

extern "C" int csmp_sp(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int64_t* idx,
                       double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || k < 1) return fail(ctx, CSMP_EINVAL, "sp: b == NULL or k < 1");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (2 * k > ctx->M) return fail(ctx, CSMP_ERANGE, "2k > length(b) is invalid for Subspace Pursuit");  // src/twostage.jl:55
    if (k > ctx->N) return fail(ctx, CSMP_ERANGE, "sp: k > number of atoms");
    SpJob& j = ctx->spjob;
    ctx->gate = nullptr;  // (one solve: nothing to queue behind)
    int rc = sp_job_begin(j, ctx, b, b_dtype, k, delta, maxiter);
    while (rc == CSMP_OK && j.phase != SpJob::DONE) {
        CHECK(sp_job_wait(j));
        rc = sp_job_advance(j);
    }
    if (rc != CSMP_OK) return rc;
    for (size_t t = 0; t < j.xi.size(); ++t) {
        if (idx) idx[t] = j.xi[t];
        if (val) val[t] = j.xv[t];
    }
    if (nnz) *nnz = (int64_t)j.xi.size();
    if (iters) *iters = j.it;
    return CSMP_OK;
}

// sp for many signals: up to four solves in flight, each on a context of its own (this one + clones on their own streams), all
// driven by the CALLING thread: it advances whichever job's pending phase has finished (hipEventQuery) and never blocks on one
// while another could move.  A Subspace Pursuit solve is two HBM-bound sweeps and a chain of short kernels with five host decisions:
// one solve leaves most of the chip idle most of the time, and another signal's solve fills it.  Signal s is solved by the very
// job csmp_sp runs: results are csmp_sp's.
extern "C" int csmp_sp_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int64_t k, double delta, int64_t maxiter,
                             int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!B || nsig < 0 || k < 1 || ldB < ctx->M || !idx || !val || !nnz) return fail(ctx, CSMP_EINVAL, "sp_batch: bad arguments");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (2 * k > ctx->M) return fail(ctx, CSMP_ERANGE, "2k > length(b) is invalid for Subspace Pursuit");  // src/twostage.jl:55
    if (k > ctx->N) return fail(ctx, CSMP_ERANGE, "sp: k > number of atoms");
    if (nsig == 0) return CSMP_OK;
    HIPCHECK(hipSetDevice(ctx->dev));
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ctx->opt_in_flight, 4), nsig));
    CHECK(twins_ensure(ctx, T - 1));
    csmp_ctx* cc[4] = {ctx, ctx->twins[0], ctx->twins[1], ctx->twins[2]};
    for (int t = 1; t < T; ++t) cc[t]->opt_screened = ctx->opt_screened;
    if (screened_on(ctx)) {  // (the twins sweep this context's image)
        if (T == 1) CHECK(screened_ensure(ctx));
        for (int t = 1; t < T; ++t) CHECK(screened_ensure_pair(ctx, cc[t]));
    }
    if (!ctx->ev_gate) HIPCHECK(hipEventCreateWithFlags(&ctx->ev_gate, hipEventDisableTiming));
    for (int t = 0; t < T; ++t) cc[t]->gate = T > 1 ? &ctx->ev_gate : nullptr;
    const size_t es = b_dtype == CSMP_F32 ? 4 : 8;
    int64_t sig_of[4] = {-1, -1, -1, -1}, next = 0, finished = 0;
    int rc = CSMP_OK;
    auto start = [&](int t) -> int {
        sig_of[t] = next++;
        return sp_job_begin(cc[t]->spjob, cc[t], (const char*)B + (size_t)sig_of[t] * (size_t)ldB * es, b_dtype, k, delta, maxiter);
    };
    for (int t = 0; t < T && rc == CSMP_OK; ++t) rc = start(t);
    int spins = 0;
    while (rc == CSMP_OK && finished < nsig) {
        bool moved = false;
        for (int t = 0; t < T && rc == CSMP_OK; ++t) {
            if (sig_of[t] < 0) continue;
            SpJob& j = cc[t]->spjob;
            if (j.phase != SpJob::DONE) {
                const hipError_t q = hipEventQuery(j.ev);
                if (q == hipErrorNotReady) continue;
                if (q != hipSuccess) { rc = fail(ctx, CSMP_EHIP, hipGetErrorString(q)); break; }
                HIPCHECK(hipSetDevice(ctx->dev));
                rc = sp_job_advance(j);
                moved = true;
                if (rc != CSMP_OK || j.phase != SpJob::DONE) continue;
            }
            const int64_t sgn = sig_of[t], n = (int64_t)j.xi.size();
            for (int64_t q = 0; q < k; ++q) {
                idx[sgn * k + q] = q < n ? j.xi[q] : -1;
                val[sgn * k + q] = q < n ? j.xv[q] : 0.0;
            }
            nnz[sgn] = std::min<int64_t>(n, k);
            if (iters) iters[sgn] = j.it;
            ++finished;
            moved = true;
            sig_of[t] = -1;
            if (next < nsig) rc = start(t);
        }
        if (!moved && ++spins > 64) {  // nothing to do right now: let the GPU work (the wait is microseconds)
            std::this_thread::yield();
            spins = 0;
        }
    }
    for (int t = 0; t < T; ++t) (void)hipStreamSynchronize(cc[t]->stream);
    for (int t = 0; t < T; ++t) cc[t]->gate = nullptr;
    for (int t = 1; t < T; ++t) {  // (the twins' screened-selection counters belong to this context's statistics)
        ctx->scr_solves += cc[t]->scr_solves;
        ctx->scr_fallbacks += cc[t]->scr_fallbacks;
        cc[t]->scr_solves = cc[t]->scr_fallbacks = 0;
        if (rc != CSMP_OK && cc[t]->err.size() && ctx->err.empty()) ctx->err = cc[t]->err;
    }
    return rc;
}

// ------------------------------------------------------------------------------------------ primitives
extern "C" int csmp_sweep(csmp_ctx* ctx, const double* r, double* abs_corr, int64_t topk, int64_t* top_idx, double* top_val) {
    if (!ctx) return CSMP_EINVAL;
    if (!r || topk < 0) return fail(ctx, CSMP_EINVAL, "sweep: bad arguments");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (topk > ctx->N) topk = ctx->N;
    if (topk > ctx->M) return fail(ctx, CSMP_ERANGE, "sweep: topk > size(A,1) is not supported (no caller of argmaxinner!(P,k) needs it)");
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(solver_ensure(ctx, (int)std::max<int64_t>(topk, 1), 1, false));
    ctx->s.begun = false;
    CHECK(upload_b(ctx, r, CSMP_F64));
    CHECK(launch_sweep(ctx, ctx->s.r, 0.0, 0, 0));
    CHECK(launch_select(ctx, 0, 0));
    if (abs_corr) {
        HIPCHECK(hipMemcpyAsync(abs_corr, ctx->s.cvec, (size_t)ctx->N * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < ctx->N; ++i) abs_corr[i] = std::fabs(abs_corr[i]);  // @. Ar = abs(Ar) on the way out
    }
    if (topk == 1) {
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, ctx->s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (top_idx) top_idx[0] = hs.cand;
        if (top_val) top_val[0] = std::fabs(hs.cval);
    } else if (topk > 1) {
        CHECK(launch_topS(ctx, (int)topk));
        std::vector<int> hc((size_t)topk);
        std::vector<double> hv((size_t)topk);
        HIPCHECK(hipMemcpyAsync(hc.data(), ctx->s.cands, (size_t)topk * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipMemcpyAsync(hv.data(), ctx->s.cvals, (size_t)topk * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        for (int64_t t = 0; t < topk; ++t) {
            if (top_idx) top_idx[t] = hc[t];
            if (top_val) top_val[t] = hv[t];
        }
    }
    return CSMP_OK;
}
