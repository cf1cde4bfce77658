// host/chain.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// solver buffers; the step chains (sweep -> select -> append); back substitution and result download.
// ------------------------------------------------------------------------------------------ solver buffers
// Largest support the on-device QR append can serve: its workgroups keep five support-length vectors in LDS.
static int qr_max_cols() {
    int k = 64;
    while (qr_lds_bytes(k + 64) <= 160 * 1024 - 512) k += 64;
    return k;
}

static int solver_alloc(csmp_ctx* ctx, Solver& s, int kcap, int outcap, int qcap) {
    const int M = (int)ctx->M;
    s.ldq = ((M + kSlabRows - 1) / kSlabRows) * kSlabRows;
    s.G = (int)(s.ldq / kSlabRows);
    s.Mpad = ((M + 255) / 256) * 256;
    const int maxgrid = ctx->prop.multiProcessorCount * 8 + 8;
    CHECK(dmalloc(ctx, &s.b, s.Mpad));
    CHECK(dmalloc(ctx, &s.r, s.Mpad));
    CHECK(dmalloc(ctx, &s.bstage, s.Mpad));
    CHECK(dmalloc(ctx, &s.avec, s.Mpad));
    CHECK(dmalloc(ctx, &s.vvec, s.Mpad));
    CHECK(dmalloc(ctx, &s.cvec, (size_t)ctx->N));
    CHECK(dmalloc(ctx, &s.pval, maxgrid));
    CHECK(dmalloc(ctx, &s.scr_val, (size_t)maxgrid * kScrCandK));
    CHECK(dmalloc(ctx, &s.scr_idx, (size_t)maxgrid * kScrCandK));
    CHECK(dmalloc(ctx, &s.scr_cb, 1));
    CHECK(dmalloc(ctx, &s.scr_flag, 2));
    CHECK(dmalloc(ctx, &s.scr_tickets, (size_t)(maxgrid / kScrPartWgs + 2) * kScrTicketStride));
    HIPCHECK(hipMemsetAsync(s.scr_tickets, 0, (size_t)(maxgrid / kScrPartWgs + 2) * kScrTicketStride * sizeof(unsigned), ctx->stream));
    s.claim_words = (size_t)(kClaimMaxWgs + 1) * kClaimStride;
    CHECK(dmalloc(ctx, &s.claim, 2 * s.claim_words));
    HIPCHECK(hipMemsetAsync(s.claim, 0, 2 * s.claim_words * sizeof(unsigned), ctx->stream));
    s.claim_par = 0;
    CHECK(dmalloc(ctx, &s.pidx, maxgrid));
    CHECK(dmalloc(ctx, &s.Q, (size_t)s.ldq * qcap));
    CHECK(dmalloc(ctx, &s.R, (size_t)qcap * qcap));
    CHECK(dmalloc(ctx, &s.z, kcap));
    CHECK(dmalloc(ctx, &s.W1, qcap));
    CHECK(dmalloc(ctx, &s.coef, kcap));
    CHECK(dmalloc(ctx, &s.P1, (size_t)s.G * qcap));
    CHECK(dmalloc(ctx, &s.P2, (size_t)s.G * qcap));
    CHECK(dmalloc(ctx, &s.P2s, (size_t)2 * s.G));
    CHECK(dmalloc(ctx, &s.P1s, (size_t)2 * s.G));
    CHECK(dmalloc(ctx, &s.scal, 8));
    CHECK(dmalloc(ctx, &s.sel, kcap));
    CHECK(dmalloc(ctx, &s.cands, kcap));
    CHECK(dmalloc(ctx, &s.ncands, 4));
    CHECK(dmalloc(ctx, &s.st, 1));
    s.top_nb = (int)((ctx->N + kTopChunk - 1) / kTopChunk);
    CHECK(dmalloc(ctx, &s.top_lv, (size_t)s.top_nb * kTopSmall));
    CHECK(dmalloc(ctx, &s.top_li, (size_t)s.top_nb * kTopSmall));
    CHECK(dmalloc(ctx, &s.cvals, kcap));
    CHECK(dmalloc(ctx, &s.rs_gt, kcap));
    CHECK(dmalloc(ctx, &s.rs_eq, kRsEqCap));
    CHECK(dmalloc(ctx, &s.rs_work, kcap));
    CHECK(dmalloc(ctx, &s.rs, 1));
    CHECK(dmalloc(ctx, &s.out_idx, outcap));
    CHECK(dmalloc(ctx, &s.out_order, outcap));
    CHECK(dmalloc(ctx, &s.out_val, outcap));
    CHECK(dmalloc(ctx, &s.out_nnz, 1));
    HIPCHECK(hipMemsetAsync(s.st, 0, sizeof(DevState), ctx->stream));
    return CSMP_OK;
}

// Buffers of the active solver slot for supports of up to kcap atoms and outcap output entries.  qr = false
// (MP, the sweep primitive): the QR arrays are not needed and stay at whatever size they have.  The slot only
// grows; a request is either served completely or leaves an EMPTY slot (kcap = 0) behind, never a half-built one.
static int solver_ensure(csmp_ctx* ctx, int kcap, int outcap, bool qr = true) {
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    Solver& s = ctx->s;
    if (s.kcap >= kcap && s.outcap >= outcap && (!qr || s.qcap == s.kcap)) return CSMP_OK;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    kcap = std::max(kcap, s.kcap);
    outcap = std::max(outcap, s.outcap);
    const int qcap = qr ? kcap : 1;
    solver_free(s);
    Solver n;
    const int rc = solver_alloc(ctx, n, kcap, outcap, qcap);
    if (rc != CSMP_OK) {
        solver_free(n);
        return rc;
    }
    n.kcap = kcap;
    n.outcap = outcap;
    n.qcap = qcap;
    s = n;
    return CSMP_OK;
}

// The column-removal kernels (csmp_downdate.hpp, csmp_tinv.hpp) address R and T with the slot's capacity as
// leading dimension and scan one support in one workgroup: at most kTMaxCols columns (kDelMaxCols for the functor's down-date).  A slot that an earlier
// call grew beyond that is rebuilt at the size this call needs.
static int solver_fit_for_removal(csmp_ctx* ctx, int kcap) {
    Solver& s = ctx->s;
    if (s.kcap > kTMaxCols && kcap <= kTMaxCols) {
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        solver_free(s);
    }
    return CSMP_OK;
}

// Small results coming back on a latency chain: the pieces land in the page-locked slot (a copy straight into pageable memory --
// a stack variable, a std::vector -- is staged by the runtime and blocks the host once per piece).
struct PinFetch {
    csmp_ctx* ctx;
    char* base = nullptr;
    size_t used = 0;
    struct Out { void* dst; size_t off, bytes; } outs[kLandMax];
    int nout = 0;
    LandSegs segs{};
    unsigned maxwords = 0;
    explicit PinFetch(csmp_ctx* c) : ctx(c) {}
    int begin(size_t total) {
        void* pv = nullptr;
        CHECK(pin_get(ctx, 1, total + 64, &pv));
        base = (char*)pv;
        used = 0;
        nout = 0;
        segs.n = 0;
        maxwords = 0;
        return CSMP_OK;
    }
    int add(void* dst, const void* dev, size_t bytes) {
        if (nout == kLandMax) return fail(ctx, CSMP_EINVAL, "PinFetch: too many pieces");
        const size_t off = (used + 7) / 8 * 8;
        if ((bytes & 3) || ((uintptr_t)dev & 3)) {  // (never in this library: every piece is whole, aligned 4-byte words)
            HIPCHECK(hipMemcpyAsync(base + off, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            segs.src[segs.n] = dev;
            segs.off[segs.n] = (unsigned)off;
            segs.words[segs.n] = (unsigned)(bytes / 4);
            maxwords = std::max(maxwords, (unsigned)(bytes / 4));
            segs.n += 1;
        }
        outs[nout++] = {dst, off, bytes};
        used = off + bytes;
        return CSMP_OK;
    }
    // ONE kernel writes all pieces into the page-locked landing area over the host link (a small hipMemcpyAsync is a blit kernel
    // of its own: 5 us on the chain for every piece), ONE wait, then the pieces are handed to their host destinations
    int wait() {
        if (segs.n > 0) {
            const int grid = (int)std::max(1u, std::min(64u, (maxwords + 255u) / 256u));
            hipLaunchKernelGGL(k_land_multi, dim3(grid), dim3(256), 0, ctx->stream, segs, base);
            HIPCHECK(hipGetLastError());
        }
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        for (int q = 0; q < nout; ++q) memcpy(outs[q].dst, base + outs[q].off, outs[q].bytes);
        return CSMP_OK;
    }
};

// b (host, any dtype) -> device Float64 b and r, state reset
static int upload_b(csmp_ctx* ctx, const void* b, int b_dtype) {
    Solver& s = ctx->s;
    const int M = (int)ctx->M;
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    // through the page-locked slot: the copy is asynchronous and the host does not wait for it.  The slot is rewritten by the
    // next upload only -- after the stream has been drained at least once (every entry point ends with its results on the host).
    void* pv = nullptr;
    CHECK(pin_get(ctx, 0, (size_t)M * sizeof(double), &pv));
    HIPCHECK(hipStreamSynchronize(ctx->stream));  // (a previous upload of a step-level caller may still be in flight)
    double* hb = (double*)pv;
    if (b_dtype == CSMP_F32)
        for (int i = 0; i < M; ++i) hb[i] = (double)((const float*)b)[i];
    else
        memcpy(hb, b, (size_t)M * sizeof(double));
    HIPCHECK(hipMemcpyAsync(s.bstage, hb, (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    s.keep_valid = false;
    s.fac_valid = false;  // (the kept Gram matrix carries A_S'b of the previous b)
    hipLaunchKernelGGL(k_init<double>, dim3(s.Mpad / 256), dim3(256), 0, ctx->stream, (const double*)s.bstage, M, s.Mpad, s.b, s.r, s.st);
    HIPCHECK(hipGetLastError());
    s.jh = 0;
    return CSMP_OK;
}

template <typename TB>
static int init_from_device_t(csmp_ctx* ctx, const TB* col) {
    Solver& s = ctx->s;
    s.keep_valid = false;
    s.fac_valid = false;
    hipLaunchKernelGGL(k_init<TB>, dim3(s.Mpad / 256), dim3(256), 0, ctx->stream, col, (int)ctx->M, s.Mpad, s.b, s.r, s.st);
    HIPCHECK(hipGetLastError());
    s.jh = 0;
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ step chains
static int launch_select(csmp_ctx* ctx, int mode, int skipmask) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_select, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.pval, (const int*)s.pidx,
                       ctx->sweep_grid, (const double*)s.cvec, (const int*)s.sel, s.st, (int)ctx->M, s.kcap, mode, skipmask);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// One atom through the append chain.  mode 1: atom = arg-max of the last sweep + OMP guards
// (src/matchingpursuit.jl:63,65-66); mode 2: atom = cands[which] + GOMP's duplicate skip
// (src/util.jl:119,129-134).  Then add_column!(AiQR, A[:, atom]) and the residual update.
static int launch_append(csmp_ctx* ctx, int mode, int which, int skipmask, bool optimistic = false, double min_d2 = 0.0, int nblk_sweep = 0,
                         const void* onecol = nullptr) {
    Solver& s = ctx->s;
    // onecol (mode 4): the atom's column is handed over as a one-column dictionary (ld = 0: every index reads it)
    const void* dA = onecol ? onecol : ctx->dA;
    const int64_t ldA = onecol ? 0 : ctx->ld;
    const int jh = std::min(s.jh, s.kcap);
    // the LDS vectors of the append kernels are sized by the support they can meet (jh bounds it), not by the capacity
    const int jpad = qr_jpad(jh);
    const size_t lds = qr_lds_bytes(jh);
    if (jh >= qr_max_cols() || lds > 160 * 1024 - 512) {
        // The append kernels keep five support-length vectors in LDS: about 3900 columns.  A support beyond that -- the
        // reference's defaults k = size(A,1) at M >= 4096 with a residual test that never fires (src/matchingpursuit.jl:54,89,108)
        // -- is served by the SPILL kernels: the same bodies with those vectors in global memory (5 jpad doubles per workgroup,
        // L2-resident), only the slab scratch in LDS.
        const int jpc = qr_jpad(s.kcap);  // (sized by the capacity once: the launches below follow jh growing to it)
        if (!s.spill || s.spill_cap < qr_spill_doubles(s.G, s.kcap)) {
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            dfree(s.spill);
            CHECK(dmalloc(ctx, &s.spill, qr_spill_doubles(s.G, s.kcap)));
            s.spill_cap = qr_spill_doubles(s.G, s.kcap);
        }
        const size_t slds = qr_spill_lds_bytes();
        if (ctx->dtype == CSMP_F32)
            hipLaunchKernelGGL(k_qr1s<float>, dim3(s.G), dim3(kQrThreads), slds, ctx->stream, (const float*)dA, ldA,
                               (int)ctx->M, (const double*)s.Q, s.ldq, s.st, s.avec, s.P1, s.G, s.kcap, jpc, mode,
                               (const double*)s.pval, (const int*)s.pidx, nblk_sweep > 0 ? nblk_sweep : ctx->sweep_grid, (const int*)s.cands,
                               (const int*)s.ncands, which, (const int*)s.sel, skipmask, (const double*)s.r, s.P1s, jh, min_d2, s.spill);
        else
            hipLaunchKernelGGL(k_qr1s<double>, dim3(s.G), dim3(kQrThreads), slds, ctx->stream, (const double*)dA, ldA,
                               (int)ctx->M, (const double*)s.Q, s.ldq, s.st, s.avec, s.P1, s.G, s.kcap, jpc, mode,
                               (const double*)s.pval, (const int*)s.pidx, nblk_sweep > 0 ? nblk_sweep : ctx->sweep_grid, (const int*)s.cands,
                               (const int*)s.ncands, which, (const int*)s.sel, skipmask, (const double*)s.r, s.P1s, jh, min_d2, s.spill);
        HIPCHECK(hipGetLastError());
        hipLaunchKernelGGL(k_qr2s, dim3(s.G), dim3(kQrThreads), slds, ctx->stream, s.Q, s.ldq, s.st, (const double*)s.avec, s.r,
                           (const double*)s.P1, (const double*)s.P1s, s.G, s.W1, s.vvec, s.P2, s.P2s, s.R, s.z, s.sel, s.kcap,
                           jpc, 0, jh, optimistic ? 1 : 0, s.spill);
        HIPCHECK(hipGetLastError());
        if (s.jh < s.kcap) s.jh += 1;
        if (optimistic) return CSMP_OK;
        hipLaunchKernelGGL(k_qr3s, dim3(s.G), dim3(kQrThreads), slds, ctx->stream, s.Q, s.ldq, s.st, (const double*)s.vvec, s.r,
                           (const double*)s.P2, (const double*)s.P2s, s.G, (const double*)s.W1, s.R, s.z, s.sel, s.kcap, jpc, s.spill);
        HIPCHECK(hipGetLastError());
        return CSMP_OK;
    }
    if (lds > 64 * 1024) {
        HIPCHECK(hipFuncSetAttribute((const void*)k_qr1<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHECK(hipFuncSetAttribute((const void*)k_qr1<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHECK(hipFuncSetAttribute((const void*)k_qr2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHECK(hipFuncSetAttribute((const void*)k_qr3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_qr1<float>, dim3(s.G), dim3(kQrThreads), lds, ctx->stream, (const float*)dA, ldA,
                           (int)ctx->M, (const double*)s.Q, s.ldq, s.st, s.avec, s.P1, s.G, s.kcap, jpad, mode,
                           (const double*)s.pval, (const int*)s.pidx, nblk_sweep > 0 ? nblk_sweep : ctx->sweep_grid, (const int*)s.cands,
                           (const int*)s.ncands, which, (const int*)s.sel, skipmask, (const double*)s.r, s.P1s, jh, min_d2);
    else
        hipLaunchKernelGGL(k_qr1<double>, dim3(s.G), dim3(kQrThreads), lds, ctx->stream, (const double*)dA, ldA,
                           (int)ctx->M, (const double*)s.Q, s.ldq, s.st, s.avec, s.P1, s.G, s.kcap, jpad, mode,
                           (const double*)s.pval, (const int*)s.pidx, nblk_sweep > 0 ? nblk_sweep : ctx->sweep_grid, (const int*)s.cands,
                           (const int*)s.ncands, which, (const int*)s.sel, skipmask, (const double*)s.r, s.P1s, jh, min_d2);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_qr2, dim3(s.G), dim3(kQrThreads), lds, ctx->stream, s.Q, s.ldq, s.st, (const double*)s.avec, s.r,
                       (const double*)s.P1, (const double*)s.P1s, s.G, s.W1, s.vvec, s.P2, s.P2s, s.R, s.z, s.sel, s.kcap,
                       jpad, 0, jh, optimistic ? 1 : 0);
    HIPCHECK(hipGetLastError());
    if (s.jh < s.kcap) s.jh += 1;
    if (optimistic) return CSMP_OK;  // k_qr3 (second Gram-Schmidt pass) only in the safe chain
    hipLaunchKernelGGL(k_qr3, dim3(s.G), dim3(kQrThreads), lds, ctx->stream, s.Q, s.ldq, s.st, (const double*)s.vvec, s.r,
                       (const double*)s.P2, (const double*)s.P2s, s.G, (const double*)s.W1, s.R, s.z, s.sel, s.kcap, jpad);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

static int launch_mp_update(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    const int grid = ((int)ctx->M + 255) / 256;
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_mp_update<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, s.r, s.st, s.sel, s.z);
    else
        hipLaunchKernelGGL(k_mp_update<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, s.r, s.st, s.sel, s.z);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// update!(P::OMP, x) + the driver's residual check of the PREVIOUS iteration (src/matchingpursuit.jl:62-70,79)
static int omp_step(csmp_ctx* ctx, double eps, int check_eps, bool optimistic) {
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    CHECK(launch_sweep(ctx, ctx->s.r, eps, check_eps, skip));
    return launch_append(ctx, 1, 0, skip, optimistic);
}

// ldiv! + SparseVector assembly into device outputs
static int launch_finish(csmp_ctx* ctx, int64_t* d_idx, double* d_val, int64_t* d_nnz, int64_t* d_order, int outcap,
                         int* d_flag = nullptr) {
    Solver& s = ctx->s;
    if (s.kcap > 256) {
        // super-blocks of 256 columns over several CUs (k_trsv_*): the host's bound on the support says how many there are; a
        // super-block beyond the true support returns at once
        const int jb = s.jh > 0 ? std::min(s.jh, s.kcap) : s.kcap;
        const int nsb = (jb + kTrsvBlk - 1) / kTrsvBlk;
        for (int sb = nsb - 1; sb >= 0; --sb) {
            const int off = sb * kTrsvBlk;
            hipLaunchKernelGGL(k_trsv_blk, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.R, (const double*)s.z, (const DevState*)s.st,
                               s.kcap, s.coef, off, sb == nsb - 1 ? 1 : 0);
            if (sb > 0)
                hipLaunchKernelGGL(k_trsv_upd, dim3(off / 64), dim3(256), 0, ctx->stream, (const double*)s.R, (const DevState*)s.st, s.kcap,
                                   s.coef, off);
        }
        const int ne = std::max(jb, outcap);
        hipLaunchKernelGGL(k_trsv_emit, dim3((ne + 255) / 256), dim3(256), (size_t)(s.kcap + 4) * sizeof(int), ctx->stream,
                           (const double*)s.coef, (const int*)s.sel, (const DevState*)s.st, d_idx, d_val, d_nnz, d_order, outcap, d_flag);
        HIPCHECK(hipGetLastError());
        return CSMP_OK;
    }
    if (s.kcap <= 1024) {  // single-wave form
        const size_t lds = (size_t)s.kcap * sizeof(int);
        if (s.kcap <= 256)
            hipLaunchKernelGGL(k_finish_w<4>, dim3(1), dim3(64), lds, ctx->stream, (const double*)s.R, (const double*)s.z,
                               (const int*)s.sel, (const DevState*)s.st, s.kcap, s.coef, d_idx, d_val, d_nnz, d_order, outcap, d_flag);
        else if (s.kcap <= 512)
            hipLaunchKernelGGL(k_finish_w<8>, dim3(1), dim3(64), lds, ctx->stream, (const double*)s.R, (const double*)s.z,
                               (const int*)s.sel, (const DevState*)s.st, s.kcap, s.coef, d_idx, d_val, d_nnz, d_order, outcap, d_flag);
        else
            hipLaunchKernelGGL(k_finish_w<16>, dim3(1), dim3(64), lds, ctx->stream, (const double*)s.R, (const double*)s.z,
                               (const int*)s.sel, (const DevState*)s.st, s.kcap, s.coef, d_idx, d_val, d_nnz, d_order, outcap, d_flag);
        HIPCHECK(hipGetLastError());
        return CSMP_OK;
    }
    const size_t lds = (size_t)(s.kcap + 2) * sizeof(double);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), lds, ctx->stream, (const double*)s.R, (const double*)s.z,
                       (const int*)s.sel, (const DevState*)s.st, s.kcap, s.coef, d_idx, d_val, d_nnz, d_order, outcap, d_flag);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

static int download_result(csmp_ctx* ctx, int outcap, int64_t* idx, double* val, int64_t* nnz, int64_t* order) {
    Solver& s = ctx->s;
    std::vector<int64_t> hi((size_t)outcap), ho((size_t)outcap);
    std::vector<double> hv((size_t)outcap);
    int64_t hn = 0;
    PinFetch f(ctx);
    CHECK(f.begin((size_t)outcap * 24 + 64));
    CHECK(f.add(hi.data(), s.out_idx, (size_t)outcap * 8));
    CHECK(f.add(hv.data(), s.out_val, (size_t)outcap * 8));
    CHECK(f.add(ho.data(), s.out_order, (size_t)outcap * 8));
    CHECK(f.add(&hn, s.out_nnz, 8));
    CHECK(f.wait());
    for (int64_t t = 0; t < hn; ++t) {
        if (idx) idx[t] = hi[t];
        if (val) val[t] = hv[t];
        if (order) order[t] = ho[t];
    }
    if (nnz) *nnz = hn;
    return CSMP_OK;
}

// Every kPollSteps steps of a long single-signal driver loop the host looks at the control block once: a solve that
// has stopped (residual test, stagnation, full support) is not followed by thousands of no-op launches -- the
// reference's defaults ask for k = size(A,1) steps (src/matchingpursuit.jl:73,126) -- and the host's bound on the
// support (jh, which sizes the append kernels' LDS vectors) snaps back to the true column count.
static constexpr int64_t kPollSteps = 256;
static int solver_poll(csmp_ctx* ctx, bool* stopped) {
    Solver& s = ctx->s;
    DevState hs;
    HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    s.jh = std::min(s.kcap, hs.nsel);
    *stopped = hs.done != 0;
    return CSMP_OK;
}
