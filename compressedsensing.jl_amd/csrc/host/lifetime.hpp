// host/lifetime.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// lifetime of a context, options, streams.
// ------------------------------------------------------------------------------------------ lifetime
extern "C" int csmp_version(void) { return 100; }

extern "C" const char* csmp_last_error(const csmp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int csmp_create(csmp_ctx** out, int device_id) {
    if (!out) return CSMP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device visible: ") + hipGetErrorString(e) +
                       " -- libcsmp has no CPU fallback";
        return CSMP_EHIP;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_err = "device_id out of range";
        return CSMP_EINVAL;
    }
    csmp_ctx* ctx = new csmp_ctx();
    ctx->dev = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_err = "hipSetDevice / hipStreamCreate failed";
        delete ctx;
        return CSMP_EHIP;
    }
    *out = ctx;
    return CSMP_OK;
}

static void batch_free(Batch& b, bool keep_dict) {
    dfree(b.Rb); dfree(b.r); dfree(b.b); dfree(b.T); dfree(b.Tt); dfree(b.z); dfree(b.sel); dfree(b.bs);
    dfree(b.cand_val); dfree(b.cand_idx); dfree(b.pick); dfree(b.R8); dfree(b.sigscale);
    b.Bcap = b.kcap = 0;
    if (!keep_dict) {
        if (b.ab_borrowed) b.Ab = nullptr;
        b.ab_borrowed = false;
        dfree(b.Ab);
        dfree(b.amax);
        if (b.a8_borrowed) b.A8 = nullptr;
        b.a8_borrowed = false;
        dfree(b.A8);
        b.a8_valid = false;
        if (b.ah_borrowed) b.Ah = nullptr;
        b.ah_borrowed = false;
        dfree(b.Ah);
        b.ah_valid = false;
        dfree(b.Gm);
        b.gram_valid = false;
        b.ab_valid = false;
        b.meta_valid = false;
        b.anorm_host = -1.f;
    }
}

// Make slot `slot` the active solver: every launch helper works on ctx->s / ctx->stream.
static void activate_slot(csmp_ctx* ctx, int slot) {
    if (ctx->active == slot) return;
    ctx->park[ctx->active] = ctx->s;
    ctx->s = ctx->park[slot];
    ctx->park[slot] = Solver();
    ctx->active = slot;
}
static hipError_t sync_all(csmp_ctx* ctx) { return hipStreamSynchronize(ctx->stream); }

static void solver_free(Solver& s) {
    dfree(s.spill); s.spill_cap = 0;
    dfree(s.b); dfree(s.r); dfree(s.cvec); dfree(s.pval); dfree(s.pidx); dfree(s.Q); dfree(s.R); dfree(s.z);
    dfree(s.W1); dfree(s.P1); dfree(s.P2); dfree(s.P2s); dfree(s.P1s); dfree(s.avec); dfree(s.vvec); dfree(s.coef);
    dfree(s.scal); dfree(s.sel); dfree(s.cands); dfree(s.ncands); dfree(s.st); dfree(s.bstage);
    dfree(s.top_lv); dfree(s.cvals); dfree(s.top_li); dfree(s.rs_gt); dfree(s.rs_eq); dfree(s.rs_work); dfree(s.rs);
    dfree(s.out_idx); dfree(s.out_order); dfree(s.out_nnz); dfree(s.out_val); dfree(s.sigflags); dfree(s.scr_val); dfree(s.scr_idx); dfree(s.scr_tickets); dfree(s.claim); dfree(s.scr_cb); dfree(s.scr_flag);
    dfree(s.Apan); dfree(s.Vpan); dfree(s.PB1); dfree(s.W1b); dfree(s.PG); dfree(s.Gsum); dfree(s.pan_atoms);
    dfree(s.rho2); dfree(s.dvec); dfree(s.frg1); dfree(s.frg2); dfree(s.frq); dfree(s.swapH); dfree(s.swapv); s.swapv_cap = 0;
    dfree(s.R2); dfree(s.Gdel); dfree(s.qdrop); dfree(s.qsave); dfree(s.bwd); dfree(s.bwd_coef); dfree(s.bwd_info); dfree(s.delmeta); dfree(s.delpos);
    dfree(s.T); dfree(s.T2); dfree(s.tpd); dfree(s.tpn); dfree(s.tmeta); dfree(s.extcol);
    dfree(s.Gm); dfree(s.Dfac); dfree(s.Gpart); dfree(s.gdiag); dfree(s.rpart); dfree(s.Acomp); dfree(s.Gkeep); dfree(s.gdkeep); dfree(s.Gkeep2); dfree(s.gdkeep2); dfree(s.Wb); dfree(s.Gin); dfree(s.Gm2); dfree(s.ytmp); dfree(s.kpos); dfree(s.rhs_part); dfree(s.rn2part);
    s = Solver();
}

static void dict_release(csmp_ctx* ctx) {
    if (ctx->share && --ctx->share->refs == 0) {
        if (ctx->share->kind == 0) (void)hipFree(ctx->share->p);
        else if (ctx->share->kind == 1) (void)hipHostFree(ctx->share->p);
        else (void)hipHostUnregister(ctx->share->p);
        delete ctx->share;
    }
    ctx->share = nullptr;
    ctx->dA = nullptr;
    ctx->ownA = false;
    ctx->streamed = false;
}

extern "C" int csmp_destroy(csmp_ctx* ctx) {
    if (!ctx) return CSMP_OK;
    (void)hipSetDevice(ctx->dev);
    (void)sync_all(ctx);
    for (int q = 2; q >= 0; --q) {
        activate_slot(ctx, q);
        solver_free(ctx->s);
    }
    batch_free(ctx->bt, false);
    if (ctx->comm) (void)csmp_comm_free(ctx);
    for (auto& t : ctx->twins) {
        if (t) (void)csmp_destroy(t);
        t = nullptr;
    }
    if (ctx->ev_twin) (void)hipEventDestroy(ctx->ev_twin);
    if (ctx->prof_ref) (void)hipEventDestroy(ctx->prof_ref);
    if (ctx->spjob.ev) (void)hipEventDestroy(ctx->spjob.ev);
    if (ctx->stream_b) (void)hipStreamDestroy(ctx->stream_b);
    for (hipEvent_t e : {ctx->ev_fork, ctx->ev_join, ctx->ev_off, ctx->ev_gate})
        if (e) (void)hipEventDestroy(e);
    dict_release(ctx);
    for (auto& e : ctx->ev) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev2) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int q = 0; q < 3; ++q)
        if (ctx->pin[q]) (void)hipHostFree(ctx->pin[q]);
    delete ctx;
    return CSMP_OK;
}

// slot `q` of the page-locked host buffers, at least `bytes` long (grown with the stream drained: nothing is in flight on it)
static int pin_get(csmp_ctx* ctx, int q, size_t bytes, void** out) {
    if (ctx->pin_bytes[q] < bytes) {
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->pin[q]) HIPCHECK(hipHostFree(ctx->pin[q]));
        ctx->pin[q] = nullptr;
        ctx->pin_bytes[q] = 0;
        const size_t want = std::max<size_t>(bytes, 64 * 1024);
        HIPCHECK(hipHostMalloc(&ctx->pin[q], want, hipHostMallocDefault));
        ctx->pin_bytes[q] = want;
    }
    *out = ctx->pin[q];
    return CSMP_OK;
}

extern "C" int csmp_set_stream(csmp_ctx* ctx, void* hip_stream) {
    if (!ctx) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(sync_all(ctx));
    activate_slot(ctx, 0);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return CSMP_OK;
}

extern "C" int csmp_sync(csmp_ctx* ctx) {
    if (!ctx) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(sync_all(ctx));
    return CSMP_OK;
}

// Options: the choices that exist only on this side of the boundary (the reference passes its own as arguments:
// src/matchingpursuit.jl:88-91,145-148, src/twostage.jl:87).  Per context; a clone starts from its parent's values.
static int* opt_slot(csmp_ctx* ctx, int key, int64_t* lo, int64_t* hi) {
    switch (key) {
        case CSMP_OPT_BATCH_CERT: *lo = 0; *hi = 1; return &ctx->opt_batch_cert;
        case CSMP_OPT_BATCH_GRAM: *lo = 0; *hi = 1; return &ctx->opt_batch_gram;
        case CSMP_OPT_BATCH_WINDOW: *lo = 0; *hi = kWinMax; return &ctx->opt_batch_window;
        case CSMP_OPT_SOLVES_IN_FLIGHT: *lo = 1; *hi = 4; return &ctx->opt_in_flight;
        case CSMP_OPT_SCREENED_SWEEP: *lo = 0; *hi = 3; return &ctx->opt_screened;
        case CSMP_OPT_BATCH_SCREEN: *lo = 0; *hi = 3; return &ctx->opt_batch_screen;
        default: return nullptr;
    }
}
static bool* opt_flag(csmp_ctx* ctx, int key) {
    switch (key) {
        case CSMP_OPT_PIPELINE: return &ctx->pipeline;
        default: return nullptr;
    }
}
extern "C" int csmp_set_option(csmp_ctx* ctx, int key, int64_t value) {
    if (!ctx) return CSMP_EINVAL;
    int64_t lo = 0, hi = 0;
    if (int* p = opt_slot(ctx, key, &lo, &hi)) {
        if (value < lo || value > hi) return fail(ctx, CSMP_EINVAL, "csmp_set_option: value out of range");
        if (key == CSMP_OPT_BATCH_GRAM && value == 0 && ctx->bt.Gm) {  // switching the Gram matrix off releases its 8 N^2 bytes
            HIPCHECK(hipSetDevice(ctx->dev));
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            dfree(ctx->bt.Gm);
            ctx->bt.gram_valid = false;
        }
        *p = (int)value;
        return CSMP_OK;
    }
    if (bool* f = opt_flag(ctx, key)) {
        if (value != 0 && value != 1) return fail(ctx, CSMP_EINVAL, "csmp_set_option: value must be 0 or 1");
        *f = value != 0;
        return CSMP_OK;
    }
    return fail(ctx, CSMP_EINVAL, "csmp_set_option: unknown key");
}
extern "C" int csmp_get_option(csmp_ctx* ctx, int key, int64_t* value) {
    if (!ctx || !value) return CSMP_EINVAL;
    int64_t lo = 0, hi = 0;
    if (int* p = opt_slot(ctx, key, &lo, &hi)) {
        *value = *p;
        return CSMP_OK;
    }
    if (bool* f = opt_flag(ctx, key)) {
        *value = *f ? 1 : 0;
        return CSMP_OK;
    }
    return fail(ctx, CSMP_EINVAL, "csmp_get_option: unknown key");
}

extern "C" int csmp_device_info(csmp_ctx* ctx, char* name, int name_len, int* compute_units, int64_t* hbm_bytes) {
    if (!ctx) return CSMP_EINVAL;
    if (name && name_len > 0) {
        std::string n = std::string(ctx->prop.name) + " (" + ctx->prop.gcnArchName + ")";
        snprintf(name, (size_t)name_len, "%s", n.c_str());
    }
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    return CSMP_OK;
}
