// host/dictionary.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// sweep configuration and launch; the resident dictionary.
// ------------------------------------------------------------------------------------------ sweep launch
// the product sweep (k_sweep_gen): U loads per unit, a ring of 32 loads, the residual in one image or staged in phases
template <typename TA, int U, int NB>
static hipError_t sweep_launch_t(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, double* cout, int64_t ncols = 0) {
    auto kern = k_sweep_gen<TA, U, NB>;
    // the LDS REQUEST may exceed what the kernel uses (configure_sweep: sweep_lds_req): it sets how many workgroups share a CU, and with
    // more workgroups than fit the rest QUEUE -- a CU that a workgroup leaves goes to the next one in line
    const size_t lds = std::max(ctx->sweep_lds, ctx->sweep_lds_req);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    Solver& s = ctx->s;
    hipLaunchKernelGGL(kern, dim3(ctx->sweep_grid), dim3(kSweepThreads), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv,
                       ncols > 0 ? ncols : ctx->N, r, cout ? cout : s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask, ctx->sweep_KP);
    return hipGetLastError();
}
// the two sets of column-pool counters of a solver slot (sweep_body_dyn): this launch's, and the one it zeroes for the slot's next sweep
static void claim_sets(Solver& s, unsigned*& cur, unsigned*& next) {
    cur = s.claim + (size_t)s.claim_par * s.claim_words;
    s.claim_par ^= 1;
    next = s.claim + (size_t)s.claim_par * s.claim_words;
}
// the sweep with its columns handed out at run time (k_sweep_dyn): one residual image only
template <typename TA, int U, int NB>
static hipError_t sweep_launch_dyn(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, double* cout) {
    auto kern = k_sweep_dyn<TA, U, NB>;
    if (ctx->sweep_lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->sweep_lds);
        if (e != hipSuccess) return e;
    }
    Solver& s = ctx->s;
    unsigned *cur, *next;
    claim_sets(s, cur, next);
    hipLaunchKernelGGL(kern, dim3(ctx->sweep_grid), dim3(kSweepDynThreads), ctx->sweep_lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv,
                       ctx->N, r, cout ? cout : s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask, ctx->sweep_KP, cur, next, ctx->claim_pools | (ctx->tune_sweep_dyn << 16));
    return hipGetLastError();
}
// a residual longer than the LDS, staged in phases (k_sweep_ph)
template <typename TA>
static hipError_t sweep_launch_ph(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, double* cout, int64_t ncols) {
    auto kern = k_sweep_ph<TA, 8, 4>;
    if (ctx->sweep_lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->sweep_lds);
        if (e != hipSuccess) return e;
    }
    Solver& s = ctx->s;
    hipLaunchKernelGGL(kern, dim3(ctx->sweep_grid), dim3(kSweepThreads), ctx->sweep_lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv,
                       ncols > 0 ? ncols : ctx->N, r, cout ? cout : s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask, ctx->sweep_KP, ctx->sweep_pcap);
    return hipGetLastError();
}
// short columns (k_sweep_short): U loads = CPU neighbouring columns of U / CPU chunks, one transposing reduction per unit
template <typename TA, int NCH, int CPU>
static hipError_t sweep_launch_short(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, double* cout, int64_t ncols) {
    auto kern = k_sweep_short<TA, NCH, CPU>;
    Solver& s = ctx->s;
    hipLaunchKernelGGL(kern, dim3(ctx->sweep_grid), dim3(kSweepThreads), ctx->short_lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv,
                       ncols > 0 ? ncols : ctx->N, r, cout ? cout : s.cvec, s.pval, s.pidx, s.st, eps, check_eps, skipmask, ctx->short_KP);
    return hipGetLastError();
}
template <typename TA>
static hipError_t sweep_product(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, double* cout, int64_t ncols = 0) {
    if (ctx->short_cpu > 0 && ctx->short_nch == 1) return sweep_launch_short<TA, 1, 4>(ctx, r, eps, check_eps, skipmask, cout, ncols);
    if (ctx->short_cpu > 0 && ctx->short_nch == 2) return sweep_launch_short<TA, 2, 4>(ctx, r, eps, check_eps, skipmask, cout, ncols);
    if (ctx->short_cpu > 0) return sweep_launch_short<TA, 4, 2>(ctx, r, eps, check_eps, skipmask, cout, ncols);
    if (ctx->sweep_dyn && ncols == 0) {
        switch (ctx->sweep_U) {
            case 16: return sweep_launch_dyn<TA, 16, 2>(ctx, r, eps, check_eps, skipmask, cout);
            case 8: return sweep_launch_dyn<TA, 8, 4>(ctx, r, eps, check_eps, skipmask, cout);
            default: return sweep_launch_dyn<TA, 4, 8>(ctx, r, eps, check_eps, skipmask, cout);
        }
    }
    if (ctx->sweep_ph) return sweep_launch_ph<TA>(ctx, r, eps, check_eps, skipmask, cout, ncols);
    switch (ctx->sweep_U) {
        case 16: return sweep_launch_t<TA, 16, 2>(ctx, r, eps, check_eps, skipmask, cout, ncols);
        case 8: return sweep_launch_t<TA, 8, 4>(ctx, r, eps, check_eps, skipmask, cout, ncols);
        default: return sweep_launch_t<TA, 4, 8>(ctx, r, eps, check_eps, skipmask, cout, ncols);
    }
}

// true when this sweep launch is one of the sampled ones
static bool prof_pick(csmp_ctx* ctx) {
    if (!ctx->prof) return false;
    const bool pick = (ctx->prof_count % ctx->prof_every) == 0;
    if (pick) {  // (csmp_profile_window: the launches between the first and the last sampled one)
        if (ctx->prof_first < 0) ctx->prof_first = ctx->prof_count;
        ctx->prof_last = ctx->prof_count;
    }
    ctx->prof_count += 1;
    return pick;
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


// cout: where c = A'r goes (default: the solver slot's correlation vector); ncols > 0: the first ncols columns only
static int launch_sweep(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, double* cout = nullptr, int64_t ncols = 0) {
    const bool timed = ncols == 0 && prof_pick(ctx);
    if (timed) CHECK(prof_mark(ctx));
    hipError_t e = ctx->dtype == CSMP_F32
                       ? sweep_product<float>(ctx, r, eps, check_eps, skipmask, cout, ncols)
                       : sweep_product<double>(ctx, r, eps, check_eps, skipmask, cout, ncols);
    HIPCHECK(e);
    if (timed) CHECK(prof_mark(ctx));
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ dictionary
// Workgroups of a sweep: `base` (a multiple of the CU count), moved within +-1/8 to the count that splits the columns over the
// waves most evenly -- only where a wave carries few columns (N = 4096 on 768 waves: most carry 5, some 6; on 820: 5 each).
static int balanced_grid(int64_t N, int64_t base) {
    const int64_t groups = (N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    if (groups <= base) return (int)std::max<int64_t>(1, groups);
    int64_t grid = base;
    if (N / (4 * base) < 32) {
        double best = 0.0;
        for (int64_t g = base + base / 8; g >= base - base / 8; --g) {
            const int64_t per_wave = (N + 4 * g - 1) / (4 * g);
            const double eff = (double)N / (double)(per_wave * 4 * g);
            if (eff > best + 5e-3) {
                best = eff;
                grid = g;
            }
        }
    }
    return (int)grid;
}

// k_sweep_gen for every shape (the reference sweeps whatever size(A) is: zeros(T, n), mul!, src/matchingpursuit.jl:54-60,183).
// One image when the residual fits the LDS: the unit size (16 / 8 / 4 loads) that pads the column's chunks least, the larger one
// on a tie.  Else phases of KP rows, 8-load units (a column has >= 10 of them there: at most one in ten is padding).
// Workgroups, measured on MI355X with 1 GiB dictionaries of M = 1000 .. 32768 rows, f32 and f64 (profiles/r05_sweep_shapes.json):
// 3/4 of the CUs (192) wherever a column is 8 KiB or longer -- every wave keeps 32 KiB in flight, and FEWER concurrent DRAM streams
// reach a higher bandwidth (256 workgroups: -3..5 %) --, three per CU for shorter columns (the per-column reduction then weighs in
// and more waves hide it).
static int configure_sweep(csmp_ctx* ctx) {
    const int vec = ctx->dtype == CSMP_F32 ? 4 : 2;
    const int rows = kWave * vec;
    const int nchunk = (ctx->Mv + rows - 1) / rows;
    const int cus = ctx->prop.multiProcessorCount;
    const size_t lds_cap = 160 * 1024 - 512;
    const int kp_cap = (int)(lds_cap / sizeof(double)) - 48;  // rows the LDS holds beside the reduction scratch and the claim rings
    ctx->sweep_ph = false;
    int bestU = 0, best_pad = 0;
    for (int u : {16, 8, 4}) {
        if (ctx->tune_sweep_U > 0 && u != ctx->tune_sweep_U) continue;
        const int pad = ((nchunk + u - 1) / u) * u;
        if (pad * rows > kp_cap) continue;
        if (!bestU || pad < best_pad) {
            bestU = u;
            best_pad = pad;
        }
    }
    const size_t col_bytes = (size_t)ctx->Mv * (ctx->dtype == CSMP_F32 ? 4 : 8);
    const int64_t base = col_bytes >= 8192 ? (int64_t)cus * 3 / 4 : (int64_t)cus * 3;
    const int maxgrid = cus * 8 + 8;  // (the per-workgroup partials pval / pidx of a solver slot are sized for it: solver_alloc)
    ctx->sweep_grid = ctx->tune_sweep_grid > 0 ? balanced_grid(ctx->N, ctx->tune_sweep_grid) : balanced_grid(ctx->N, base);
    if (ctx->tune_sweep_grid > 0 && ctx->tune_sweep_grid < ctx->sweep_grid + ctx->sweep_grid / 4) {
        const int64_t groups = (ctx->N + 3) / 4;
        ctx->sweep_grid = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->tune_sweep_grid, groups));  // (an override is taken literally)
    }
    // inside the tick kernel (csmp_omp_batch) the sweep shares the CUs with the append stages of two other signals
    ctx->tick_grid = balanced_grid(ctx->N, col_bytes >= 8192 ? (int64_t)cus * 11 / 16 : (int64_t)cus * 3);
    ctx->sweep_grid = std::min(ctx->sweep_grid, maxgrid);
    ctx->tick_grid = std::min(ctx->tick_grid, maxgrid);
    ctx->sweep_pcap = 0;
    if (bestU) {
        ctx->sweep_U = bestU;
        ctx->sweep_KP = best_pad * rows;
    } else {
        // phases: beside the image the LDS holds the partial sums of every wave's columns (sweep_body_ph), sized for the smallest
        // grid the sweep is launched on (the tick kernel's, or an override's)
        int ming = std::min(ctx->sweep_grid, ctx->tick_grid);
        if (ctx->tick_nblk > 0) ming = std::min(ming, ctx->tick_nblk);
        const int64_t pcap64 = (ctx->N + (int64_t)ming * 4 - 1) / ((int64_t)ming * 4);
        const int64_t spare = (int64_t)(lds_cap / sizeof(double)) - 48 - 4 * pcap64;
        const int ur = 8 * rows;
        if (spare < ur) return fail(ctx, CSMP_ERANGE, "set_dictionary: too many columns per sweep workgroup for a residual staged in phases");
        ctx->sweep_pcap = (int)pcap64;
        int kp_max = (int)(spare / ur) * ur;
        if (ctx->tune_phase_rows > 0) kp_max = std::max(ur, std::min(kp_max, (ctx->tune_phase_rows / ur) * ur));
        const int nph = (ctx->Mv + kp_max - 1) / kp_max;
        const int per = (ctx->Mv + nph - 1) / nph;
        ctx->sweep_U = 8;
        ctx->sweep_ph = true;
        ctx->sweep_KP = ((per + ur - 1) / ur) * ur;
    }
    // columns handed out at run time (sweep_body_dyn): only on csmp_tune(CSMP_TUNE_SWEEP_DYN, 1) -- measured 1-3 % slower than the static split
    ctx->sweep_dyn = !ctx->sweep_ph && ctx->tune_sweep_dyn >= 1;
    if (ctx->sweep_grid > kClaimMaxWgs || std::max(ctx->tick_grid, ctx->tick_nblk) > kClaimMaxWgs) ctx->sweep_dyn = false;  // (one counter per workgroup)
    ctx->sweep_lds = ctx->sweep_ph ? sweep_ph_lds_bytes(ctx->sweep_KP, ctx->sweep_pcap)
                                   : ctx->sweep_dyn ? sweep_dyn_lds_bytes(ctx->sweep_KP) : sweep_gen_lds_bytes(ctx->sweep_KP);
    ctx->sweep_lds_req = (size_t)ctx->tune_sweep_lds_kib * 1024;
    // short columns: the stand-alone sweep takes several columns per unit (k_sweep_short); the tick kernel keeps the one-column body
    // (the same bits).  Units of eight loads: eight columns of one chunk (two sets of four), four of two chunks, two of three or four
    // chunks.  csmp_tune(CSMP_TUNE_SWEEP_SHORT, 1): never.
    ctx->short_cpu = ctx->short_nch = 0;
    if (ctx->tune_sweep_short != 1 && nchunk <= 4 && ctx->N >= 8) {
        ctx->short_nch = nchunk <= 2 ? nchunk : 4;
        ctx->short_cpu = nchunk <= 2 ? 4 : 2;
        ctx->short_KP = ctx->short_nch * rows;
        ctx->short_lds = sweep_gen_lds_bytes(ctx->short_KP);
    }
    return CSMP_OK;
}

// the context lets go of its dictionary and of everything sized by it
static int dict_forget(csmp_ctx* ctx) {
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
    return CSMP_OK;
}

// A call that fails after the previous dictionary has been let go leaves the context with NO dictionary (dA == NULL, M = N = 0) --
// never with the new dimensions over nothing: the entry points that test dA then say "no dictionary set".
static int dict_failed(csmp_ctx* ctx, int rc) {
    if (rc != CSMP_OK) {
        const std::string keep = ctx->err;
        (void)hipGetLastError();
        dict_release(ctx);
        ctx->M = ctx->N = ctx->ld = 0;
        ctx->Mv = 0;
        ctx->err = keep;
    }
    return rc;
}
static int set_dictionary_impl(csmp_ctx* ctx, const void* A, int64_t M, int64_t N, int64_t ldA, int dtype, int loc);
extern "C" int csmp_set_dictionary(csmp_ctx* ctx, const void* A, int64_t M, int64_t N, int64_t ldA, int dtype, int loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!A || M < 1 || N < 1 || ldA < M) return fail(ctx, CSMP_EDIM, "set_dictionary: need A != NULL, M,N >= 1, ldA >= M");
    if (dtype != CSMP_F32 && dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "set_dictionary: dtype must be CSMP_F32 or CSMP_F64");
    if (M > (int64_t)1 << 30 || N > (int64_t)1 << 31) return fail(ctx, CSMP_ERANGE, "set_dictionary: M or N too large");
    if (loc != CSMP_HOST && loc != CSMP_DEVICE && loc != CSMP_HOST_STREAMED)
        return fail(ctx, CSMP_EINVAL, "set_dictionary: loc must be CSMP_HOST, CSMP_DEVICE or CSMP_HOST_STREAMED");
    return dict_failed(ctx, set_dictionary_impl(ctx, A, M, N, ldA, dtype, loc));
}
static int set_dictionary_impl(csmp_ctx* ctx, const void* A, int64_t M, int64_t N, int64_t ldA, int dtype, int loc) {
    CHECK(dict_forget(ctx));
    const size_t es = dtype == CSMP_F32 ? 4 : 8;
    const int vec = 16 / (int)es;
    const bool aligned = ((uintptr_t)A % 16 == 0) && (M % vec == 0) && (ldA % vec == 0);
    const bool borrow = loc == CSMP_DEVICE && aligned;
    ctx->dtype = dtype;
    ctx->M = M;
    ctx->N = N;
    ctx->col_offset = 0;
    if (loc == CSMP_HOST_STREAMED) {
        // The dictionary stays in HOST memory, page-locked and mapped into the device's address space; ctx->dA is its device
        // alias, and every kernel that reads A (the sweeps, the column gathers of the appends) reads it over the host link.
        // An aligned array is registered where it lies (no second copy in host memory: the point of the mode is a dictionary
        // that does not fit anywhere twice); anything else gets a padded page-locked copy.
        void* hostp = const_cast<void*>(A);
        int kind = 2;
        int64_t ld = ldA;
        if (aligned) {
            // (a column-major view with ldA > M owns ldA (N - 1) + M elements, not ldA N: registering more could run past the allocation)
            const hipError_t e = hipHostRegister(hostp, ((size_t)ldA * (size_t)(N - 1) + (size_t)M) * es, hipHostRegisterMapped | hipHostRegisterPortable);
            if (e == hipErrorHostMemoryAlreadyRegistered) {
                (void)hipGetLastError();
                kind = -1;  // (page-locked by the caller already -- hipHostMalloc, or registered: nothing to undo later)
            } else {
                HIPCHECK(e);
            }
        } else {
            ld = ((M + vec - 1) / vec) * vec;
            HIPCHECK(hipHostMalloc(&hostp, (size_t)ld * (size_t)N * es, hipHostMallocMapped | hipHostMallocPortable));
            kind = 1;
            memset(hostp, 0, (size_t)ld * (size_t)N * es);
            for (int64_t c = 0; c < N; ++c) memcpy((char*)hostp + (size_t)c * (size_t)ld * es, (const char*)A + (size_t)c * (size_t)ldA * es, (size_t)M * es);
        }
        void* dalias = nullptr;
        const hipError_t e2 = hipHostGetDevicePointer(&dalias, hostp, 0);
        if (e2 != hipSuccess) {
            if (kind == 2) (void)hipHostUnregister(hostp);
            if (kind == 1) (void)hipHostFree(hostp);
            HIPCHECK(e2);
        }
        ctx->dA = dalias;
        ctx->ld = ld;
        ctx->Mv = aligned ? (int)M : (int)ld;
        if (kind >= 1) ctx->share = new DictShare{hostp, 1, kind};
        ctx->streamed = true;
        return configure_sweep(ctx);
    }
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

// ------------------------------------------------------------------------------------------ dictionary files
// SURVEY §8(f-4), second half: a dictionary on disk, and one larger than HBM.  The file is the array as the kernels want it --
// a 64-byte header, then N columns of ld elements each (ld = M rounded up to 16 bytes, the padding zero), little-endian IEEE:
//     bytes 0-7 "CSMPDICT" | u32 version = 1 | u32 dtype (CSMP_F32 / CSMP_F64) | i64 M | i64 N | i64 ld | 24 bytes zero
// csmp_set_dictionary_file reads it straight to where it will live: HBM (CSMP_DEVICE: through two page-locked staging buffers,
// the read of one chunk under the upload of the other) or page-locked host memory mapped into the device (CSMP_HOST_STREAMED:
// every sweep then crosses the host link -- ~50 GB/s against HBM's 6.6 TB/s; the mode for a dictionary that does not fit).
// (DictFileHeader, csmp_dictionary_file_write / _info and dict_file_open: host/hostonly.hpp -- no context, no HIP)
static int set_dictionary_file_impl(csmp_ctx* ctx, FILE* f, const DictFileHeader& h, int loc);
extern "C" int csmp_set_dictionary_file(csmp_ctx* ctx, const char* path, int loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!path) return fail(ctx, CSMP_EINVAL, "set_dictionary_file: path == NULL");
    if (loc != CSMP_DEVICE && loc != CSMP_HOST_STREAMED)
        return fail(ctx, CSMP_EINVAL, "set_dictionary_file: loc must be CSMP_DEVICE (resident in HBM) or CSMP_HOST_STREAMED");
    FILE* f = nullptr;
    DictFileHeader h;
    if (dict_file_open(path, &f, &h) != CSMP_OK) return fail(ctx, CSMP_EIO, std::string("set_dictionary_file: cannot read a dictionary from ") + path);
    struct Closer {
        FILE* f;
        ~Closer() { if (f) fclose(f); }
    } closer{f};
    if (h.M > (int64_t)1 << 30 || h.N > (int64_t)1 << 31) return fail(ctx, CSMP_ERANGE, "set_dictionary_file: M or N too large");
    return dict_failed(ctx, set_dictionary_file_impl(ctx, f, h, loc));  // (a short file, no memory: the context ends with no dictionary)
}
static int set_dictionary_file_impl(csmp_ctx* ctx, FILE* f, const DictFileHeader& h, int loc) {
    const size_t es = h.dtype == (uint32_t)CSMP_F32 ? 4 : 8;
    const size_t total = (size_t)h.ld * (size_t)h.N * es;
    CHECK(dict_forget(ctx));
    ctx->dtype = (int)h.dtype;
    ctx->M = h.M;
    ctx->N = h.N;
    ctx->col_offset = 0;
    ctx->ld = h.ld;
    ctx->Mv = (int)h.ld;  // (the padding rows are zero in the file)
    if (loc == CSMP_HOST_STREAMED) {
        void* hostp = nullptr;
        HIPCHECK(hipHostMalloc(&hostp, total, hipHostMallocMapped | hipHostMallocPortable));
        size_t got = 0;
        while (got < total) {
            const size_t n = fread((char*)hostp + got, 1, std::min<size_t>(total - got, (size_t)64 << 20), f);
            if (n == 0) break;
            got += n;
        }
        void* dalias = nullptr;
        const hipError_t e2 = got == total ? hipHostGetDevicePointer(&dalias, hostp, 0) : hipSuccess;
        if (got != total || e2 != hipSuccess) {
            (void)hipHostFree(hostp);
            if (got != total) return fail(ctx, CSMP_EIO, "set_dictionary_file: file shorter than its header says");
            HIPCHECK(e2);
        }
        ctx->dA = dalias;
        ctx->share = new DictShare{hostp, 1, 1};
        ctx->streamed = true;
        return configure_sweep(ctx);
    }
    void* d = nullptr;
    HIPCHECK(hipMalloc(&d, total));
    ctx->dA = d;
    ctx->ownA = true;
    ctx->share = new DictShare{d, 1, 0};
    const size_t chunk = std::min<size_t>(total, (size_t)64 << 20);
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = CSMP_OK;
    auto cleanup = [&]() {
        for (int q = 0; q < 2; ++q) {
            if (stage[q]) (void)hipHostFree(stage[q]);
            if (done[q]) (void)hipEventDestroy(done[q]);
        }
    };
    for (int q = 0; q < 2 && rc == CSMP_OK; ++q) {
        if (hipHostMalloc(&stage[q], chunk, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&done[q], hipEventDisableTiming) != hipSuccess)
            rc = fail(ctx, CSMP_ENOMEM, "set_dictionary_file: no page-locked staging memory");
    }
    size_t off = 0;
    for (int q = 0; rc == CSMP_OK && off < total; q ^= 1) {
        if (hipEventSynchronize(done[q]) != hipSuccess) { rc = fail(ctx, CSMP_EHIP, "set_dictionary_file: staging event"); break; }  // (buffer q's previous upload)
        const size_t want = std::min(chunk, total - off);
        if (fread(stage[q], 1, want, f) != want) { rc = fail(ctx, CSMP_EIO, "set_dictionary_file: file shorter than its header says"); break; }
        if (hipMemcpyAsync((char*)d + off, stage[q], want, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(done[q], ctx->stream) != hipSuccess) { rc = fail(ctx, CSMP_EHIP, "set_dictionary_file: upload"); break; }
        off += want;
    }
    (void)hipStreamSynchronize(ctx->stream);
    cleanup();
    if (rc != CSMP_OK) {
        dict_release(ctx);
        return rc;
    }
    return configure_sweep(ctx);
}
