// csmp.hip -- host side of libcsmp.so: the C ABI of include/csmp.h over the gfx950 kernels in
// csmp_kernels.hpp.  One ctx = one GPU + one HIP stream.  A solve is a chain of asynchronous
// launches with all control state (support, stop flags) in device memory: the host never
// synchronises inside a solve, only when results are copied back.
#include "../../include/csmp.h"
#include "csmp_kernels.hpp"
#include "csmp_batched.hpp"
#include "csmp_block.hpp"
#include "csmp_forward.hpp"
#include "csmp_downdate.hpp"
#include "csmp_tinv.hpp"
#include "csmp_shard.hpp"
#include "csmp_gram.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace csmp;

static std::string g_create_err;
static constexpr int kRsEqCap = 4096;   // entries of the bucket list (exact ties beyond it: in-order scan)
static constexpr int kRsSettle = 256;   // a bucket this small ends the passes (k_rs_finish ranks it in LDS)

struct Solver {
    int kcap = 0, outcap = 0;
    int qcap = 0;  // capacity of the QR arrays: kcap, or 1 for a slot that so far served MP / sweep-only calls
    int64_t ldq = 0;
    int G = 0, Mpad = 0;
    double *b = nullptr, *r = nullptr, *cvec = nullptr, *pval = nullptr;
    int* pidx = nullptr;
    double *Q = nullptr, *R = nullptr, *z = nullptr, *W1 = nullptr, *P1 = nullptr, *P2 = nullptr, *P2s = nullptr, *P1s = nullptr;
    double *avec = nullptr, *vvec = nullptr, *coef = nullptr, *scal = nullptr;
    int *sel = nullptr, *cands = nullptr, *ncands = nullptr;
    // top-S selection scratch
    double *top_lv = nullptr, *cvals = nullptr;
    int *top_li = nullptr, *rs_gt = nullptr, *rs_eq = nullptr, *rs_work = nullptr;
    RsState* rs = nullptr;
    int top_nb = 0;
    DevState* st = nullptr;
    double* bstage = nullptr;  // Mpad doubles: host-uploaded b
    int64_t *out_idx = nullptr, *out_order = nullptr, *out_nnz = nullptr;
    double* out_val = nullptr;
    int algo = -1;
    bool begun = false;
    int jh = 0;          // host upper bound on the QR column count (appends launched since the last reset)
    bool capped = false; // an append was withheld: the support reached what the on-device QR append can hold (qr_max_cols)
    int jh_last = 0;     // jh used by the most recent k_qr1 stage (the matching k_qr2 stage reuses it)
    // multi-column append (csmp_block.hpp), allocated on first use
    double *Apan = nullptr, *Vpan = nullptr, *PB1 = nullptr, *W1b = nullptr, *PG = nullptr, *Gsum = nullptr;
    int* pan_atoms = nullptr;
    int blk_kcap = 0;
    int* sigflags = nullptr;  // per-signal stop flags of a batch (optimistic-chain verification)
    double *rho2 = nullptr, *dvec = nullptr;  // forward regression: OLS rescaling and δ² scores (N each), allocated on first use
    int fr_grid = 0;
    // column removal (csmp_downdate.hpp), allocated on first use
    double *R2 = nullptr, *Gdel = nullptr, *qdrop = nullptr, *qsave = nullptr, *bwd = nullptr, *bwd_coef = nullptr, *bwd_info = nullptr;
    int *delmeta = nullptr, *delpos = nullptr;
    // explicit inverse factor of the two-stage solvers (csmp_tinv.hpp)
    double *T = nullptr, *T2 = nullptr, *tpd = nullptr, *tpn = nullptr;
    int* tmeta = nullptr;
    int sigcap = 0;
    // whole-set least squares (csmp_gram.hpp), allocated on first use
    double *Gm = nullptr, *Gpart = nullptr, *gdiag = nullptr, *rpart = nullptr, *Dfac = nullptr;
    // the last bordered Gram matrix that was COMPUTED (before its factorisation), for the sets that are subsets of it: SP solves
    // on T = S + k new atoms and then on the k atoms of T it keeps -- the second system is a principal submatrix of the first
    double *Gkeep = nullptr, *gdkeep = nullptr, *rhs_part = nullptr, *rn2part = nullptr;
    int* kpos = nullptr;
    std::vector<int> keep_cols;
    int keep_n = 0, keep_np = 0;
    bool keep_valid = false;
    void* Acomp = nullptr;  // the set's columns, contiguous (np columns of Mv elements of the dictionary's type)
    int gram_np = 0, gram_split = 0;
    void* extcol = nullptr;  // column-sharded OMP (csmp_shard.hpp): the winning column of a step, Mv elements of the dictionary's type
};

// device state of the batched (MFMA-screened) path
struct Batch {
    __bf16* Ab = nullptr;  // dictionary as bf16 [Npad][Mk]
    bool ab_valid = false;
    int Mk = 0;
    int64_t Npad = 0;
    int n_atiles = 0;
    float* amax = nullptr;  // max |A_ij| (device scalar) for the screening error bound
    float amax_host = 0.f;
    float anorm_host = -1.f;  // max column 2-norm (the deterministic bound, CSMP_CERT=rigorous), computed on first use
    // per-batch buffers
    int Bcap = 0, kcap = 0, Mr = 0;
    __bf16* Rb = nullptr;
    double *r = nullptr, *b = nullptr, *T = nullptr, *Tt = nullptr, *z = nullptr;
    int* sel = nullptr;
    BState* bs = nullptr;
    BPick* pick = nullptr;   // k_b_pick -> k_b_append hand-off, one per signal
    double* Gm = nullptr;    // G = A'A (upper triangle of N x N), the option CSMP_OPT_BATCH_GRAM
    int64_t Ng = 0;
    bool gram_valid = false;
    float* cand_val = nullptr;
    int* cand_idx = nullptr;
    int64_t last_signals = 0, last_resolved = 0, last_uncertain = 0, last_illcond = 0;
    int last_mode = 0;  // screening kernel of the last batch (kScreen128 / kScreen256 / kScreenCo)
    int64_t last_screen_signals = 0;  // signal columns of one (timed) screening launch of the last batch
    int last_streams = 1;
};

// A library-owned copy of the dictionary is shared by the context that uploaded it and its clones (csmp_clone): the memory
// lives until the LAST of them lets go (csmp_destroy, or csmp_set_dictionary replacing it), so a functor never sweeps freed
// memory.  A BORROWED device pointer (zero-copy) stays the caller's to keep alive.
struct DictShare {
    void* p;
    int refs;
};

struct csmp_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipDeviceProp_t prop{};
    std::string err;
    // page-locked host buffers for the small transfers on the latency chains (slot 0: the signal going up, slot 1: results and
    // control words coming down): a copy from or to pageable memory is staged by the runtime and blocks the host for every piece
    void* pin[3] = {nullptr, nullptr, nullptr};  // (slot 2: column lists going up)
    size_t pin_bytes[3] = {0, 0, 0};
    // dictionary
    void* dA = nullptr;
    bool ownA = false;
    struct DictShare* share = nullptr;  // library-owned dictionary memory, shared with the clones (reference counted)
    csmp_ctx* twins[3] = {nullptr, nullptr, nullptr};  // clones on their own streams: the other solves in flight of csmp_gomp_batch / csmp_sp_batch
    int opt_in_flight = 3;        // CSMP_OPT_SOLVES_IN_FLIGHT (csmp_sp_batch)
    hipEvent_t ev_twin = nullptr;
    int dtype = CSMP_F32;
    int64_t M = 0, N = 0, ld = 0;
    int64_t col_offset = 0;  // global index of local column 0 (column-sharded OMP; 0 otherwise)
    int Mv = 0;  // M rounded up to the 16-byte vector (zero rows in our own copy)
    int sweep_grid = 0, sweep_U = 1;
    int tick_U = 1;  // load-block size of the sweep inside the tick kernel (8 where it tiles, else sweep_U)
    bool sweep_full = false, sweep_nt = false;
    bool force_reorth = false;  // CSMP_OPT_FORCE_REORTH (test switch): always run the second Gram-Schmidt pass
    // options (csmp_set_option, include/csmp.h)
    int opt_batch_cert = 0;        // CSMP_OPT_BATCH_CERT: 0 statistical, 1 rigorous
    int opt_batch_gram = 0;        // CSMP_OPT_BATCH_GRAM: resident G = A'A for csmp_omp_batch_mfma
    int opt_batch_window = 0;      // CSMP_OPT_BATCH_WINDOW: rescoring window capacity, 0 = default
    bool opt_ls_gram = true;       // CSMP_OPT_LS_GRAM: whole-set least squares by Gram + Cholesky
    bool opt_ls_gram_reuse = true; // CSMP_OPT_LS_GRAM_REUSE
    int opt_twostage_update = 0;   // CSMP_OPT_TWOSTAGE_UPDATE: 0 explicit inverse, 1 Givens down-date of R, 2 refactorise
    size_t sweep_lds = 0;
    Solver s;        // the ACTIVE solver slot (see activate_slot)
    Solver park[3];  // parked slots (park[active] is unused): three signals are pipelined in csmp_omp_batch
    int active = 0;
    bool pipeline = true;
    int tick_wg_per_cu = 2;  // sweep workgroups per CU inside the tick kernel (CSMP_TICK_WGS)
    bool tick_pf = true;     // software-pipelined sweep inside the tick kernel (CSMP_TICK_PF)
    int tick_nblk = 0;       // absolute override of the sweep workgroup count (CSMP_TICK_NBLK), 0 = per-CU rule
    bool tick_sweep_first = false;  // dispatch the sweep workgroups ahead of the append stages (CSMP_TICK_ORDER=1)
    Batch bt;
    // profiling
    bool prof = false;
    int prof_every = 1;       // time every n-th sweep launch only (an event pair costs a few us of stream time)
    int64_t prof_count = 0;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    int64_t prof_n = 0;
    double prof_ms = 0.0;
    // second event pool: the batched path's screening GEMM
    std::vector<hipEvent_t> ev2;
    size_t ev2_used = 0;
    int64_t prof2_n = 0;
    double prof2_ms = 0.0;
};

#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char buf_[512];                                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                       \
            ctx->err = buf_;                                                                    \
            return CSMP_EHIP;                                                                   \
        }                                                                                       \
    } while (0)

#define CHECK(expr)                  \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != CSMP_OK) return rc_; \
    } while (0)

// Tuning switches exist only in the experiments build (`make experiments`, tools/probe_*.py); the product library reads no
// environment variable: every behavioural choice is an argument or a csmp_set_option key (include/csmp.h).
static const char* tune_env(const char* name) {
#ifdef CSMP_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

static int fail(csmp_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}

template <typename T>
static int dmalloc(csmp_ctx* ctx, T** p, size_t n) {
    HIPCHECK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    return CSMP_OK;
}
template <typename T>
static void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}
// device temporary of one call: released on every return path (hipFree waits for the work that uses it)
struct DevTmp {
    void* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 1)); }
};

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
    if (const char* tw = tune_env("CSMP_TICK_WGS")) ctx->tick_wg_per_cu = std::max(1, atoi(tw));
    if (const char* tn = tune_env("CSMP_TICK_NBLK")) ctx->tick_nblk = std::max(0, atoi(tn));
    if (const char* to = tune_env("CSMP_TICK_ORDER")) ctx->tick_sweep_first = atoi(to) != 0;
    if (const char* tp = tune_env("CSMP_TICK_PF")) ctx->tick_pf = tp[0] != '0';
    *out = ctx;
    return CSMP_OK;
}

static void batch_free(Batch& b, bool keep_dict) {
    dfree(b.Rb); dfree(b.r); dfree(b.b); dfree(b.T); dfree(b.Tt); dfree(b.z); dfree(b.sel); dfree(b.bs);
    dfree(b.cand_val); dfree(b.cand_idx); dfree(b.pick);
    b.Bcap = b.kcap = 0;
    if (!keep_dict) {
        dfree(b.Ab);
        dfree(b.amax);
        dfree(b.Gm);
        b.gram_valid = false;
        b.ab_valid = false;
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
    dfree(s.b); dfree(s.r); dfree(s.cvec); dfree(s.pval); dfree(s.pidx); dfree(s.Q); dfree(s.R); dfree(s.z);
    dfree(s.W1); dfree(s.P1); dfree(s.P2); dfree(s.P2s); dfree(s.P1s); dfree(s.avec); dfree(s.vvec); dfree(s.coef);
    dfree(s.scal); dfree(s.sel); dfree(s.cands); dfree(s.ncands); dfree(s.st); dfree(s.bstage);
    dfree(s.top_lv); dfree(s.cvals); dfree(s.top_li); dfree(s.rs_gt); dfree(s.rs_eq); dfree(s.rs_work); dfree(s.rs);
    dfree(s.out_idx); dfree(s.out_order); dfree(s.out_nnz); dfree(s.out_val); dfree(s.sigflags);
    dfree(s.Apan); dfree(s.Vpan); dfree(s.PB1); dfree(s.W1b); dfree(s.PG); dfree(s.Gsum); dfree(s.pan_atoms);
    dfree(s.rho2); dfree(s.dvec);
    dfree(s.R2); dfree(s.Gdel); dfree(s.qdrop); dfree(s.qsave); dfree(s.bwd); dfree(s.bwd_coef); dfree(s.bwd_info); dfree(s.delmeta); dfree(s.delpos);
    dfree(s.T); dfree(s.T2); dfree(s.tpd); dfree(s.tpn); dfree(s.tmeta); dfree(s.extcol);
    dfree(s.Gm); dfree(s.Dfac); dfree(s.Gpart); dfree(s.gdiag); dfree(s.rpart); dfree(s.Acomp); dfree(s.Gkeep); dfree(s.gdkeep); dfree(s.kpos); dfree(s.rhs_part); dfree(s.rn2part);
    s = Solver();
}

static void dict_release(csmp_ctx* ctx) {
    if (ctx->share && --ctx->share->refs == 0) {
        (void)hipFree(ctx->share->p);
        delete ctx->share;
    }
    ctx->share = nullptr;
    ctx->dA = nullptr;
    ctx->ownA = false;
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
    for (auto& t : ctx->twins) {
        if (t) (void)csmp_destroy(t);
        t = nullptr;
    }
    if (ctx->ev_twin) (void)hipEventDestroy(ctx->ev_twin);
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
        case CSMP_OPT_TWOSTAGE_UPDATE: *lo = 0; *hi = 2; return &ctx->opt_twostage_update;
        case CSMP_OPT_SOLVES_IN_FLIGHT: *lo = 1; *hi = 4; return &ctx->opt_in_flight;
        default: return nullptr;
    }
}
static bool* opt_flag(csmp_ctx* ctx, int key) {
    switch (key) {
        case CSMP_OPT_PIPELINE: return &ctx->pipeline;
        case CSMP_OPT_FORCE_REORTH: return &ctx->force_reorth;
        case CSMP_OPT_LS_GRAM: return &ctx->opt_ls_gram;
        case CSMP_OPT_LS_GRAM_REUSE: return &ctx->opt_ls_gram_reuse;
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

#ifdef CSMP_EXPERIMENTS  // kernel variants kept only for the tuning probes (tools/probe_sweep*.py, `make experiments`)
template <typename TA, typename TACC>
static hipError_t sweep_dispatch(csmp_ctx* ctx, int U, bool full, bool nt, int grid, size_t lds, const double* r,
                                 double eps, int check_eps, int skipmask) {
    if (!full) return sweep_launch_t<TA, TACC, 1, false, false>(ctx, grid, lds, r, eps, check_eps, skipmask);
    if (U == 4) return nt ? sweep_launch_t<TA, TACC, 4, true, true>(ctx, grid, lds, r, eps, check_eps, skipmask)
                          : sweep_launch_t<TA, TACC, 4, true, false>(ctx, grid, lds, r, eps, check_eps, skipmask);
    if (U == 2) return nt ? sweep_launch_t<TA, TACC, 2, true, true>(ctx, grid, lds, r, eps, check_eps, skipmask)
                          : sweep_launch_t<TA, TACC, 2, true, false>(ctx, grid, lds, r, eps, check_eps, skipmask);
    return nt ? sweep_launch_t<TA, TACC, 1, true, true>(ctx, grid, lds, r, eps, check_eps, skipmask)
              : sweep_launch_t<TA, TACC, 1, true, false>(ctx, grid, lds, r, eps, check_eps, skipmask);
}

#endif

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

#ifdef CSMP_EXPERIMENTS
// one sweep with the product configuration (or an explicit experimental one)
static int launch_sweep_cfg(csmp_ctx* ctx, const double* r, double eps, int check_eps, int skipmask, int U, bool nt,
                            bool f32acc, int grid) {
    const int vec = ctx->dtype == CSMP_F32 ? 4 : 2;
    const int rows = kWave * vec;
    bool full = (ctx->Mv % (rows * U)) == 0;
    if (!full && (ctx->Mv % rows) == 0) {  // fall back to the largest U that divides
        for (int u : {2, 1})
            if (u < U && ctx->Mv % (rows * u) == 0) {
                U = u;
                full = true;
                break;
            }
    }
    if (ctx->prof) CHECK(prof_mark(ctx));
    hipError_t e;
    if (ctx->dtype == CSMP_F32)
        e = f32acc ? sweep_dispatch<float, float>(ctx, U, full, nt, grid, ctx->sweep_lds, r, eps, check_eps, skipmask)
                   : sweep_dispatch<float, double>(ctx, U, full, nt, grid, ctx->sweep_lds, r, eps, check_eps, skipmask);
    else
        e = sweep_dispatch<double, double>(ctx, U, full, nt, grid, ctx->sweep_lds, r, eps, check_eps, skipmask);
    HIPCHECK(e);
    if (ctx->prof) CHECK(prof_mark(ctx));
    return CSMP_OK;
}

#endif

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
    const char* su = tune_env("CSMP_SWEEP_U");  // tuning knob: cap the load-block size
    const int umax = su ? atoi(su) : 16;
    if (ctx->sweep_full)
        for (int u : {16, 8, 4, 2})
            if (u <= umax && nchunk % u == 0) {
                ctx->sweep_U = u;
                break;
            }
    ctx->tick_U = ctx->sweep_U;
    if (ctx->sweep_full && ctx->sweep_U == 16 && !tune_env("CSMP_TICK_U16")) ctx->tick_U = 8;  // (16 | nchunk implies 8 | nchunk)
    ctx->sweep_nt = true;
    int per_cu = ctx->sweep_U == 16 ? 3 : 4;
    per_cu = (int)std::min<size_t>((size_t)per_cu, (160 * 1024) / ctx->sweep_lds);
    if (per_cu < 1) per_cu = 1;
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    int64_t grid = (int64_t)ctx->prop.multiProcessorCount * per_cu;
    if (ctx->sweep_full && ctx->sweep_U == 16) grid = (int64_t)ctx->prop.multiProcessorCount * 3 / 4;  // pipelined kernel
    if (ctx->sweep_full && ctx->sweep_U == 8) grid = (int64_t)ctx->prop.multiProcessorCount;
    if (const char* sn = tune_env("CSMP_SWEEP_NBLK")) grid = std::max(1, atoi(sn));  // tuning knob
    if (const char* sp = tune_env("CSMP_SWEEP_LDS"))  // tuning knob: request at least this much LDS per workgroup
        ctx->sweep_lds = std::max(ctx->sweep_lds, (size_t)atoi(sp));
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
// leading dimension and scan one support in one workgroup: at most kDelMaxCols columns.  A slot that an earlier
// call grew beyond that is rebuilt at the size this call needs.
static int solver_fit_for_removal(csmp_ctx* ctx, int kcap) {
    Solver& s = ctx->s;
    if (s.kcap > kDelMaxCols && kcap <= kDelMaxCols) {
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        solver_free(s);
    }
    return CSMP_OK;
}

// Small results coming back on a latency chain: every piece is copied into the page-locked slot (truly asynchronous, back
// to back), ONE wait, then the pieces are handed to their host destinations.  (A copy straight into pageable memory -- a stack
// variable, a std::vector -- is staged by the runtime and blocks the host once per piece.)
struct PinFetch {
    csmp_ctx* ctx;
    char* base = nullptr;
    size_t used = 0;
    struct Out { void* dst; size_t off, bytes; } outs[8];
    int nout = 0;
    explicit PinFetch(csmp_ctx* c) : ctx(c) {}
    int begin(size_t total) {
        void* pv = nullptr;
        CHECK(pin_get(ctx, 1, total + 64, &pv));
        base = (char*)pv;
        used = 0;
        nout = 0;
        return CSMP_OK;
    }
    int add(void* dst, const void* dev, size_t bytes) {
        const size_t off = (used + 7) / 8 * 8;
        HIPCHECK(hipMemcpyAsync(base + off, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        outs[nout++] = {dst, off, bytes};
        used = off + bytes;
        return CSMP_OK;
    }
    int wait() {
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
    s.keep_valid = false;  // (the kept Gram matrix carries A_S'b of the previous b)
    hipLaunchKernelGGL(k_init<double>, dim3(s.Mpad / 256), dim3(256), 0, ctx->stream, (const double*)s.bstage, M, s.Mpad, s.b, s.r, s.st);
    HIPCHECK(hipGetLastError());
    s.jh = 0;
    s.capped = false;
    return CSMP_OK;
}

template <typename TB>
static int init_from_device_t(csmp_ctx* ctx, const TB* col) {
    Solver& s = ctx->s;
    s.keep_valid = false;
    hipLaunchKernelGGL(k_init<TB>, dim3(s.Mpad / 256), dim3(256), 0, ctx->stream, col, (int)ctx->M, s.Mpad, s.b, s.r, s.st);
    HIPCHECK(hipGetLastError());
    s.jh = 0;
    s.capped = false;
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
        // The append kernels keep five support-length vectors in LDS: about 3900 columns.  A solve that gets there (the
        // reference's defaults k = size(A,1) at M = 4096 with a residual test that never fires) STOPS there: the step is
        // withheld, the solution reached so far stays valid, and the driver reports CSMP_WCAPACITY / CSMP_STOP_CAPACITY.
        s.capped = true;
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
                       jpad, ctx->force_reorth ? 1 : 0, jh, optimistic ? 1 : 0);
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
    if (s.kcap > 256 && !tune_env("CSMP_FINISH_W") && !tune_env("CSMP_FINISH_B")) {
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
    if (s.kcap > 256 && !tune_env("CSMP_FINISH_W")) {  // blocked form in ONE workgroup: one memory round trip per 64 columns
        const size_t lds = (size_t)(s.kcap + 64) * sizeof(double) + (size_t)s.kcap * sizeof(int);
        if (lds > 64 * 1024) HIPCHECK(hipFuncSetAttribute((const void*)k_finish_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_finish_b, dim3(1), dim3(256), lds, ctx->stream, (const double*)s.R, (const double*)s.z,
                           (const int*)s.sel, (const DevState*)s.st, s.kcap, s.coef, d_idx, d_val, d_nnz, d_order, outcap, d_flag);
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

// ------------------------------------------------------------------------------------------ tick kernel (3 signals in flight)
template <typename TA>
static TickSweep<TA> tick_sweep_params(csmp_ctx* ctx, const Solver& s, double eps, int check_eps, int skipmask, int nblk, int active) {
    TickSweep<TA> p;
    p.A = (const TA*)ctx->dA; p.ld = ctx->ld; p.Mv = ctx->Mv; p.N = ctx->N;
    p.r = s.r; p.cvec = s.cvec; p.pval = s.pval; p.pidx = s.pidx; p.st = s.st;
    p.eps = eps; p.check_eps = check_eps; p.skipmask = skipmask; p.nblk = nblk; p.active = active;
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
    p.kcap = s.kcap; p.jpad = qr_jpad(jh); p.force_reorth = ctx->force_reorth ? 1 : 0; p.jh = jh; p.optimistic = optimistic;
    p.active = active;
    return p;
}

template <typename TA, int U, bool PF, bool STEADY = false>
static hipError_t tick_launch_t(csmp_ctx* ctx, const TickSweep<TA>& sw, const TickQr1<TA>& q1, const TickQr2& q2, int G, size_t lds) {
    auto kern = k_tick<TA, U, PF, STEADY>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(2 * G + sw.nblk), dim3(kSweepThreads), lds, ctx->stream, sw, q1, q2, G, ctx->tick_sweep_first ? 1 : 0);
    return hipGetLastError();
}
// steady: all three stages of this tick are live (the launches the bench's roofline is quoted on)
template <typename TA>
static hipError_t tick_launch(csmp_ctx* ctx, const TickSweep<TA>& sw, const TickQr1<TA>& q1, const TickQr2& q2, int G, size_t lds, bool steady) {
    switch (ctx->tick_U) {
        case 16:
            if (!ctx->tick_pf) return tick_launch_t<TA, 16, false>(ctx, sw, q1, q2, G, lds);
            return steady ? tick_launch_t<TA, 16, true, true>(ctx, sw, q1, q2, G, lds) : tick_launch_t<TA, 16, true, false>(ctx, sw, q1, q2, G, lds);
        case 8:
            if (!ctx->tick_pf) return tick_launch_t<TA, 8, false>(ctx, sw, q1, q2, G, lds);
            return steady ? tick_launch_t<TA, 8, true, true>(ctx, sw, q1, q2, G, lds) : tick_launch_t<TA, 8, true, false>(ctx, sw, q1, q2, G, lds);
        case 4: return tick_launch_t<TA, 4, false>(ctx, sw, q1, q2, G, lds);
        case 2: return tick_launch_t<TA, 2, false>(ctx, sw, q1, q2, G, lds);
        default: return tick_launch_t<TA, 1, false>(ctx, sw, q1, q2, G, lds);
    }
}

// OMP for up to three signals (solver slots 0..2, already initialised with their b) advanced
// together: at tick n slot n%3 sweeps, slot (n-1)%3 runs its k_qr1 stage, slot (n-2)%3 its k_qr2
// stage.  k steps per signal = 3k+2 ticks.  present[q] == false leaves slot q idle.
template <typename TA>
static int omp_ticks(csmp_ctx* ctx, const bool present[3], int64_t k, double eps, bool optimistic) {
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    activate_slot(ctx, 0);
    Solver* sl[3] = {&ctx->s, &ctx->park[1], &ctx->park[2]};
    const int G = sl[0]->G;
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    // Measured at 4096 x 65536 f32: 8-chunk load blocks on ONE workgroup per CU (the append stages of the other two
    // signals share those CUs) 160.4 us per tick; 16-chunk blocks on 176 workgroups (11/12 of the stand-alone sweep's
    // optimum of 192) 162.6 us.
    const int64_t auto_nblk = ctx->tick_U == 8 ? (int64_t)ctx->prop.multiProcessorCount
                              : ctx->tick_U == 16 ? (int64_t)ctx->sweep_grid * 11 / 12
                                                  : (int64_t)ctx->prop.multiProcessorCount * ctx->tick_wg_per_cu;
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->tick_nblk > 0 ? ctx->tick_nblk : auto_nblk, groups));
    if (k > qr_max_cols()) {  // the appends stop at the support the QR kernels can hold (see launch_append): CSMP_WCAPACITY
        k = qr_max_cols();
        for (int q = 0; q < 3; ++q)
            if (present[q]) sl[q]->capped = true;
    }
    const size_t lds = std::max(ctx->sweep_lds, qr_lds_bytes((int)std::min<int64_t>(k, sl[0]->kcap)));  // (jh never exceeds k here)
    for (int64_t n = 0; n < 3 * k + 2; ++n) {
        const int zs = (int)(n % 3), ys = (int)((n + 2) % 3), xs = (int)((n + 1) % 3);  // sweep, qr1, qr2 slots
        const int64_t tz = (n - zs) / 3, ty = (n - 1 - ys) / 3, tx = (n - 2 - xs) / 3;
        const bool az = present[zs] && n >= zs && tz < k;
        const bool ay = present[ys] && n >= 1 + ys && ty < k && (n - 1 - ys) % 3 == 0;
        const bool ax = present[xs] && n >= 2 + xs && tx < k && (n - 2 - xs) % 3 == 0;
        if (!az && !ay && !ax) continue;
        int jh1 = 0;
        if (ay) {
            jh1 = std::min(sl[ys]->jh, sl[ys]->kcap);
            sl[ys]->jh_last = jh1;
            if (sl[ys]->jh < sl[ys]->kcap) sl[ys]->jh += 1;
        }
        const auto sw = tick_sweep_params<TA>(ctx, *sl[zs], eps, tz > 0 ? 1 : 0, skip, nblk, az ? 1 : 0);
        const auto q1 = tick_qr1_params<TA>(ctx, *sl[ys], skip, nblk, jh1, ay ? 1 : 0);
        const auto q2 = tick_qr2_params(ctx, *sl[xs], sl[xs]->jh_last, optimistic ? 1 : 0, ax ? 1 : 0);
        const bool steady = az && ay && ax;
        const bool timed = steady && prof_pick(ctx);  // steady-state ticks only
        if (timed) CHECK(prof_mark(ctx));
        HIPCHECK(tick_launch<TA>(ctx, sw, q1, q2, G, lds, steady));
        if (timed) CHECK(prof_mark(ctx));
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
    // optimistic two-kernel append chain first; if any column failed the DGKS test (flagged on the
    // device, nothing committed) the solve is repeated with the second Gram-Schmidt pass enabled
    bool capacity_stop = false;
    for (int pass = 0; pass < 2; ++pass) {
        const bool optimistic = pass == 0 && !ctx->force_reorth;
        CHECK(upload_b(ctx, b, b_dtype));
        for (int64_t t = 0; t < k && !ctx->s.capped; ++t) {
            CHECK(omp_step(ctx, eps, t > 0, optimistic));
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
            capacity_stop = ctx->s.capped && !(hs.done & (STOP_EPS | STOP_STAG | STOP_FULL));
            break;
        }
    }
    CHECK(download_result(ctx, ctx->s.outcap, idx, val, nnz, order));
    return capacity_stop ? CSMP_WCAPACITY : CSMP_OK;
}

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
        default: {
            auto kern = k_fr_update4<TA, U, FULL>;
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, ctx->stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, ctx->N,
                               ps.q1, ps.qstride, ps.s1, ctx->s.rho2);
            return hipGetLastError();
        }
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
        const char* fu = tune_env("CSMP_FR_U");  // tuning knob: cap the load-block size
        const int umax = fu ? atoi(fu) : 8;
        for (int u : {16, 8})
            if (u <= umax && nchunk % u == 0) {
                U = u;
                full = true;
                break;
            }
    }
    lds = fr_sweep_lds_bytes(ctx->Mv, vec, U, nq);
    const int cus = ctx->prop.multiProcessorCount;
    int64_t g = U == 16 ? (int64_t)cus * 3 / 4 : (int64_t)cus;  // as the OMP sweep (configure_sweep)
    if (const char* sn = tune_env("CSMP_FR_NBLK")) g = std::max(1, atoi(sn));  // tuning knob
    const int64_t groups = (ctx->N + (kSweepThreads / kWave) - 1) / (kSweepThreads / kWave);
    grid = (int)std::max<int64_t>(1, std::min<int64_t>(g, groups));
}

static int fr_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    int U; bool full; size_t lds;
    fr_config(ctx, 1, U, full, lds, s.fr_grid);
    if (lds > 160 * 1024 - 512) return fail(ctx, CSMP_ERANGE, "fr: M too large (r and q must both fit the 160 KiB LDS)");
    if (!s.rho2) CHECK(dmalloc(ctx, &s.rho2, (size_t)ctx->N));
    if (!s.dvec) CHECK(dmalloc(ctx, &s.dvec, (size_t)ctx->N));
    return CSMP_OK;
}

// forward_δ! + the residual-norm guard of forward_step! (src/forward.jl:59-61,75-82)
static int launch_fr_pass(csmp_ctx* ctx, const FrPass& ps, double max_eps, int skipmask) {
    int U, grid; bool full; size_t lds;
    fr_config(ctx, ps.nq, U, full, lds, grid);
    if (lds > 160 * 1024 - 512) return fail(ctx, CSMP_ERANGE, "forward-regression sweep: M too large for the LDS images");
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
template <typename TA>
static int fr_ticks(csmp_ctx* ctx, const bool present[3], int64_t k, double max_eps, double min_d2, bool optimistic) {
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL | STOP_REORTH;
    activate_slot(ctx, 0);
    Solver* sl[3] = {&ctx->s, &ctx->park[1], &ctx->park[2]};
    const int G = sl[0]->G;
    int U, grid; bool full; size_t flds;
    fr_config(ctx, 1, U, full, flds, grid);
    int nblk = grid;
    if (const char* tn = tune_env("CSMP_FR_TICK_NBLK")) nblk = std::max(1, atoi(tn));
    if (k > qr_max_cols()) {  // (as omp_ticks)
        k = qr_max_cols();
        for (int q = 0; q < 3; ++q)
            if (present[q]) sl[q]->capped = true;
    }
    const size_t lds = std::max(flds, qr_lds_bytes((int)std::min<int64_t>(k, sl[0]->kcap)));
    for (int64_t n = 0; n < 3 * k + 2; ++n) {
        const int zs = (int)(n % 3), ys = (int)((n + 2) % 3), xs = (int)((n + 1) % 3);
        const int64_t tz = (n - zs) / 3, ty = (n - 1 - ys) / 3, tx = (n - 2 - xs) / 3;
        const bool az = present[zs] && n >= zs && tz < k;
        const bool ay = present[ys] && n >= 1 + ys && ty < k && (n - 1 - ys) % 3 == 0;
        const bool ax = present[xs] && n >= 2 + xs && tx < k && (n - 2 - xs) % 3 == 0;
        if (!az && !ay && !ax) continue;
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
        sw.sel = z.sel; sw.st = z.st; sw.max_eps = max_eps; sw.skipmask = skip; sw.nblk = nblk; sw.active = az ? 1 : 0;
        auto q1 = tick_qr1_params<TA>(ctx, *sl[ys], skip, nblk, jh1, ay ? 1 : 0);
        q1.mode = 3;
        const auto q2 = tick_qr2_params(ctx, *sl[xs], sl[xs]->jh_last, optimistic ? 1 : 0, ax ? 1 : 0);
        const bool timed = az && ay && ax && prof_pick(ctx);
        if (timed) CHECK(prof_mark(ctx));
        hipError_t e;
        if (U == 16)
            e = tz == 0 ? tick_fr_launch_t<TA, 16, -1>(ctx, sw, q1, q2, G, lds, min_d2) : tick_fr_launch_t<TA, 16, 1>(ctx, sw, q1, q2, G, lds, min_d2);
        else
            e = tz == 0 ? tick_fr_launch_t<TA, 8, -1>(ctx, sw, q1, q2, G, lds, min_d2) : tick_fr_launch_t<TA, 8, 1>(ctx, sw, q1, q2, G, lds, min_d2);
        HIPCHECK(e);
        if (timed) CHECK(prof_mark(ctx));
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
    bool capacity_stop = false;
    for (int pass = 0; pass < 2; ++pass) {  // optimistic append chain, repeated with re-orthogonalisation if flagged (see csmp_omp)
        const bool optimistic = pass == 0 && !ctx->force_reorth;
        CHECK(upload_b(ctx, b, b_dtype));
        for (int64_t t = 0; t < k && !ctx->s.capped; ++t) {
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
            capacity_stop = ctx->s.capped && !(hs.done & (STOP_EPS | STOP_STAG | STOP_FULL));
            break;
        }
    }
    CHECK(download_result(ctx, ctx->s.outcap, idx, val, nnz, order));
    return capacity_stop ? CSMP_WCAPACITY : CSMP_OK;
}

// omp (algo = CSMP_ALGO_OMP: p1 = eps) or fr (CSMP_ALGO_FR: p1 = max_eps, p2 = min_delta^2) for every column of B
static int batch_impl(csmp_ctx* ctx, int algo, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                      double eps, double p2, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    const bool isfr = algo == CSMP_ALGO_FR;
    bool capacity_stop = false;
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
    const bool opt = !ctx->force_reorth;
    bool pipe = ctx->pipeline && nsig >= 2 && ctx->sweep_full;
    if (isfr) {  // the tick kernel exists for the exact-tiling FR sweeps only
        int U, g; bool full; size_t l;
        fr_config(ctx, 1, U, full, l, g);
        pipe = pipe && full;
    }
    auto init_sig = [&](int64_t sgn) -> int {
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        return b_dtype == CSMP_F32 ? init_from_device_t<float>(ctx, (const float*)col)
                                   : init_from_device_t<double>(ctx, (const double*)col);
    };
    int64_t sgn = 0;
    if (pipe) {
        for (int q = 1; q < 3 && rc == CSMP_OK; ++q) {
            activate_slot(ctx, q);
            rc = solver_ensure(ctx, kc, (int)k);
            if (rc == CSMP_OK && isfr) rc = fr_ensure(ctx);
        }
        activate_slot(ctx, 0);
        for (; sgn < nsig && rc == CSMP_OK; sgn += 3) {
            bool present[3];
            for (int q = 0; q < 3 && rc == CSMP_OK; ++q) {
                present[q] = sgn + q < nsig;
                if (!present[q]) continue;
                activate_slot(ctx, q);
                rc = init_sig(sgn + q);
            }
            if (rc == CSMP_OK && isfr)
                rc = ctx->dtype == CSMP_F32 ? fr_ticks<float>(ctx, present, k, eps, p2, opt) : fr_ticks<double>(ctx, present, k, eps, p2, opt);
            else if (rc == CSMP_OK)
                rc = ctx->dtype == CSMP_F32 ? omp_ticks<float>(ctx, present, k, eps, opt) : omp_ticks<double>(ctx, present, k, eps, opt);
            for (int q = 0; q < 3 && rc == CSMP_OK; ++q) {
                if (!present[q]) continue;
                activate_slot(ctx, q);
                rc = launch_finish(ctx, d_idx + (sgn + q) * k, d_val + (sgn + q) * k, d_nnz + sgn + q, nullptr, (int)k, sigflags + sgn + q);
            }
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
        if (k > qr_max_cols())  // a signal that no stopping rule ended was cut at the QR append's capacity
            for (int64_t sgn = 0; sgn < nsig; ++sgn) capacity_stop |= !(hf[sgn] & (STOP_EPS | STOP_STAG | STOP_FULL));
    }
    if (out_loc == CSMP_HOST) {
        if (rc == CSMP_OK) {
            HIPCHECK(hipMemcpyAsync(idx, d_idx, (size_t)k * nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(val, d_val, (size_t)k * nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(nnz, d_nnz, (size_t)nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHECK(hipStreamSynchronize(ctx->stream));
    }
    return rc == CSMP_OK && capacity_stop ? CSMP_WCAPACITY : rc;
}

extern "C" int csmp_omp_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                              double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    return batch_impl(ctx, CSMP_ALGO_OMP, B, b_dtype, ldB, nsig, b_loc, k, eps, 0.0, idx, val, nnz, out_loc);
}

// fr(A, B[:,s], max_eps, min_delta, k) for every column of B: the forward-regression sweeps of three signals
// at a time are pipelined against one another's append stages exactly like csmp_omp_batch's
extern "C" int csmp_fr_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                             double max_eps, double min_delta, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
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
    CHECK(upload_b(ctx, b, b_dtype));
    if (nnz0 > 0) CHECK(upload_support(ctx, idx0, val0, nnz0));
    for (int64_t t = 0; t < k; ++t) CHECK(mp_step(ctx));
    return mp_collect(ctx, idx0, val0, nnz0, idx, val, nnz);
}

// ------------------------------------------------------------------------------------------ step-level API
extern "C" int csmp_solver_begin(csmp_ctx* ctx, int algo, const void* b, int b_dtype, int64_t kcap, const int64_t* idx0,
                                 const double* val0, int64_t nnz0) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || kcap < 1) return fail(ctx, CSMP_EINVAL, "solver_begin: bad arguments");
    if (algo != CSMP_ALGO_MP && algo != CSMP_ALGO_OMP && algo != CSMP_ALGO_GOMP && algo != CSMP_ALGO_FR) return fail(ctx, CSMP_EINVAL, "solver_begin: unknown algo");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int kc = algo == CSMP_ALGO_MP ? (int)kcap : (int)std::min<int64_t>(kcap, ctx->M);
    CHECK(solver_ensure(ctx, kc, kc, algo != CSMP_ALGO_MP));
    if (algo == CSMP_ALGO_FR) CHECK(fr_ensure(ctx));
    CHECK(upload_b(ctx, b, b_dtype));
    if (nnz0 > 0) {
        if (algo != CSMP_ALGO_MP) return fail(ctx, CSMP_EINVAL, "warm start is only defined for MP (src/matchingpursuit.jl:34)");
        CHECK(upload_support(ctx, idx0, val0, nnz0));
    }
    ctx->s.algo = algo;
    ctx->s.begun = true;
    return CSMP_OK;
}

static int gomp_update(csmp_ctx* ctx, int64_t l, double eps, int check_eps, int skipmask, bool block);

extern "C" int csmp_solver_step(csmp_ctx* ctx, int64_t l) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_step: no solver begun");
    HIPCHECK(hipSetDevice(ctx->dev));
    int rc = CSMP_OK;
    switch (ctx->s.algo) {
        case CSMP_ALGO_MP: return mp_step(ctx);
        case CSMP_ALGO_OMP: {
            // update!(P::OMP, x) alone: no eps logic (that belongs to the omp driver)
            CHECK(launch_sweep(ctx, ctx->s.r, 0.0, 0, STOP_FULL));
            rc = launch_append(ctx, 1, 0, STOP_FULL);
            break;
        }
        case CSMP_ALGO_FR: {
            // update!(P::FR, x): nnz < n guard, acquisition_index! = argmax δ², addindex!, solve (src/forward.jl:88-95)
            const int skip = STOP_FULL | STOP_STAG;  // (a step that found no finite score would otherwise downdate rho2 twice)
            CHECK(launch_fr_sweep(ctx, ctx->s.jh == 0, -HUGE_VAL, skip));
            rc = launch_append(ctx, 3, 0, skip, false, -1.0, ctx->s.fr_grid);
            break;
        }
        default: rc = gomp_update(ctx, l, 0.0, 0, STOP_FULL, false);
    }
    // a step the QR append could not take (support at its capacity): x is unchanged, the caller is told
    return rc == CSMP_OK && ctx->s.capped ? CSMP_WCAPACITY : rc;
}

// ------------------------------------------------------------------------------------------ shared dictionary
// A second context on the same GPU that BORROWS the resident dictionary of `src` (no copy): the independent
// P objects of the reference -- P1 = OMP(A, b1); P2 = OMP(A, b2) share A and nothing else
// (src/matchingpursuit.jl:44-60).  `src` must outlive the clone and keep its dictionary.
extern "C" int csmp_clone(csmp_ctx* src, csmp_ctx** out) {
    if (!src || !out) return CSMP_EINVAL;
    *out = nullptr;
    if (!src->dA) return fail(src, CSMP_ESTATE, "clone: no dictionary set (csmp_set_dictionary)");
    csmp_ctx* c = nullptr;
    const int rc = csmp_create(&c, src->dev);
    if (rc != CSMP_OK) {
        src->err = g_create_err;
        return rc;
    }
    c->dA = src->dA;
    c->ownA = false;
    c->share = src->share;  // (null for a borrowed device pointer: the caller keeps that alive)
    if (c->share) c->share->refs += 1;
    c->pipeline = src->pipeline;
    c->force_reorth = src->force_reorth;
    c->opt_batch_cert = src->opt_batch_cert;
    c->opt_batch_window = src->opt_batch_window;
    c->opt_ls_gram = src->opt_ls_gram;
    c->opt_ls_gram_reuse = src->opt_ls_gram_reuse;
    c->opt_twostage_update = src->opt_twostage_update;
    c->dtype = src->dtype;
    c->M = src->M;
    c->N = src->N;
    c->ld = src->ld;
    c->Mv = src->Mv;
    c->col_offset = src->col_offset;
    const int rc2 = configure_sweep(c);
    if (rc2 != CSMP_OK) {
        src->err = c->err;
        csmp_destroy(c);
        return rc2;
    }
    *out = c;
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ column-sharded OMP
// See csmp_shard.hpp.  The ctx holds columns [col_offset, col_offset + N) of the global dictionary; a solve is
// csmp_solver_begin(CSMP_ALGO_OMP) on every rank, then per step csmp_shard_sweep -> the caller's all_gather of
// one record per rank -> csmp_shard_append, and csmp_solver_state at the end (identical on every rank).
extern "C" int csmp_shard_config(csmp_ctx* ctx, int64_t col_offset) {
    if (!ctx) return CSMP_EINVAL;
    if (col_offset < 0 || col_offset + ctx->N > 0x7fffffff) return fail(ctx, CSMP_ERANGE, "shard_config: global column indices must fit 31 bits");
    ctx->col_offset = col_offset;
    return CSMP_OK;
}

extern "C" int64_t csmp_shard_record_bytes(const csmp_ctx* ctx) {
    if (!ctx || !ctx->dA) return 0;
    return (int64_t)shard_record_bytes(ctx->Mv, ctx->dtype == CSMP_F32 ? 4 : 8);
}

static int shard_ready(csmp_ctx* ctx, const char* who) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun || ctx->s.algo != CSMP_ALGO_OMP)
        return fail(ctx, CSMP_ESTATE, (std::string(who) + ": begin the solve with csmp_solver_begin(CSMP_ALGO_OMP)").c_str());
    return CSMP_OK;
}

// steps 1-2: argmaxinner!(P) over the local columns (+ the driver's residual test of the previous iteration,
// src/matchingpursuit.jl:79, when check_eps != 0) and the rank's record, written to DEVICE memory at rec_dev
extern "C" int csmp_shard_sweep(csmp_ctx* ctx, double eps, int check_eps, void* rec_dev) {
    CHECK(shard_ready(ctx, "shard_sweep"));
    if (!rec_dev || !(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "shard_sweep: rec_dev == NULL or eps < 0");
    HIPCHECK(hipSetDevice(ctx->dev));
    Solver& s = ctx->s;
    const int skip = STOP_EPS | STOP_STAG | STOP_FULL;
    CHECK(launch_sweep(ctx, s.r, eps, check_eps, skip));
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_shard_pack<float>, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.pval, (const int*)s.pidx,
                           ctx->sweep_grid, (const double*)s.cvec, (const float*)ctx->dA, ctx->ld, ctx->Mv, ctx->col_offset,
                           (const DevState*)s.st, skip, (char*)rec_dev);
    else
        hipLaunchKernelGGL(k_shard_pack<double>, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.pval, (const int*)s.pidx,
                           ctx->sweep_grid, (const double*)s.cvec, (const double*)ctx->dA, ctx->ld, ctx->Mv, ctx->col_offset,
                           (const DevState*)s.st, skip, (char*)rec_dev);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

// steps 4-5: global arg-max over the nrec gathered records (DEVICE memory, csmp_shard_record_bytes apart),
// then update!(P::OMP, x)'s guards, add_column! and the residual update on the winning column
extern "C" int csmp_shard_append(csmp_ctx* ctx, const void* recs_dev, int nrec) {
    CHECK(shard_ready(ctx, "shard_append"));
    if (!recs_dev || nrec < 1) return fail(ctx, CSMP_EINVAL, "shard_append: recs_dev == NULL or nrec < 1");
    HIPCHECK(hipSetDevice(ctx->dev));
    Solver& s = ctx->s;
    const size_t es = ctx->dtype == CSMP_F32 ? 4 : 8;
    if (!s.extcol) HIPCHECK(hipMalloc(&s.extcol, (size_t)ctx->Mv * es));
    const int64_t rb = (int64_t)shard_record_bytes(ctx->Mv, es);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_shard_pick<float>, dim3(1), dim3(256), 0, ctx->stream, (const char*)recs_dev, nrec, rb, ctx->Mv,
                           (float*)s.extcol, s.cands, s.ncands, s.st);
    else
        hipLaunchKernelGGL(k_shard_pick<double>, dim3(1), dim3(256), 0, ctx->stream, (const char*)recs_dev, nrec, rb, ctx->Mv,
                           (double*)s.extcol, s.cands, s.ncands, s.st);
    HIPCHECK(hipGetLastError());
    return launch_append(ctx, 4, 0, STOP_EPS | STOP_STAG | STOP_FULL, false, 0.0, 0, s.extcol);
}

// ------------------------------------------------------------------------------------------ signal sharding helpers
// The data path of the signal-sharded batch has ONE exchange: every rank's results.  These
// three host-side helpers fix its layout so that any host language can run it over its own collective
// (torch.distributed / RCCL here, MPI.jl from Julia): contiguous blocks of signals per rank, and per signal one row
// of 2k + 1 Float64 = [idx_0 .. idx_{k-1} | val_0 .. val_{k-1} | nnz] (indices are exact in Float64 below 2^53).
extern "C" int csmp_shard_range(int64_t nsig, int rank, int world, int64_t* lo, int64_t* hi) {
    if (nsig < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return CSMP_EINVAL;
    const int64_t base = nsig / world, extra = nsig % world;  // block sizes differ by at most one
    *lo = rank * base + std::min<int64_t>(rank, extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
    return CSMP_OK;
}
extern "C" int csmp_pack_results(const int64_t* idx, const double* val, const int64_t* nnz, int64_t k, int64_t nsig, double* packed) {
    if (!idx || !val || !nnz || !packed || k < 0 || nsig < 0) return CSMP_EINVAL;
    const int64_t w = 2 * k + 1;
    for (int64_t s = 0; s < nsig; ++s) {
        for (int64_t t = 0; t < k; ++t) {
            packed[s * w + t] = (double)idx[s * k + t];
            packed[s * w + k + t] = val[s * k + t];
        }
        packed[s * w + 2 * k] = (double)nnz[s];
    }
    return CSMP_OK;
}
extern "C" int csmp_unpack_results(const double* packed, int64_t k, int64_t nsig, int64_t* idx, double* val, int64_t* nnz) {
    if (!idx || !val || !nnz || !packed || k < 0 || nsig < 0) return CSMP_EINVAL;
    const int64_t w = 2 * k + 1;
    for (int64_t s = 0; s < nsig; ++s) {
        for (int64_t t = 0; t < k; ++t) {
            idx[s * k + t] = (int64_t)packed[s * w + t];
            val[s * k + t] = packed[s * w + k + t];
        }
        nnz[s] = (int64_t)packed[s * w + 2 * k];
    }
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ column removal
static int del_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.kcap > kDelMaxCols) return fail(ctx, CSMP_ERANGE, "column removal supports at most 1023 columns");
    if (s.R2) return CSMP_OK;
    CHECK(dmalloc(ctx, &s.R2, (size_t)s.kcap * s.kcap));
    CHECK(dmalloc(ctx, &s.Gdel, (size_t)2 * s.kcap + 2));
    CHECK(dmalloc(ctx, &s.qdrop, s.Mpad));
    CHECK(dmalloc(ctx, &s.bwd, s.kcap));
    CHECK(dmalloc(ctx, &s.bwd_coef, s.kcap));
    CHECK(dmalloc(ctx, &s.bwd_info, 2));
    CHECK(dmalloc(ctx, &s.delmeta, 4));
    CHECK(dmalloc(ctx, &s.qsave, s.Mpad));
    CHECK(dmalloc(ctx, &s.delpos, 1));
    return CSMP_OK;
}

// remove_column!(AiQR, *delpos) -- the insertion position is read from device memory (-1: nothing happens)
static int launch_delete(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    const int threads = std::min(1024, ((s.kcap + 1 + 63) / 64) * 64);
    hipLaunchKernelGGL(k_qrdel_r, dim3(1), dim3(threads), 0, ctx->stream, (const double*)s.R, s.R2, s.kcap, s.z, s.sel, s.st,
                       (const int*)s.delpos, s.Gdel, s.scal, s.delmeta);
    HIPCHECK(hipGetLastError());
    std::swap(s.R, s.R2);
    hipLaunchKernelGGL(k_qrdel_q, dim3(s.G), dim3(64), 0, ctx->stream, s.Q, s.ldq, (const double*)s.Gdel, (const double*)s.scal,
                       (const int*)s.delmeta, s.r, s.qdrop, s.qsave);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

static int launch_delete_atom(csmp_ctx* ctx, int atom) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_find_pos, dim3(1), dim3(256), 0, ctx->stream, (const int*)s.sel, (const DevState*)s.st, atom, s.delpos);
    HIPCHECK(hipGetLastError());
    return launch_delete(ctx);
}

// ---- explicit-inverse mode (csmp_tinv.hpp): T = R^-1 kept next to R by the two-stage solvers
static int tinv_ensure(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    CHECK(del_ensure(ctx));
    if (s.T) return CSMP_OK;
    const size_t nch = (size_t)(s.kcap + kTChunk - 1) / kTChunk;
    CHECK(dmalloc(ctx, &s.T, (size_t)s.kcap * s.kcap));
    CHECK(dmalloc(ctx, &s.T2, (size_t)s.kcap * s.kcap));
    CHECK(dmalloc(ctx, &s.tpd, nch * s.kcap));
    CHECK(dmalloc(ctx, &s.tpn, nch * s.kcap));
    CHECK(dmalloc(ctx, &s.tmeta, 2));
    return CSMP_OK;
}
// T = R^-1 for the columns factorised so far
static int launch_tinv_build(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    if (s.kcap <= 257)
        hipLaunchKernelGGL((k_tinv_build<4, 4>), dim3(s.kcap), dim3(64), 0, ctx->stream, (const double*)s.R, s.kcap,
                           (const DevState*)s.st, s.T, s.tmeta);
    else
        hipLaunchKernelGGL((k_tinv_build<16, 2>), dim3(s.kcap), dim3(64), 0, ctx->stream, (const double*)s.R, s.kcap,
                           (const DevState*)s.st, s.T, s.tmeta);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}
static int launch_tinv_mv(csmp_ctx* ctx, int mode) {
    Solver& s = ctx->s;
    const dim3 grid((s.kcap + 63) / 64, (s.kcap + kTChunk - 1) / kTChunk);
    hipLaunchKernelGGL(k_tinv_matvec, grid, dim3(64), 0, ctx->stream, (const double*)s.T, s.kcap, (const DevState*)s.st,
                       (const int*)s.tmeta, (const double*)s.z, (const double*)s.R, mode, s.tpd, s.tpn);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_tinv_fin, dim3(1), dim3(256), 0, ctx->stream, s.T, s.kcap, (const DevState*)s.st, s.tmeta,
                       (const double*)s.R, mode, (const double*)s.tpd, (const double*)s.tpn, s.bwd_coef, s.bwd);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}
// after launch_append: the column the append may have added enters T (no-op if it added none)
static int launch_tinv_append(csmp_ctx* ctx) { return launch_tinv_mv(ctx, 1); }
// x = T z (insertion order, s.bwd_coef) and the backward scores x^2 / gamma (s.bwd)
static int launch_tinv_solve(csmp_ctx* ctx) { return launch_tinv_mv(ctx, 0); }
// remove_column!(AiQR, *delpos) with the rotations taken from T
static int launch_delete_t(csmp_ctx* ctx) {
    Solver& s = ctx->s;
    const int threads = std::min(1024, ((s.kcap + 1 + 63) / 64) * 64);
    hipLaunchKernelGGL(k_tdel_prep, dim3(1), dim3(threads), 0, ctx->stream, (const double*)s.T, s.kcap, (const double*)s.z, s.sel,
                       s.st, (const int*)s.delpos, s.Gdel, s.scal, s.delmeta, s.tmeta);
    HIPCHECK(hipGetLastError());
    const int NB = (s.kcap + 63) / 64;
    hipLaunchKernelGGL(k_tdel_apply, dim3(s.G + 2 * NB + 1), dim3(64), 0, ctx->stream, s.Q, s.ldq, s.G, (const double*)s.T, s.T2,
                       (const double*)s.R, s.R2, s.kcap, NB, s.z, (const double*)s.Gdel, (const double*)s.scal,
                       (const int*)s.delmeta, s.r, s.qdrop, s.qsave);
    HIPCHECK(hipGetLastError());
    std::swap(s.T, s.T2);
    std::swap(s.R, s.R2);
    return CSMP_OK;
}
static int launch_delete_atom_t(csmp_ctx* ctx, int atom) {
    Solver& s = ctx->s;
    hipLaunchKernelGGL(k_find_pos, dim3(1), dim3(256), 0, ctx->stream, (const int*)s.sel, (const DevState*)s.st, atom, s.delpos);
    HIPCHECK(hipGetLastError());
    return launch_delete_t(ctx);
}
// fetch_sorted in explicit-inverse mode: coefficients from T z, emitted in index order
// (resnorm != NULL: the residual norm travels in the same synchronisation)
static int fetch_sorted_t(csmp_ctx* ctx, std::vector<int64_t>& idx, std::vector<double>& val, double* resnorm = nullptr) {
    Solver& s = ctx->s;
    if (resnorm) {
        hipLaunchKernelGGL(k_norm2, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.r, (int)ctx->M, s.scal + 1);
        HIPCHECK(hipGetLastError());
    }
    CHECK(launch_tinv_solve(ctx));
    hipLaunchKernelGGL(k_emit_sorted, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.bwd_coef, (const int*)s.sel,
                       (const DevState*)s.st, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap);
    HIPCHECK(hipGetLastError());
    idx.assign((size_t)s.outcap, 0);
    val.assign((size_t)s.outcap, 0.0);
    std::vector<int64_t> hi((size_t)s.outcap);
    std::vector<double> hv((size_t)s.outcap);
    int64_t hn = 0;
    double n2 = 0.0;
    PinFetch f(ctx);
    CHECK(f.begin((size_t)s.outcap * 16 + 64));
    CHECK(f.add(hi.data(), s.out_idx, (size_t)s.outcap * 8));
    CHECK(f.add(hv.data(), s.out_val, (size_t)s.outcap * 8));
    CHECK(f.add(&hn, s.out_nnz, 8));
    if (resnorm) CHECK(f.add(&n2, s.scal + 1, 8));
    CHECK(f.wait());
    idx.assign(hi.begin(), hi.begin() + hn);
    val.assign(hv.begin(), hv.begin() + hn);
    if (resnorm) *resnorm = std::sqrt(n2);
    return CSMP_OK;
}

// dropindex!(x, AiQR, i) on the step-level solver (src/util.jl:137-161): atom leaves the support
extern "C" int csmp_solver_remove(csmp_ctx* ctx, int64_t atom) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_remove: no solver begun");
    if (ctx->s.algo == CSMP_ALGO_MP) return fail(ctx, CSMP_EINVAL, "solver_remove: MP keeps no factorisation");
    if (ctx->s.algo == CSMP_ALGO_FR) return fail(ctx, CSMP_EINVAL, "solver_remove: use csmp_srr / the backward step for FR");
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(del_ensure(ctx));
    return launch_delete_atom(ctx, (int)atom);
}

extern "C" int csmp_fr_scores(csmp_ctx* ctx, double* delta2) {
    if (!ctx || !delta2) return CSMP_EINVAL;
    if (!ctx->s.dvec) return fail(ctx, CSMP_ESTATE, "fr_scores: no forward-regression step has run");
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(hipMemcpyAsync(delta2, ctx->s.dvec, (size_t)ctx->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return CSMP_OK;
}

extern "C" int csmp_solver_state(csmp_ctx* ctx, int64_t* idx, double* val, int64_t* nnz, double* resnorm, int64_t* order,
                                 int* stop) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->s.begun) return fail(ctx, CSMP_ESTATE, "solver_state: no solver begun");
    HIPCHECK(hipSetDevice(ctx->dev));
    Solver& s = ctx->s;
    if (resnorm) {
        hipLaunchKernelGGL(k_norm2, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.r, (int)ctx->M, s.scal);
        HIPCHECK(hipGetLastError());
        double n2 = 0.0;
        HIPCHECK(hipMemcpyAsync(&n2, s.scal, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        *resnorm = std::sqrt(n2);
    }
    {
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (s.algo != CSMP_ALGO_MP) s.jh = std::min(s.kcap, hs.nsel);  // the host's support bound snaps to the true count
        if (s.capped && s.jh < qr_max_cols()) s.capped = false;  // (the bound was loose: no-op steps had been counted)
        if (stop) *stop = (hs.done & (STOP_EPS | STOP_STAG | STOP_FULL)) | (s.capped ? CSMP_STOP_CAPACITY : 0);
    }
    if (s.algo == CSMP_ALGO_MP) return mp_collect(ctx, nullptr, nullptr, 0, idx, val, nnz);
    CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));
    return download_result(ctx, s.outcap, idx, val, nnz, order);
}

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
    for (int pass = 0; pass < kRsPasses; ++pass)  // (a settled selection turns the remaining launches into no-ops)
        hipLaunchKernelGGL(k_rs_hist, dim3(hgrid), dim3(256), 0, ctx->stream, (const double*)s.cvec, ctx->N, s.rs, kRsSettle);
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
    // that fails its DGKS test flags the solve, which is then repeated with the column-wise chain
    bool capacity_stop = false;
    for (int pass = 0; pass < 2; ++pass) {
        const bool block = pass == 0 && !ctx->force_reorth && l <= kPanelMax;
        CHECK(upload_b(ctx, b, b_dtype));
        const int main_skip = STOP_EPS | STOP_FULL | STOP_REORTH;
        for (int64_t it = 0; it < k / l && !ctx->s.capped; ++it) {  // :130-133
            CHECK(gomp_update(ctx, l, eps, it > 0, main_skip, block));
            if ((it + 1) % kPollSteps == 0 && it + 1 < k / l) {
                bool stopped = false;
                CHECK(solver_poll(ctx, &stopped));
                if (stopped) break;  // (the remainder step below still runs, as in the reference)
            }
        }
        const int64_t rem = k % l;                                                                             // :134
        if (rem > 0) CHECK(gomp_update(ctx, rem, 0.0, 0, STOP_FULL | STOP_REORTH, block));  // :135-137: runs even after an eps-break
        CHECK(launch_finish(ctx, ctx->s.out_idx, ctx->s.out_val, ctx->s.out_nnz, ctx->s.out_order, ctx->s.outcap));
        DevState hs;
        HIPCHECK(hipMemcpyAsync(&hs, ctx->s.st, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (!(hs.done & STOP_REORTH)) {
            capacity_stop = ctx->s.capped && !(hs.done & (STOP_EPS | STOP_STAG | STOP_FULL));
            break;
        }
    }
    CHECK(download_result(ctx, ctx->s.outcap, idx, val, nnz, order));
    return capacity_stop ? CSMP_WCAPACITY : CSMP_OK;
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
        csmp_ctx* c = ctx->twins[t];
        c->force_reorth = ctx->force_reorth;
        c->opt_ls_gram = ctx->opt_ls_gram;
        c->opt_ls_gram_reuse = ctx->opt_ls_gram_reuse;
        c->opt_twostage_update = ctx->opt_twostage_update;
    }
    return CSMP_OK;
}

static int gomp_enqueue(csmp_ctx* c, const void* col_dev, int b_dtype, int64_t l, int64_t k, double eps, bool block, int64_t* d_idx,
                        double* d_val, int64_t* d_nnz, int* d_flag, hipEvent_t after_first_sweep) {
    int rc = b_dtype == CSMP_F32 ? init_from_device_t<float>(c, (const float*)col_dev) : init_from_device_t<double>(c, (const double*)col_dev);
    if (rc != CSMP_OK) return rc;
    const int main_skip = STOP_EPS | STOP_FULL | STOP_REORTH;
    for (int64_t it = 0; it < k / l && !c->s.capped; ++it) {  // src/matchingpursuit.jl:130-133
        rc = gomp_update(c, l, eps, it > 0, main_skip, block);
        if (rc != CSMP_OK) return rc;
        if (it == 0 && after_first_sweep && hipEventRecord(after_first_sweep, c->stream) != hipSuccess) return CSMP_EHIP;
    }
    const int64_t rem = k % l;
    if (rem > 0) {  // :134-137: runs even after an eps-break
        rc = gomp_update(c, rem, 0.0, 0, STOP_FULL | STOP_REORTH, block);
        if (rc != CSMP_OK) return rc;
    }
    return launch_finish(c, d_idx, d_val, d_nnz, nullptr, (int)k, d_flag);
}

extern "C" int csmp_gomp_batch(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t l, int64_t k,
                               double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");  // src/matchingpursuit.jl:127
    if (!B || nsig < 0 || k < 1 || l < 1 || l > k || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "gomp_batch: bad arguments (needs 1 <= l <= k)");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (nsig == 0) return CSMP_OK;
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(twins_ensure(ctx, 1));
    csmp_ctx* cc[2] = {ctx, ctx->twins[0]};
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));  // at most k atoms are ever added (GOMP's own capacity is M: :108)
    for (int q = 0; q < 2; ++q) {
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
    if (!ctx->ev_twin) HIPCHECK(hipEventCreateWithFlags(&ctx->ev_twin, hipEventDisableTiming));
    const bool block = !ctx->force_reorth && l <= kPanelMax;
    for (int64_t sgn = 0; sgn < nsig; ++sgn) {
        csmp_ctx* c = cc[sgn & 1];
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        if (sgn == 1) HIPCHECK(hipStreamWaitEvent(c->stream, ctx->ev_twin, 0));  // the twin starts one sweep late: out of phase
        const int rc = gomp_enqueue(c, col, b_dtype, l, k, eps, block, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, d_flag + sgn,
                                    sgn == 0 ? ctx->ev_twin : nullptr);
        if (rc != CSMP_OK) {
            if (c != ctx) ctx->err = c->err;
            (void)hipStreamSynchronize(cc[0]->stream);
            (void)hipStreamSynchronize(cc[1]->stream);
            return rc;
        }
    }
    HIPCHECK(hipStreamSynchronize(cc[1]->stream));
    std::vector<int> hf((size_t)nsig);
    HIPCHECK(hipMemcpyAsync(hf.data(), d_flag, (size_t)nsig * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    // a panel that failed its DGKS test flagged the solve (nothing committed): that signal again, column by column
    int rc = CSMP_OK;
    bool capacity_stop = false;
    for (int64_t sgn = 0; sgn < nsig && rc == CSMP_OK; ++sgn) {
        if (hf[sgn] & STOP_REORTH) {
            const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
            rc = gomp_enqueue(ctx, col, b_dtype, l, k, eps, false, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, d_flag + sgn, nullptr);
            if (rc == CSMP_OK) {
                HIPCHECK(hipMemcpyAsync(&hf[sgn], d_flag + sgn, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHECK(hipStreamSynchronize(ctx->stream));
                capacity_stop |= ctx->s.capped && !(hf[sgn] & (STOP_EPS | STOP_STAG | STOP_FULL));
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
    return rc == CSMP_OK && capacity_stop ? CSMP_WCAPACITY : rc;
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
    if ((int)cols.size() > s.kcap) return fail(ctx, CSMP_ERANGE, "least squares: more columns than the QR capacity");
    CHECK(solver_restart(ctx));
    const int n = (int)cols.size();
    HIPCHECK(hipMemcpyAsync(s.cands, cols.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipMemcpyAsync(s.ncands, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    if (n > 1 && !ctx->force_reorth) {  // panels of 32 columns; verified through the device flag
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
static int gram_split_for(const csmp_ctx* ctx, int np) {
    // pieces of k_gram on or above the diagonal; the rows are split so that ONE round of workgroups (two per CU) covers them:
    // a second, partly filled round would cost as much as a full one
    const int TJ = np / kGramWgJ;
    int pieces = 0;
    for (int J = 0; J < TJ; ++J) pieces += (J * kGramWgJ + kGramWgJ - 1) / kGramWgI + 1;
    const int slots = (ctx->dtype == CSMP_F32 ? 3 : 2) * ctx->prop.multiProcessorCount;  // k_gram's workgroups per CU
    int nsplit = std::max(1, slots / std::max(1, pieces));
    nsplit = std::min<int>(nsplit, std::max<int>(1, (int)(ctx->M / 64)));  // at least four 16-row blocks each
    if (const char* e = tune_env("CSMP_GRAM_SPLIT")) nsplit = std::max(1, atoi(e));  // tuning / debugging knob
    return std::min(nsplit, 32);
}
static int gram_ensure(csmp_ctx* ctx, int np, int nsplit) {
    Solver& s = ctx->s;
    if (s.gram_np >= np && s.gram_split >= nsplit) return CSMP_OK;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    np = std::max(np, s.gram_np);
    nsplit = std::max(nsplit, s.gram_split);
    dfree(s.Gm); dfree(s.Dfac); dfree(s.Gpart); dfree(s.gdiag); dfree(s.rpart); dfree(s.Acomp); dfree(s.Gkeep); dfree(s.gdkeep); dfree(s.kpos); dfree(s.rhs_part); dfree(s.rn2part);
    s.gram_np = s.gram_split = 0;
    s.keep_valid = false;
    CHECK(dmalloc(ctx, &s.Gkeep, (size_t)np * np));
    CHECK(dmalloc(ctx, &s.gdkeep, (size_t)np));
    CHECK(dmalloc(ctx, &s.kpos, (size_t)np));
    CHECK(dmalloc(ctx, &s.rn2part, (size_t)(ctx->M + 255) / 256));
    CHECK(dmalloc(ctx, &s.rhs_part, (size_t)np * (size_t)(((ctx->M + 15) / 16 * 16 + 255) / 256)));
    CHECK(dmalloc(ctx, &s.Gm, (size_t)np * np));
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
template <typename TA>
static int ls_gram_t(csmp_ctx* ctx, const std::vector<int>& cols) {
    Solver& s = ctx->s;
    const int n = (int)cols.size(), M = (int)ctx->M;
    const int np = ((n + 1 + kGramTile - 1) / kGramTile) * kGramTile;
    const int nsplit = gram_split_for(ctx, np);
    CHECK(gram_ensure(ctx, np, nsplit));
    CHECK(solver_restart(ctx));
    // the column list (and, for a subset, its positions in the kept set) go up from a page-locked buffer that lives until the
    // next call -- every caller drains the stream before it comes back here
    void* pcv = nullptr;
    CHECK(pin_get(ctx, 2, (size_t)(2 * n + 2) * 4, &pcv));
    int* pcols = (int*)pcv;
    int* ppos = pcols + n + 1;
    for (int t = 0; t < n; ++t) pcols[t] = cols[t];
    pcols[n] = n;
    HIPCHECK(hipMemcpyAsync(s.cands, pcols, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipMemcpyAsync(s.ncands, pcols + n, 4, hipMemcpyHostToDevice, ctx->stream));
    // A set inside the last computed one: its bordered Gram matrix is a principal submatrix of the kept one -- gathered, not recomputed
    bool subset = s.keep_valid && n <= s.keep_n && ctx->opt_ls_gram_reuse;
    if (subset) {
        std::vector<std::pair<int, int>> where((size_t)s.keep_n);
        for (int t = 0; t < s.keep_n; ++t) where[t] = {s.keep_cols[t], t};
        std::sort(where.begin(), where.end());
        for (int t = 0; t < n && subset; ++t) {
            auto it = std::lower_bound(where.begin(), where.end(), std::make_pair(cols[t], 0));
            if (it == where.end() || it->first != cols[t]) subset = false;
            else ppos[t] = it->second;
        }
    }
    if (subset) {
        HIPCHECK(hipMemcpyAsync(s.kpos, ppos, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        const int64_t nel = (int64_t)np * np;
        hipLaunchKernelGGL(k_gram_subset, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gkeep, s.keep_np, s.keep_n,
                           (const double*)s.gdkeep, (const int*)s.kpos, n, np, s.Gm, s.gdiag);
        HIPCHECK(hipGetLastError());
    } else {
        const int blk = 16;
        const int rps = (((M + nsplit - 1) / nsplit + blk - 1) / blk) * blk;
        const int64_t ldo = ((int64_t)M + 15) / 16 * 16;
        const int nchunk = (int)((ldo + 255) / 256);
        hipLaunchKernelGGL(k_gather_cols<TA>, dim3((unsigned)nchunk, np), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M,
                           (const int*)s.cands, n, (TA*)s.Acomp, ldo, (const double*)s.b, np, s.rhs_part);
        hipLaunchKernelGGL(k_gram<TA>, dim3(np / kGramWgJ, (np + kGramWgI - 1) / kGramWgI, nsplit), dim3(256), 0, ctx->stream, (const TA*)s.Acomp, ldo, np,
                           rps, s.Gpart);
        HIPCHECK(hipGetLastError());
        const int64_t nel = (int64_t)np * np;
        hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gpart, nsplit, n, np,
                           s.Gm, s.gdiag, (const double*)s.rhs_part, nchunk, s.Gkeep, s.gdkeep);
        HIPCHECK(hipGetLastError());
        s.keep_cols.assign(cols.begin(), cols.end());
        s.keep_n = n;
        s.keep_np = np;
        s.keep_valid = true;
    }
    // Block rows 0 .. ceil(n / 32) - 1 are all that is needed: the bordered column n is a column of their row panels (or of
    // the last diagonal block when n is not a multiple of 32); the corner b'b - z'z and the identity padding are never read.
    const int nsteps = (n + kCholNB - 1) / kCholNB;
    {
        const int left0 = np - kCholNB;
        hipLaunchKernelGGL(k_chol_row, dim3(std::max(1, (left0 + kCholRowCols - 1) / kCholRowCols)), dim3(kCholThreads), 0, ctx->stream, s.Gm, np, n,
                           0, (const double*)s.gdiag, s.st, s.Dfac);
    }
    for (int kb = 0; kb + 1 < nsteps; ++kb) {  // one launch per step: trailing update of panel kb + block row kb + 1
        const int left = np - (kb + 1) * kCholNB;   // columns from the next block row on
        const int left2 = left - kCholNB;           // columns to the right of the next diagonal block
        const int Tt = (left + kGramTile - 1) / kGramTile;
        const int ntrail = left > kCholNB ? Tt * (Tt + 1) / 2 : 0;
        const int nrow = std::max(1, (left2 + kCholRowCols - 1) / kCholRowCols);
        hipLaunchKernelGGL(k_chol_step, dim3(nrow + ntrail), dim3(kCholThreads), 0, ctx->stream, s.Gm, np, n, kb, (const double*)s.gdiag, s.st,
                           nrow, s.Dfac);
    }
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_gram_export, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)s.Gm, np, n,
                       (const int*)s.cands, s.R, s.kcap, s.z, s.sel, s.st, (const double*)s.Dfac);
    HIPCHECK(hipGetLastError());
    s.jh = std::min(s.kcap, n);
    CHECK(launch_finish(ctx, s.out_idx, s.out_val, s.out_nnz, s.out_order, s.outcap));  // x = R^-1 z (s.coef: the order of cols) + sorted emission
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
    return n >= 64 && !ctx->force_reorth && ctx->opt_ls_gram;
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

// Least squares on `cols` + the sorted solution on the host (+ ||r|| when asked) in ONE synchronisation.  Large sets go
// through the whole-set path (csmp_gram.hpp); if their DGKS test fails, or for small sets, the append chain does it.
static int ls_fetch(csmp_ctx* ctx, const std::vector<int>& cols, std::vector<int64_t>& idx, std::vector<double>& val, double* resnorm) {
    Solver& s = ctx->s;
    if (gram_applicable(ctx, cols.size())) {
        CHECK(ls_gram(ctx, cols));
        const size_t n = cols.size();
        const size_t nshare = (size_t)(ctx->M + 255) / 256;  // |r|^2 comes back as the residual kernel's per-workgroup shares
        // one page-locked landing area for everything that comes back: [idx n | val n | control block | shares of ||r||^2]
        const size_t need = n * 16 + sizeof(DevState) + 16 + nshare * 8;
        void* pv = nullptr;
        CHECK(pin_get(ctx, 1, need, &pv));
        int64_t* pi = (int64_t*)pv;
        double* pvv = (double*)(pi + n);
        DevState* phs = (DevState*)(pvv + n);
        double* pn2 = (double*)((char*)phs + ((sizeof(DevState) + 7) / 8) * 8);
        HIPCHECK(hipMemcpyAsync(pi, s.out_idx, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipMemcpyAsync(pvv, s.out_val, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipMemcpyAsync(phs, s.st, sizeof(DevState), hipMemcpyDeviceToHost, ctx->stream));
        if (resnorm) HIPCHECK(hipMemcpyAsync(pn2, s.rn2part, nshare * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        const DevState hs = *phs;
        if (!(hs.done & STOP_REORTH) && hs.nsel == (int)n) {
            idx.assign(pi, pi + n);
            val.assign(pvv, pvv + n);
            if (resnorm) {
                double n2 = 0.0;
                for (size_t q = 0; q < nshare; ++q) n2 += pn2[q];
                *resnorm = std::sqrt(n2);
            }
            return CSMP_OK;
        }
    }
    CHECK(ls_on_columns(ctx, cols));
    CHECK(fetch_sorted(ctx, idx, val));
    if (resnorm) CHECK(residual_norm(ctx, resnorm));
    return CSMP_OK;
}

// sp_acquisition!(P, x, k): src/twostage.jl:67-72 -- sweep on the current residual, union the k best
// atoms into the support, least squares on the union
static int sp_acquire(csmp_ctx* ctx, int k, std::vector<int64_t>& idx, std::vector<double>& val, double* resnorm = nullptr) {
    Solver& s = ctx->s;
    CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
    CHECK(launch_topS(ctx, k));
    void* pv = nullptr;
    CHECK(pin_get(ctx, 1, (size_t)k * 4 + 16, &pv));  // (page-locked: the two small copies do not block the host one by one)
    int* top = (int*)pv;
    int* pnt = top + k;
    HIPCHECK(hipMemcpyAsync(top, s.cands, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipMemcpyAsync(pnt, s.ncands, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    const int nt = *pnt;
    std::vector<int> cols;
    for (auto i : idx) cols.push_back((int)i);
    for (int t = 0; t < nt; ++t) cols.push_back(top[t]);  // @. x[i] = NaN (:70)
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    return ls_fetch(ctx, cols, idx, val, resnorm);  // solve! (:71)
}

extern "C" int csmp_sp(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int64_t* idx,
                       double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || k < 1) return fail(ctx, CSMP_EINVAL, "sp: b == NULL or k < 1");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (2 * k > ctx->M) return fail(ctx, CSMP_ERANGE, "2k > length(b) is invalid for Subspace Pursuit");  // src/twostage.jl:55
    if (k > ctx->N) return fail(ctx, CSMP_ERANGE, "sp: k > number of atoms");
    if (maxiter < 0) maxiter = 16 * k;  // :87
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(solver_ensure(ctx, (int)(2 * k), (int)(2 * k)));
    ctx->s.begun = false;
    CHECK(upload_b(ctx, b, b_dtype));
    std::vector<int64_t> xi;
    std::vector<double> xv;
    double resnorm = 0.0;
    CHECK(sp_acquire(ctx, (int)k, xi, xv, &resnorm));  // :90-91
    int64_t it = 0;
    while (it < maxiter) {                   // :92
        const double oldnorm = resnorm;
        // update!(P::SP, x): :75-83
        CHECK(sp_acquire(ctx, (int)k, xi, xv));  // :77
        const int64_t drop = (int64_t)xi.size() - k;
        if (drop > 0) {  // :78-81: delete the (nnz-k) smallest |coef|, ties by position
            std::vector<int> pos(xi.size());
            for (size_t t = 0; t < pos.size(); ++t) pos[t] = (int)t;
            std::stable_sort(pos.begin(), pos.end(), [&](int a, int c) { return std::fabs(xv[a]) < std::fabs(xv[c]); });
            std::vector<char> kill(xi.size(), 0);
            for (int64_t t = 0; t < drop; ++t) kill[pos[t]] = 1;
            std::vector<int64_t> keep;
            for (size_t t = 0; t < xi.size(); ++t)
                if (!kill[t]) keep.push_back(xi[t]);
            xi.swap(keep);
        }
        std::vector<int> cols;
        for (auto i : xi) cols.push_back((int)i);
        CHECK(ls_fetch(ctx, cols, xi, xv, &resnorm));       // :82, :95
        ++it;
        if (resnorm <= delta || oldnorm <= resnorm) break;   // :96
    }
    for (size_t t = 0; t < xi.size(); ++t) {
        if (idx) idx[t] = xi[t];
        if (val) val[t] = xv[t];
    }
    if (nnz) *nnz = (int64_t)xi.size();
    if (iters) *iters = it;
    return CSMP_OK;
}

// sp for many signals: up to four solves in flight, each on a context (this one + clones on their own streams) driven by its own
// host thread.  A Subspace Pursuit solve is two HBM-bound sweeps and a long chain of short kernels with five host round trips
// (factorisations, selections, the pruning decision): one solve leaves most of the chip idle most of the time, and another
// signal's solve fills it.  Signal s is solved by context s mod T with the single-signal driver itself: results are csmp_sp's.
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
    const size_t es = b_dtype == CSMP_F32 ? 4 : 8;
    int rcs[4] = {CSMP_OK, CSMP_OK, CSMP_OK, CSMP_OK};
    auto work = [&](int t) {
        std::vector<int64_t> ti((size_t)2 * k);
        std::vector<double> tv((size_t)2 * k);
        for (int64_t sgn = t; sgn < nsig && rcs[t] == CSMP_OK; sgn += T) {
            int64_t n = 0, it = 0;
            const char* col = (const char*)B + (size_t)sgn * (size_t)ldB * es;
            rcs[t] = csmp_sp(cc[t], col, b_dtype, k, delta, maxiter, ti.data(), tv.data(), &n, &it);
            if (rcs[t] != CSMP_OK) break;
            for (int64_t q = 0; q < k; ++q) {
                idx[sgn * k + q] = q < n ? ti[q] : -1;
                val[sgn * k + q] = q < n ? tv[q] : 0.0;
            }
            nnz[sgn] = std::min<int64_t>(n, k);
            if (iters) iters[sgn] = it;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < T; ++t)
        if (rcs[t] != CSMP_OK) {
            if (t) ctx->err = cc[t]->err;
            return rcs[t];
        }
    return CSMP_OK;
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

// ------------------------------------------------------------------------------------------ OMP with replacement
// ompr(A,b,k,delta;maxiter): src/twostage.jl:110-202, x starting empty.  The support is filled by
// oblivious_acquisition! (src/matchingpursuit.jl:207-216); every update! (:134-180) is one sweep +
// arg-max on the device, the tiny "which entry leaves" decision on k+1 numbers on the host, and --
// when the support changes -- remove_column! as a Givens down-date of the on-device QR
// (csmp_downdate.hpp) followed by the usual append (k > 1023: a fresh panel factorisation instead).
extern "C" int csmp_ompr(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int64_t* idx,
                         double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || k < 1) return fail(ctx, CSMP_EINVAL, "ompr: b == NULL or k < 1");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (k > ctx->N || k > ctx->M) return fail(ctx, CSMP_ERANGE, "ompr: k exceeds size(A)");
    if (maxiter < 0) maxiter = ctx->M;  // :185
    HIPCHECK(hipSetDevice(ctx->dev));
    const bool want_downdate = k <= kDelMaxCols && ctx->opt_twostage_update != 2;
    if (want_downdate) CHECK(solver_fit_for_removal(ctx, (int)k));
    CHECK(solver_ensure(ctx, (int)k, (int)k));
    ctx->s.begun = false;
    Solver& s = ctx->s;
    const bool use_downdate = k <= kDelMaxCols && ctx->opt_twostage_update != 2;  // (option 2: refactorise instead)
    const bool tmode = use_downdate && ctx->opt_twostage_update == 0;  // explicit inverse next to R (csmp_tinv.hpp)
    if (use_downdate) CHECK(del_ensure(ctx));
    if (tmode) CHECK(tinv_ensure(ctx));
    CHECK(upload_b(ctx, b, b_dtype));
    // oblivious_acquisition!(P, x, k): the k atoms best correlated with b, least squares on them
    CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
    CHECK(launch_topS(ctx, (int)k));
    std::vector<int> top((size_t)k);
    HIPCHECK(hipMemcpyAsync(top.data(), s.cands, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    std::sort(top.begin(), top.end());
    CHECK(ls_on_columns(ctx, top));
    std::vector<int64_t> xi;
    std::vector<double> xv;
    if (tmode) {
        CHECK(launch_tinv_build(ctx));
        CHECK(fetch_sorted_t(ctx, xi, xv));
    } else {
        CHECK(fetch_sorted(ctx, xi, xv));
    }
    double resnorm = 0.0;
    CHECK(residual_norm(ctx, &resnorm));  // :192
    int64_t it = 0;
    std::vector<double> cs((size_t)k), call;
    while (it < maxiter) {  // :193
        const double oldnorm = resnorm;
        bool have_norm = false;
        // update!(P, x): Ar = x + A'r (eta = 1), arg-max over atoms outside the support
        CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
        CHECK(launch_select(ctx, 0, 0));
        std::vector<int> cur(xi.begin(), xi.end());
        HIPCHECK(hipMemcpyAsync(s.cands, cur.data(), cur.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_gather, dim3(((int)k + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.cvec, (const int*)s.cands, (int)k, s.coef);
        HIPCHECK(hipGetLastError());
        DevState hs;
        {
            PinFetch f(ctx);
            CHECK(f.begin((size_t)k * 8 + sizeof hs + 16));
            CHECK(f.add(cs.data(), s.coef, (size_t)k * 8));
            CHECK(f.add(&hs, s.st, sizeof hs));
            CHECK(f.wait());
        }
        int64_t cand = hs.cand;
        double ccand = hs.cval;
        if (std::binary_search(xi.begin(), xi.end(), cand)) {
            // degenerate: the overall arg-max lies inside the support; scan the correlations on the host
            call.resize((size_t)ctx->N);
            HIPCHECK(hipMemcpy(call.data(), s.cvec, (size_t)ctx->N * 8, hipMemcpyDeviceToHost));
            cand = -1;
            double m = 0.0;
            for (int64_t j = 0; j < ctx->N; ++j) {
                if (std::binary_search(xi.begin(), xi.end(), j)) continue;
                const double f = std::fabs(call[j]);
                if (f > m) {  // strict '>' from m = 0: first maximum, none if everything is zero (:139-155)
                    m = f;
                    cand = j;
                }
            }
            if (cand >= 0) ccand = call[cand];
        } else if (!(std::fabs(ccand) > 0.0)) {
            cand = -1;
        }
        ++it;
        if (cand >= 0) {
            // x[i] = NaN; x.nzval = Ar[x.nzind]; drop the first entry of smallest magnitude (:158-169)
            const size_t pos = (size_t)(std::lower_bound(xi.begin(), xi.end(), cand) - xi.begin());
            size_t jmin = 0;
            double vmin = 0.0;
            for (size_t t = 0; t <= xi.size(); ++t) {
                const double v = t == pos ? ccand : (t < pos ? xv[t] + cs[t] : xv[t - 1] + cs[t - 1]);
                if (t == 0 || std::fabs(v) < vmin) {
                    vmin = std::fabs(v);
                    jmin = t;
                }
            }
            if (jmin != pos) {  // qr_i != j (:171): the support really changes
                const int leaving = (int)(jmin < pos ? xi[jmin] : xi[jmin - 1]);
                if (use_downdate) {
                    // remove_column! + add_column! (:172-176) as a Givens down-date and a Gram-Schmidt append
                    if (tmode)
                        CHECK(launch_delete_atom_t(ctx, leaving));
                    else
                        CHECK(launch_delete_atom(ctx, leaving));
                    const int one = 1, ci = (int)cand;
                    HIPCHECK(hipMemcpyAsync(s.cands, &ci, 4, hipMemcpyHostToDevice, ctx->stream));
                    HIPCHECK(hipMemcpyAsync(s.ncands, &one, 4, hipMemcpyHostToDevice, ctx->stream));
                    CHECK(launch_append(ctx, 2, 0, 0));
                    if (tmode) CHECK(launch_tinv_append(ctx));
                } else {
                    std::vector<int> cols;
                    for (size_t t = 0; t <= xi.size(); ++t) {
                        if (t == jmin) continue;
                        cols.push_back(t == pos ? (int)cand : (int)(t < pos ? xi[t] : xi[t - 1]));
                    }
                    CHECK(ls_on_columns(ctx, cols));  // :178
                }
                if (tmode) {
                    CHECK(fetch_sorted_t(ctx, xi, xv, &resnorm));  // :178 and :196 in one synchronisation
                    have_norm = true;
                } else {
                    CHECK(fetch_sorted(ctx, xi, xv));
                }
            }
        }
        if (!have_norm) CHECK(residual_norm(ctx, &resnorm));  // :196
        if (resnorm <= delta || oldnorm <= resnorm) break;   // :197
    }
    for (size_t t = 0; t < xi.size(); ++t) {
        if (idx) idx[t] = xi[t];
        if (val) val[t] = xv[t];
    }
    if (nnz) *nnz = (int64_t)xi.size();
    if (iters) *iters = it;
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ stepwise regression object
// StepwiseRegression = ForwardRegression (src/forward.jl:14-32) kept on the device and driven from the
// host: forward_step! (src/forward.jl:56-73) = one dictionary sweep + one append, backward_step!
// (src/backward.jl:51-67) = scores from T = R^-1 + column removal.  The OLS rescaling rho2 follows the
// support through rank-one corrections folded into the NEXT sweep: -<a,q>^2 for the column a forward
// step appended, +<a,q_drop>^2 for the direction a backward step rotated out (csmp_forward.hpp,
// NQ = 2), so a forward/backward pair streams the dictionary once.  The host reads the 48-byte control
// block after every step (it must know which steps changed the support).
struct Stepwise {
    struct Pend { const double* q; double sgn; };
    csmp_ctx* ctx = nullptr;
    std::vector<Pend> pend;  // corrections rho2 still lacks; q == nullptr: the last Q column (device look-up)
    bool unmark = false;     // delmeta[2] names an atom that left the support and needs its rho2 re-seeded
    bool rho_ready = false;  // rho2 has been initialised (|a_j|^2 at least)
    int n = 0;               // atoms in the support
    double last_max_d2 = 0.0;  // maximum(P.δ²) of the last forward step
    int last_added = -1, last_removed = -1;  // atoms moved by the last successful forward / backward step
    DevState hs;

    int read_state(int* also_int = nullptr, const int* also_dev = nullptr) {
        Solver& s = ctx->s;
        PinFetch f(ctx);
        CHECK(f.begin(sizeof hs + 16));
        if (also_int) CHECK(f.add(also_int, also_dev, sizeof(int)));
        CHECK(f.add(&hs, s.st, sizeof hs));
        return f.wait();
    }
    int clear_flags() {
        static const int zero = 0;
        HIPCHECK(hipMemcpyAsync(&ctx->s.st->done, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        return CSMP_OK;
    }
    FrPass pass_of(int update_only) const {
        const Solver& s = ctx->s;
        FrPass ps;
        ps.nq = rho_ready ? (int)pend.size() : -1;
        ps.update_only = update_only;
        if (pend.size() >= 1) { ps.q1 = pend[0].q; ps.s1 = pend[0].sgn; }
        if (pend.size() >= 2) { ps.q2 = pend[1].q; ps.s2 = pend[1].sgn; }
        ps.unmark = unmark ? s.delmeta + 2 : nullptr;  // (the direction that was rotated out is always the last one)
        return ps;
    }
    // buffers for at most kcap atoms, b uploaded, empty support
    int begin(csmp_ctx* c, const void* b, int b_dtype, int kcap) {
        ctx = c;
        CHECK(solver_fit_for_removal(ctx, kcap));
        CHECK(solver_ensure(ctx, kcap, kcap));
        CHECK(fr_ensure(ctx));
        CHECK(tinv_ensure(ctx));
        Solver& s = ctx->s;
        s.begun = false;
        CHECK(upload_b(ctx, b, b_dtype));
        HIPCHECK(hipMemsetAsync(s.tmeta, 0, 2 * sizeof(int), ctx->stream));
        pend.clear();
        unmark = false;
        rho_ready = false;
        n = 0;
        return CSMP_OK;
    }
    // forward_step!(P, x, max_eps, min_delta); guarded == false: update!(P::FR, x) (src/forward.jl:88-95)
    int forward(double max_eps, double min_d2, bool guarded, bool* ok) {
        Solver& s = ctx->s;
        const int skipF = STOP_EPS | STOP_STAG | STOP_FULL;
        if (!guarded) {
            max_eps = -HUGE_VAL;
            min_d2 = -1.0;
        }
        CHECK(launch_fr_pass(ctx, pass_of(0), max_eps, skipF));
        CHECK(launch_append(ctx, 3, 0, skipF, false, min_d2, s.fr_grid));
        CHECK(launch_tinv_append(ctx));
        CHECK(read_state());
        if (hs.done & skipF) {
            // the step failed.  A residual-norm stop returns before rho2 is touched; the other guards act
            // after the sweep, which has then consumed the pending corrections.
            if (!(hs.done & STOP_EPS)) {
                pend.clear();
                unmark = false;
                rho_ready = true;
                last_max_d2 = hs.cval;
            }
            CHECK(clear_flags());
            *ok = false;
            return CSMP_OK;
        }
        last_max_d2 = hs.cval;
        last_added = hs.cand;
        rho_ready = true;
        pend.clear();
        unmark = false;
        pend.push_back({nullptr, -1.0});
        n = hs.nsel;
        *ok = true;
        return CSMP_OK;
    }
    // applies the pending corrections now (needed before a second removal reuses the q_drop buffer)
    int flush() {
        if (pend.empty() && !unmark) return CSMP_OK;
        CHECK(launch_fr_pass(ctx, pass_of(1), 0.0, 0));
        pend.clear();
        unmark = false;
        return CSMP_OK;
    }
    // backward_step!(P, x, max_eps, max_delta); lace: LACE's candidate rule (least |x_i|)
    int backward(double max_eps, double max_d2, bool* ok, bool lace = false) {
        Solver& s = ctx->s;
        *ok = false;
        if (n <= 0) return CSMP_OK;
        bool has_drop = false;
        for (const Pend& e : pend) has_drop |= e.q == s.qdrop;
        if (has_drop) CHECK(flush());
        CHECK(launch_tinv_solve(ctx));
        hipLaunchKernelGGL(k_bwd_pick, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.bwd, (const int*)s.sel,
                           (const DevState*)s.st, (const double*)s.r, (int)ctx->M, max_eps, max_d2, s.delpos, s.bwd_info,
                           lace ? (const double*)s.bwd_coef : (const double*)nullptr);
        HIPCHECK(hipGetLastError());
        CHECK(launch_delete_t(ctx));
        CHECK(read_state(&last_removed, s.delmeta + 2));
        if (hs.nsel == n) return CSMP_OK;  // the thresholds (or the lack of a finite score) kept every atom
        for (Pend& e : pend)
            if (!e.q) e.q = s.qsave;  // the appended column has been rotated; k_tdel_apply kept a copy
        pend.push_back({s.qdrop, 1.0});
        unmark = true;
        n = hs.nsel;
        *ok = true;
        return CSMP_OK;
    }
    int result(int64_t* idx, double* val, int64_t* nnz) {
        std::vector<int64_t> xi;
        std::vector<double> xv;
        CHECK(fetch_sorted_t(ctx, xi, xv));
        for (size_t t = 0; t < xi.size(); ++t) {
            if (idx) idx[t] = xi[t];
            if (val) val[t] = xv[t];
        }
        if (nnz) *nnz = (int64_t)xi.size();
        return CSMP_OK;
    }
};

static int stepwise_args(csmp_ctx* ctx, const void* b, const char* who) {
    if (!ctx) return CSMP_EINVAL;
    if (!b) return fail(ctx, CSMP_EINVAL, (std::string(who) + ": b == NULL").c_str());
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    return CSMP_OK;
}

// srr(A,b,k,delta; maxiter=4k, initialization, l): src/twostage.jl:3-33, x starting empty.  initialization 3
// (random_acquisition!, src/matchingpursuit.jl:195-204) takes its k atoms from `init` (sorted, distinct): the draw is the
// caller's -- the reference takes it from the host language's RNG.
static int srr_impl(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int initialization,
                    const std::vector<int>* init, int64_t l, int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    CHECK(stepwise_args(ctx, b, "srr"));
    if (k < 1 || l < 1) return fail(ctx, CSMP_EINVAL, "srr: k < 1 or l < 1");
    if (initialization != 1 && initialization != 2 && !(initialization == 3 && init))
        return fail(ctx, CSMP_EINVAL, "srr: initialization must be 1 (oblivious) or 2 (forward regression); 3 (random) through csmp_srr_from");
    if (k > ctx->N || k + l > ctx->M) return fail(ctx, CSMP_ERANGE, "srr: k exceeds size(A)");
    if (k + l > kTMaxCols) return fail(ctx, CSMP_ERANGE, "srr: k + l exceeds 1023");
    if (maxiter < 0) maxiter = 4 * k;  // :5
    HIPCHECK(hipSetDevice(ctx->dev));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, (int)(k + l)));
    Solver& s = ctx->s;
    if (initialization == 1 || initialization == 3) {
        std::vector<int> top((size_t)k);
        if (initialization == 1) {
            // oblivious_acquisition!(P, x, k): src/matchingpursuit.jl:207-216
            CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
            CHECK(launch_topS(ctx, (int)k));
            HIPCHECK(hipMemcpyAsync(top.data(), s.cands, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            std::sort(top.begin(), top.end());
        } else {
            top = *init;  // random_acquisition!(P, x, k): :195-204 -- the caller's draw, sorted
        }
        CHECK(ls_on_columns(ctx, top));
        // rho2_j = |a_j|^2 - |Q'a_j|^2 for the k columns just factorised: the norms, then four columns per pass
        FrPass p0;
        p0.nq = -1;
        p0.update_only = 1;
        CHECK(launch_fr_pass(ctx, p0, 0.0, 0));
        if (!tune_env("CSMP_FR_REBUILD_SWEEPS")) {
            // Q'A on the Float64 matrix cores, 128 directions per pass (csmp_forward.hpp, k_fr_rebuild)
            const int grid = (int)((ctx->N + 127) / 128);  // 4 waves x 32 atoms
            for (int64_t t = 0; t < k; t += 128) {
                const int nd = (int)std::min<int64_t>(128, k - t);
                if (ctx->dtype == CSMP_F32)
                    hipLaunchKernelGGL(k_fr_rebuild<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld,
                                       (int)ctx->M, ctx->N, (const double*)s.Q, s.ldq, (int)t, nd, s.rho2);
                else
                    hipLaunchKernelGGL(k_fr_rebuild<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld,
                                       (int)ctx->M, ctx->N, (const double*)s.Q, s.ldq, (int)t, nd, s.rho2);
                HIPCHECK(hipGetLastError());
            }
        } else {
        int U4, g4; bool f4; size_t lds4;
            fr_config(ctx, 4, U4, f4, lds4, g4);
            const bool four = lds4 <= 160 * 1024 - 512;  // four direction images fit the LDS (M <= ~5000)
            for (int64_t t = 0; t < k;) {
                FrPass ps;
                ps.update_only = 1;
                ps.q1 = s.Q + t * s.ldq;
                ps.s1 = -1.0;
                if (four && t + 4 <= k) {
                    ps.nq = 4;
                    ps.qstride = s.ldq;
                    t += 4;
                } else if (t + 2 <= k) {
                    ps.nq = 2;
                    ps.q2 = s.Q + (t + 1) * s.ldq;
                    ps.s2 = -1.0;
                    t += 2;
                } else {
                    ps.nq = 1;
                    t += 1;
                }
                CHECK(launch_fr_pass(ctx, ps, 0.0, 0));
            }
        }
        hipLaunchKernelGGL(k_mark_inf, dim3(1), dim3(256), 0, ctx->stream, s.rho2, (const int*)s.sel, (const DevState*)s.st);
        HIPCHECK(hipGetLastError());
        CHECK(launch_tinv_build(ctx));
        CHECK(P.read_state());
        P.n = P.hs.nsel;
        P.rho_ready = true;
        if (P.hs.done) CHECK(P.clear_flags());
    } else {
        // k times update!(P::FR, x) (:12-15; src/forward.jl:88-95): enqueued back to back, no host round trips
        const int skip = STOP_FULL | STOP_STAG;
        for (int64_t t = 0; t < k; ++t) {
            CHECK(launch_fr_sweep(ctx, t == 0, -HUGE_VAL, skip));
            CHECK(launch_append(ctx, 3, 0, skip, false, -1.0, s.fr_grid));
        }
        CHECK(launch_tinv_build(ctx));
        CHECK(P.read_state());
        P.n = P.hs.nsel;
        P.rho_ready = true;
        if (P.n > 0) P.pend.push_back({nullptr, -1.0});  // the last appended column has not reached rho2 yet
        if (P.hs.done) CHECK(P.clear_flags());
    }
    double resnorm = 0.0;
    CHECK(residual_norm(ctx, &resnorm));  // :18
    int64_t it = 0;
    while (it < maxiter) {  // :19
        const double oldnorm = resnorm;
        std::vector<int> added, removed;
        for (int64_t f = 0; f < l; ++f) {  // :21-23  forward_step!(P, x, 0, 0) || break
            bool ok;
            CHECK(P.forward(0.0, 0.0, true, &ok));
            if (!ok) break;
            added.push_back(P.last_added);
        }
        while (P.n > k) {  // :24-26  backward_step!(P, x, Inf, Inf)
            bool ok;
            CHECK(P.backward((double)HUGE_VAL, (double)HUGE_VAL, &ok));
            if (!ok) break;
            removed.push_back(P.last_removed);
        }
        std::sort(added.begin(), added.end());
        std::sort(removed.begin(), removed.end());
        // An iteration that removed exactly the atoms it added left x where it was: the residual norm is the
        // old one (:27-28 then stops).  Measuring it instead would compare two roundings of the same number.
        if (added == removed)
            resnorm = oldnorm;
        else
            CHECK(residual_norm(ctx, &resnorm));  // :27
        ++it;
        if (resnorm <= delta || oldnorm <= resnorm) break;  // :28-30
    }
    if (iters) *iters = it;
    return P.result(idx, val, nnz);
}

// ------------------------------------------------------------------------------------------ relevance matching pursuit, FoBa
// src/stepwise.jl (x starting empty): loops over the two steps of the object above.  kmax bounds the
// support the forward stage may build (at most 1023; the reference's only bound is size(A,1)): a
// forward stage that needs more atoms than that ends with CSMP_ERANGE rather than a truncated answer.
static int stepwise_cap(csmp_ctx* ctx, int64_t kmax, int* kcap) {
    const int64_t lim = std::min<int64_t>(std::min<int64_t>(ctx->M, ctx->N), kTMaxCols);
    if (kmax <= 0) kmax = lim;
    *kcap = (int)std::min<int64_t>(kmax, lim);
    return CSMP_OK;
}
static int stepwise_full(csmp_ctx* ctx, const Stepwise& P, int kcap) {
    if (P.n >= kcap && kcap < std::min<int64_t>(ctx->M, ctx->N))
        return fail(ctx, CSMP_ERANGE, "stepwise regression: the forward stage filled the support capacity (kmax, at most 1023 atoms)");
    return CSMP_OK;
}
// !(xt ≈ x): isapprox with Julia's default rtol = sqrt(eps) on the sparse vectors
static bool x_changed(const std::vector<int64_t>& i0, const std::vector<double>& v0, const std::vector<int64_t>& i1,
                      const std::vector<double>& v1) {
    double d2 = 0.0, na = 0.0, nb = 0.0;
    size_t i = 0, j = 0;
    while (i < i0.size() || j < i1.size()) {
        double a = 0.0, b = 0.0;
        if (j >= i1.size() || (i < i0.size() && i0[i] < i1[j])) a = v0[i++];
        else if (i >= i0.size() || i1[j] < i0[i]) b = v1[j++];
        else { a = v0[i++]; b = v1[j++]; }
        d2 += (a - b) * (a - b);
        na += a * a;
        nb += b * b;
    }
    return !(std::sqrt(d2) <= 1.4901161193847656e-08 * std::sqrt(std::max(na, nb)));
}

// rmp(A, b, delta, maxiter): src/stepwise.jl:5-26

extern "C" int csmp_srr(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int initialization,
                        int64_t l, int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (initialization == 3) return fail(ctx, CSMP_EINVAL, "srr: initialization 3 (random) needs the drawn atoms: csmp_srr_from");
    return srr_impl(ctx, b, b_dtype, k, delta, maxiter, initialization, nullptr, l, idx, val, nnz, iters);
}

// srr with initialization = 3: init[0..k) are the k distinct atoms random_acquisition! would have drawn (any order)
extern "C" int csmp_srr_from(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, const int64_t* init,
                             int64_t l, int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!init || k < 1) return fail(ctx, CSMP_EINVAL, "srr_from: init == NULL or k < 1");
    std::vector<int> top((size_t)k);
    for (int64_t t = 0; t < k; ++t) {
        if (init[t] < 0 || init[t] >= ctx->N) return fail(ctx, CSMP_ERANGE, "srr_from: atom index out of range");
        top[t] = (int)init[t];
    }
    std::sort(top.begin(), top.end());  // sort!(ind) (:197)
    if (std::adjacent_find(top.begin(), top.end()) != top.end()) return fail(ctx, CSMP_EINVAL, "srr_from: duplicate atom");
    return srr_impl(ctx, b, b_dtype, k, delta, maxiter, 3, &top, l, idx, val, nnz, iters);
}
extern "C" int csmp_rmp_delta(csmp_ctx* ctx, const void* b, int b_dtype, double delta, int64_t maxiter, int64_t kmax, int64_t* idx,
                              double* val, int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "rmp"));
    if (maxiter < 0) maxiter = 1;
    HIPCHECK(hipSetDevice(ctx->dev));
    int kcap;
    CHECK(stepwise_cap(ctx, kmax, &kcap));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, kcap));
    const double d2 = delta * delta;
    std::vector<int64_t> xi0, xi;
    std::vector<double> xv0, xv;
    for (int64_t it = 0; it < maxiter; ++it) {  // :10
        for (int64_t f = 0; f < ctx->M; ++f) {  // :12-14
            bool ok;
            CHECK(P.forward(0.0, d2, true, &ok));
            if (!ok) break;
        }
        CHECK(stepwise_full(ctx, P, kcap));
        CHECK(fetch_sorted_t(ctx, xi, xv));
        if (!x_changed(xi0, xv0, xi, xv)) break;  // :15
        xi0 = xi;
        xv0 = xv;
        for (int t = P.n; t >= 1; --t) {  // :18-20
            bool ok;
            CHECK(P.backward((double)HUGE_VAL, d2, &ok));
            if (!ok) break;
        }
        CHECK(fetch_sorted_t(ctx, xi, xv));
        if (!x_changed(xi0, xv0, xi, xv)) break;  // :21
        xi0 = xi;
        xv0 = xv;
    }
    return P.result(idx, val, nnz);
}

// rmp(A, b, k): src/stepwise.jl:32-43
extern "C" int csmp_rmp_k(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, int64_t kmax, int64_t* idx, double* val,
                          int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "rmp"));
    if (k < 0) return fail(ctx, CSMP_EINVAL, "rmp: k < 0");
    HIPCHECK(hipSetDevice(ctx->dev));
    int kcap;
    CHECK(stepwise_cap(ctx, kmax, &kcap));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, kcap));
    for (int64_t f = 0; f < ctx->M; ++f) {  // :36-38
        bool ok;
        CHECK(P.forward(0.0, 0.0, true, &ok));
        if (!ok) break;
    }
    CHECK(stepwise_full(ctx, P, kcap));
    for (int t = P.n; t >= k + 1; --t) {  // :39-41
        bool ok;
        CHECK(P.backward((double)HUGE_VAL, (double)HUGE_VAL, &ok));
        if (!ok) break;
    }
    return P.result(idx, val, nnz);
}

// foba(A, b, delta): src/stepwise.jl:47-56
extern "C" int csmp_foba(csmp_ctx* ctx, const void* b, int b_dtype, double delta, int64_t kmax, int64_t* idx, double* val,
                         int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "foba"));
    HIPCHECK(hipSetDevice(ctx->dev));
    int kcap;
    CHECK(stepwise_cap(ctx, kmax, &kcap));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, kcap));
    const double d2 = delta * delta;
    for (int64_t f = 0; f < ctx->M; ++f) {  // :50
        bool ok;
        CHECK(P.forward(0.0, d2, true, &ok));  // :51
        if (!ok) break;
        const double half = std::sqrt(P.last_max_d2) / 2.0;  // :52-53
        for (;;) {
            CHECK(P.backward((double)HUGE_VAL, half * half, &ok));
            if (!ok) break;
        }
    }
    CHECK(stepwise_full(ctx, P, kcap));
    return P.result(idx, val, nnz);
}

// br(A,b,max_eps,max_delta,k) (src/backward.jl:27-35; fbr :154-162 is the same algorithm on the normal
// equations) and, with lace != 0, lace(A,b,eps,delta,k) (:233-270): the least-squares solution on ALL
// N <= M columns, then backward steps until k atoms are left or a threshold stops them.
extern "C" int csmp_br(csmp_ctx* ctx, const void* b, int b_dtype, double max_eps, double max_delta, int64_t k, int lace,
                       int64_t* idx, double* val, int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "br"));
    if (k < 0) return fail(ctx, CSMP_EINVAL, "br: k < 0");
    if (max_eps != max_eps || max_delta != max_delta) return fail(ctx, CSMP_EINVAL, "br: threshold is NaN");
    if (ctx->N > ctx->M) return fail(ctx, CSMP_ERANGE, "br: A needs to be overdetermined (size(A,2) <= size(A,1))");  // :218
    if (ctx->N > kTMaxCols) return fail(ctx, CSMP_ERANGE, "br: more than 1023 columns");
    HIPCHECK(hipSetDevice(ctx->dev));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, (int)ctx->N));
    std::vector<int> all((size_t)ctx->N);
    for (int64_t j = 0; j < ctx->N; ++j) all[(size_t)j] = (int)j;
    CHECK(ls_on_columns(ctx, all));  // UpdatableQR(A); x = AiQR \ b   (:11,:30)
    CHECK(launch_tinv_build(ctx));
    CHECK(P.read_state());
    P.n = P.hs.nsel;
    if (P.hs.done) CHECK(P.clear_flags());
    const double d2 = max_delta * max_delta;
    for (int t = P.n; t >= k + 1; --t) {  // :31-33
        bool ok;
        CHECK(P.backward(max_eps, d2, &ok, lace != 0));
        if (!ok) break;
    }
    return P.result(idx, val, nnz);
}

// ------------------------------------------------------------------------------------------ batched (MFMA-screened) OMP
__global__ void k_absmax_f32(const float* __restrict__ A, int64_t n, float* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(A[i]));
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
}
__global__ void k_absmax_f64(const double* __restrict__ A, int64_t n, float* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, (float)fabs(A[i]));
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m * 1.0000002f));
}

// max_j |a_j|_2 (rounded up), one wave per column
template <typename TA>
__global__ __launch_bounds__(256) void k_colnorm_max(const TA* __restrict__ A, int64_t ld, int M, int64_t N, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    double acc = 0.0;
    for (int m = lane; m < M; m += 64) {
        const double v = (double)A[col * ld + m];
        acc = fma(v, v, acc);
    }
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint((float)sqrt(acc) * 1.0000002f));
}

static int batch_dict(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.ab_valid) return CSMP_OK;
    // K is padded (zeros) to an even number of 64-deep tiles, at least four: what the eight-phase screening kernel needs
    b.Mk = (int)std::max<int64_t>(256, ((ctx->M + 127) / 128) * 128);
    b.Npad = ((ctx->N + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT);  // whole 256-atom tiles
    b.n_atiles = (int)(b.Npad / kBT);
    HIPCHECK(hipMalloc((void**)&b.Ab, (size_t)b.Npad * b.Mk * sizeof(__bf16)));
    HIPCHECK(hipMalloc((void**)&b.amax, sizeof(float)));
    HIPCHECK(hipMemsetAsync(b.amax, 0, sizeof(float), ctx->stream));
    const int64_t total = b.Npad * (b.Mk / 8);
    const int grid = (int)((total + 255) / 256);
    const int64_t nel = ctx->ld * ctx->N;
    if (ctx->dtype == CSMP_F32) {
        hipLaunchKernelGGL(k_b_convert<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.Ab, b.Mk, b.Npad);
        hipLaunchKernelGGL(k_absmax_f32, dim3(2048), dim3(256), 0, ctx->stream, (const float*)ctx->dA, nel, b.amax);
    } else {
        hipLaunchKernelGGL(k_b_convert<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.Ab, b.Mk, b.Npad);
        hipLaunchKernelGGL(k_absmax_f64, dim3(2048), dim3(256), 0, ctx->stream, (const double*)ctx->dA, nel, b.amax);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(&b.amax_host, b.amax, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.ab_valid = true;
    return CSMP_OK;
}

// max_j |a_j|_2 (the deterministic screening bound), computed on first use
static int batch_colnorm(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.anorm_host >= 0.f) return CSMP_OK;
    HIPCHECK(hipMemsetAsync(b.amax, 0, sizeof(float), ctx->stream));
    const unsigned grid = (unsigned)((ctx->N + 3) / 4);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_colnorm_max<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.amax);
    else
        hipLaunchKernelGGL(k_colnorm_max<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.amax);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(&b.anorm_host, b.amax, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return CSMP_OK;
}

static int batch_ensure(csmp_ctx* ctx, int nsig, int kcap) {
    Batch& b = ctx->bt;
    const int Bpad = ((nsig + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT);
    if (b.Bcap >= Bpad && b.kcap >= kcap) return CSMP_OK;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    const int nb = std::max(Bpad, b.Bcap), nk = std::max(kcap, b.kcap);
    batch_free(b, true);
    b.Bcap = nb;
    b.kcap = nk;
    b.Mr = (int)(((ctx->M + 3) / 4) * 4);
    CHECK(dmalloc(ctx, &b.Rb, (size_t)nb * b.Mk));
    CHECK(dmalloc(ctx, &b.r, (size_t)nb * b.Mr));
    CHECK(dmalloc(ctx, &b.b, (size_t)nb * b.Mr));
    CHECK(dmalloc(ctx, &b.T, (size_t)nb * nk * nk));
    CHECK(dmalloc(ctx, &b.Tt, (size_t)nb * nk * nk));
    CHECK(dmalloc(ctx, &b.z, (size_t)nb * nk));
    CHECK(dmalloc(ctx, &b.sel, (size_t)nb * nk));
    CHECK(dmalloc(ctx, &b.bs, (size_t)nb));
    CHECK(dmalloc(ctx, &b.pick, (size_t)nb));
    CHECK(dmalloc(ctx, &b.cand_val, (size_t)nb * b.n_atiles * kTileCand));
    CHECK(dmalloc(ctx, &b.cand_idx, (size_t)nb * b.n_atiles * kTileCand));
    return CSMP_OK;
}

// G = A'A, Float64 products of the exactly promoted dictionary values, upper triangle (row <= column) of an N x N array:
// the option CSMP_OPT_BATCH_GRAM.  8 N^2 bytes (32 GiB at N = 65536) and 2 M N^2 / 2 flops on the Float64 matrix cores
// (k_gram, csmp_gram.hpp: the dictionary is its own "compact copy") -- once per dictionary, like the bf16 image.
static int batch_gram(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.gram_valid) return CSMP_OK;
    const int64_t N = ctx->N;
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)N * (size_t)N * sizeof(double);
    if (need + ((size_t)1 << 30) > free_b) return fail(ctx, CSMP_ENOMEM, "CSMP_OPT_BATCH_GRAM: 8 N^2 bytes of HBM are not available");
    HIPCHECK(hipMalloc((void**)&b.Gm, need));
    // k_gram tiles are 128 x 64 over np columns; np = N need not be a multiple of the tile: rows / columns >= np are clamped and
    // never stored.  One slice of the rows (no partials): rows_per_split = the whole column, which the kernel walks in blocks of
    // 16 rows -- a dictionary whose leading dimension is not a multiple of 16 goes through a zero-padded temporary copy.
    const int np = (int)N;
    const size_t es = ctx->dtype == CSMP_F32 ? 4 : 8;
    const int rows = (int)((ctx->M + 15) / 16 * 16);
    const void* src = ctx->dA;
    int64_t ldo = ctx->ld;
    DevTmp padded;
    if (ctx->ld % 16 != 0) {
        ldo = rows;
        HIPCHECK(padded.alloc((size_t)ldo * (size_t)N * es));
        HIPCHECK(hipMemsetAsync(padded.p, 0, (size_t)ldo * (size_t)N * es, ctx->stream));
        HIPCHECK(hipMemcpy2DAsync(padded.p, (size_t)ldo * es, ctx->dA, (size_t)ctx->ld * es, (size_t)ctx->M * es, (size_t)N, hipMemcpyDeviceToDevice,
                                  ctx->stream));
        src = padded.p;
    }
    const dim3 grid((unsigned)((np + kGramWgJ - 1) / kGramWgJ), (unsigned)((np + kGramWgI - 1) / kGramWgI), 1);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_gram<float>, grid, dim3(256), 0, ctx->stream, (const float*)src, ldo, np, rows, b.Gm);
    else
        hipLaunchKernelGGL(k_gram<double>, grid, dim3(256), 0, ctx->stream, (const double*)src, ldo, np, rows, b.Gm);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.Ng = N;
    b.gram_valid = true;
    return CSMP_OK;
}

template <typename TA>
static hipError_t b_pick_launch(csmp_ctx* ctx, hipStream_t stream, int sig0, int nsig, double eps, int check_eps, double cert_abs, double cert_rel,
                                int kwin) {
    Batch& b = ctx->bt;
    constexpr int U = sizeof(TA) == 4 ? 16 : 8;  // 64-lane chunks of a column in flight per wave (16 bytes per lane each)
    const size_t lds = b_pick_lds_bytes(ctx->Mv, (int)(16 / sizeof(TA)));
    auto kern = k_b_pick<TA, U>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nsig), dim3(256), lds, stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, (const float*)b.cand_val,
                       (const int*)b.cand_idx, b.n_atiles * kTileCand, (const int*)b.sel, b.bs, b.pick, (const double*)b.r, b.Mr, b.kcap, (int)ctx->M, eps,
                       check_eps, cert_abs, cert_rel, kwin, sig0);
    return hipGetLastError();
}
// DEPTH of the append kernel: columns whose loads are issued together (registers: DEPTH x NI x 16 bytes per lane)
template <typename TA, int NI, bool GRAM>
static hipError_t b_append_launch(csmp_ctx* ctx, hipStream_t stream, int sig0, int nsig) {
    Batch& b = ctx->bt;
    constexpr int DEPTH = (NI >= 8 || (sizeof(TA) == 8 && NI >= 4)) ? 2 : (NI >= 4 || sizeof(TA) == 8) ? 2 : 4;
    const size_t lds = b_append_lds_bytes(ctx->Mv, (int)(16 / sizeof(TA)), b.kcap);
    auto kern = k_b_append<TA, NI, DEPTH, GRAM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nsig), dim3(256), lds, stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, (const double*)b.Gm, b.Ng, (const BPick*)b.pick, b.T,
                       b.Tt, b.z, b.sel, b.bs, b.r, b.Mr, b.Rb, b.Mk, b.kcap, (int)ctx->M, sig0);
    return hipGetLastError();
}
template <typename TA>
static hipError_t b_step_dispatch(csmp_ctx* ctx, hipStream_t stream, int sig0, int nsig, double eps, int check_eps, double cert_abs, double cert_rel,
                                  int kwin, bool gram) {
    const int groups = (ctx->Mv + 1023) / 1024;
    hipError_t e = b_pick_launch<TA>(ctx, stream, sig0, nsig, eps, check_eps, cert_abs, cert_rel, kwin);
    if (e != hipSuccess) return e;
#define CSMP_BSTEP(NI)                                                                                                  \
    return gram ? b_append_launch<TA, NI, true>(ctx, stream, sig0, nsig) : b_append_launch<TA, NI, false>(ctx, stream, sig0, nsig);
    if (groups <= 1) { CSMP_BSTEP(1) }
    if (groups <= 2) { CSMP_BSTEP(2) }
    if (groups <= 4) { CSMP_BSTEP(4) }
    CSMP_BSTEP(8)
#undef CSMP_BSTEP
}

extern "C" int csmp_omp_batch_mfma(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                                   double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");
    if (!B || nsig < 1 || k < 1 || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "omp_batch_mfma: bad arguments");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (ctx->Mv > 8192) return fail(ctx, CSMP_ERANGE, "omp_batch_mfma: M > 8192 not supported (use csmp_omp_batch)");
    if (nsig > (1 << 20)) return fail(ctx, CSMP_ERANGE, "omp_batch_mfma: too many signals in one call");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    CHECK(batch_dict(ctx));
    CHECK(batch_ensure(ctx, (int)nsig, kc));
    CHECK(solver_ensure(ctx, kc, (int)k));  // the exact path re-solves flagged signals
    ctx->s.begun = false;
    Batch& b = ctx->bt;
    const bool gram = ctx->opt_batch_gram != 0;
    if (gram) CHECK(batch_gram(ctx));
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
    const int Bpad = (int)(((nsig + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT));  // whole 256-signal tiles
    if (b_dtype == CSMP_F32)
        hipLaunchKernelGGL(k_b_init<float>, dim3(Bpad), dim3(256), 0, ctx->stream, (const float*)dB, ldB, (int)ctx->M, (int)nsig, b.r, b.b, b.Mr, b.Rb, b.Mk, b.bs);
    else
        hipLaunchKernelGGL(k_b_init<double>, dim3(Bpad), dim3(256), 0, ctx->stream, (const double*)dB, ldB, (int)ctx->M, (int)nsig, b.r, b.b, b.Mr, b.Rb, b.Mk, b.bs);
    HIPCHECK(hipGetLastError());
    // Screening error bound  | |<a_n, r>| - s_n | <= cert_abs |r| + cert_rel s_n  (k_b_pick, csmp_batched.hpp).
    // Statistical (default): 8 standard deviations of the bf16 rounding model -- independent roundings of the M products,
    // sigma = sqrt(2/3) 2^-9 max|A_ij| |r| -- PLUS the fully coherent case the independent model misses: an operand whose
    // entries all round the same way is a scaled operand, (1 + a)(1 + b) s with |a|, |b| <= 2^-8 (few-valued and one-magnitude
    // dictionaries: every entry of a column rounds alike), i.e. 2^-7 s, plus the 2^-15 the packed candidate keys drop.
    // Rigorous (CSMP_OPT_BATCH_CERT = 1): |<a,r> - screened| <= (2^-7 (1 + 2^-9) + Mk 2^-24) |a|_2 |r|_2 (bf16 unit roundoff
    // 2^-8 on both operands, Float32 accumulation) with the largest column norm, and the key truncation: a proof, about nine
    // times wider on a Gaussian dictionary -- the window holds more candidates (64 instead of 16), more signals overflow it.
    double cert_abs, cert_rel;
    int kwin;
    if (ctx->opt_batch_cert == 1) {
        CHECK(batch_colnorm(ctx));
        cert_abs = (std::ldexp(1.0, -7) * (1.0 + std::ldexp(1.0, -9)) + (double)b.Mk * std::ldexp(1.0, -24)) * (double)b.anorm_host;
        cert_rel = std::ldexp(1.0, -14);
        kwin = kWinMax;  // 128
    } else {
        cert_abs = 8.0 * std::sqrt(2.0 / 3.0) * std::ldexp(1.0, -9) * (double)b.amax_host;
        cert_rel = std::ldexp(1.0, -7) * 1.01 + std::ldexp(1.0, -14);
        kwin = kWinMax / 2;  // 64
    }
    if (ctx->opt_batch_window > 0) kwin = std::min<int>(kWinMax, (int)ctx->opt_batch_window);
    if (tune_env("CSMP_CERT_NOREL")) cert_rel = std::ldexp(1.0, -14);  // (experiments build: the round-2 bound, for tools/probe_structured.py)
    b.last_mode = kScreen256p;
    b.last_streams = 1;
    b.last_screen_signals = Bpad;
    for (int64_t t = 0; t < k; ++t) {
        const bool timed = ctx->prof;  // (HIP events around the screening launch)
        if (timed) {
            if (ctx->ev2_used == ctx->ev2.size()) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); ctx->ev2.push_back(e); }
            HIPCHECK(hipEventRecord(ctx->ev2[ctx->ev2_used++], ctx->stream));
        }
        HIPCHECK(launch_screen(ctx->stream, (const __bf16*)b.Ab, (const __bf16*)b.Rb, b.Mk, b.n_atiles, Bpad / kBT, ctx->N, b.cand_val, b.cand_idx));
        if (timed) {
            if (ctx->ev2_used == ctx->ev2.size()) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); ctx->ev2.push_back(e); }
            HIPCHECK(hipEventRecord(ctx->ev2[ctx->ev2_used++], ctx->stream));
        }
        hipError_t e = ctx->dtype == CSMP_F32 ? b_step_dispatch<float>(ctx, ctx->stream, 0, (int)nsig, eps, t > 0, cert_abs, cert_rel, kwin, gram)
                                              : b_step_dispatch<double>(ctx, ctx->stream, 0, (int)nsig, eps, t > 0, cert_abs, cert_rel, kwin, gram);
        HIPCHECK(e);
    }
    hipLaunchKernelGGL(k_b_finish, dim3((int)nsig), dim3(256), (size_t)(b.kcap + 2) * 8, ctx->stream, (const double*)b.T,
                       (const double*)b.z, (const int*)b.sel, (const BState*)b.bs, b.kcap, (int)k, d_idx, d_val, d_nnz);
    HIPCHECK(hipGetLastError());
    // signals whose screen could not be certified (or whose support turned ill-conditioned) are
    // re-solved by the exact single-signal path
    std::vector<BState> hs((size_t)nsig);
    HIPCHECK(hipMemcpyAsync(hs.data(), b.bs, (size_t)nsig * sizeof(BState), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.last_signals = nsig;
    b.last_resolved = b.last_uncertain = b.last_illcond = 0;
    int rc = CSMP_OK;
    for (int64_t sgn = 0; sgn < nsig && rc == CSMP_OK; ++sgn) {
        if (!hs[sgn].uncertain && !hs[sgn].illcond) continue;
        b.last_resolved += 1;
        b.last_uncertain += hs[sgn].uncertain ? 1 : 0;
        b.last_illcond += hs[sgn].illcond ? 1 : 0;
        if (tune_env("CSMP_BATCH_DEBUG"))
            fprintf(stderr, "signal %lld: uncertain %d illcond %d nsel %d | first failed certificate at step %d: window %d (cap %d), best exact %.6f, bound %.6f, top screened %.6f, |r| %.4f\n",
                    (long long)sgn, hs[sgn].uncertain, hs[sgn].illcond, hs[sgn].nsel, hs[sgn].unc_step, hs[sgn].unc_nall, kwin, hs[sgn].unc_best,
                    hs[sgn].unc_cb, hs[sgn].unc_s1, std::sqrt(hs[sgn].rnorm2));
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        rc = b_dtype == CSMP_F32 ? init_from_device_t<float>(ctx, (const float*)col)
                                 : init_from_device_t<double>(ctx, (const double*)col);
        for (int64_t t = 0; t < k && rc == CSMP_OK; ++t) rc = omp_step(ctx, eps, t > 0, false);
        if (rc == CSMP_OK) rc = launch_finish(ctx, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, nullptr, (int)k);
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

// name of the screening kernel the last csmp_omp_batch_mfma call used (for the bench's roofline line)
extern "C" const char* csmp_batch_screen_kernel(const csmp_ctx* ctx) { return ctx ? screen_kernel_name(ctx->bt.last_mode) : ""; }

// how the last csmp_omp_batch_mfma call was laid out: signal columns per screening launch, streams used
extern "C" int csmp_batch_layout(const csmp_ctx* ctx, int64_t* screen_signals, int* streams) {
    if (!ctx) return CSMP_EINVAL;
    if (screen_signals) *screen_signals = ctx->bt.last_screen_signals;
    if (streams) *streams = ctx->bt.last_streams;
    return CSMP_OK;
}

extern "C" int csmp_batch_stats(csmp_ctx* ctx, int64_t* signals, int64_t* resolved_exactly, int64_t* uncertain, int64_t* illcond,
                                int64_t* screen_launches, double* screen_ms) {
    if (!ctx) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 1 < ctx->ev2_used; i += 2) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev2[i], ctx->ev2[i + 1]));
        ctx->prof2_ms += ms;
        ctx->prof2_n += 1;
    }
    ctx->ev2_used = 0;
    if (signals) *signals = ctx->bt.last_signals;
    if (resolved_exactly) *resolved_exactly = ctx->bt.last_resolved;
    if (uncertain) *uncertain = ctx->bt.last_uncertain;
    if (illcond) *illcond = ctx->bt.last_illcond;
    if (screen_launches) *screen_launches = ctx->prof2_n;
    if (screen_ms) *screen_ms = ctx->prof2_ms;
    ctx->prof2_n = 0;
    ctx->prof2_ms = 0.0;
    return CSMP_OK;
}

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
