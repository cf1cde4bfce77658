// host/ctx.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// context, solver and batch state; error / allocation helpers.
static std::string g_create_err;
static constexpr int kRsEqCap = 4096;   // entries of the bucket list (exact ties beyond it: in-order scan)
static constexpr int kRsSettle = 256;   // a bucket this small ends the passes (k_rs_finish ranks it in LDS)

struct Solver {
    std::vector<int> hpos;           // host-side atom -> position marks (whole-set solves' set algebra: host/gomp_sp.hpp)
    std::vector<unsigned> hstamp;
    unsigned hgen = 0;
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
    float* scr_val = nullptr;  // screened sweep: the sweep workgroups' candidates (4 per workgroup)
    int* scr_idx = nullptr;
    unsigned long long* scr_cb = nullptr;  // sp_select_screened: the bound of everything that was not rescored (bits of a double)
    int* scr_flag = nullptr;               // ... and 1 when the selection could not be certified
    unsigned* claim = nullptr;  // sweep_body_dyn's column pools: two sets of claim_words words, used in turn (claim_par: the next launch's)
    int claim_par = 0;
    size_t claim_words = 0;
    unsigned* scr_tickets = nullptr;  // k_sweep_bf16's ticket counters, one per partition of workgroups, kScrTicketStride words apart
    double* spill = nullptr;  // supports beyond the LDS append kernels' ~3900 columns: their five support-length vectors per workgroup (launch_append)
    size_t spill_cap = 0;
    int jh_last = 0;     // jh used by the most recent k_qr1 stage (the matching k_qr2 stage reuses it)
    // multi-column append (csmp_block.hpp), allocated on first use
    double *Apan = nullptr, *Vpan = nullptr, *PB1 = nullptr, *W1b = nullptr, *PG = nullptr, *Gsum = nullptr;
    int* pan_atoms = nullptr;
    int blk_kcap = 0;
    int* sigflags = nullptr;  // per-signal stop flags of a batch (optimistic-chain verification)
    double *rho2 = nullptr, *dvec = nullptr;  // forward regression: OLS rescaling and δ² scores (N each), allocated on first use
    int fr_grid = 0;
    double *frg1 = nullptr, *frg2 = nullptr, *frq = nullptr;  // tall dictionaries (launch_fr_pass_tall): g = A'q per direction, the last Q column
    // column removal (csmp_downdate.hpp), allocated on first use
    double *R2 = nullptr, *Gdel = nullptr, *qdrop = nullptr, *qsave = nullptr, *bwd = nullptr, *bwd_coef = nullptr, *bwd_info = nullptr;
    int *delmeta = nullptr, *delpos = nullptr;
    // explicit inverse factor of the two-stage solvers (csmp_tinv.hpp)
    double *T = nullptr, *T2 = nullptr, *tpd = nullptr, *tpn = nullptr;
    int* tmeta = nullptr;
    double *swapH = nullptr, *swapv = nullptr;  // OMPR's exchanges on the inverse Gram matrix (csmp_swap.hpp): H (kcap x kcap) and its vectors
    size_t swapv_cap = 0;
    int sigcap = 0;
    // whole-set least squares (csmp_gram.hpp), allocated on first use
    double *Gm = nullptr, *Gpart = nullptr, *gdiag = nullptr, *rpart = nullptr, *Dfac = nullptr, *Gm2 = nullptr, *ytmp = nullptr;
    // the last bordered Gram matrix that was COMPUTED (before its factorisation), for the sets that are subsets of it: SP solves
    // on T = S + k new atoms and then on the k atoms of T it keeps -- the second system is a principal submatrix of the first
    double *Gkeep = nullptr, *gdkeep = nullptr, *rhs_part = nullptr, *rn2part = nullptr;
    int* kpos = nullptr;
    // bordered extension (ls_gram_extend_t): the second kept matrix (ping-pong), W's workspace, and the set whose factor the slot holds
    double *Gkeep2 = nullptr, *gdkeep2 = nullptr, *Wb = nullptr, *Gin = nullptr;
    int64_t fac_gen = 0, tt_gen = -1;  // (R^-1)' of the factor confirmed as number tt_gen sits in Gm's augmented columns:
    int tt_col0 = 0, tt_ld = 0;        //   Gm + tt_col0 * tt_ld, leading dimension tt_ld
    bool tt_pending = false;
    std::vector<int> fac_cols, fac_pending;  // factor order of the last CONFIRMED whole-set solve / of the one in flight
    bool fac_valid = false;                  // R, z, sel hold fac_cols' factor on the current b (dropped by anything that rewrites them)
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
    bool meta_valid = false;  // Mk, Npad, n_atiles, amax_host are those of the current dictionary
    bool ab_valid = false;
    bool ab_borrowed = false;  // a twin sweeping its parent's image (host/screened.hpp): not this context's to free
    int Mk = 0;
    int64_t Npad = 0;
    int n_atiles = 0;
    float* amax = nullptr;  // max |A_ij| (device scalar) for the screening error bound
    float amax_host = 0.f;
    float arms_host = 0.f;   // root mean square of the dictionary's entries (how flat it is: the int8 screen's applicability)
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
    // int8 screen (CSMP_OPT_BATCH_SCREEN = 1): the dictionary as int8 [Npad][Mk8] under one step, the residual images, the
    // factor of every signal's integer dot products
    signed char* A8 = nullptr;
    bool a8_valid = false;
    bool a8_borrowed = false;
    int Mk8 = 0;
    float astep = 0.f;
    signed char* R8 = nullptr;
    float* sigscale = nullptr;
    // binary16 screen (CSMP_OPT_BATCH_SCREEN = 3, the default): the dictionary as binary16 [Npad][Mk] of ascale16 * A (a power of two
    // that puts max|A| in [2^14, 2^15)); the residual images live in Rb under per-signal scales
    _Float16* Ah = nullptr;
    bool ah_valid = false;
    bool ah_borrowed = false;  // (a twin sweeping its parent's image)
    float ascale16 = 1.f;
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
    int kind = 0;  // how p is given back: 0 hipFree (HBM), 1 hipHostFree (library-owned page-locked memory), 2 hipHostUnregister (the caller's)
};

struct csmp_ctx;
// Subspace Pursuit as a resumable job: see host/gomp_sp.hpp
struct SpJob {
    csmp_ctx* c = nullptr;
    int64_t k = 0, maxiter = 0, it = 0;
    double delta = 0.0, resnorm = 0.0, oldnorm = 0.0;
    std::vector<int64_t> xi;  // the support (sorted) and its coefficients
    std::vector<double> xv;
    std::vector<int64_t> prev_xi;  // ... as they were before the current acquisition
    std::vector<double> prev_xv;
    std::vector<int> cols;    // the set whose least squares is in flight
    enum Phase { IDLE, SELECT, LS_FIRST, LS_UNION, LS_PRUNED, DONE } phase = IDLE;
    bool screened = false, sel_screened = false, want_norm = false, gram_inflight = false;
    hipEvent_t ev = nullptr;
    int rc = CSMP_OK;
};


struct csmp_ctx {
    SpJob spjob;  // the Subspace Pursuit solve this context is carrying (csmp_sp, csmp_sp_batch, the SP functor)
    std::shared_ptr<void> omprjob;  // the OMPR object of csmp_ompr / the OMPR functor (OmprJob, host/twostage.hpp), made on first use
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
    hipEvent_t* gate = nullptr;  // csmp_sp_batch: the completion of the most recently enqueued sweep of ANY solve in flight (owned by the parent)
    hipEvent_t ev_gate = nullptr;
    bool streamed = false;  // the dictionary lives in HOST memory mapped into the device's address space: every sweep crosses the host link
    csmp_ctx* twins[3] = {nullptr, nullptr, nullptr};  // clones on their own streams: the other solves in flight of csmp_gomp_batch / csmp_sp_batch
    int opt_in_flight = 3;        // CSMP_OPT_SOLVES_IN_FLIGHT (csmp_sp_batch)
    hipEvent_t ev_twin = nullptr;
    hipStream_t stream_b = nullptr;  // second stream of the batched path (half-batches out of phase)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_off = nullptr;
    int dtype = CSMP_F32;
    int64_t M = 0, N = 0, ld = 0;
    int64_t col_offset = 0;  // global index of local column 0 (column-sharded OMP; 0 otherwise)
    int Mv = 0;  // M rounded up to the 16-byte vector (zero rows in our own copy)
    int sweep_grid = 0, sweep_U = 1;
    // the product sweep's configuration for this dictionary (configure_sweep): sweep_U loads per unit (16 / 8 / 4; the ring holds 32)
    bool sweep_ph = false;  // the residual is staged in phases of sweep_KP rows (it exceeds the LDS)
    int sweep_KP = 0;       // rows of the residual image in the LDS
    int sweep_pcap = 0;     // phases: columns per wave whose partial sums the LDS holds (sweep_body_ph)
    int tick_grid = 0;      // sweep workgroups inside the tick kernel
    int tune_sweep_grid = 0, tune_sweep_U = 0;
    bool sweep_dyn = false;  // the product sweep hands its columns out at run time (k_sweep_dyn, the DYN tick)
    int tune_sweep_dyn = 0;  // csmp_tune: 1 = the columns handed out at run time where one image holds the residual (measured slower: DESIGN.md section 0)
    int tune_pair_lds_kib = 0;  // csmp_tune: dynamic LDS (KiB) of the ticks of two pipelines side by side, 0 = kPairLdsKiB (one workgroup per CU)
    int tune_pair_split = 0;    // csmp_tune: 1 = two pipelines side by side keep the fused tick (one launch), default: append stages and sweep in two launches
    int tune_fail_alloc = 0;    // csmp_tune (test hook): the n-th device allocation of a solver slot from now fails (dmalloc)
    int tune_pipelines = 0;  // csmp_tune: 1 = csmp_omp_batch keeps ONE pipeline of three signals (default: two side by side from two signals on)
    int claim_pools = 8;     // counters a workgroup of the dynamic sweep finds empty in a row before it stops (its own, then the following workgroups')
    int tune_rebuild_direct = 0;  // csmp_tune: the oblivious start's Q'A pass reads its directions from L2 (k_fr_rebuild) instead of the LDS
    int tune_swap_refuse = 0;  // csmp_tune: OMPR's inverse-Gram exchanges fail their guard (tests walk the fallback to the QR path)
    int tune_diag_split = 0;  // csmp_tune: fused kernels run as one launch per part (a kernel trace then shows the parts)
    int64_t tune_batch_budget_mib = 0;  // csmp_tune: HBM the batched path's per-signal state may take (MiB), 0 = what is free  // csmp_tune (include/csmp_internal.h): measurement overrides, 0 = automatic
    // options (csmp_set_option, include/csmp.h)
    void* comm = nullptr;          // ncclComm_t of the signal-sharded solve (csmp_comm_init, host/rccl.hpp); this rank and the group's size
    int comm_rank = 0, comm_world = 1;
    int opt_batch_cert = 1;        // CSMP_OPT_BATCH_CERT: 1 rigorous (default), 0 statistical (opt-in)
    int opt_batch_gram = 0;        // CSMP_OPT_BATCH_GRAM: resident G = A'A for csmp_omp_batch_mfma
    int opt_batch_window = 0;      // CSMP_OPT_BATCH_WINDOW: rescoring window capacity, 0 = default
    int64_t scr_solves = 0, scr_fallbacks = 0;  // screened solves made / repeated with the exact sweep (csmp_screened_stats)
    int scr_grid = 0;                           // workgroups of k_sweep_bf16
    double scr_cert_abs = 0.0, scr_cert_rel = 0.0, scr_cert_abs2 = 0.0;
    int scr_kwin = 0;
    int scr_image = 1;       // the image the screened sweeps of this context read: 1 bf16, 2 int8 (option 2 on a flat dictionary)
    bool scr_lone = false;   // the solve in progress runs alone on the GPU (csmp_omp, csmp_mp): the pick kernel may take a whole CU
    int opt_screened = 0;         // CSMP_OPT_SCREENED_SWEEP: csmp_omp / csmp_omp_batch / csmp_gomp sweep the bf16 image and certify (csmp_screened.hpp)
    int opt_batch_screen = 3;     // CSMP_OPT_BATCH_SCREEN: 3 (default) binary16 operands, 0 bf16, 1 int8, 2 int8 where the dictionary is flat (int8: statistical certificate only)
    size_t sweep_lds = 0;
    int short_cpu = 0, short_nch = 0, short_KP = 0;  // k_sweep_short (stand-alone sweep of short columns): columns per reduction, chunks per column, image rows; 0 = the one-column body
    size_t short_lds = 0;
    int tune_phase_rows = 0;   // csmp_tune: most rows of a stage of the phased sweep (0: what the LDS holds)
    int tune_screen_static = 0;  // csmp_tune: 1 = the screened sweep deals its column groups out statically (a measurement switch; tickets are the default)
    int tune_sweep_short = 0;  // csmp_tune: 1 = the one-column body for every shape
    size_t sweep_lds_req = 0;    // the stand-alone sweep's LDS request when larger than sweep_lds (residency control)
    int tune_sweep_lds_kib = 0;  // csmp_tune
    Solver s;        // the ACTIVE solver slot (see activate_slot)
    Solver park[3];  // parked slots (park[active] is unused): three signals are pipelined in csmp_omp_batch
    int active = 0;
    bool pipeline = true;
    int tick_nblk = 0;       // absolute override of the sweep workgroup count (CSMP_TICK_NBLK), 0 = per-CU rule
    bool tick_sweep_first = false;  // dispatch the sweep workgroups ahead of the append stages (CSMP_TICK_ORDER=1)
    Batch bt;
    // profiling
    bool prof = false;
    int prof_every = 1;       // time every n-th sweep launch only (an event pair costs a few us of stream time)
    int64_t prof_count = 0;
    int64_t prof_first = -1, prof_last = -1;  // launch numbers of the first and the last sampled launch since the events were last read
    hipEvent_t prof_ref = nullptr;            // recorded by csmp_profile_enable: the origin of csmp_profile_window's clock
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
            (void)hipGetLastError(); /* reported HERE: left pending, the next launch's check would report it again, as its own */ \
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

// The library reads no environment variable: every behavioural choice is an argument or a csmp_set_option key (include/csmp.h).
static int fail(csmp_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    if (code == CSMP_EHIP || code == CSMP_ENOMEM) (void)hipGetLastError();  // (a HIP failure is reported once: not again by the next launch's check)
    return code;
}
static int fail(csmp_ctx* ctx, int code, const std::string& msg) { return fail(ctx, code, msg.c_str()); }

template <typename T>
static int dmalloc(csmp_ctx* ctx, T** p, size_t n) {
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    // test hook (csmp_tune, CSMP_TUNE_FAIL_ALLOC = n): the n-th allocation from now asks for an impossible size -- a REAL hipMalloc
    // failure, with hipErrorOutOfMemory left pending for the next hipGetLastError, as a full device would produce it
    if (ctx->tune_fail_alloc > 0 && --ctx->tune_fail_alloc == 0) bytes = (size_t)1 << 60;
    HIPCHECK(hipMalloc((void**)p, bytes));
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

// defined in host/batched.hpp and host/screened.hpp (included later); used by the omp drivers
static int batch_dict(csmp_ctx* ctx);
static int batch_dict8(csmp_ctx* ctx);
static int batch_dict16(csmp_ctx* ctx);
static int batch_meta(csmp_ctx* ctx);
static int batch_colnorm(csmp_ctx* ctx);
static int screened_ensure(csmp_ctx* ctx);
// the screened sweep is asked for and this dictionary fits its kernels (else: the exact sweep, silently -- the results are the same)
static bool screened_on(const csmp_ctx* ctx) {
    return ctx->opt_screened != 0 && ctx->dA && ctx->Mv <= 16384 && ctx->N >= 1;  // (pick kernels: the Float64 residual image in LDS, 8 M bytes)
}
static int twins_ensure(csmp_ctx* ctx, int n);
static int gomp_update_screened(csmp_ctx* ctx, int64_t l, double eps, int check_eps, int skipmask, bool block);
static int screened_ensure_pair(csmp_ctx* ctx, csmp_ctx* twin);
static int launch_block_appends(csmp_ctx* ctx, int n, int skipmask);
static int sp_select_screened(csmp_ctx* ctx, int k);
static int mp_step_screened(csmp_ctx* ctx);
static int ompr_sweep_screened(csmp_ctx* ctx, const int* cols_dev, int n);
static int launch_topS(csmp_ctx* ctx, int S);
static int omp_step_screened(csmp_ctx* ctx, double eps, int check_eps, bool optimistic);
static int omp_batch_screened(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k, double eps,
                              int64_t* idx, double* val, int64_t* nnz, int out_loc);
