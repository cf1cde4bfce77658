/*
 * csmp.h -- C ABI of libcsmp.so: the MI355X (gfx950) matching-pursuit path of
 * CompressedSensing.jl (mp / omp / gomp / sp and their step primitives).
 *
 * The reference is pure Julia and has no FFI seam of its own; the seam is the set of methods
 * dispatched on AbstractMatchingPursuit plus the UpdatableQR API they call.
 * Each entry point below names the reference interface it replaces (paths relative to the
 * reference repository).  A Julia host binds these with `ccall` (see INTEGRATION.md and
 * compressedsensing.jl_amd/julia/CompressedSensingAMD.jl); the Python host mirror binds them
 * with ctypes (compressedsensing.jl_amd/_lib.py).
 *
 * Conventions
 *   - A: dense column-major, M rows (signal length) x N columns (atoms), leading dimension ldA
 *     in elements, element type CSMP_F32 or CSMP_F64.  NOTE the reference names these
 *     `n, m = size(A)` (src/matchingpursuit.jl:20); BASELINE.json uses m=rows, n=cols.
 *   - All selection arithmetic is Float64 on the exactly promoted dictionary values, so an f32
 *     dictionary yields the support a Float64 run of the reference would select on the same values.
 *   - Results follow SparseVector{Float64,Int64} (src/matchingpursuit.jl:76): indices sorted
 *     ascending (0-BASED here; the Julia wrapper adds 1), values aligned, nnz may be < k.
 *   - Every call returns a status (0 = ok, negative = error);
 *     csmp_last_error() gives the text of an error.
 *   - A ctx is bound to one GPU and one HIP stream and is not thread-safe (the reference is
 *     single-threaded too).  Caller owns every buffer passed in; the library owns device memory
 *     behind the opaque ctx.  `loc` says where a caller buffer lives: CSMP_HOST or CSMP_DEVICE.
 */
#ifndef CSMP_H
#define CSMP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CSMP_OK 0
#define CSMP_EINVAL (-1) /* eps < 0 (reference: throw("eps has to be non-negative"), src/matchingpursuit.jl:74,127), bad argument */
#define CSMP_EDIM (-2)   /* length(b) != size(A,1) and the like */
#define CSMP_ERANGE (-3) /* 2k > M for SP (reference: error(...), src/twostage.jl:55); k out of range */
#define CSMP_EHIP (-4)   /* HIP runtime failure / no gfx950 device */
#define CSMP_ESTATE (-5) /* no dictionary set / no solver begun */
#define CSMP_ENOMEM (-6)
#define CSMP_ERCCL (-7)  /* RCCL could not be loaded, or a communicator / collective call failed */
#define CSMP_EIO (-8)    /* a dictionary file could not be read or written */
#define CSMP_F32 0
#define CSMP_F64 1
#define CSMP_HOST 0
#define CSMP_DEVICE 1
#define CSMP_HOST_STREAMED 2 /* csmp_set_dictionary / csmp_set_dictionary_file only: the dictionary stays in host memory (below) */

#define CSMP_ALGO_MP 0
#define CSMP_ALGO_OMP 1
#define CSMP_ALGO_GOMP 2
#define CSMP_ALGO_FR 3 /* update!(P::FR, x): src/forward.jl:88-95 */
#define CSMP_ALGO_SP 4   /* SP(A,b,k), update!(P::SP, x): src/twostage.jl:42-83 */
#define CSMP_ALGO_OMPR 5 /* OMPR(A,b,k), update!(P::OMPR, x) with eta = 1: src/twostage.jl:110-180 */

/* why a solve stopped early (csmp_solver_state: *stop) */
#define CSMP_STOP_NONE 0
#define CSMP_STOP_EPS 1    /* norm(residual) < eps: src/matchingpursuit.jl:79,132 */
#define CSMP_STOP_STAG 2   /* arg-max atom already selected: src/matchingpursuit.jl:66 */
#define CSMP_STOP_FULL 4   /* nnz(x) == size(A,1): src/matchingpursuit.jl:63,117 */

typedef struct csmp_ctx csmp_ctx;

/* ------------------------------------------------------------------ lifetime */
int csmp_version(void);
int csmp_create(csmp_ctx **out, int device_id);
int csmp_destroy(csmp_ctx *ctx);
const char *csmp_last_error(const csmp_ctx *ctx); /* ctx may be NULL: last create error */
/* borrow a caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int csmp_set_stream(csmp_ctx *ctx, void *hip_stream);
int csmp_sync(csmp_ctx *ctx);
/* name (<=255 chars), compute-unit count and total HBM bytes of the bound device */
int csmp_device_info(csmp_ctx *ctx, char *name, int name_len, int *compute_units, int64_t *hbm_bytes);

/* ------------------------------------------------------------------ dictionary
 * Replaces the `A` field of MP/OMP/GOMP/SP (src/matchingpursuit.jl:10-24,44-60,95-114;
 * src/twostage.jl:42-61).  Uploaded once, stays resident in HBM.  A CSMP_DEVICE pointer that is
 * 16-byte aligned with M and ldA multiples of 16 bytes is borrowed without a copy.
 * The entries must be FINITE.  The sweeps pad a column's last load by re-reading its own last 16 bytes against zeros of the
 * residual image (no predicate in the load stream): an Inf or NaN there gives 0 * Inf = NaN for THAT column's product, where
 * the reference's mul! would leave an Inf -- no other column is touched, and a column with a non-finite entry has no
 * meaningful correlation in the reference either. */
int csmp_set_dictionary(csmp_ctx *ctx, const void *A, int64_t M, int64_t N, int64_t ldA, int dtype, int loc);
/* A dictionary LARGER THAN HBM (not in the reference, whose A is whatever the host's memory holds):
 * loc = CSMP_HOST_STREAMED leaves A in HOST memory, page-locked and mapped into the device's address space, and every kernel that
 * reads A -- the sweeps, the column gathers of the appends -- reads it over the host link (PCIe 5 x16: ~50 GB/s against
 * 6.6 TB/s from HBM; the results are those of the resident dictionary, bit for bit).  A 16-byte aligned array with M and ldA
 * multiples of 16 bytes is registered where it lies (hipHostRegister: no second copy; the caller keeps it alive and unchanged
 * until the context is destroyed or given another dictionary); anything else is copied into page-locked memory of the library's.
 *
 * Dictionary files: a 64-byte header ("CSMPDICT", u32 version 1, u32 dtype, i64 M, i64 N, i64 ld, 24 zero bytes) followed by
 * the N columns, ld elements each (M rounded up to 16 bytes, zero padded), little-endian.  csmp_dictionary_file_write writes one
 * from host memory; csmp_set_dictionary_file reads one to where it will live -- HBM (loc = CSMP_DEVICE: chunked upload, the file
 * never sits in host memory whole) or mapped host memory (loc = CSMP_HOST_STREAMED). */
int csmp_dictionary_file_write(const char *path, const void *A, int64_t M, int64_t N, int64_t ldA, int dtype);
int csmp_dictionary_file_info(const char *path, int64_t *M, int64_t *N, int *dtype);
int csmp_set_dictionary_file(csmp_ctx *ctx, const char *path, int loc);
/* A second context on the same GPU that borrows (does not copy) src's resident dictionary: the reference's P
 * objects are independent of one another and share only A -- P1 = OMP(A, b1); P2 = OMP(A, b2)
 * (src/matchingpursuit.jl:44-60) -- so every step-level solver (csmp_solver_begin) that must live beside
 * another one gets a clone.  A dictionary the library copied (host pointer, or an unaligned device pointer) is reference
 * counted: it lives until the last context holding it is destroyed or given another dictionary, whatever the order.  A
 * BORROWED device pointer (zero-copy, see csmp_set_dictionary) stays the caller's to keep alive. */
int csmp_clone(csmp_ctx *src, csmp_ctx **out);

/* ------------------------------------------------------------------ drivers (synchronous)
 * b: length M, element type b_dtype, host memory.  idx/val/order: caller arrays of capacity
 * given per function; *nnz receives the count.  order (may be NULL): atoms in selection order. */

/* mp(A,b,k,x): src/matchingpursuit.jl:26-40.  x0 = optional warm start (idx0 sorted or not).
 * idx/val capacity k + nnz0. */
int csmp_mp(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, const int64_t *idx0, const double *val0,
            int64_t nnz0, int64_t *idx, double *val, int64_t *nnz);

/* omp(A,b,eps,k): src/matchingpursuit.jl:62-82 (update! :62-70).  The caller supplies eps
 * (the Julia/Python wrappers default it to eps(eltype(A)): :85,89).  Capacity k. */
int csmp_omp(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, double eps, int64_t *idx, double *val,
             int64_t *nnz, int64_t *order);

/* gomp(A,b,l,eps,k): src/matchingpursuit.jl:116-139, including the remainder step :134-137.
 * Capacity k + l. */
int csmp_gomp(csmp_ctx *ctx, const void *b, int b_dtype, int64_t l, int64_t k, double eps, int64_t *idx,
              double *val, int64_t *nnz, int64_t *order);

/* sp(A,b,k,delta;maxiter): src/twostage.jl:54-107.  maxiter < 0 selects the default 16k.
 * Capacity 2k.  *iters (may be NULL) = number of update! calls made. */
int csmp_sp(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, double delta, int64_t maxiter, int64_t *idx,
            double *val, int64_t *nnz, int64_t *iters);

/* ompr(A,b,k,delta;maxiter): OMP with replacement, src/twostage.jl:110-202, with x starting empty
 * (the support is filled by oblivious_acquisition!, src/matchingpursuit.jl:207-216).  maxiter < 0
 * selects the default size(A,1).  Capacity k.  *iters (may be NULL) = number of update! calls. */
int csmp_ompr(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, double delta, int64_t maxiter, int64_t *idx,
              double *val, int64_t *nnz, int64_t *iters);

/* fr(A,b,max_eps,min_delta,k) = ols = oomp = ormp: forward regression / orthogonal least squares,
 * src/forward.jl:44-54 (forward_step! :56-73, forward_δ! :75-82, ols_rescaling! :99-114), with x
 * starting empty.  Each step adds the atom maximising <a_j,r>^2 / (|a_j|^2 - |Q_S' a_j|^2); stops
 * when norm(r) <= max_eps, when the best score does not exceed min_delta^2, or at k atoms / nnz = M.
 * Capacity k.  (Any M: columns whose two LDS images, 16 M bytes, exceed the LDS -- M beyond ~10 000 -- are swept once per image.) */
int csmp_fr(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, double max_eps, double min_delta, int64_t *idx,
            double *val, int64_t *nnz, int64_t *order);
/* P.δ² of the most recent forward-regression step (src/forward.jl:11,75-82; foba reads its maximum,
 * src/stepwise.jl:52): delta2 receives N Float64 scores (host). */
int csmp_fr_scores(csmp_ctx *ctx, double *delta2);

/* srr(A,b,k,delta; maxiter,initialization,l): stepwise regression with replacement, src/twostage.jl:3-33,
 * x starting empty.  initialization 1 = oblivious_acquisition! (src/matchingpursuit.jl:207-216),
 * 2 = k forward-regression steps; each iteration takes l forward steps (src/forward.jl:56-73) and then
 * backward steps (src/backward.jl:51-83) until k atoms remain.  maxiter < 0 selects the default 4k.
 * Capacity k + l (at most 4095).  *iters (may be NULL) = iterations made. */
int csmp_srr(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, double delta, int64_t maxiter, int initialization,
             int64_t l, int64_t *idx, double *val, int64_t *nnz, int64_t *iters);
/* srr with initialization = 3 (random_acquisition!, src/matchingpursuit.jl:195-204; src/twostage.jl:14-16): init[0..k) are the k
 * distinct atoms of the initial support (0-based, any order).  The reference draws them with sample(1:n, k, replace = false) from
 * the host language's RNG; here the host draws and passes them, everything after the draw is the reference's.  The same call is
 * the warm start srr(A, b, k, delta, x) of a full support, nnz(x) == k (src/twostage.jl:7-9: P is built on x.nzind and the
 * acquisition of k - nnz(x) = 0 further atoms changes nothing). */
int csmp_srr_from(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, double delta, int64_t maxiter, const int64_t *init,
                  int64_t l, int64_t *idx, double *val, int64_t *nnz, int64_t *iters);

/* rmp(A,b,delta,maxiter) (src/stepwise.jl:5-26), rmp(A,b,k) (:32-43) and foba(A,b,delta) (:47-56), x
 * starting empty: loops over forward_step! (src/forward.jl:56-73) and backward_step!
 * (src/backward.jl:51-67) of the StepwiseRegression object.  kmax (<= 0: min(M, N, 4095)) is the largest
 * support the forward stage may build; reaching it below min(M,N) is CSMP_ERANGE.  Capacity of idx/val:
 * kmax.  maxiter < 0 selects the default 1. */
int csmp_rmp_delta(csmp_ctx *ctx, const void *b, int b_dtype, double delta, int64_t maxiter, int64_t kmax,
                   int64_t *idx, double *val, int64_t *nnz);
int csmp_rmp_k(csmp_ctx *ctx, const void *b, int b_dtype, int64_t k, int64_t kmax, int64_t *idx, double *val,
               int64_t *nnz);
int csmp_foba(csmp_ctx *ctx, const void *b, int b_dtype, double delta, int64_t kmax, int64_t *idx, double *val,
              int64_t *nnz);

/* br(A,b,max_eps,max_delta,k) = fbr (src/backward.jl:27-35,154-162) and, with lace != 0,
 * lace(A,b,eps,delta,k) (:233-270): backward regression from the least-squares solution on all N <= M
 * columns (N <= 4095).  Infinite thresholds are passed as HUGE_VAL.  Capacity of idx/val: N. */
int csmp_br(csmp_ctx *ctx, const void *b, int b_dtype, double max_eps, double max_delta, int64_t k, int lace,
            int64_t *idx, double *val, int64_t *nnz);

/* Many independent signals sharing the resident dictionary: omp(A, B[:,s], eps, k) for
 * s = 0..nsig-1 (the loop a caller of the reference writes around omp; signals are independent,
 * SURVEY.md section 8e).  B: M x nsig column-major (ldB elements) on host or device (b_loc);
 * outputs idx (k x nsig, int64, unused tail = -1), val (k x nsig, f64, unused tail = 0),
 * nnz (nsig) on host or device (out_loc).  All signals are enqueued back to back without host
 * synchronisation; the call then synchronises ONCE (to verify the per-signal factorisation flags,
 * see DESIGN.md "optimistic chain") and returns with the results complete. */
int csmp_omp_batch(csmp_ctx *ctx, const void *B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                   double eps, int64_t *idx, double *val, int64_t *nnz, int out_loc);

/* fr(A, B[:,s], max_eps, min_delta, k) for s = 0..nsig-1 (src/forward.jl:44-54 in the caller's loop): same
 * conventions, pipelining and single synchronisation as csmp_omp_batch. */
int csmp_fr_batch(csmp_ctx *ctx, const void *B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                  double max_eps, double min_delta, int64_t *idx, double *val, int64_t *nnz, int out_loc);

/* gomp(A, B[:,s], l, eps, k) for s = 0..nsig-1 (src/matchingpursuit.jl:116-139 in the caller's loop), 1 <= l <= k: the conventions
 * of csmp_omp_batch (idx / val: k x nsig).  TWO solves are in flight, one on the context's stream and one on an internal clone's,
 * out of phase: a signal's short stages (top-l selection, the l-column panel append) run under the other signal's dictionary
 * sweep.  Results are csmp_gomp's, signal by signal.  One synchronisation at the end. */
int csmp_gomp_batch(csmp_ctx *ctx, const void *B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t l, int64_t k,
                    double eps, int64_t *idx, double *val, int64_t *nnz, int out_loc);

/* sp(A, B[:,s], k, delta; maxiter) for s = 0..nsig-1 (src/twostage.jl:87-101 in the caller's loop).  B: HOST matrix M x nsig
 * (ldB elements); idx / val: k x nsig host arrays (tail -1 / 0), nnz and iters (may be NULL): nsig.  Up to
 * CSMP_OPT_SOLVES_IN_FLIGHT solves run at once, each on its own context (this one and internal clones) and stream, all driven by
 * the CALLING thread: a Subspace Pursuit solve is two dictionary sweeps and a chain of short kernels with a handful of host
 * decisions (which atoms join, which leave, whether to go on); a solve whose pending device phase has not finished is skipped and
 * another one advanced, so one signal's chain runs under another's sweeps.  Signal s is solved by the very job csmp_sp runs:
 * identical results.  No threads are started; the ctx must not be used by another thread while the call runs. */
int csmp_sp_batch(csmp_ctx *ctx, const void *B, int b_dtype, int64_t ldB, int64_t nsig, int64_t k, double delta, int64_t maxiter,
                  int64_t *idx, double *val, int64_t *nnz, int64_t *iters);

/* The same contract, solved by the batched variant (BASELINE configs 3/4): the residual sweeps of
 * all signals become ONE MFMA GEMM per step (A' [r_1 .. r_B] on 16-bit images -- binary16 by default, bf16 or int8 by option --,
 * f32 accumulate) that only SCREENS: per signal the candidates whose screened value could still be the exact maximum are rescored in
 * Float64 from the f32/f64 master dictionary, and a certificate (exact best > an upper bound on the exact value of
 * every atom that was not rescored) guards every step: a signal that fails it once is re-solved by the exact path
 * before returning, so a certified result equals csmp_omp_batch's.  The error bound behind the certificate is
 * chosen by CSMP_OPT_BATCH_CERT (below).  Dictionaries of more than 8192 rows, and support capacities min(k, M) whose per-signal
 * vectors exceed the LDS (about 5000 columns at M = 4096), are solved by csmp_omp_batch's exact sweeps (same results;
 * the internal csmp_batch_screen_kernel then reports "none").  The per-signal state is two k x k Float64 factors: a batch
 * whose state does not fit the free HBM is solved in chunks of whole 256-signal tiles (csmp_batch_stats adds them up).  With out_loc == CSMP_DEVICE this call still
 * synchronises once (to read the per-signal certificates). */
int csmp_omp_batch_mfma(csmp_ctx *ctx, const void *B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                        double eps, int64_t *idx, double *val, int64_t *nnz, int out_loc);
/* statistics of the last csmp_omp_batch_mfma call: signals, how many were re-solved exactly (and
 * why), and -- when profiling is enabled -- the number and total duration (ms) of screening GEMMs */
int csmp_batch_stats(csmp_ctx *ctx, int64_t *signals, int64_t *resolved_exactly, int64_t *uncertain, int64_t *illcond,
                     int64_t *screen_launches, double *screen_ms);
/* ------------------------------------------------------------------ options
 * The reference passes every behavioural choice as an argument (src/matchingpursuit.jl:88-91,145-148;
 * src/twostage.jl:87); those are arguments here too.  The choices that exist only on this side of the boundary are
 * per-context options ("Config / flags").  The library reads NO environment variable.  A clone
 * (csmp_clone) starts from its parent's values (the resident Gram matrix excepted: 8 N^2 bytes per context are asked for, not
 * inherited).  Unknown keys and out-of-range values: CSMP_EINVAL.  (Keys 5-8 of rounds 2-3 -- a test switch and three choices with
 * one sane value each -- are gone: the library takes the default they had.) */
#define CSMP_OPT_BATCH_CERT 1      /* certificate of csmp_omp_batch_mfma (and of the screened sweeps, CSMP_OPT_SCREENED_SWEEP).
                                      1 (default): rigorous -- a deterministic bound on the screen's error (unit roundoff of both
                                      operands' images, Float32 accumulation, key truncation) with the largest column norm: a
                                      passed certificate PROVES the pick; no residual, however constructed, can slip through
                                      (tests: test_batched_certificate_against_adversarial_residuals).
                                      0 (opt-in, UNSAFE): statistical -- 8 standard deviations of independent roundings + a
                                      coherent term; narrower windows (fewer rescored candidates, int8 operands allowed), holds
                                      for generic data, NOT a proof: a residual aligned with the rounding errors of a near-tied
                                      atom passes the certificate with the WRONG atom and nothing reports it (the library's own
                                      adversarial test constructs 24 such signals out of 24).  Not measured in the default bench
                                      line; use it only where a wrong pick on a crafted input is acceptable */
#define CSMP_OPT_BATCH_GRAM 2      /* 1: csmp_omp_batch_mfma keeps G = A'A resident (Float64, 8 N^2 bytes: 32 GiB at N = 65536;
                                      built on first use, 2 M N^2 / 2 flops on the Float64 matrix cores) and takes A_S'a from it
                                      instead of streaming the support's columns: half the append traffic.  0 (default) frees it */
#define CSMP_OPT_BATCH_WINDOW 3    /* capacity of the rescoring window, 1..128; 0 (default) = 64 statistical / 128 rigorous */
#define CSMP_OPT_PIPELINE 4        /* csmp_omp_batch / csmp_fr_batch: 1 (default) three signals in flight, 0 one at a time */
#define CSMP_OPT_SOLVES_IN_FLIGHT 9 /* csmp_sp_batch: solves in flight (contexts on their own streams, one host thread), 1..4, default 3 */
#define CSMP_OPT_SCREENED_SWEEP 10 /* csmp_mp, csmp_omp(_batch), csmp_gomp(_batch) (l <= 16), csmp_sp(_batch) and csmp_ompr (k <= 4096): 0 (default) every sweep
                                      reads the f32/f64 dictionary (exact, 4 / 8 bytes per element).  1 / 3 / 2: the sweep reads an IMAGE of the
                                      dictionary and only SCREENS -- the best candidates are rescored in Float64 from the master dictionary under the
                                      batched variant's certificate (CSMP_OPT_BATCH_CERT; gomp: the whole top-l set and its order are certified; sp:
                                      every acquisition's top-k SET), and a solve (sp: an acquisition) with an uncertified step is repeated with the
                                      exact sweep before the call returns: results are those of option 0.
                                      3: the binary16 image (2 bytes per element under one power-of-two scale, f32 accumulate): with the rigorous
                                      certificate (the default) its bound is 2^-11 |a||r| -- narrow enough to certify (nearly) every step on the
                                      benchmark dictionaries: a PROVABLY exact solve at half the bytes per atom.
                                      1: the bf16 image (2 bytes, bound 2^-8 |a||r|: under the rigorous certificate most solves fall back; useful with
                                      CSMP_OPT_BATCH_CERT = 0).  2: the int8 image where the dictionary is flat (max|A| <= 8 rms of its entries;
                                      otherwise the bf16 image): 1 byte per element under one step max|A|/127, the residual quantised per sweep,
                                      exact integer accumulation; statistical certificate only.
                                      Costs the image (2 Mk N or Mk N bytes) beside the dictionary.  Default off: the headline path stays the
                                      exact sweep ("Screened single-signal sweep"). */
#define CSMP_OPT_BATCH_SCREEN 11   /* csmp_omp_batch_mfma: operands of the screening GEMM.
                                      3 (default): binary16 images (v_mfma_f32_16x16x32_f16; the dictionary and every residual under
                                      exact power-of-two scales): the same rate and bytes as bf16 with eleven significand bits --
                                      the rigorous bound is 2^-10 |a||r| instead of 2^-7 |a||r|, windows as narrow as bf16's
                                      statistical ones.  0: bf16 images (v_mfma_f32_16x16x32_bf16), the form of rounds 1-3.
                                      1: int8 images -- the dictionary under one step max|A|/127, every residual under its own -- on
                                      v_mfma_i32_16x16x64_i8 (half the K-loop, exact integer accumulation); 2: int8 where the
                                      dictionary is flat (max|A| <= 8 x the root mean square of its entries), binary16 otherwise.
                                      The int8 screen has a statistical certificate only: it runs under CSMP_OPT_BATCH_CERT = 0;
                                      under the rigorous certificate 1 and 2 run binary16.  Only the image in use is built
                                      (2 M N bytes; int8: M N). */
int csmp_set_option(csmp_ctx *ctx, int key, int64_t value);
int csmp_get_option(csmp_ctx *ctx, int key, int64_t *value);

/* screened solves (CSMP_OPT_SCREENED_SWEEP) made through this ctx and how many of them had to be repeated with the exact sweep
 * (a step whose certificate failed); reset != 0 zeroes both counters.  Measurement / tests. */
int csmp_screened_stats(csmp_ctx *ctx, int64_t *solves, int64_t *fallbacks, int reset);

/* ------------------------------------------------------------------ step-level API
 * Mirrors the Update functors: P = OMP(A,b,k) / MP(A,b) / GOMP(A,b,l) / FR(A,b) then update!(P,x)
 * (src/CompressedSensing.jl:22-23; src/matchingpursuit.jl:26,62,116; src/forward.jl:88-95).  The solver state
 * (residual, on-device QR, support) lives in the ctx. */
int csmp_solver_begin(csmp_ctx *ctx, int algo, const void *b, int b_dtype, int64_t kcap,
                      const int64_t *idx0, const double *val0, int64_t nnz0);
/* CSMP_ALGO_SP / CSMP_ALGO_OMPR: kcap is the k of SP(A,b,k) (2k <= M or CSMP_ERANGE: src/twostage.jl:55) / OMPR(A,b,k).  The
 * solver owns x.  SP: (idx0, val0) is the x the host would hand to update! -- any vector of at most 2k atoms; its values count (the
 * acquisition starts from residual!(P, x), :68).  OMPR: x starts empty (nnz0 = 0): its factorisation does (:124-129).
 * one update!: MP/OMP ignore l; GOMP adds the l best atoms; SP / OMPR: update!(P, x) (:75-83 / :134-180, eta = 1), which requires
 * nnz(x) == k -- otherwise CSMP_ESTATE with the reference's message "nnz(x) = .. != .. = k" (:76, :135) */
int csmp_solver_step(csmp_ctx *ctx, int64_t l);
/* SP: sp_acquisition!(P, x, k) (src/twostage.jl:67-72) -- the k atoms best correlated with the residual of x join it, least squares
 * on the union.  OMPR / OMP / GOMP: oblivious_acquisition!(P, x, k) (src/matchingpursuit.jl:207-216) -- the same with the
 * updatable QR (OMPR: on the empty x, k = the k of OMPR(A,b,k): how ompr fills the support, src/twostage.jl:190). */
int csmp_solver_acquire(csmp_ctx *ctx, int64_t k);
/* dropindex!(x, AiQR, i) (src/util.jl:137-161): atom leaves the support of an OMP/GOMP solver -- Givens
 * down-date of the on-device QR (remove_column!), residual and coefficients follow.  No-op if absent.
 * Any capacity: up to 1023 columns one workgroup walks R, up to 4095 the rotations come from the explicit inverse, beyond that the
 * factorisation is rebuilt from the columns that stay. */
int csmp_solver_remove(csmp_ctx *ctx, int64_t atom);
/* current x (sorted), ||b - A x||_2, selection order, stop reason.  Any pointer may be NULL. */
int csmp_solver_state(csmp_ctx *ctx, int64_t *idx, double *val, int64_t *nnz, double *resnorm,
                      int64_t *order, int *stop);

/* ------------------------------------------------------------------ one signal, columns sharded over GPUs
 * SURVEY.md section 8e/8f-4 (not in the reference: its omp takes one b and one A, src/matchingpursuit.jl:73-82).
 * Each rank's ctx holds columns [col_offset, col_offset + N) of the global dictionary and a full replica of
 * the solver state.  A solve: csmp_solver_begin(ctx, CSMP_ALGO_OMP, b, ...) on every rank; then k times
 *     csmp_shard_sweep (argmaxinner!(P), :181-185, over the local columns; t > 0: the driver's residual test :79)
 *     -> the caller's all_gather of ONE record per rank (csmp_shard_record_bytes each, device memory)
 *     -> csmp_shard_append (global arg-max, ties to the lower global index; update!'s guards :63,66;
 *        add_column! and the residual update on the winner's column, which travels inside its record);
 * then csmp_solver_state (indices are global; identical on every rank).  Nothing synchronises with the host
 * between those calls: with csmp_set_stream they are stream-ordered with the collective. */
int csmp_shard_config(csmp_ctx *ctx, int64_t col_offset);
int64_t csmp_shard_record_bytes(const csmp_ctx *ctx);
int csmp_shard_sweep(csmp_ctx *ctx, double eps, int check_eps, void *rec_dev);
int csmp_shard_append(csmp_ctx *ctx, const void *recs_dev, int nrec);

/* ------------------------------------------------------------------ many signals sharded over GPUs
 * SURVEY.md section 8e, BASELINE configs[3]: signals are independent given A (the loop a caller of the reference writes around
 * omp, src/matchingpursuit.jl:73-82), so rank r -- ONE PROCESS PER GPU, A replicated -- solves the contiguous block
 * [lo, hi) = csmp_shard_range(nsig, r, world) and ONE collective moves the results: per signal a row of 2k + 1 Float64
 * [idx_0..idx_{k-1} | val_0..val_{k-1} | nnz].
 *
 * csmp_omp_sharded does all of it inside the library: the block's solves (method 0: csmp_omp_batch, 1: csmp_omp_batch_mfma),
 * the packing on the device, ONE ncclAllGather over xGMI on the context's stream (device memory to device memory), and the
 * unpacking into global signal order.  B: THIS RANK'S block, M x (hi - lo) column-major (ldB elements), host or device (b_loc);
 * idx / val (k x nsig, tail -1 / 0) and nnz (nsig): ALL signals, on every rank, host or device (out_loc).  nsig is the
 * global count and must be the same on every rank; the call is collective.
 * The communicator: rank 0 calls csmp_comm_id (ncclGetUniqueId, CSMP_COMM_ID_BYTES of host memory), the host passes those
 * bytes to the other ranks by whatever started them (a file, a socket, Julia's Distributed, torch.distributed's store), and
 * every rank calls csmp_comm_init(ctx, id, rank, world) (ncclCommInitRank on the ctx's GPU; collective).  The host language
 * needs no collective library of its own.  RCCL is bound lazily (librccl.so.1) by these calls only.  csmp_destroy frees the
 * communicator; csmp_comm_free does it earlier. */
#define CSMP_COMM_ID_BYTES 128
int csmp_comm_id(void *id);
int csmp_comm_init(csmp_ctx *ctx, const void *id, int rank, int world);
int csmp_comm_free(csmp_ctx *ctx);
int csmp_omp_sharded(csmp_ctx *ctx, const void *B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k, double eps,
                     int method, int64_t *idx, double *val, int64_t *nnz, int out_loc);
/* The device-side halves of that exchange on their own, stream-ordered on the ctx (device pointers throughout): rows of 2k + 1 Float64
 * from a block's results (idx, val: k x nloc; nnz: nloc; `rows` >= nloc rows are written, the surplus zeroed -- every rank packs
 * ceil(nsig / world) rows so that the blocks gather), and the gathered world x rows x (2k + 1) array into idx / val (k x nsig) and nnz
 * (nsig) in global signal order.  For a host that keeps results on the device under a collective of its own. */
int csmp_pack_block_device(csmp_ctx *ctx, const int64_t *idx, const double *val, const int64_t *nnz, int64_t k, int64_t nloc,
                           int64_t rows, double *packed);
int csmp_unpack_gathered_device(csmp_ctx *ctx, const double *gathered, int64_t k, int64_t nsig, int world, int64_t *idx, double *val,
                                int64_t *nnz);
/* The same wire layout for hosts that bring their own collective (torch.distributed all_gather in sharded.py's gloo tests and
 * single-signal solver families; MPI): host memory, no ctx. */
int csmp_shard_range(int64_t nsig, int rank, int world, int64_t *lo, int64_t *hi);
int csmp_pack_results(const int64_t *idx, const double *val, const int64_t *nnz, int64_t k, int64_t nsig, double *packed);
int csmp_unpack_results(const double *packed, int64_t k, int64_t nsig, int64_t *idx, double *val, int64_t *nnz);

/* ------------------------------------------------------------------ primitives
 * argmaxinner!(P) / argmaxinner!(P,k): src/matchingpursuit.jl:181-193.  r: length-M Float64
 * host vector.  abs_corr (may be NULL): receives |A' r| (length N).  top_idx/top_val: the
 * topk atoms, descending by |<a_i, r>|, ties by ascending index (partialsortperm, :192). */
int csmp_sweep(csmp_ctx *ctx, const double *r, double *abs_corr, int64_t topk, int64_t *top_idx, double *top_val);

/* A[:, cols] \ b by the on-device QR: the UpdatableQR solve pinned by test/forward.jl:23-28
 * ("P.AiQR \ y ~ A[:, nzind] \ y").  cols in any order; coef aligned with cols. */
int csmp_lstsq(csmp_ctx *ctx, const int64_t *cols, int64_t ncols, const void *b, int b_dtype, double *coef);

#ifdef __cplusplus
}
#endif
#endif
