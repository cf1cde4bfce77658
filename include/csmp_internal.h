/*
 * csmp_internal.h -- measurement and test hooks of libcsmp.so.  NOT part of the drop-in boundary: nothing a host of the
 * reference's API (mp / omp / gomp / sp, the Update functors) needs is declared here, and the Julia wrapper binds none of it.
 * bench.py, tools/ and tests/ bind these through compressedsensing.jl_amd/_lib.py (INTERNAL_SIGNATURES).
 */
#ifndef CSMP_INTERNAL_H
#define CSMP_INTERNAL_H
#include "csmp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* on = 1: every sweep launch is bracketed by HIP events on the ctx stream; on = n > 1: every
 * n-th launch only (an event pair costs a few microseconds of stream time); 0 = off. */
int csmp_profile_enable(csmp_ctx *ctx, int on);
/* number of sweep launches timed and the sum of their durations (ms); reset != 0 clears */
int csmp_profile_read(csmp_ctx *ctx, int64_t *sweep_launches, double *sweep_ms, int reset);
/* the timed launches of this context and of the second pipeline of csmp_omp_batch (its twin) as ONE window: launches from the first
 * to the last sampled one on each stream, the time from the earliest start event to the latest end event, the mean duration of a
 * sampled launch, the number of streams that carried timed launches.  Before csmp_profile_read, which consumes the events. */
int csmp_profile_window(csmp_ctx *ctx, int64_t *launches, double *window_ms, double *mean_launch_ms, int *streams);
/* average reading (ms) of an event pair with nothing between its two records: the share of a timed launch's bracket that is the
 * bracket itself */
int csmp_profile_overhead(csmp_ctx *ctx, int reps, double *avg_ms);
/* what the library holds at this moment, process-wide: bytes and blocks of device memory, bytes of page-locked host memory, host ranges
 * registered with the device, events, streams.  Every one is zero when no context exists (tests/test_gpu_leaks.py).  Any pointer may be NULL. */
int csmp_live_resources(int64_t *device_bytes, int64_t *device_blocks, int64_t *pinned_bytes, int64_t *registered_ranges, int64_t *events,
                        int64_t *streams);
/* sweep bandwidth probe: `reps` product sweeps (argmaxinner!(P), src/matchingpursuit.jl:181-185) of a random residual,
 * bracketed by one HIP event pair on the ctx stream; returns the average ms per sweep.  variant must be 0. */
int csmp_bench_sweep(csmp_ctx *ctx, int variant, int reps, double *avg_ms);
/* what configure_sweep chose for the resident dictionary: loads per unit of k_sweep_gen (16 / 8 / 4); phases the residual is
 * staged in (1: one LDS image); workgroups of a stand-alone sweep and of the sweep inside the tick kernel; dynamic LDS bytes;
 * dynamic = 1: the columns are handed out at run time (k_sweep_dyn and the DYN tick), 0: split statically; columns_per_unit: 2 or 4
 * where the stand-alone sweep takes short columns several at a time (k_sweep_short), else 1.  Any pointer may be NULL. */
int csmp_sweep_config(const csmp_ctx *ctx, int *unit_loads, int *phases, int *workgroups, int *tick_workgroups, int64_t *lds_bytes, int *dynamic,
                      int *columns_per_unit);
/* measurement overrides of that choice, applied to the resident dictionary at once and to later ones: 0 = automatic */
#define CSMP_TUNE_SWEEP_GRID 2   /* workgroups of the product sweep */
#define CSMP_TUNE_SWEEP_UNIT 3   /* loads per unit (16, 8 or 4) */
#define CSMP_TUNE_TICK_GRID 4    /* sweep workgroups inside the tick kernel of csmp_omp_batch */
#define CSMP_TUNE_SWEEP_DYN 9     /* 1: the product sweep hands its columns out at run time (k_sweep_dyn; one LDS image, grids up to 512 workgroups); n = 2..64: only the last 1 / n of a workgroup's columns, after a static head; default 0: the static split */
#define CSMP_TUNE_PIPELINES 12    /* 1: csmp_omp_batch keeps one pipeline of three signals; 2: two pipelines side by side whatever the sizes; default 0: two from two signals and a 4-MiB dictionary on (rounds of 3 + 3 signals, the remainder 1 + 1) */
#define CSMP_TUNE_TICK_ORDER 10   /* 1: the tick kernel's sweep workgroups are dispatched ahead of its append stages' */
#define CSMP_TUNE_CLAIM_POOLS 11  /* the dynamic sweep: column pools a workgroup may claim from (its own first) */
#define CSMP_TUNE_PAIR_LDS_KIB 13 /* dynamic LDS (KiB) requested by the ticks of two pipelines side by side: above 80 = one workgroup per CU (default 81), 1 = what the kernels need */
#define CSMP_TUNE_PAIR_SPLIT 14   /* 1: two pipelines side by side keep the fused tick (append stages + sweep in ONE launch under the large LDS request); default 0: two launches per tick */
#define CSMP_TUNE_SWEEP_LDS_KIB 15 /* dynamic LDS (KiB) the stand-alone product sweep REQUESTS when that is more than it uses: above 80 = one workgroup per CU, 54 = two */
#define CSMP_TUNE_SWEEP_SHORT 16  /* 1: the stand-alone sweep keeps one column per unit for every shape (default 0: columns of up to four 1-KiB chunks go two or four to a unit, k_sweep_short) */
#define CSMP_TUNE_PHASE_ROWS 17   /* the phased sweep (a residual longer than the LDS): most rows of one stage; 0 = as many as the LDS holds */
#define CSMP_TUNE_SCREEN_STATIC 19 /* 1: the screened (image) sweeps deal their column groups out statically; default 0: by ticket counters (measured: 12 % faster for a lone 2-GiB sweep, level with three solves in flight) */
#define CSMP_TUNE_FAIL_ALLOC 18   /* test hook: the n-th device allocation of solver state from now fails for real (hipMalloc of an impossible size: hipErrorOutOfMemory stays pending); 0 = off */
#define CSMP_TUNE_REBUILD_DIRECT 8 /* 1: the oblivious start of csmp_srr forms Q'A with its directions read from L2 per wave (k_fr_rebuild), not staged in the LDS */
#define CSMP_TUNE_SWAP_REFUSE 7 /* 1: every exchange of csmp_ompr on the inverse Gram matrix fails its guard: the fallback to the QR path runs */
#define CSMP_TUNE_DIAG_SPLIT 6   /* 1: kernels that fuse independent parts run one launch per part (same results; a kernel trace shows the parts) */
#define CSMP_TUNE_BATCH_BUDGET_MIB 5 /* csmp_omp_batch_mfma: HBM (MiB) its per-signal state may take -- a test's stand-in for a full device */
int csmp_tune(csmp_ctx *ctx, int key, int64_t value);

/* layout of the last csmp_omp_batch_mfma call: signal columns of one screening launch (the batch
 * padded to whole 256-signal tiles) and the number of streams used (1: the screening GEMM and the per-signal
 * kernels alternate on the context's stream) */
int csmp_batch_layout(const csmp_ctx *ctx, int64_t *screen_signals, int *streams);
/* name of the screening kernel the last csmp_omp_batch_mfma call ran ("none": the exact sweeps served it) */
const char *csmp_batch_screen_kernel(const csmp_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
