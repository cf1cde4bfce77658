// csmp_kernels.hpp -- gfx950 (MI355X, CDNA4, wave64) kernels of the matching-pursuit path.
//
// Reference primitives replaced (paths relative to the reference repository):
//   k_sweep     mul!(Ar, A', r); abs; argmax            src/matchingpursuit.jl:181-185
//   k_select    argmax + "i not in x.nzind" guard       src/matchingpursuit.jl:63,65-66
//   k_qr1/2/3   add_column!(AiQR, a, pos)               src/util.jl:118-126 (UpdatableQR)
//               + residual!                             src/matchingpursuit.jl:152-161
//   k_finish    ldiv!(AiQR, r) + sorted-index order     src/matchingpursuit.jl:170-176
//   k_residual  residual!(r, A, x, b)                   src/matchingpursuit.jl:158-161
//   k_mp_update x[i] += dot(A[:,i], r)                  src/matchingpursuit.jl:29
//
// Numerics: every product and sum is Float64 on the exactly promoted dictionary value
// (v_cvt_f64_f32 + v_fma_f64).  The sweep is HBM-bound: per 1 KiB wave-load of an f32
// dictionary the SIMD spends 4 cvt + 4 fma DP instructions, far below the time HBM needs to
// deliver it, so Float64 selection costs no bandwidth ("sweep").
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace csmp {

constexpr int kWave = 64;
constexpr int kSweepThreads = 256;  // 4 waves
constexpr int kCPW = 4;             // dictionary columns a wave reduces together
constexpr int kSlabRows = 64;       // rows of Q owned by one QR workgroup
constexpr int kQrThreads = 256;

enum : int { STOP_EPS = 1, STOP_STAG = 2, STOP_FULL = 4, STOP_REORTH = 8 };  // REORTH: internal, see k_qr2
constexpr int STOP_UNCERTAIN = 256;  // (only in the flag word the finish kernels hand to the batch drivers: a screened pick failed its certificate)

// Control block of one solve, in device memory.  Written only by single-workgroup control
// kernels (k_select / k_init) and by workgroup 0 of k_qr1 / k_qr3 / k_mp_update (fields no
// workgroup of the same launch depends on), so no launch races with itself.
struct DevState {
    int nsel;       // columns in the QR / atoms in the support (MP: steps taken)
    int j;          // nsel frozen for the current step's QR kernels
    int cand;       // atom chosen for the current step
    int go;         // 1: the current step's append kernels run
    int done;       // STOP_* bits
    int steps;      // update! calls that changed x
    int go2;        // 1: k_qr2 asked for the re-orthogonalisation pass (k_qr3)
    int pcount;     // atoms accepted into the current panel (multi-column append, csmp_block.hpp)
    double rnorm2;  // ||r||^2 seen by the last sweep prologue
    double cval;    // signed <a_cand, r> (MP coefficient, src/matchingpursuit.jl:29)
    int uncertain;  // screened sweep (csmp_screened.hpp): steps whose pick could not be certified
    float rstep;    // int8 screened sweep: the quantisation step of the residual image of the last sweep
};

// up to eight pieces of device memory -> one page-locked landing area (host memory mapped into the device), in one launch
constexpr int kLandMax = 8;
struct LandSegs {
    const void* src[kLandMax];
    unsigned off[kLandMax], words[kLandMax];
    int n;
};
__global__ __launch_bounds__(256) void k_land_multi(const LandSegs g, char* __restrict__ dst) {
    for (int q = 0; q < g.n; ++q) {
        const int* s_ = reinterpret_cast<const int*>(g.src[q]);
        int* d_ = reinterpret_cast<int*>(dst + g.off[q]);
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < g.words[q]; i += gridDim.x * 256) d_[i] = s_[i];
    }
}
__global__ __launch_bounds__(256) void k_put_ints(const int* __restrict__ src, int n, int* __restrict__ dst) {
    const int t = (int)blockIdx.x * 256 + threadIdx.x;
    if (t < n) dst[t] = src[t];
}
// two ints from kernel arguments (a candidate list of one: the host knows the atom)
__global__ void k_set_pair(int* __restrict__ a, int va, int* __restrict__ b, int vb) {
    if (threadIdx.x == 0) {
        *a = va;
        *b = vb;
    }
}

using f32x4 = float __attribute__((ext_vector_type(4)));
using f64x2 = double __attribute__((ext_vector_type(2)));
template <typename TA> struct Vec;
template <> struct Vec<float> { using type = f32x4; static constexpr int n = 4; };
template <> struct Vec<double> { using type = f64x2; static constexpr int n = 2; };

// sum over the 256 threads of a workgroup in a fixed order; every thread gets the result
__device__ __forceinline__ double block_sum256(double v, double* scratch4) {
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, kWave);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch4[0] + scratch4[1]) + (scratch4[2] + scratch4[3]);
}

__device__ __forceinline__ double shx(double v, int m) { return __shfl_xor(v, m, kWave); }
// The wave-wide butterfly sum v += xor 32, 16, 8, 4, 2, 1 (every lane ends with the total) WITHOUT LDS trips: __shfl_xor is a
// ds_bpermute -- six dependent LDS round trips per column of the sweep, during which the wave's ring is one unit short.
// gfx950's v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows of two registers; handed the same
// value twice they leave (own, partner) or (partner, own) in the pair, and own + partner is the butterfly step either way.  The
// four in-row steps are DPP moves (row_ror:8; row_shl:4 / row_shr:4 under bank masks; two quad permutations).  The same pairs in
// the same order as the shuffle form and commutative additions: bit-identical to it (tools/probes/xsum_probe.hip).
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ double dpp_mov_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}
// one butterfly step across the wave's halves (xor 32) / across neighbouring rows (xor 16) on TWO inputs: x + its partner where the
// result keeps x's column, y + its partner where it keeps y's -- lanes 0..31 (rows 0 and 2) end with x's pair sums, lanes 32..63
// (rows 1 and 3) with y's.  x == y is the plain step.  (swap: X' = [X_lo | Y_lo], Y' = [X_hi | Y_hi]; X' + Y' pairs lane i with i + 32.)
__device__ __forceinline__ double xs32(double x, double y) {
    const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xs16(double x, double y) {
    const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// the four steps inside a row of 16 lanes (xor 8, 4, 2, 1)
__device__ __forceinline__ double row_xsum(double v) {
    v += dpp_mov_f64<0x128>(v, v);                                    // xor 8: row_ror:8
    v += dpp_mov_f64<0x114, 0xa>(dpp_mov_f64<0x104, 0x5>(v, v), v);   // xor 4: lanes with bit 2 clear read lane + 4 (row_shl:4), the others lane - 4 (row_shr:4)
    v += dpp_mov_f64<0x4E>(v, v);                                     // xor 2: quad_perm [2,3,0,1]
    v += dpp_mov_f64<0xB1>(v, v);                                     // xor 1: quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ double wave_xsum(double v) {
    v = xs32(v, v);
    v = xs16(v, v);
    return row_xsum(v);
}
// value of lane `src` (uniform) in every lane: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding
// global load and store (s_waitcnt vmcnt(0)), which would serialise a latency chain with the
// look-ahead loads it is supposed to overlap
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// tools/probes/ph_trace.hip: wall-clock stamps (100 MHz) of a sweep workgroup's waves; nothing in the product build
#ifndef CSMP_C_INDEX
#define CSMP_C_INDEX(c) (c)  // (tools/probes/ph_trace.hip redirects the short body's c stores)
#endif
#ifdef CSMP_PH_TRACE
__device__ unsigned long long* g_ph_trace;
#define PH_STAMP(slot) do { if ((threadIdx.x & 63) == 0) g_ph_trace[((size_t)bid * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PH_STAMP(slot) do { } while (0)
#endif

// lexicographic "better": larger value wins, ties go to the LOWER index (Julia argmax = first max)
__device__ __forceinline__ bool better(double v, int i, double bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}

// LDS image of the residual: lane l of chunk t reads its rows' r values as double2 slots.
// f32 dictionary (4 rows per lane-load): two planes so that consecutive lanes read consecutive
// 16-B slots (conflict-free ds_read_b128); f64 dictionary (2 rows): the linear layout already is.
template <int VEC>
__device__ __forceinline__ int r_slot(int m) {  // index in doubles
    if constexpr (VEC == 4) {
        const int t = m >> 8, l = (m & 255) >> 2, e = m & 3;
        return ((((t << 1) + (e >> 1)) << 6) + l) * 2 + (e & 1);
    } else {
        return m;
    }
}

// ---------------------------------------------------------------------------------------------
// Sweep: c = A' r (Float64), fused |.| + arg-max partials (argmaxinner!(P), src/matchingpursuit.jl:181-185).  One wave owns ONE
// whole column at a time (16 KiB contiguous at M = 4096 f32), lanes stride the rows with 16-byte non-temporal loads (A is streamed
// once per sweep and exceeds every cache), r lives in the LDS.  Grid-stride over columns; one (max |c|, first index) pair per
// workgroup.  dynamic LDS: the r image + 32 doubles of reduction scratch.
// The pipelined product sweep for any size(A) whose residual one LDS image holds (the reference allocates zeros(T, n) and calls
// mul! whatever n is, :54-60); longer residuals: sweep_body_ph, columns of up to four chunks: sweep_body_short.
//   unit    U consecutive 64-lane loads (U KiB) of ONE column: the grain of the pipeline.  One wave owns one column at a time.
//   ring    NB units in flight per wave, consumed oldest first; a consumed buffer is refilled at once with the unit NB ahead.
//           In the steady loop every load is unconditional (no branch around a load), so the wait the compiler places in front
//           of a unit's arithmetic is s_waitcnt vmcnt((NB-1)*U): the wave never drains below (NB-1)*U KiB in flight.
//   ragged  a lane whose rows lie past the column's end CLAMPS its vector index to the column's last vector -- a load of valid
//           memory, no predicate, no extra DRAM line -- and multiplies it with a zero of the residual image.
// KP: rows of the residual image (a multiple of U*64*VEC, >= Mv).  dynamic LDS: KP doubles + 32 doubles of scratch.
template <typename TA, int U, int NB>
__device__ __forceinline__ void sweep_body_gen(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, const int bid, const int nblk, const int KP, double* lds) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    constexpr int UR = U * ROWS;  // rows per unit
    constexpr int NW = kSweepThreads / kWave;
    static_assert((NB - 1) * U < 64, "the ring must fit the 6-bit vmcnt");
    if (st->done & skipmask) return;
    PH_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nvec = Mv / VEC;  // 16-byte vectors per column (Mv is a multiple of VEC: the leading dimension is padded to 16 bytes)
    double* red = lds + KP;
    double* redv = red + 8;
    int* redi = reinterpret_cast<int*>(redv + 4 * NW);
    const f64x2* rs = reinterpret_cast<const f64x2*>(lds);

    const int64_t col0 = (int64_t)bid * NW + wave, stride = (int64_t)nblk * NW;
    const int64_t ncol = col0 < N ? (N - 1 - col0) / stride + 1 : 0;
    double bestv = -1.0;
    int besti = 0x7fffffff;
    const int nunit = (Mv + UR - 1) / UR;  // units per column
    const int Mst = nunit * UR;            // rows of the image in use (zero beyond Mv)
    VT buf[NB][U];
    // c values are STAGED: lane s keeps the total of the wave's s-th finished column and the wave writes 64 of them with one
    // store instruction.  A store per column sits in the wave's in-order memory queue among the ring's loads: measured 1.5-3 us of
    // a 155-us sweep at 16-KiB columns, 5-7 % at 8-KiB Float64 columns, 13 us of 170 with one store per 4 KiB (short columns).
    double cst = 0.0;
    int64_t ccst = -1;
    int cslot = 0;
    int64_t icol = col0, ccol = col0;  // issue / consume pointers: (column, unit within the column)
    int ib = 0, cb = 0;
    const int64_t T = ncol * nunit;
    int64_t ileft = T, cleft = T;
    double acc = 0.0;
    auto issue = [&](VT(&b)[U]) {
        const VT* pc = reinterpret_cast<const VT*>(A + icol * ld);
        const int vb = ib * (U * kWave) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = vb + u * kWave;
            b[u] = __builtin_nontemporal_load(pc + (v < nvec ? v : nvec - 1));
        }
        if (++ib == nunit) {
            ib = 0;
            icol += stride;
        }
        --ileft;
    };
    // the ring's first loads go out BEFORE the residual image is staged: they do not depend on it, and the trip to HBM and the
    // image's trip to L2 then overlap (tools/probes/ph_trace.hip: the image and ||r||^2 take 6.5-9.7 us of a long column's launch)
    // (a wave with fewer than NB units re-reads its last column -- or column N - 1 -- for the rest: no load sits under a branch)
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        if (ileft <= 0) {
            icol = (ncol > 0 ? col0 + (ncol - 1) * stride : N - 1);
            ib = 0;
        }
        issue(buf[d]);
    }
    {
        // the residual image, and ||r||^2 in a fixed per-thread order (identical in every workgroup).
        // The loads go out 16 at a time (a rolled loop waits for each in turn: ~6 us of a 150 us kernel at M = 4096).
        constexpr int RP = 16;
        double n2 = 0.0;
        for (int m0 = tid; m0 < Mst; m0 += RP * kSweepThreads) {
            double rv[RP];
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int m = m0 + q * kSweepThreads;
                rv[q] = m < Mv ? r[m] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int m = m0 + q * kSweepThreads;
                if (m < Mst) lds[r_slot<VEC>(m)] = rv[q];
                n2 = fma(rv[q], rv[q], n2);
            }
        }
        for (int s = 32; s >= 1; s >>= 1) n2 += __shfl_xor(n2, s, kWave);
        __syncthreads();
        if (lane == 0) red[wave] = n2;
        __syncthreads();
        n2 = (red[0] + red[1]) + (red[2] + red[3]);
        if (bid == 0 && tid == 0) st->rnorm2 = n2;
        if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break  (:79,:132)
            if (bid == 0 && tid == 0) st->done |= STOP_EPS;
            return;
        }
    }
    PH_STAMP(1);
    auto consume = [&](const VT(&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = cb * U + u;
            if constexpr (VEC == 4) {
                const f64x2 r01 = rs[(t * 2 + 0) * kWave + lane];
                const f64x2 r23 = rs[(t * 2 + 1) * kWave + lane];
                acc = fma((double)b[u].x, r01.x, acc);
                acc = fma((double)b[u].y, r01.y, acc);
                acc = fma((double)b[u].z, r23.x, acc);
                acc = fma((double)b[u].w, r23.y, acc);
            } else {
                const f64x2 r01 = rs[t * kWave + lane];
                acc = fma((double)b[u].x, r01.x, acc);
                acc = fma((double)b[u].y, r01.y, acc);
            }
        }
        if (++cb == nunit) {  // the column's last unit
            acc = wave_xsum(acc);
            if (lane == cslot) {
                cst = acc;
                ccst = ccol;
            }
            if (++cslot == kWave) {
                if (ccst >= 0) cvec[ccst] = cst;
                ccst = -1;
                cslot = 0;
            }
            const double av = fabs(acc);
            if (av > bestv) {  // columns arrive in increasing order: '>' keeps the first maximum
                bestv = av;
                besti = (int)ccol;
            }
            acc = 0.0;
            cb = 0;
            ccol += stride;
        }
        --cleft;
    };
    if (T >= 2 * NB) {
        const int64_t groups = T / NB - 1;
        for (int64_t g = 0; g < groups; ++g) {
#pragma unroll
            for (int d = 0; d < NB; ++d) {
                consume(buf[d]);
                issue(buf[d]);
            }
        }
    }
    while (cleft > 0) {  // the last NB .. 2 NB - 1 units of the wave (or all of them, when there are fewer)
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            if (cleft == 0) break;
            consume(buf[d]);
            if (ileft > 0) issue(buf[d]);
        }
    }
    if (ccst >= 0) cvec[ccst] = cst;  // (the columns staged since the last full store)
    PH_STAMP(5);
    if ((lane & 15) == 0) {
        redv[wave * 4 + (lane >> 4)] = bestv;
        redi[wave * 4 + (lane >> 4)] = besti;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = redv[0];
        int bi = redi[0];
        for (int q = 1; q < 4 * NW; ++q)
            if (better(redv[q], redi[q], bv, bi)) {
                bv = redv[q];
                bi = redi[q];
            }
        pval[bid] = bv;
        pidx[bid] = bi;
    }
}
template <typename TA, int U, int NB>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_gen(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, int KP) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sweep_body_gen<TA, U, NB>(A, ld, Mv, N, r, cvec, pval, pidx, st, eps, check_eps, skipmask, (int)blockIdx.x, (int)gridDim.x, KP, lds);
}
inline size_t sweep_gen_lds_bytes(int KP) { return (size_t)(KP + 8 + 16 + 8) * sizeof(double); }

// ---------------------------------------------------------------------------------------------
// A residual LONGER than the LDS (8 M bytes > 160 KiB, M beyond ~20 000): the image is staged KP rows at a time and the workgroup
// runs ALL its columns against one stage before the next is loaded.  Round 5 did this inside sweep_body_gen with every stage a
// pipeline of its own (the ring drained before the stage's barrier and refilled after the reload) and the columns' partial sums
// parked in c[col]: 0.765-0.78 of the roofline at M = 32 768.  Here the dictionary stream does not stop at a stage boundary:
//   issue     ONE stream of units over (stage, column, unit), NB units ahead of the arithmetic whatever stage that is in: while the
//             wave multiplies the last units of stage p its ring already holds the first units of stage p + 1, and they stay in
//             flight across the barrier and the reload of the image (lds_barrier: no drain).  Every ring buffer carries its column
//             ordinal and unit number; each stage's unit count is padded to a multiple of NB with loads of ONE vector that nobody
//             consumes (after the last stage too), so every stage starts on buffer 0, no load sits under a branch, and the
//             compiler's counted waits stay exact.
//   partials  a column's sum over the stages so far waits in the LDS (pcap doubles per wave, the wave's columns in order), not in
//             c[col]: nothing is read back from memory, and c is written once, 64 columns to a store instruction.
//             c[col] = ((stage 0) + stage 1) + ... as before: the same bits.
// dynamic LDS: KP + 40 + 4 pcap doubles.  Every wave joins every barrier (a wave without columns too).
inline size_t sweep_ph_lds_bytes(int KP, int pcap) { return (size_t)(KP + 8 + 16 + 8 + 8 + 4 * (size_t)pcap) * sizeof(double); }
template <typename TA, int U, int NB>
__device__ __forceinline__ void sweep_body_ph(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, const int bid, const int nblk, const int KP, const int pcap, double* lds) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    constexpr int UR = U * ROWS;
    constexpr int NW = kSweepThreads / kWave;
    static_assert((NB - 1) * U < 64, "the ring must fit the 6-bit vmcnt");
    if (st->done & skipmask) return;
    PH_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nvec = Mv / VEC;
    const int nph = (Mv + KP - 1) / KP;
    double* red = lds + KP;
    double* redv = red + 8;
    int* redi = reinterpret_cast<int*>(redv + 4 * NW);
    double* part = redv + 4 * NW + 8 + 8 + (size_t)wave * pcap;
    const f64x2* rs = reinterpret_cast<const f64x2*>(lds);
    const int64_t col0 = (int64_t)bid * NW + wave, stride = (int64_t)nblk * NW;
    const int ncol = col0 < N ? (int)((N - 1 - col0) / stride + 1) : 0;  // (<= pcap: the host sized the partials for it)
    auto units_of = [&](int ph) {  // units per column in stage ph
        const int rows_here = (Mv - ph * KP) < KP ? (Mv - ph * KP) : KP;
        return (rows_here + UR - 1) / UR;
    };
    // the image of stage ph: pairs of rows (16-byte loads; a pair is one 16-byte LDS slot in either layout), 16 loads in flight per
    // thread -- one by one a stage of 16 384 rows is 64 dependent trips per thread, and the stage's barrier waits for all of them
    auto stage_image = [&](int ph, double& n2, auto rp) {
        const int k0 = ph * KP, Mst = units_of(ph) * UR;
        const int Mend = Mst;  // (||r||^2 grows stage by stage: thread t adds the pairs of rows with (m / 2) mod 256 = t in increasing order)
        constexpr int RP = decltype(rp)::value;
        for (int m0 = 2 * tid; m0 < Mend; m0 += RP * 2 * kSweepThreads) {
            f64x2 rv[RP];
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int m = k0 + m0 + q * 2 * kSweepThreads;  // (even: Mv, KP and k0 are)
                const f64x2 x = *reinterpret_cast<const f64x2*>(r + (m < Mv ? m : Mv - 2));  // (no load under a branch: they all go out together)
                rv[q] = m < Mv ? x : f64x2{0.0, 0.0};
            }
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int m = m0 + q * 2 * kSweepThreads;
                if (m < Mst) *reinterpret_cast<f64x2*>(lds + r_slot<VEC>(m)) = rv[q];
                n2 = fma(rv[q].x, rv[q].x, n2);
                n2 = fma(rv[q].y, rv[q].y, n2);
            }
        }
    };
    double bestv = -1.0;
    int besti = 0x7fffffff;
    VT buf[NB][U];
    int ms[NB], mb[NB];  // column ordinal of the unit in the buffer (-1: nobody consumes it) and its number within the column
    // the issue stream: irem real units and then ipad padding units left in its stage; (icol, ib) always names a valid unit -- a
    // padding unit re-reads it (one vector per lane from lines the wave has just had) and nobody consumes it
    int iph = 0, is = 0, ib = 0;
    int inunit = units_of(0);
    int64_t irem = (int64_t)ncol * inunit;
    int ipad = (int)((NB - irem % NB) % NB);
    int64_t icol = col0 < N ? col0 : N - 1;
    int ik0v = 0;
    auto issue = [&](VT(&b)[U], int& s_, int& b_) {
        const VT* pc = reinterpret_cast<const VT*>(A + icol * ld);
        const int vb = ik0v + ib * (U * kWave) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = vb + u * kWave;
            b[u] = __builtin_nontemporal_load(pc + (v < nvec ? v : nvec - 1));
        }
        const bool real = irem > 0;
        s_ = real ? is : -1;
        b_ = ib;
        if (real) {
            --irem;
            if (++ib == inunit) {
                ib = 0;
                ++is;
                if (irem > 0) icol += stride;
            }
        } else if (ipad > 0) {
            --ipad;
        }
        if (irem == 0 && ipad == 0 && iph + 1 < nph) {  // on to the next stage's units
            ++iph;
            inunit = units_of(iph);
            irem = (int64_t)ncol * inunit;
            ipad = (int)((NB - irem % NB) % NB);
            is = 0;
            ib = 0;
            icol = col0 < N ? col0 : N - 1;
            ik0v = (iph * KP) / VEC;
        }
    };
    double acc = 0.0;
    double cst = 0.0;  // c values staged 64 to a store instruction (see sweep_body_gen)
    int64_t ccst = -1;
    int cslot = 0;
    // the ring's first loads go out before the first image is staged (see sweep_body_gen)
#pragma unroll
    for (int d = 0; d < NB; ++d) issue(buf[d], ms[d], mb[d]);
    // ||r||^2 is complete when the LAST stage's image has been read (every stage adds its rows: no stage reads rows it does not
    // keep -- the whole residual in the first stage was 2 of its 4 trips, ~4.5 us of a 165-us launch at M = 32768); the eps test
    // waits for it.  A sweep that turns out not to be wanted has then run all but its last stage: nobody looks at its results.
    double n2 = 0.0;
    auto norm_done = [&]() -> bool {  // (after the image's stores, before the barrier that publishes it)
        for (int s = 32; s >= 1; s >>= 1) n2 += __shfl_xor(n2, s, kWave);
        if (lane == 0) red[wave] = n2;
        __syncthreads();
        n2 = (red[0] + red[1]) + (red[2] + red[3]);
        if (bid == 0 && tid == 0) st->rnorm2 = n2;
        if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break  (:79,:132)
            if (bid == 0 && tid == 0) st->done |= STOP_EPS;
            return true;
        }
        return false;
    };
    stage_image(0, n2, std::integral_constant<int, 16>());
    if (nph == 1) {
        if (norm_done()) return;
    } else {
        __syncthreads();
    }
    PH_STAMP(1);
    for (int ph = 0; ph < nph; ++ph) {
        if (ph > 0) {
            PH_STAMP(2);
            lds_barrier();  // everyone is done with the previous image (the ring's loads stay in flight)
            PH_STAMP(3);
            stage_image(ph, n2, std::integral_constant<int, 16>());
            if (ph + 1 == nph) {
                if (norm_done()) return;
            } else {
                __syncthreads();
            }
            PH_STAMP(4);
        }
        const int nunit = units_of(ph);
        const bool lastph = ph + 1 == nph;
        const int64_t Tpad = ((int64_t)ncol * nunit + NB - 1) / NB * NB;
        for (int64_t g = 0; g < Tpad / NB; ++g) {
#pragma unroll
            for (int d = 0; d < NB; ++d) {
                if (ms[d] >= 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int t = mb[d] * U + u;
                        if constexpr (VEC == 4) {
                            const f64x2 r01 = rs[(t * 2 + 0) * kWave + lane];
                            const f64x2 r23 = rs[(t * 2 + 1) * kWave + lane];
                            acc = fma((double)buf[d][u].x, r01.x, acc);
                            acc = fma((double)buf[d][u].y, r01.y, acc);
                            acc = fma((double)buf[d][u].z, r23.x, acc);
                            acc = fma((double)buf[d][u].w, r23.y, acc);
                        } else {
                            const f64x2 r01 = rs[t * kWave + lane];
                            acc = fma((double)buf[d][u].x, r01.x, acc);
                            acc = fma((double)buf[d][u].y, r01.y, acc);
                        }
                    }
                    if (mb[d] == nunit - 1) {  // the column's last unit of this stage
                        acc = wave_xsum(acc);
                        if (ph > 0) acc = part[ms[d]] + acc;
                        if (!lastph) {
                            if (lane == 0) part[ms[d]] = acc;
                        } else {
                            const int64_t col = col0 + (int64_t)ms[d] * stride;
                            if (lane == cslot) {
                                cst = acc;
                                ccst = col;
                            }
                            if (++cslot == kWave) {
                                if (ccst >= 0) cvec[ccst] = cst;
                                ccst = -1;
                                cslot = 0;
                            }
                            const double av = fabs(acc);
                            if (av > bestv) {  // columns arrive in increasing order: '>' keeps the first maximum
                                bestv = av;
                                besti = (int)col;
                            }
                        }
                        acc = 0.0;
                    }
                }
                issue(buf[d], ms[d], mb[d]);
            }
        }
    }
    if (ccst >= 0) cvec[ccst] = cst;
    PH_STAMP(5);
    if ((lane & 15) == 0) {
        redv[wave * 4 + (lane >> 4)] = bestv;
        redi[wave * 4 + (lane >> 4)] = besti;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = redv[0];
        int bi = redi[0];
        for (int q = 1; q < 4 * NW; ++q)
            if (better(redv[q], redi[q], bv, bi)) {
                bv = redv[q];
                bi = redi[q];
            }
        pval[bid] = bv;
        pidx[bid] = bi;
    }
}
template <typename TA, int U, int NB>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_ph(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, int KP, int pcap) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sweep_body_ph<TA, U, NB>(A, ld, Mv, N, r, cvec, pval, pidx, st, eps, check_eps, skipmask, (int)blockIdx.x, (int)gridDim.x, KP, pcap, lds);
}

// ---------------------------------------------------------------------------------------------
// SHORT columns (the reference sweeps whatever size(A) is: its own tests are 32 x 48, configs[0] is 256 x 1024): when a column is
// one or two 1-KiB wave loads, one column per unit leaves the ring's loads clamped duplicates and pays a reduction per KiB.  Here a
// unit of eight loads covers 8 / NCH NEIGHBOURING columns of NCH chunks each; the lanes keep one partial sum per column and ONE
// transposing butterfly reduces CPU of them together (two such sets where a unit holds eight columns): the xor-32 step keeps columns 0 / 1 in the wave's lower half and 2 / 3 in the upper
// (CPU = 2: column 0 below, 1 above), the xor-16 step leaves column q in row q, the four in-row steps finish all of them at once --
// the pairs and their order are those of wave_xsum for every column, so c[col] has the bits of the one-column body.
// Lane 16 q (CPU = 2: lane 32 q) writes column q; every lane tracks the maximum of ITS column sequence (increasing indices).
// One residual image (NCH chunks); a wave owns the column groups g0, g0 + stride, ...; a ragged last group clamps its column
// addresses into the dictionary and neither writes nor ranks the surplus.  KP = NCH * 64 * VEC.
// Measured, 1 GiB dictionaries (fraction of 8 TB/s, one column per unit -> this body): M = 256 f32 0.52 -> 0.79+, f64 0.75 -> 0.81;
// M = 512 f32 0.72 -> 0.80+; M = 1000 f32 0.78 -> 0.83; M = 64 f32 0.17 -> 0.45.
template <typename TA, int NCH, int CPU>
__device__ __forceinline__ void sweep_body_short(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, const int bid, const int nblk, const int KP, double* lds) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int U = 8, NB = 32 / U;       // units of eight loads, a ring of four
    constexpr int CU_ = U / NCH;            // columns per unit
    constexpr int SETS = CU_ / CPU;         // ... reduced CPU at a time
    constexpr int NW = kSweepThreads / kWave;
    static_assert(CPU == 2 || CPU == 4, "two or four columns per transposing reduction");
    static_assert(U % NCH == 0 && CU_ % CPU == 0 && SETS >= 1, "whole columns per unit, whole sets per unit");
    if (st->done & skipmask) return;
    PH_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nvec = Mv / VEC;
    double* red = lds + KP;
    double* redv = red + 8;
    int* redi = reinterpret_cast<int*>(redv + 4 * NW);
    const f64x2* rs = reinterpret_cast<const f64x2*>(lds);
    const int64_t ngrp = (N + CU_ - 1) / CU_;
    const int64_t g0 = (int64_t)bid * NW + wave, stride = (int64_t)nblk * NW;
    const int64_t T = g0 < ngrp ? (ngrp - 1 - g0) / stride + 1 : 0;
    const int q = CPU == 4 ? (lane >> 4) : (lane >> 5);  // the column of its group this lane ends up holding
    double bestv = -1.0;
    int besti = 0x7fffffff;
    VT buf[NB][U];
    constexpr int LW = kWave / CPU, SLOTS = LW / SETS;
    constexpr int KS = 6;  // registers of staged c values per lane: the wave stores every 64 KS columns (see `flush` below)
    double cst[KS];
    int ccst[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        cst[k] = 0.0;
        ccst[k] = -1;
    }
    int cslot = 0, kslot = 0;
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (ccst[k] >= 0) cvec[CSMP_C_INDEX(ccst[k])] = cst[k];
            ccst[k] = -1;
        }
        kslot = 0;
    };
    int64_t ig = g0, cg = g0;
    int64_t ileft = T, cleft = T;
    auto issue = [&](VT(&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t col = ig * CU_ + u / NCH;
            const VT* pc = reinterpret_cast<const VT*>(A + (col < N ? col : N - 1) * ld);
            const int v = (u % NCH) * kWave + lane;
            b[u] = __builtin_nontemporal_load(pc + (v < nvec ? v : nvec - 1));
        }
        ig += stride;
        --ileft;
    };
    // the ring's first loads go out before the residual image is staged (see sweep_body_gen; past the wave's last group `issue`
    // re-reads column N - 1: no load sits under a branch)
#pragma unroll
    for (int d = 0; d < NB; ++d) issue(buf[d]);
    {
        double n2 = 0.0;
        for (int m = tid; m < KP; m += kSweepThreads) {
            const double rv = m < Mv ? r[m] : 0.0;
            lds[r_slot<VEC>(m)] = rv;
            n2 = fma(rv, rv, n2);
        }
        for (int s = 32; s >= 1; s >>= 1) n2 += __shfl_xor(n2, s, kWave);
        __syncthreads();
        if (lane == 0) red[wave] = n2;
        __syncthreads();
        n2 = (red[0] + red[1]) + (red[2] + red[3]);
        if (bid == 0 && tid == 0) st->rnorm2 = n2;
        if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break  (:79,:132)
            if (bid == 0 && tid == 0) st->done |= STOP_EPS;
            return;
        }
    }
    PH_STAMP(1);
    auto consume = [&](const VT(&b)[U]) {
        double a[CU_];
#pragma unroll
        for (int c = 0; c < CU_; ++c) a[c] = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = u / NCH, t = u % NCH;
            if constexpr (VEC == 4) {
                const f64x2 r01 = rs[(t * 2 + 0) * kWave + lane];
                const f64x2 r23 = rs[(t * 2 + 1) * kWave + lane];
                a[c] = fma((double)b[u].x, r01.x, a[c]);
                a[c] = fma((double)b[u].y, r01.y, a[c]);
                a[c] = fma((double)b[u].z, r23.x, a[c]);
                a[c] = fma((double)b[u].w, r23.y, a[c]);
            } else {
                const f64x2 r01 = rs[t * kWave + lane];
                a[c] = fma((double)b[u].x, r01.x, a[c]);
                a[c] = fma((double)b[u].y, r01.y, a[c]);
            }
        }
        double vs[SETS];
#pragma unroll
        for (int set = 0; set < SETS; ++set) {
            double v;
            if constexpr (CPU == 4) {
                v = xs16(xs32(a[set * 4 + 0], a[set * 4 + 2]), xs32(a[set * 4 + 1], a[set * 4 + 3]));
            } else {
                v = xs32(a[set * 2 + 0], a[set * 2 + 1]);
                v = xs16(v, v);
            }
            v = row_xsum(v);
            vs[set] = v;
            const int64_t col = cg * CU_ + set * CPU + q;
            if (col < N) {
                const double av = fabs(v);
                if (av > bestv) {  // a lane's columns arrive in increasing order: '>' keeps the first maximum
                    bestv = av;
                    besti = (int)col;
                }
            }
        }
        // the c values are staged (see sweep_body_gen): the LW = 64 / CPU lanes that hold a column's total are SETS x SLOTS; lane
        // (q, set, slot) keeps column set * CPU + q of the unit whose number is slot mod SLOTS, and every SLOTS units all 64 lanes store
        {
            const int set = (lane & (LW - 1)) / SLOTS, slot = (lane & (LW - 1)) % SLOTS;
            if (slot == cslot) {
                double v = vs[0];
                if constexpr (SETS == 2) v = set == 1 ? vs[1] : vs[0];
                const int64_t col = cg * CU_ + set * CPU + q;
#pragma unroll
                for (int k = 0; k < KS; ++k)
                    if (k == kslot) {
                        cst[k] = v;
                        ccst[k] = col < N ? (int)col : -1;
                    }
            }
            // 64 KS columns per lane set, then KS stores back to back: c writes that trickle out among the reads cost the DRAM far
            // more than their bytes (M = 256 Float32, 8 MiB of c beside 1 GiB of A: 168 us, 159 us with the stores aimed at one
            // L2-resident line -- tools/probes/ph_trace.hip); a wave of the 1-GiB table keeps ALL its columns until its end
            if (++cslot == SLOTS) {
                cslot = 0;
                if (++kslot == KS) flush();
            }
        }
        cg += stride;
        --cleft;
    };
    if (T >= 2 * NB) {
        const int64_t groups = T / NB - 1;
        for (int64_t g = 0; g < groups; ++g) {
#pragma unroll
            for (int d = 0; d < NB; ++d) {
                consume(buf[d]);
                issue(buf[d]);
            }
        }
    }
    while (cleft > 0) {
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            if (cleft == 0) break;
            consume(buf[d]);
            if (ileft > 0) issue(buf[d]);
        }
    }
    flush();
    PH_STAMP(5);
    if ((lane & 15) == 0) {
        redv[wave * 4 + (lane >> 4)] = bestv;
        redi[wave * 4 + (lane >> 4)] = besti;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = redv[0];
        int bi = redi[0];
        for (int qq = 1; qq < 4 * NW; ++qq)
            if (better(redv[qq], redi[qq], bv, bi)) {
                bv = redv[qq];
                bi = redi[qq];
            }
        pval[bid] = bv;
        pidx[bid] = bi;
    }
}
template <typename TA, int NCH, int CPU>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_short(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, int KP) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sweep_body_short<TA, NCH, CPU>(A, ld, Mv, N, r, cvec, pval, pidx, st, eps, check_eps, skipmask, (int)blockIdx.x, (int)gridDim.x, KP, lds);
}

// ---------------------------------------------------------------------------------------------
// The same sweep with its columns handed out AT RUN TIME (one residual image: !PH).  With a static split the workgroups of one
// 1-GiB launch finish between 139 and 160 us (which ones are slow changes from launch to launch), so the
// launch lasts 3.5-6 % longer than its average workgroup.  Round 5 measured what does NOT work: claims made by the streaming
// waves themselves (an atomic's answer returns in order with the loads of the same wave: +29 us) and one counter for the chip
// (~7 ns per claim).  Here a FIFTH wave per workgroup does the claiming and nothing else:
//   pools    every workgroup owns the groups of four neighbouring columns that the static split would give it (group g' belongs to
//            workgroup g' mod grid) and a counter over them, on a line of its own kClaimStride words from the next one: an
//            agent-scope atomic is a read-modify-write at the memory side, ~90 ns of ONE channel beside the stream -- 8 counters
//            for the chip made the launch 181-205 us, 22 made it 161-165 (the static split: 155); one per workgroup spreads
//            the ~93 claims of a workgroup's share over as many channels.  A workgroup claims from its own counter and, once
//            that is empty, STEALS from the following workgroups' (b + 1, b + 2, ...: eight consecutive block ids sit on eight
//            XCDs under the round-robin dispatch, and the XCDs differ by several percent in speed) until npools - 1 of them in
//            a row had nothing left.  The N mod 4 last columns go to workgroup 0 up front.
//            With dyn_div >= 2 (csmp_tune sweep_dyn = n) a pool has a STATIC head: its first groups go to its own workgroup
//            without a claim, only the last 1 / dyn_div of it (at least two groups) is claimed and can be stolen.  Measured level
//            with the static split at 1/16 (the claims then cost nothing, and the launch is no shorter: its tail is the memory
//            system draining, not work a free wave could take over -- DESIGN.md section 0.1).
//   claimer  two claims in flight (agent-scope atomic adds: ~1.1-1.3 us each beside the stream), each published as ONE column
//            per streaming wave into that wave's ring of kClaimQ slots in the LDS -- single producer, single consumer, no LDS
//            atomics; a slot is EMPTY, a column index, or END.
//   stream   waves 0..3 run sweep_body_gen's ring of NB units; at a column's first unit the wave takes its next column from its
//            slot ring (a broadcast ds_read; it frees the slot at once).  Every buffer of the ring carries its column and unit
//            index, so consuming needs no queue.  After END the ring is refilled with loads of ONE 16-byte vector (all lanes the
//            same address) that nobody consumes: no load sits under a branch and the compiler's counted waits stay exact.
// Every column's sum is formed by one wave in the same lane order as in the static body, and the maxima are compared
// lexicographically (value, then lower index): the results do not depend on who swept which column -- bit-identical.
// The counters: two sets per solver slot; a launch claims from one and workgroup 0 zeroes the other for the slot's next sweep
// (kernel boundary in between).  dynamic LDS: KP + 48 doubles.
constexpr int kClaimStride = 1088;  // words between two counters: 4 KiB + 256 B, so that neighbours differ in every plausible channel-interleave bit
constexpr int kClaimMaxWgs = 512;   // largest grid the dynamic sweep is launched on (a solver slot holds two sets of that many counters)
constexpr int kClaimQ = 8;        // slots per streaming wave (a power of two)
constexpr int kClEmpty = -2, kClEnd = -1;
constexpr int kSweepDynThreads = kSweepThreads + kWave;
inline size_t sweep_dyn_lds_bytes(int KP) { return (size_t)(KP + 8 + 16 + 8 + 16) * sizeof(double); }
using lds_int_ptr = __attribute__((address_space(3))) int*;
__device__ __forceinline__ int q_load(lds_int_ptr p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void q_store(lds_int_ptr p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename TA, int U, int NB>
__device__ __forceinline__ void sweep_body_dyn(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, const int bid, const int nblk, const int KP, unsigned* __restrict__ claim,
    unsigned* __restrict__ claim_next, const int npools_div, double* lds) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    constexpr int UR = U * ROWS;
    constexpr int NW = kSweepThreads / kWave;
    constexpr int Q = kClaimQ;
    static_assert((NB - 1) * U < 64, "the ring must fit the 6-bit vmcnt");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = nblk;  // pools: one per workgroup
    const int npools = npools_div & 0xffff, dyn_div = npools_div >> 16;  // (dyn_div >= 2: only the last 1 / dyn_div of a pool is claimed)
    if (bid == 0 && wave == NW)  // (before any way out)
        for (int p = lane; p < P; p += kWave) claim_next[p * kClaimStride] = 0u;
    if (st->done & skipmask) return;
    const int nvec = Mv / VEC;
    double* red = lds + KP;
    double* redv = red + 8;
    int* redi = reinterpret_cast<int*>(redv + 4 * NW);
    lds_int_ptr q = (lds_int_ptr)(redi + 16);
    const f64x2* rs = reinterpret_cast<const f64x2*>(lds);
    const int nunit = (Mv + UR - 1) / UR;
    const int Mst = nunit * UR;
    if (tid < NW * Q) q_store(q + tid, kClEmpty);
    {
        constexpr int RP = 16;
        double n2 = 0.0;
        if (wave < NW) {
            for (int m0 = tid; m0 < Mst; m0 += RP * kSweepThreads) {
                double rv[RP];
#pragma unroll
                for (int qq = 0; qq < RP; ++qq) {
                    const int m = m0 + qq * kSweepThreads;
                    rv[qq] = m < Mv ? r[m] : 0.0;
                }
#pragma unroll
                for (int qq = 0; qq < RP; ++qq) {
                    const int m = m0 + qq * kSweepThreads;
                    if (m < Mst) lds[r_slot<VEC>(m)] = rv[qq];
                    n2 = fma(rv[qq], rv[qq], n2);
                }
            }
            for (int s = 32; s >= 1; s >>= 1) n2 += __shfl_xor(n2, s, kWave);
        }
        __syncthreads();
        if (lane == 0 && wave < NW) red[wave] = n2;
        __syncthreads();
        n2 = (red[0] + red[1]) + (red[2] + red[3]);
        if (bid == 0 && tid == 0) st->rnorm2 = n2;
        if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break  (:79,:132)
            if (bid == 0 && tid == 0) st->done |= STOP_EPS;
            return;
        }
    }
    if (wave == NW) {
        // ---- the claimer: lane w < 4 serves streaming wave w
        const int64_t NG = N >> 2;  // full groups of four columns
        int wpos = 0;
        auto publish = [&](int col) {  // (lanes < NW: each its own slot ring)
            lds_int_ptr slot = q + lane * Q + (wpos & (Q - 1));
            while (q_load(slot) != kClEmpty) __builtin_amdgcn_s_sleep(2);
            q_store(slot, col);
            ++wpos;
        };
        if (bid == 0 && lane < (int)(N & 3)) publish((int)(NG * 4 + lane));
        const int ntry = npools < P ? npools : P;  // empty counters in a row that end the search
        int pcur = bid, tried = 0;
        unsigned g0 = 0, g1 = 0;
        int p0, p1;
        // the STATIC head of a pool: its first groups go to the pool's own workgroup without a claim (no atomics, the columns the
        // static split would give it); only the last 1 / dyn_div of every pool -- at least two groups -- is claimed and can be stolen
        auto head_of = [&](int pp) -> int64_t {
            const int64_t cnt = NG > pp ? (NG - pp + P - 1) / P : 0;
            if (dyn_div < 2) return 0;
            const int64_t tail = cnt / dyn_div > 2 ? cnt / dyn_div : 2;
            return cnt > tail ? cnt - tail : 0;
        };
        {
            const int64_t head = head_of(bid);
            for (int64_t gg = 0; gg < head; ++gg)
                if (lane < NW) publish((int)(((gg * P + bid) << 2) + lane));
        }
        auto ask = [&](unsigned& g, int& pp) {
            pp = pcur;
            if (lane == 0) g = __hip_atomic_fetch_add(claim + pcur * kClaimStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto take = [&](unsigned& g, int& pp) {  // the answer of the older claim; then the next claim goes out in its place
            const int64_t gg = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)g) + head_of(pp);
            const int64_t cnt = NG > pp ? (NG - pp + P - 1) / P : 0;
            if (gg < cnt) {
                if (lane < NW) publish((int)(((gg * P + pp) << 2) + lane));
                if (pp == pcur) tried = 0;
            } else if (pp == pcur) {
                ++tried;
                pcur = pcur + 1 == P ? 0 : pcur + 1;
            }
            if (tried < ntry) ask(g, pp);
        };
        ask(g0, p0);
        ask(g1, p1);
        while (tried < ntry) {
            take(g0, p0);
            if (tried < ntry) take(g1, p1);
        }
        if (lane < NW) publish(kClEnd);
        return;
    }
    // ---- the streaming waves
    double bestv = -1.0;
    int besti = 0x7fffffff;
    VT buf[NB][U];
    int bc[NB], bu[NB];
    int icol = 0, ib = 0, ipos = 0;
    bool live = true;
    double acc = 0.0;
    double cst = 0.0;  // c values staged 64 to a store instruction (see sweep_body_gen)
    int ccst = -1, cslot = 0;
    lds_int_ptr myq = q + wave * Q;
    auto issue = [&](VT(&b)[U], int& c_, int& u_) {
        if (ib == 0 && live) {
            lds_int_ptr slot = myq + (ipos & (Q - 1));
            int c = q_load(slot);
            while (c == kClEmpty) {
                __builtin_amdgcn_s_sleep(1);
                c = q_load(slot);
            }
            c = __builtin_amdgcn_readfirstlane(c);
            if (c == kClEnd) {
                live = false;
            } else {
                if (lane == 0) q_store(slot, kClEmpty);
                ++ipos;
                icol = c;
            }
        }
        const VT* pc = reinterpret_cast<const VT*>(A + (live ? (int64_t)icol * ld : 0));
        const int vb = ib * (U * kWave) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = vb + u * kWave;
            b[u] = __builtin_nontemporal_load(pc + (live ? (v < nvec ? v : nvec - 1) : 0));
        }
        c_ = live ? icol : -1;
        u_ = ib;
        if (++ib == nunit) ib = 0;
    };
    auto consume = [&](const VT(&b)[U], const int c_, const int u_) {
        if (c_ < 0) return;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = u_ * U + u;
            if constexpr (VEC == 4) {
                const f64x2 r01 = rs[(t * 2 + 0) * kWave + lane];
                const f64x2 r23 = rs[(t * 2 + 1) * kWave + lane];
                acc = fma((double)b[u].x, r01.x, acc);
                acc = fma((double)b[u].y, r01.y, acc);
                acc = fma((double)b[u].z, r23.x, acc);
                acc = fma((double)b[u].w, r23.y, acc);
            } else {
                const f64x2 r01 = rs[t * kWave + lane];
                acc = fma((double)b[u].x, r01.x, acc);
                acc = fma((double)b[u].y, r01.y, acc);
            }
        }
        if (u_ == nunit - 1) {  // the column's last unit
            acc = wave_xsum(acc);
            if (lane == cslot) {
                cst = acc;
                ccst = c_;
            }
            if (++cslot == kWave) {
                if (ccst >= 0) cvec[ccst] = cst;
                ccst = -1;
                cslot = 0;
            }
            const double av = fabs(acc);
            if (better(av, c_, bestv, besti)) {
                bestv = av;
                besti = c_;
            }
            acc = 0.0;
        }
    };
#pragma unroll
    for (int d = 0; d < NB; ++d) issue(buf[d], bc[d], bu[d]);
    while (live) {
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            consume(buf[d], bc[d], bu[d]);
            issue(buf[d], bc[d], bu[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < NB; ++d) consume(buf[d], bc[d], bu[d]);
    if (ccst >= 0) cvec[ccst] = cst;
    if (lane == 0) {
        redv[wave] = bestv;
        redi[wave] = besti;
    }
    lds_barrier();  // (the claimer has left: a barrier counts the waves that are still running)
    if (tid == 0) {
        double bv = redv[0];
        int bi = redi[0];
        for (int qq = 1; qq < NW; ++qq)
            if (better(redv[qq], redi[qq], bv, bi)) {
                bv = redv[qq];
                bi = redi[qq];
            }
        pval[bid] = bv;
        pidx[bid] = bi;
    }
}
template <typename TA, int U, int NB>
__global__ __launch_bounds__(kSweepDynThreads) void k_sweep_dyn(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask, int KP, unsigned* __restrict__ claim, unsigned* __restrict__ claim_next, int npools_div) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sweep_body_dyn<TA, U, NB>(A, ld, Mv, N, r, cvec, pval, pidx, st, eps, check_eps, skipmask, (int)blockIdx.x, (int)gridDim.x, KP, claim,
                              claim_next, npools_div, lds);
}

// ---------------------------------------------------------------------------------------------
// block-wide lexicographic arg-max over (v, i) pairs held one per thread (256 threads)
__device__ __forceinline__ void block_argmax(double& v, int& i, double* sv, int* si) {
    const int tid = threadIdx.x;
    sv[tid] = v;
    si[tid] = i;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s && better(sv[tid + s], si[tid + s], sv[tid], si[tid])) {
            sv[tid] = sv[tid + s];
            si[tid] = si[tid + s];
        }
        __syncthreads();
    }
    v = sv[0];
    i = si[0];
}

// Control kernel after a sweep (ONE workgroup): final arg-max over the workgroup partials, then
// the reference's guards.  mode 1 (OMP): nnz < M (:63) and "i not in x.nzind" (:66) -- a failed
// guard makes every later update! the same no-op, so the solve is flagged done.  mode 0 (MP):
// no guards, atoms may repeat (:28-29).
__global__ __launch_bounds__(256) void k_select(const double* __restrict__ pval, const int* __restrict__ pidx,
                                                int nblk, const double* __restrict__ cvec,
                                                const int* __restrict__ sel, DevState* st, int M, int kcap,
                                                int mode, int skipmask) {
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ int found;
    const int tid = threadIdx.x;
    if (st->done & skipmask) {
        if (tid == 0) st->go = 0;
        return;
    }
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int q = tid; q < nblk; q += 256)
        if (better(pval[q], pidx[q], bv, bi)) {
            bv = pval[q];
            bi = pidx[q];
        }
    block_argmax(bv, bi, sv, si);
    const int nsel = st->nsel;
    if (tid == 0) found = 0;
    __syncthreads();
    if (mode == 1) {
        for (int q = tid; q < nsel; q += 256)
            if (sel[q] == bi) found = 1;
    }
    __syncthreads();
    if (tid == 0) {
        int go = 1;
        if (mode == 1) {
            if (nsel >= M || nsel >= kcap) {
                st->done |= STOP_FULL;
                go = 0;
            } else if (found) {
                st->done |= STOP_STAG;
                go = 0;
            }
        }
        st->cand = bi;
        st->cval = cvec[bi];
        st->j = nsel;
        st->go = go;
    }
}

// ---------------------------------------------------------------------------------------------
// Top-S selection over c = A'r: partialsortperm(abs(Ar), 1:S, rev=true) (src/matchingpursuit.jl:192)
// -- descending by |c|, ties by ascending index.  Small S (GOMP's l): each workgroup takes the S
// best of its chunk by S rounds of block arg-max in LDS, a single workgroup merges.  Large S (SP's
// k): radix select on the IEEE bits of |c| (order-preserving for non-negative doubles), then a
// rank sort of the survivors.
constexpr int kTopChunk = 2048;  // |c| values per workgroup in the local stage
constexpr int kTopSmall = 16;    // largest S served by the arg-max path

__global__ __launch_bounds__(256) void k_top_local(const double* __restrict__ cvec, int64_t N, int S,
                                                   double* __restrict__ lv, int* __restrict__ li) {
    __shared__ double v[kTopChunk];
    __shared__ double sv[256];
    __shared__ int si[256];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kTopChunk;
    for (int t = tid; t < kTopChunk; t += 256) v[t] = (base + t < N) ? fabs(cvec[base + t]) : -1.0;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int t = tid; t < kTopChunk; t += 256)
            if (v[t] > bv) {  // ascending t per thread: '>' keeps the lowest index on ties
                bv = v[t];
                bi = t;
            }
        block_argmax(bv, bi, sv, si);
        if (tid == 0) {
            lv[(int64_t)blockIdx.x * S + s] = bv;
            li[(int64_t)blockIdx.x * S + s] = (bv >= 0.0) ? (int)(base + bi) : -1;
            if (bv >= 0.0) v[bi] = -1.0;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_top_merge(const double* __restrict__ lv, const int* __restrict__ li, int n,
                                                   int S, int* __restrict__ cands, double* __restrict__ cvals,
                                                   int* __restrict__ ncands) {
    extern __shared__ __attribute__((aligned(16))) double mv[];  // n values
    __shared__ double sv[256];
    __shared__ int si[256];
    const int tid = threadIdx.x;
    for (int t = tid; t < n; t += 256) mv[t] = (li[t] >= 0) ? lv[t] : -1.0;
    __syncthreads();
    int cnt = 0;
    for (int s = 0; s < S; ++s) {
        double bv = -1.0;
        int bi = 0x7fffffff, bp = -1;
        for (int t = tid; t < n; t += 256)
            if (better(mv[t], li[t], bv, bi) && mv[t] >= 0.0) {
                bv = mv[t];
                bi = li[t];
                bp = t;
            }
        // arg-max on (value, atom index); carry the position through a second pass
        double rv = bv;
        int ri = bi;
        block_argmax(rv, ri, sv, si);
        if (rv >= 0.0 && bv == rv && bi == ri && bp >= 0) mv[bp] = -1.0;  // the unique owner retires it
        if (tid == 0 && rv >= 0.0) {
            cands[s] = ri;
            cvals[s] = rv;
        }
        cnt += (rv >= 0.0);
        __syncthreads();
    }
    if (tid == 0) *ncands = cnt;
}

// ---- radix select (large S): the S largest |c| of N, S up to the support capacity.
// Keys are the bit patterns of |c| (monotone for finite values).  The digits are 11 bits wide (the last one 8): pass 0 fixes
// the exponent, pass 1 the leading 11 mantissa bits -- after which the bucket that holds the S-th largest key carries a
// handful of entries for anything but massively tied data.  As soon as that bucket holds at most `settle` keys the selection is
// SETTLED: the later histogram launches (enqueued blindly by the host, which never reads anything back) return at once,
// k_rs_collect gathers the keys above the bucket and the bucket itself, and k_rs_finish picks the `remaining` best of the
// bucket exactly.  The scan of a pass is done by the LAST workgroup of its histogram launch (ticket counter): one launch per
// pass.  Typical cost at N = 131072, S = 1024: 2 live passes + the tail launch (k_rs_tail: the remaining passes, a no-op once settled) + collect + finish, ~55 us (it was 19
// launches, ~200 us, with 8-bit digits and a single-workgroup rank sort).
constexpr int kRsDigit = 11, kRsBins = 1 << kRsDigit, kRsPasses = 6;
struct RsState {
    unsigned long long prefix;  // bits fixed so far (high bits)
    int pass;                   // next digit, 0 = most significant
    int remaining;              // how many still to take among keys matching the prefix
    int n_gt, n_eq;             // append counters of k_rs_collect
    int settled;                // the prefix bucket is small enough: no further passes
    unsigned int ticket;        // workgroups of the current histogram launch that are done
    unsigned int hist[kRsBins];
};

__device__ __forceinline__ unsigned long long abs_key(double c) {
    return (unsigned long long)__double_as_longlong(fabs(c));  // NaN sorts above inf; inputs are finite
}
// bits fixed after `pass` passes (the sign bit counts as fixed: it is 0), and the mask of those bits
__device__ __forceinline__ int rs_fixed_bits(int pass) { return pass >= kRsPasses ? 64 : 1 + kRsDigit * pass; }
__device__ __forceinline__ unsigned long long rs_himask(int fixed) { return fixed >= 64 ? ~0ull : ~(~0ull >> fixed); }

__global__ __launch_bounds__(256) void k_rs_init(RsState* rs, int S) {
    if (threadIdx.x == 0) {
        rs->prefix = 0;
        rs->pass = 0;
        rs->remaining = S;
        rs->n_gt = 0;
        rs->n_eq = 0;
        rs->settled = 0;
        rs->ticket = 0;
    }
    for (int b = threadIdx.x; b < kRsBins; b += 256) rs->hist[b] = 0;
}

// One pass: histogram of the current digit over the keys that match the prefix; the last workgroup to finish walks the
// histogram from the largest digit down, extends the prefix and decides whether the selection is settled.
__device__ __forceinline__ void rs_hist_body(const double* __restrict__ cvec, int64_t N, RsState* rs, int settle, unsigned int* h,
                                             unsigned int* part, unsigned int& last) {
    const int tid = threadIdx.x;
    if (rs->settled || rs->pass >= kRsPasses) return;
    const int pass = rs->pass;
    const int fixed = rs_fixed_bits(pass), width = (pass + 1 < kRsPasses) ? kRsDigit : 64 - fixed, shift = 64 - fixed - width;
    const unsigned long long prefix = rs->prefix, himask = rs_himask(fixed);
    const unsigned int dmask = (1u << width) - 1u;
    for (int b = tid; b < kRsBins; b += 256) h[b] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < N; i += (int64_t)gridDim.x * 256) {
        const unsigned long long key = abs_key(cvec[i]);
        if ((key & himask) == prefix) atomicAdd(&h[(unsigned int)(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int b = tid; b < kRsBins; b += 256)
        if (h[b]) atomicAdd(&rs->hist[b], h[b]);
    __threadfence();
    __syncthreads();
    if (tid == 0) last = (atomicAdd(&rs->ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // thread t owns the 8 digits 2047 - 8t .. 2040 - 8t (descending); prefix sums over the threads, then the owner of the
    // crossing walks its 8 bins
    constexpr int PER = kRsBins / 256;
    unsigned int mine[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        mine[u] = __hip_atomic_load(&rs->hist[kRsBins - 1 - (tid * PER + u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += mine[u];
    }
    part[tid] = sum;
    __syncthreads();
    unsigned int before = 0;
    for (int t = 0; t < tid; ++t) before += part[t];  // (256 LDS reads per thread at most: a microsecond, once per pass)
    const unsigned int rem = (unsigned int)rs->remaining;
    __syncthreads();
    if (before < rem && before + sum >= rem) {  // exactly one thread, unless fewer than `remaining` keys exist at all
        unsigned int acc = before;
        int u = 0;
        for (; u < PER - 1; ++u) {
            if (acc + mine[u] >= rem) break;
            acc += mine[u];
        }
        const int b = kRsBins - 1 - (tid * PER + u);
        rs->prefix = prefix | ((unsigned long long)b << shift);
        rs->remaining = (int)(rem - acc);
        rs->pass = pass + 1;
        if ((int)mine[u] <= settle) rs->settled = 1;
    }
    if (tid == 255 && before + sum < rem) {  // N < S cannot happen (S is clamped to N); keep the state consistent anyway
        rs->pass = pass + 1;
        rs->remaining = (int)(rem - (before + sum)) < 0 ? 0 : (int)(rem - (before + sum));
    }
    for (int b = tid; b < kRsBins; b += 256) rs->hist[b] = 0;
    if (tid == 0) rs->ticket = 0;
}
__global__ __launch_bounds__(256) void k_rs_hist(const double* __restrict__ cvec, int64_t N, RsState* rs, int settle) {
    __shared__ unsigned int h[kRsBins];
    __shared__ unsigned int part[256];
    __shared__ unsigned int last;
    rs_hist_body(cvec, N, rs, settle, h, part, last);
}
// The passes after the first two, in ONE launch of ONE workgroup: for anything but massively tied data the selection is settled
// by then and this returns at once (it used to be four empty launches on the chain); a selection that is not settled is carried
// through its remaining digits here, a pass at a time over all N keys (slow, rare, exact).
__global__ __launch_bounds__(256) void k_rs_tail(const double* __restrict__ cvec, int64_t N, RsState* rs, int settle) {
    __shared__ unsigned int h[kRsBins];
    __shared__ unsigned int part[256];
    __shared__ unsigned int last;
    for (int it = 2; it < kRsPasses; ++it) {
        rs_hist_body(cvec, N, rs, settle, h, part, last);
        __threadfence();   // (the state the next pass reads was written by threads of this workgroup)
        __syncthreads();
    }
}

// keys above the prefix bucket -> gt_idx, keys inside it -> eq_idx (at most eq_cap are kept; more than that only when the
// selection never settled, i.e. after all passes: then the bucket is one exact key and k_rs_finish scans for the ties)
__global__ __launch_bounds__(256) void k_rs_collect(const double* __restrict__ cvec, int64_t N, RsState* rs,
                                                    int* __restrict__ gt_idx, int* __restrict__ eq_idx, int eq_cap) {
    const unsigned long long T = rs->prefix, himask = rs_himask(rs_fixed_bits(rs->pass));
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < N; i0 += stride) {  // (whole waves stay in the loop: ballots below)
        const int64_t i = i0 + threadIdx.x;
        const unsigned long long kh = i < N ? abs_key(cvec[i]) & himask : 0ull;
        const bool gt = i < N && kh > T, eq = i < N && kh == T;
        // one counter update per wave and list: the lanes that append take consecutive slots
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (mg) {
            int base = 0;
            if (lane == __ffsll((long long)mg) - 1) base = atomicAdd(&rs->n_gt, __popcll(mg));
            base = __shfl(base, __ffsll((long long)mg) - 1, kWave);
            if (gt) gt_idx[base + __popcll(mg & below)] = (int)i;
        }
        if (me) {
            int base = 0;
            if (lane == __ffsll((long long)me) - 1) base = atomicAdd(&rs->n_eq, __popcll(me));
            base = __shfl(base, __ffsll((long long)me) - 1, kWave);
            const int p = base + __popcll(me & below);
            if (eq && p < eq_cap) eq_idx[p] = (int)i;
        }
    }
}

// The keys above the bucket plus the `remaining` best of the bucket -- (|c| descending, index ascending) -- rank-sorted by
// the same order into cands.  (The append order above is arbitrary; the sort makes the output deterministic.)  Every
// workgroup builds the whole survivor list (in LDS when lds_pairs of (value, index) fit, else in `work`, which every workgroup
// fills with the same values) and ranks 256 of its entries.  If more keys tie at T than eq_idx holds (e.g. r = 0: every |c|
// is 0), the ties are taken by an in-order scan of c instead.
__global__ __launch_bounds__(256) void k_rs_finish(const double* __restrict__ cvec, int64_t N, RsState* rs,
                                                   const int* __restrict__ gt_idx, const int* __restrict__ eq_idx,
                                                   int eq_cap, int* __restrict__ work /*S ints*/,
                                                   int* __restrict__ cands, double* __restrict__ cvals,
                                                   int* __restrict__ ncands, int lds_pairs) {
    __shared__ int wcnt[4];
    __shared__ int taken;
    extern __shared__ __attribute__((aligned(16))) double rs_lds[];  // lds_pairs x (value, index) | eq_cap x (value, index)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngt = rs->n_gt, neq = rs->n_eq, take = min(rs->remaining, neq);
    const unsigned long long T = rs->prefix;
    const int n = ngt + take;
    const bool in_lds = n <= lds_pairs;
    double* sv = rs_lds;
    int* si = reinterpret_cast<int*>(rs_lds + lds_pairs);
    int* list = in_lds ? si : work;
    for (int t = tid; t < ngt; t += 256) list[t] = gt_idx[t];
    if (neq <= eq_cap) {
        // the bucket, staged with its values; rank by (|c| descending, index ascending), keep the `take` first
        double* ev = reinterpret_cast<double*>(si + lds_pairs + (lds_pairs & 1));
        int* ei = reinterpret_cast<int*>(ev + eq_cap);
        for (int t = tid; t < neq; t += 256) {
            const int o = eq_idx[t];
            ei[t] = o;
            ev[t] = fabs(cvec[o]);
        }
        __syncthreads();
        for (int t = tid; t < neq; t += 256) {
            const int me = ei[t];
            const double mv = ev[t];
            int rank = 0;
            for (int u = 0; u < neq; ++u) rank += (ev[u] > mv) || (ev[u] == mv && ei[u] < me);
            if (rank < take) list[ngt + rank] = me;
        }
    } else {
        if (tid == 0) taken = 0;
        __syncthreads();
        for (int64_t base = 0; base < N; base += 256) {
            const int64_t i = base + tid;
            const bool hit = i < N && abs_key(cvec[i]) == T;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            int off = taken;
            for (int w = 0; w < wave; ++w) off += wcnt[w];
            off += __popcll(m & ((1ull << lane) - 1ull));
            if (hit && off < take) list[ngt + off] = (int)i;
            __syncthreads();
            if (tid == 0) taken += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
            if (taken >= take) break;
        }
    }
    __syncthreads();
    // rank sort of the n survivors, 256 per workgroup: n^2 comparisons on global gathers cost 200 us at n = 512 in one
    // workgroup, hence the LDS staging and the split
    if (in_lds) {
        for (int t = tid; t < n; t += 256) sv[t] = fabs(cvec[si[t]]);
        __syncthreads();
        for (int t = blockIdx.x * 256 + tid; t < n; t += gridDim.x * 256) {
            const int me = si[t];
            const double mv = sv[t];
            int rank = 0;
            int u = 0;
            for (; u + 4 <= n; u += 4) {  // (branch-free: the short-circuit form costs a branch and an LDS round trip per entry)
                const f64x2 v01 = *reinterpret_cast<const f64x2*>(sv + u), v23 = *reinterpret_cast<const f64x2*>(sv + u + 2);
                const int i0 = si[u], i1 = si[u + 1], i2 = si[u + 2], i3 = si[u + 3];
                rank += (int)(v01.x > mv) | ((int)(v01.x == mv) & (int)(i0 < me));
                rank += (int)(v01.y > mv) | ((int)(v01.y == mv) & (int)(i1 < me));
                rank += (int)(v23.x > mv) | ((int)(v23.x == mv) & (int)(i2 < me));
                rank += (int)(v23.y > mv) | ((int)(v23.y == mv) & (int)(i3 < me));
            }
            for (; u < n; ++u) rank += (int)(sv[u] > mv) | ((int)(sv[u] == mv) & (int)(si[u] < me));
            cands[rank] = me;
            cvals[rank] = mv;
        }
    } else {
        for (int t = blockIdx.x * 256 + tid; t < n; t += gridDim.x * 256) {
            const int me = work[t];
            const double mv = fabs(cvec[me]);
            int rank = 0;
            for (int u = 0; u < n; ++u) {
                const int o = work[u];
                const double ov = fabs(cvec[o]);
                rank += (ov > mv) || (ov == mv && o < me);
            }
            cands[rank] = me;
            cvals[rank] = mv;
        }
    }
    if (tid == 0 && blockIdx.x == 0) *ncands = n;
}

// ---------------------------------------------------------------------------------------------
// On-device QR append (classical Gram-Schmidt with one re-orthogonalisation, "CGS2", in its
// two-reduction form): Q is M x kcap Float64 column-major, split into slabs of 64 rows, one
// workgroup per slab.  A grid-wide sum is a kernel boundary (cheaper on MI355X than an in-kernel
// grid barrier): partial sums are written per slab and re-summed in a fixed order by every
// workgroup of the next kernel, so results are bitwise reproducible.
//   k_qr1:  pick the atom (arg-max partials or a candidate list) + the reference's guards;
//           a = A[:,cand];                 P1[:,g] = Q_g' a_g
//   k_qr2:  w1 = sum_g P1[:,g];  v = a - Q w1;  P2[:,g] = Q_g' v_g,  |v_g|^2,  v_g' r_g
//   k_qr3:  w2 = sum_g P2[:,g];  rho^2 = |v|^2 - |w2|^2;  q = (v - Q w2)/rho;  z_j = v'r/rho;
//           r -= q z_j;  R[:,j] = [w1 + w2; rho];  support += cand
// (r is orthogonal to Q, so q'b == q'r up to rounding; z accumulates Q'b for the final solve.)
// The kernels are latency-bound (a slab is 128 KiB at j = 256), so every phase issues all of its
// loads before the first use: one L2 round trip per phase instead of one per column.

// partial Q_g' x for this workgroup's slab: wave w covers rows 16w..16w+15 (one 128-B line per
// column), lane <-> column, four 64-column chunks (32 x 16-B loads) in flight per lane.
// out is laid out [column][slab] so the consumer reads its G partials contiguously.
__device__ __forceinline__ void slab_qt_x(const double* __restrict__ Q, int64_t ldq, int g, int G, int j,
                                          const double* xs /*LDS, 64*/, double* part /*LDS 4*jpad*/, int jpad,
                                          double* __restrict__ out /*global [kcap][G]*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double xr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xr[i] = xs[wave * 16 + i];
    const double* base = Q + g * kSlabRows + wave * 16;
    for (int c0 = 0; c0 < j; c0 += 4 * kWave) {
        f64x2 v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * kWave + lane;
            const f64x2* q = reinterpret_cast<const f64x2*>(base + (int64_t)(c < j ? c : 0) * ldq);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[u][i] = (c < j) ? q[i] : (f64x2)0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * kWave + lane;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = fma(v[u][i].x, xr[2 * i], a0);
                a1 = fma(v[u][i].y, xr[2 * i + 1], a1);
            }
            if (c < j) part[wave * jpad + c] = a0 + a1;
        }
    }
    __syncthreads();
    for (int c = tid; c < j; c += kQrThreads)
        out[(int64_t)c * G + g] = (part[c] + part[jpad + c]) + (part[2 * jpad + c] + part[3 * jpad + c]);
    __syncthreads();
}

// ---- prefetched forms: the Q-slab operands of a phase are requested at kernel entry, before the
// control block and the partial sums are known (jh >= j is the host's upper bound on the column
// count), so the whole kernel is about one L2 round trip deep instead of one per phase.
// W / NX set how many columns are requested up front: the stand-alone kernels use deep windows
// (latency matters, registers do not); inside k_tick the stages run hidden under another signal's
// sweep and must stay below 128 VGPRs so that the fused kernel keeps 4 waves per SIMD.
template <int W>
struct QtPre {
    f64x2 v[W][8];  // columns c = u*64 + lane (u < W), rows 16w .. 16w+15 of the slab
};
template <int W>
__device__ __forceinline__ void slab_qt_prefetch(QtPre<W>& P, const double* __restrict__ Q, int64_t ldq, int g, int jh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* base = Q + g * kSlabRows + wave * 16;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int c = u * kWave + lane;
        const f64x2* q = reinterpret_cast<const f64x2*>(base + (int64_t)(c < jh ? c : 0) * ldq);
#pragma unroll
        for (int i = 0; i < 8; ++i) P.v[u][i] = (c < jh) ? q[i] : (f64x2)0.0;
    }
    asm volatile("" ::: "memory");  // keep the requests ahead of everything that follows
}
// partial Q_g' x with the first W*64 columns taken from the prefetch (see slab_qt_x)
template <int W>
__device__ __forceinline__ void slab_qt_x_pre(const QtPre<W>& P, const double* __restrict__ Q, int64_t ldq, int g, int G,
                                              int j, const double* xs, double* part, int jpad,
                                              double* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double xr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xr[i] = xs[wave * 16 + i];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int c = u * kWave + lane;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a0 = fma(P.v[u][i].x, xr[2 * i], a0);
            a1 = fma(P.v[u][i].y, xr[2 * i + 1], a1);
        }
        if (c < j) part[wave * jpad + c] = a0 + a1;
    }
    const double* base = Q + g * kSlabRows + wave * 16;
    for (int c0 = W * kWave; c0 < j; c0 += W * kWave) {  // columns beyond the prefetch window
        f64x2 v[W][8];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int c = c0 + u * kWave + lane;
            const f64x2* q = reinterpret_cast<const f64x2*>(base + (int64_t)(c < j ? c : 0) * ldq);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[u][i] = (c < j) ? q[i] : (f64x2)0.0;
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int c = c0 + u * kWave + lane;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = fma(v[u][i].x, xr[2 * i], a0);
                a1 = fma(v[u][i].y, xr[2 * i + 1], a1);
            }
            if (c < j) part[wave * jpad + c] = a0 + a1;
        }
    }
    __syncthreads();
    for (int c = tid; c < j; c += kQrThreads)
        out[(int64_t)c * G + g] = (part[c] + part[jpad + c]) + (part[2 * jpad + c] + part[3 * jpad + c]);
    __syncthreads();
}

template <int NX>
struct XmPre {
    f64x2 v[NX];  // columns c = stream + 8 i (i < NX), rows 2*l32, 2*l32+1 of the slab
};
template <int NX>
__device__ __forceinline__ void slab_xm_prefetch(XmPre<NX>& P, const double* __restrict__ Q, int64_t ldq, int g, int jh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stream = wave * 2 + (lane >> 5), l32 = lane & 31;
    const double* q = Q + g * kSlabRows + 2 * l32;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int c = stream + 8 * i;
        P.v[i] = (c < jh) ? *reinterpret_cast<const f64x2*>(q + (int64_t)c * ldq) : (f64x2)0.0;
    }
    asm volatile("" ::: "memory");
}
// x_g -= Q_g w with the first 8*NX columns taken from the prefetch (see slab_x_minus_qw)
template <int NX>
__device__ __forceinline__ void slab_x_minus_qw_pre(const XmPre<NX>& P, const double* __restrict__ Q, int64_t ldq, int g,
                                                    int j, const double* ws, double* xs, double* tmp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stream = wave * 2 + (lane >> 5), l32 = lane & 31;
    const double* q = Q + g * kSlabRows + 2 * l32;
    double ax0 = 0.0, ay0 = 0.0, ax1 = 0.0, ay1 = 0.0;
#pragma unroll
    for (int i = 0; i < NX; i += 2) {
        const int c0 = stream + 8 * i, c1 = stream + 8 * (i + 1);
        const double w0 = (c0 < j) ? ws[c0] : 0.0, w1 = (c1 < j) ? ws[c1] : 0.0;
        ax0 = fma(P.v[i].x, w0, ax0);
        ay0 = fma(P.v[i].y, w0, ay0);
        ax1 = fma(P.v[i + 1].x, w1, ax1);
        ay1 = fma(P.v[i + 1].y, w1, ay1);
    }
    for (int cb = 8 * NX; cb < j; cb += 8 * NX) {  // columns beyond the prefetch window, NX loads in flight
        f64x2 v[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = cb + stream + 8 * i;
            v[i] = (c < j) ? *reinterpret_cast<const f64x2*>(q + (int64_t)c * ldq) : (f64x2)0.0;
        }
#pragma unroll
        for (int i = 0; i < NX; i += 2) {
            const int c0 = cb + stream + 8 * i, c1 = cb + stream + 8 * (i + 1);
            const double w0 = (c0 < j) ? ws[c0] : 0.0, w1 = (c1 < j) ? ws[c1] : 0.0;
            ax0 = fma(v[i].x, w0, ax0);
            ay0 = fma(v[i].y, w0, ay0);
            ax1 = fma(v[i + 1].x, w1, ax1);
            ay1 = fma(v[i + 1].y, w1, ay1);
        }
    }
    tmp[stream * kSlabRows + 2 * l32] = ax0 + ax1;
    tmp[stream * kSlabRows + 2 * l32 + 1] = ay0 + ay1;
    __syncthreads();
    if (tid < kSlabRows) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t) s += tmp[t * kSlabRows + tid];
        xs[tid] -= s;
    }
    __syncthreads();
}

// ws[c] = sum_g P[c][g] (fixed order) for c < j; returns this thread's sum of squares
__device__ __forceinline__ double sum_partials(const double* __restrict__ P, int G, int j, double* ws) {
    double sq = 0.0;
    for (int c = threadIdx.x; c < j; c += kQrThreads) {
        const double* p = P + (int64_t)c * G;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int gg = 0;
        for (; gg + 16 <= G; gg += 16) {  // 8 x 16-B loads in flight
            f64x2 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = reinterpret_cast<const f64x2*>(p + gg)[i];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                s0 += t[i].x;
                s1 += t[i].y;
                s2 += t[i + 1].x;
                s3 += t[i + 1].y;
            }
        }
        for (; gg < G; ++gg) s0 += p[gg];
        const double s = (s0 + s1) + (s2 + s3);
        ws[c] = s;
        sq = fma(s, s, sq);
    }
    return sq;
}

// x_g -= Q_g w for this slab.  8 column streams (4 waves x 2 half-waves), each lane owns 2 rows
// (16-B loads, 512 B contiguous per column), 16 loads in flight per lane; result in xs (LDS).
__device__ __forceinline__ void slab_x_minus_qw(const double* __restrict__ Q, int64_t ldq, int g, int j,
                                                const double* ws /*LDS, j*/, double* xs /*LDS 64, in/out*/,
                                                double* tmp /*LDS 8*64*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stream = wave * 2 + (lane >> 5), l32 = lane & 31;
    const double* q = Q + g * kSlabRows + 2 * l32;
    double ax0 = 0.0, ay0 = 0.0, ax1 = 0.0, ay1 = 0.0;
    int c = stream;
    for (; c + 8 * 15 < j; c += 8 * 16) {
        f64x2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f64x2*>(q + (int64_t)(c + 8 * i) * ldq);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const double w0 = ws[c + 8 * i], w1 = ws[c + 8 * (i + 1)];
            ax0 = fma(v[i].x, w0, ax0);
            ay0 = fma(v[i].y, w0, ay0);
            ax1 = fma(v[i + 1].x, w1, ax1);
            ay1 = fma(v[i + 1].y, w1, ay1);
        }
    }
    for (; c < j; c += 8) {
        const f64x2 v = *reinterpret_cast<const f64x2*>(q + (int64_t)c * ldq);
        const double w0 = ws[c];
        ax0 = fma(v.x, w0, ax0);
        ay0 = fma(v.y, w0, ay0);
    }
    tmp[stream * kSlabRows + 2 * l32] = ax0 + ax1;
    tmp[stream * kSlabRows + 2 * l32 + 1] = ay0 + ay1;
    __syncthreads();
    if (tid < kSlabRows) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t) s += tmp[t * kSlabRows + tid];
        xs[tid] -= s;
    }
    __syncthreads();
}

// LDS carve shared by the QR kernels (dynamic): part[4*jpad] | ws[jpad] | xs[64] | tmp[512] | sc[8]
// spill (k_qr1s / k_qr2s / k_qr3s): supports beyond what 160 KiB of LDS hold (five vectors of ~3900 Float64) keep the five
// support-length vectors of workgroup g in GLOBAL memory (L2-resident: 5 jpad doubles per workgroup), only the slab scratch in
// LDS.  The bodies order their accesses to these vectors with __syncthreads() only (a workgroup-scope fence for global memory
// as well), so the same code runs on either kind of pointer; spill == nullptr is a compile-time constant in the LDS kernels.
__device__ __forceinline__ void qr_carve(double* base, int jpad, double*& part, double*& ws, double*& xs,
                                         double*& tmp, double*& sc, double* spill = nullptr, int g = 0) {
    if (spill) {
        part = spill + (size_t)g * 5 * (size_t)jpad;
        ws = part + 4 * jpad;
        xs = base;
    } else {
        part = base;
        ws = part + 4 * jpad;
        xs = ws + jpad;
    }
    tmp = xs + kSlabRows;
    sc = tmp + 8 * kSlabRows;
}
inline int qr_jpad(int kcap) { return ((kcap + 63) / 64) * 64 + 2; }
inline size_t qr_lds_bytes(int kcap) {
    return (size_t)(5 * qr_jpad(kcap) + kSlabRows + 8 * kSlabRows + 8) * sizeof(double);
}
inline size_t qr_spill_lds_bytes() { return (size_t)(kSlabRows + 8 * kSlabRows + 8) * sizeof(double); }
inline size_t qr_spill_doubles(int G, int kcap) { return (size_t)G * 5 * (size_t)qr_jpad(kcap); }

// mode 1 (OMP): atom = arg-max over the sweep's workgroup partials; guards nnz < M (:63) and
//   "i not in x.nzind" (:66) -- a failed guard makes every later update! the same no-op: done.
// mode 2 (GOMP): atom = cands[which] (the l best of the sweep, src/util.jl:129-134), skipped if
//   already in the support (util.jl:119); only a full support stops anything (:117).
// mode 3 (forward regression): atom = arg-max of the δ² scores of k_fr_sweep (findmax, src/forward.jl:63);
//   the step fails -- and fr stops -- unless min_δ^2 < max δ² (:64); nnz < n guard (:58).
// mode 4 (column-sharded OMP, csmp_shard_append): atom = cands[0], a LABEL (global column index) whose column
//   the caller supplies as a one-column dictionary (ld = 0); guards as mode 1.
// Every workgroup derives the same decision from the same device data; workgroup 0 publishes it
// (cand, j, go) for k_qr2 / k_qr3, which no workgroup of THIS launch reads.
template <typename TA, int W>
__device__ __forceinline__ void qr1_body(const TA* __restrict__ A, int64_t ld, int M,
                                                    const double* __restrict__ Q, int64_t ldq, DevState* st,
                                                    double* __restrict__ avec, double* __restrict__ P1, int G, int kcap,
                                                    int jpad, int mode, const double* __restrict__ pval,
                                                    const int* __restrict__ pidx, int nblk,
                                                    const int* __restrict__ cands, const int* __restrict__ ncands,
                                                    int which, const int* __restrict__ sel, int skipmask,
                                                    const double* __restrict__ r, double* __restrict__ P1s, int jh, const int g, double* lds,
                                                    const double min_d2 = 0.0, double* spill = nullptr) {
    double *part, *ws, *xs, *tmp, *sc;
    qr_carve(lds, jpad, part, ws, xs, tmp, sc, spill, g);
    const int tid = threadIdx.x;
    QtPre<W> pre;
    slab_qt_prefetch<W>(pre, Q, ldq, g, jh);
    if (st->done & skipmask) {
        if (g == 0 && tid == 0) st->go = 0;
        return;
    }
    const int nsel = st->nsel;
    int cand;
    bool low = false;  // mode 3: the best score does not exceed min_δ^2
    if (mode == 1 || mode == 3) {
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int q = tid; q < nblk; q += kQrThreads)
            if (better(pval[q], pidx[q], bv, bi)) {
                bv = pval[q];
                bi = pidx[q];
            }
        block_argmax(bv, bi, tmp, reinterpret_cast<int*>(tmp + kQrThreads));
        cand = bi;
        low = mode == 3 && !(min_d2 < bv);
        if (mode == 3 && g == 0 && tid == 0) st->cval = bv;  // maximum(P.δ²) (foba reads it: src/stepwise.jl:52)
    } else {
        cand = (which < *ncands) ? cands[which] : -1;
    }
    int found = 0;
    for (int q = tid; q < nsel; q += kQrThreads) found |= (sel[q] == cand);
    found = __syncthreads_or(found);
    const bool full = nsel >= M || nsel >= kcap;
    const bool go = cand >= 0 && cand < 0x7fffffff && !found && !full && !low;
    if (g == 0 && tid == 0) {
        st->cand = cand;
        st->j = nsel;
        st->go = go ? 1 : 0;
        if (full)
            st->done |= STOP_FULL;
        else if ((mode == 1 || mode == 3 || mode == 4) && (found || low || cand < 0 || cand == 0x7fffffff))
            st->done |= STOP_STAG;
    }
    if (!go) return;
    if (tid < kSlabRows) {
        const int row = g * kSlabRows + tid;
        const double a = (row < M) ? (double)A[(int64_t)cand * ld + row] : 0.0;
        xs[tid] = a;
        avec[row] = a;
        double n2 = a * a, ar = a * r[row];
        for (int s = 32; s >= 1; s >>= 1) {
            n2 += shx(n2, s);
            ar += shx(ar, s);
        }
        if (tid == 0) {
            P1s[2 * g] = n2;      // |a_g|^2
            P1s[2 * g + 1] = ar;  // a_g' r_g
        }
    }
    __syncthreads();
    slab_qt_x_pre<W>(pre, Q, ldq, g, G, nsel, xs, part, jpad, P1);
}

template <typename TA>
__global__ __launch_bounds__(kQrThreads) void k_qr1(const TA* __restrict__ A, int64_t ld, int M,
                                                    const double* __restrict__ Q, int64_t ldq, DevState* st,
                                                    double* __restrict__ avec, double* __restrict__ P1, int G, int kcap,
                                                    int jpad, int mode, const double* __restrict__ pval,
                                                    const int* __restrict__ pidx, int nblk,
                                                    const int* __restrict__ cands, const int* __restrict__ ncands,
                                                    int which, const int* __restrict__ sel, int skipmask,
                                                    const double* __restrict__ r, double* __restrict__ P1s, int jh,
                                                    double min_d2) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    qr1_body<TA, 4>(A, ld, M, Q, ldq, st, avec, P1, G, kcap, jpad, mode, pval, pidx, nblk, cands, ncands, which, sel, skipmask, r, P1s, jh, (int)blockIdx.x, lds, min_d2);
}
template <typename TA>
__global__ __launch_bounds__(kQrThreads) void k_qr1s(const TA* __restrict__ A, int64_t ld, int M,
                                                     const double* __restrict__ Q, int64_t ldq, DevState* st,
                                                     double* __restrict__ avec, double* __restrict__ P1, int G, int kcap,
                                                     int jpad, int mode, const double* __restrict__ pval,
                                                     const int* __restrict__ pidx, int nblk,
                                                     const int* __restrict__ cands, const int* __restrict__ ncands,
                                                     int which, const int* __restrict__ sel, int skipmask,
                                                     const double* __restrict__ r, double* __restrict__ P1s, int jh,
                                                     double min_d2, double* __restrict__ spill) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    qr1_body<TA, 4>(A, ld, M, Q, ldq, st, avec, P1, G, kcap, jpad, mode, pval, pidx, nblk, cands, ncands, which, sel, skipmask, r, P1s, jh, (int)blockIdx.x, lds, min_d2, spill);
}

// Publishes the new column (shared by the accept path of k_qr2 and by k_qr3).
__device__ __forceinline__ void qr_commit(double* __restrict__ Q, int64_t ldq, DevState* st, double* __restrict__ r,
                                          double* __restrict__ R, double* __restrict__ z, int* __restrict__ sel,
                                          int kcap, int g, int j, const double* xs, double rr, double rho, double zj,
                                          const double* W1 /*global or null*/, const double* ws /*LDS*/) {
    const int tid = threadIdx.x;
    if (tid < kSlabRows) {
        const int row = g * kSlabRows + tid;
        const double q = (rho > 0.0) ? xs[tid] / rho : 0.0;
        Q[(int64_t)j * ldq + row] = q;
        r[row] = fma(-q, zj, rr);
    }
    if (g == 0) {
        for (int c = tid; c < j; c += kQrThreads) R[(int64_t)j * kcap + c] = W1 ? W1[c] + ws[c] : ws[c];
        if (tid == 0) {
            R[(int64_t)j * kcap + j] = (rho > 0.0) ? rho : 1.0;  // degenerate column: coefficient 0
            z[j] = zj;
            sel[j] = st->cand;
            st->nsel = j + 1;
            st->steps += 1;
        }
    }
}

// Second kernel of the append.  With rho^2 = |a|^2 - |w1|^2 (Pythagoras) the Daniel-Gragg-Kaufman-
// Stewart test rho^2 >= |a|^2 / 2 says whether the first Gram-Schmidt pass lost accuracy to
// cancellation.  If it did not (always, for incoherent dictionaries with k << M) the column is
// committed here and k_qr3 returns at once; otherwise the second pass runs as before.
template <int NX>
__device__ __forceinline__ void qr2_body(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                    const double* __restrict__ avec, double* __restrict__ r,
                                                    const double* __restrict__ P1, const double* __restrict__ P1s,
                                                    int G, double* __restrict__ W1, double* __restrict__ vvec,
                                                    double* __restrict__ P2, double* __restrict__ P2s,
                                                    double* __restrict__ R, double* __restrict__ z,
                                                    int* __restrict__ sel, int kcap, int jpad, int force_reorth,
                                                    int jh, int optimistic, const int g, double* lds, double* spill = nullptr) {
    XmPre<NX> pre;
    slab_xm_prefetch<NX>(pre, Q, ldq, g, jh);
    if (!st->go) return;
    double *part, *ws, *xs, *tmp, *sc;
    qr_carve(lds, jpad, part, ws, xs, tmp, sc, spill, g);
    const int tid = threadIdx.x, j = st->j;
    double rr = 0.0, na2 = 0.0, ar = 0.0;
    if (tid < kSlabRows) {
        xs[tid] = avec[g * kSlabRows + tid];
        rr = r[g * kSlabRows + tid];
    }
    for (int gg = tid; gg < G; gg += kQrThreads) {
        na2 += P1s[2 * gg];
        ar += P1s[2 * gg + 1];
    }
    double w1sq = sum_partials(P1, G, j, ws);  // w1
    w1sq = block_sum256(w1sq, sc);
    na2 = block_sum256(na2, sc);
    ar = block_sum256(ar, sc);
    const double rho2 = na2 - w1sq;
    const bool accept = !force_reorth && rho2 >= 0.5 * na2 && rho2 > 0.0;
    if (!accept && optimistic) {
        // optimistic chain (k_qr3 is not launched): nothing is committed, the solve is flagged and the
        // host repeats it with the full three-kernel chain
        if (g == 0 && tid == 0) {
            st->done |= STOP_REORTH;
            st->go2 = 0;
        }
        return;
    }
    slab_x_minus_qw_pre<NX>(pre, Q, ldq, g, j, ws, xs, tmp);  // v_g = a_g - Q_g w1
    if (accept) {
        const double rho = sqrt(rho2);
        qr_commit(Q, ldq, st, r, R, z, sel, kcap, g, j, xs, rr, rho, ar / rho, nullptr, ws);
        if (g == 0 && tid == 0) st->go2 = 0;
        return;
    }
    if (g == 0) {
        for (int c = tid; c < j; c += kQrThreads) W1[c] = ws[c];
        if (tid == 0) st->go2 = 1;
    }
    if (tid < kSlabRows) {
        const double v = xs[tid];
        vvec[g * kSlabRows + tid] = v;
        double n2 = v * v, vr = v * rr;
        for (int s = 32; s >= 1; s >>= 1) {
            n2 += shx(n2, s);
            vr += shx(vr, s);
        }
        if (tid == 0) {
            P2s[2 * g] = n2;
            P2s[2 * g + 1] = vr;
        }
    }
    slab_qt_x(Q, ldq, g, G, j, xs, part, jpad, P2);
}

__global__ __launch_bounds__(kQrThreads) void k_qr2(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                    const double* __restrict__ avec, double* __restrict__ r,
                                                    const double* __restrict__ P1, const double* __restrict__ P1s,
                                                    int G, double* __restrict__ W1, double* __restrict__ vvec,
                                                    double* __restrict__ P2, double* __restrict__ P2s,
                                                    double* __restrict__ R, double* __restrict__ z,
                                                    int* __restrict__ sel, int kcap, int jpad, int force_reorth,
                                                    int jh, int optimistic) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    qr2_body<32>(Q, ldq, st, avec, r, P1, P1s, G, W1, vvec, P2, P2s, R, z, sel, kcap, jpad, force_reorth, jh, optimistic, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(kQrThreads) void k_qr2s(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                     const double* __restrict__ avec, double* __restrict__ r,
                                                     const double* __restrict__ P1, const double* __restrict__ P1s,
                                                     int G, double* __restrict__ W1, double* __restrict__ vvec,
                                                     double* __restrict__ P2, double* __restrict__ P2s,
                                                     double* __restrict__ R, double* __restrict__ z,
                                                     int* __restrict__ sel, int kcap, int jpad, int force_reorth,
                                                     int jh, int optimistic, double* __restrict__ spill) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    qr2_body<32>(Q, ldq, st, avec, r, P1, P1s, G, W1, vvec, P2, P2s, R, z, sel, kcap, jpad, force_reorth, jh, optimistic, (int)blockIdx.x, lds, spill);
}

__device__ __forceinline__ void qr3_body(double* __restrict__ Q, int64_t ldq, DevState* st,
                                         const double* __restrict__ vvec, double* __restrict__ r,
                                         const double* __restrict__ P2, const double* __restrict__ P2s,
                                         int G, const double* __restrict__ W1, double* __restrict__ R,
                                         double* __restrict__ z, int* __restrict__ sel, int kcap, int jpad, double* lds, double* spill = nullptr) {
    if (!st->go || !st->go2) return;
    double *part, *ws, *xs, *tmp, *sc;
    qr_carve(lds, jpad, part, ws, xs, tmp, sc, spill, (int)blockIdx.x);
    const int tid = threadIdx.x, g = blockIdx.x, j = st->j;
    double rr = 0.0, n2 = 0.0, vr = 0.0;
    if (tid < kSlabRows) {
        xs[tid] = vvec[g * kSlabRows + tid];
        rr = r[g * kSlabRows + tid];
    }
    for (int gg = tid; gg < G; gg += kQrThreads) {
        n2 += P2s[2 * gg];
        vr += P2s[2 * gg + 1];
    }
    double w2sq = sum_partials(P2, G, j, ws);  // w2
    w2sq = block_sum256(w2sq, sc);             // |w2|^2, fixed order
    n2 = block_sum256(n2, sc);
    vr = block_sum256(vr, sc);
    const double rho2 = n2 - w2sq;
    const double rho = (rho2 > 0.0) ? sqrt(rho2) : 0.0;
    const double zj = (rho > 0.0) ? vr / rho : 0.0;  // z_j = q_j' r
    slab_x_minus_qw(Q, ldq, g, j, ws, xs, tmp);      // v_g - Q_g w2
    qr_commit(Q, ldq, st, r, R, z, sel, kcap, g, j, xs, rr, rho, zj, W1, ws);
}
__global__ __launch_bounds__(kQrThreads) void k_qr3(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                    const double* __restrict__ vvec, double* __restrict__ r,
                                                    const double* __restrict__ P2, const double* __restrict__ P2s,
                                                    int G, const double* __restrict__ W1, double* __restrict__ R,
                                                    double* __restrict__ z, int* __restrict__ sel, int kcap, int jpad) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    qr3_body(Q, ldq, st, vvec, r, P2, P2s, G, W1, R, z, sel, kcap, jpad, lds);
}
__global__ __launch_bounds__(kQrThreads) void k_qr3s(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                     const double* __restrict__ vvec, double* __restrict__ r,
                                                     const double* __restrict__ P2, const double* __restrict__ P2s,
                                                     int G, const double* __restrict__ W1, double* __restrict__ R,
                                                     double* __restrict__ z, int* __restrict__ sel, int kcap, int jpad,
                                                     double* __restrict__ spill) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    qr3_body(Q, ldq, st, vvec, r, P2, P2s, G, W1, R, z, sel, kcap, jpad, lds, spill);
}

// ---------------------------------------------------------------------------------------------
// Tick kernel of the pipelined batch: THREE independent signals are in flight, one in each stage
// of an OMP step, and one launch advances all of them:
//     workgroups [0, G)        : k_qr2 stage of signal X   (commit the atom chosen two ticks ago)
//     workgroups [G, 2G)       : k_qr1 stage of signal Y   (select + first projection)
//     workgroups [2G, 2G + S)  : sweep of signal Z         (the HBM-bound part: S persistent workgroups)
// The short latency-bound chain stages run on ~128 CU slots while the sweep streams the
// dictionary, so a whole atom costs one sweep and one kernel boundary.  The stages of one signal
// are separated by kernel boundaries exactly as in the stand-alone chain (sweep -> qr1 -> qr2 in
// three consecutive ticks), so results are bit-identical to it.  A stage with active == 0 idles.
template <typename TA>
struct TickSweep {
    const TA* A; int64_t ld; int Mv; int64_t N;
    const double* r; double* cvec; double* pval; int* pidx; DevState* st;
    double eps; int check_eps, skipmask, nblk, active;
    int KP;    // rows of the residual image
    int pcap;  // PH: columns per wave the LDS holds partial sums for (sweep_body_ph)
    int npools;                    // DYN: empty pools in a row that end a workgroup's search, | (dyn_div << 16): 1 / dyn_div of a pool is claimed
    unsigned *claim, *claim_next;  // DYN: the column pools of this sweep, and the set to zero for the slot's next one (sweep_body_dyn)
};
template <typename TA>
struct TickQr1 {
    const TA* A; int64_t ld; int M;
    const double* Q; int64_t ldq; DevState* st; double* avec; double* P1;
    int G, kcap, jpad, mode;
    const double* pval; const int* pidx; int nblk_sweep;
    const int* cands; const int* ncands; int which; const int* sel; int skipmask;
    const double* r; double* P1s; int jh, active;
};
struct TickQr2 {
    double* Q; int64_t ldq; DevState* st; const double* avec; double* r;
    const double* P1; const double* P1s; int G;
    double* W1; double* vvec; double* P2; double* P2s; double* R; double* z; int* sel;
    int kcap, jpad, force_reorth, jh, optimistic, active;
};

// U, PH: the sweep body's unit size (16 / 8 / 4 loads) and whether the residual is staged in phases (sweep_body_gen).
// STEADY only names the kernel: the launches in which all three stages are live (every tick of a batch except the
// 2 + 2 that fill and drain the pipeline of a signal triple) get a symbol of their own, so that a kernel trace
// (rocprofv3 --kernel-trace --stats) reports the sweep-carrying ticks as one clean row.
// DYN: the sweep's columns are handed out at run time (sweep_body_dyn: a fifth wave per workgroup claims; in the append stages'
// workgroups that wave leaves at once).
template <typename TA, int U, bool PH, bool STEADY = false, bool DYN = false>
__global__ __launch_bounds__(DYN ? kSweepDynThreads : kSweepThreads) void k_tick(const TickSweep<TA> sw, const TickQr1<TA> q1, const TickQr2 q2,
                                                        const int G, const int sweep_first) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // Workgroups are dispatched in index order.  sweep_first: the persistent sweep
    // workgroups take their CUs at t = 0 and the short append stages fill what is left, instead of the sweep
    // tail starting only when the stages have drained.
    int bid = (int)blockIdx.x;
    if (sweep_first) bid = bid < sw.nblk ? bid + 2 * G : bid - sw.nblk;
    if constexpr (DYN) {
        if (bid < 2 * G && threadIdx.x >= kSweepThreads) return;
    }
    if (bid < G) {
        if (q2.active)
            qr2_body<8>(q2.Q, q2.ldq, q2.st, q2.avec, q2.r, q2.P1, q2.P1s, q2.G, q2.W1, q2.vvec, q2.P2, q2.P2s, q2.R, q2.z,
                     q2.sel, q2.kcap, q2.jpad, q2.force_reorth, q2.jh, q2.optimistic, bid, lds);
    } else if (bid < 2 * G) {
        if (q1.active)
            qr1_body<TA, 2>(q1.A, q1.ld, q1.M, q1.Q, q1.ldq, q1.st, q1.avec, q1.P1, q1.G, q1.kcap, q1.jpad, q1.mode, q1.pval,
                         q1.pidx, q1.nblk_sweep, q1.cands, q1.ncands, q1.which, q1.sel, q1.skipmask, q1.r, q1.P1s, q1.jh,
                         bid - G, lds);
    } else {
        if (sw.active) {
            if constexpr (DYN)
                sweep_body_dyn<TA, U, 32 / U>(sw.A, sw.ld, sw.Mv, sw.N, sw.r, sw.cvec, sw.pval, sw.pidx, sw.st, sw.eps, sw.check_eps,
                                              sw.skipmask, bid - 2 * G, sw.nblk, sw.KP, sw.claim, sw.claim_next, sw.npools, lds);
            else if constexpr (PH)
                sweep_body_ph<TA, U, 32 / U>(sw.A, sw.ld, sw.Mv, sw.N, sw.r, sw.cvec, sw.pval, sw.pidx, sw.st, sw.eps, sw.check_eps,
                                             sw.skipmask, bid - 2 * G, sw.nblk, sw.KP, sw.pcap, lds);
            else
                sweep_body_gen<TA, U, 32 / U>(sw.A, sw.ld, sw.Mv, sw.N, sw.r, sw.cvec, sw.pval, sw.pidx, sw.st, sw.eps, sw.check_eps,
                                              sw.skipmask, bid - 2 * G, sw.nblk, sw.KP, lds);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Final solve + SparseVector assembly (ONE workgroup): c = R^{-1} z by column-oriented back
// substitution, then (index, coefficient) pairs in ascending index order (rank sort), as
// ldiv!(AiQR, r) returns them (src/matchingpursuit.jl:175; sorted insert, src/util.jl:122).
__global__ __launch_bounds__(256) void k_finish(const double* __restrict__ R, const double* __restrict__ z,
                                                const int* __restrict__ sel, const DevState* st, int kcap,
                                                double* __restrict__ coef /*kcap scratch*/, int64_t* __restrict__ out_idx,
                                                double* __restrict__ out_val, int64_t* __restrict__ out_nnz,
                                                int64_t* __restrict__ out_order, int outcap, int* __restrict__ flag_out) {
    extern __shared__ __attribute__((aligned(16))) double y[];  // kcap + 2
    double& ci = y[kcap];
    const int tid = threadIdx.x, j = st->nsel;
    for (int t = tid; t < j; t += 256) y[t] = z[t];
    __syncthreads();
    // column-oriented back substitution; column i-1 of R is requested while column i is applied
    // (threads cover rows t = tid + 256 u, u < 4: kcap <= 1024)
    double rc[4], rn[4];
    double dcur = 1.0, dnext = 1.0;
    if (j > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u;
            rc[u] = (t < j - 1) ? R[(int64_t)(j - 1) * kcap + t] : 0.0;
        }
        dcur = R[(int64_t)(j - 1) * kcap + (j - 1)];
    }
    for (int i = j - 1; i >= 0; --i) {
        if (i > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = tid + 256 * u;
                rn[u] = (t < i - 1) ? R[(int64_t)(i - 1) * kcap + t] : 0.0;
            }
            dnext = R[(int64_t)(i - 1) * kcap + (i - 1)];
        }
        if (tid == 0) {
            ci = y[i] / dcur;
            y[i] = ci;
        }
        __syncthreads();
        const double c = ci;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u;
            if (t < i) y[t] = fma(-rc[u], c, y[t]);
        }
        for (int t = tid + 1024; t < i; t += 256) y[t] = fma(-R[(int64_t)i * kcap + t], c, y[t]);  // kcap > 1024
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) rc[u] = rn[u];
        dcur = dnext;
    }
    for (int t = tid; t < j; t += 256) coef[t] = y[t];
    for (int t = tid; t < outcap; t += 256) {
        out_idx[t] = -1;
        out_val[t] = 0.0;
        if (out_order) out_order[t] = (t < j) ? sel[t] : -1;
    }
    __syncthreads();
    for (int t = tid; t < j; t += 256) {
        const int me = sel[t];
        int rank = 0;
        for (int u = 0; u < j; ++u) rank += (sel[u] < me);
        out_idx[rank] = me;
        out_val[rank] = y[t];
    }
    if (tid == 0) {
        *out_nnz = j;
        if (flag_out) *flag_out = st->done | (st->uncertain ? STOP_UNCERTAIN : 0);
    }
}

// The same for kcap <= 64*NU as ONE wave with y in registers (lane l owns rows l + 64u): no barrier and
// no LDS in the chain -- the coefficient of column i is formed by its owner and broadcast with
// v_readlane -- and reciprocal diagonals formed up front, so the division leaves the chain too.
// ~0.05 us per column instead of ~0.5: this kernel closes every solve and every OMPR iteration.
template <int NU>
__global__ __launch_bounds__(64) void k_finish_w(const double* __restrict__ R, const double* __restrict__ z,
                                                 const int* __restrict__ sel, const DevState* st, int kcap,
                                                 double* __restrict__ coef, int64_t* __restrict__ out_idx,
                                                 double* __restrict__ out_val, int64_t* __restrict__ out_nnz,
                                                 int64_t* __restrict__ out_order, int outcap, int* __restrict__ flag_out) {
    extern __shared__ __attribute__((aligned(16))) int ssel[];  // kcap
    const int lane = threadIdx.x, j = st->nsel;
    double yr[NU], rdr[NU], rc[NU], rn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int t = lane + 64 * u;
        yr[u] = t < j ? z[t] : 0.0;
        rdr[u] = t < j ? 1.0 / R[(int64_t)t * kcap + t] : 0.0;
    }
    for (int t = lane; t < j; t += 64) ssel[t] = sel[t];
    auto fetch = [&](double* dst, int i) {  // column i above the diagonal: rows lane + 64u < i (zero elsewhere)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int t = lane + 64 * u;
            dst[u] = (i >= 0 && t < i) ? R[(int64_t)i * kcap + t] : 0.0;
        }
    };
    fetch(rc, j - 1);
    for (int i = j - 1; i >= 0; --i) {
        fetch(rn, i - 1);
        const int su = i >> 6, sl = i & 63;
        double mine = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (u == su) mine = yr[u] * rdr[u];
        const double c = readlane_f64(mine, sl);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            yr[u] = fma(-rc[u], c, yr[u]);
            if (u == su && lane == sl) yr[u] = c;
            rc[u] = rn[u];
        }
    }
    for (int t = lane; t < outcap; t += 64) {
        out_idx[t] = -1;
        out_val[t] = 0.0;
        if (out_order) out_order[t] = (t < j) ? sel[t] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int t = lane + 64 * u;
        if (t < j) {
            coef[t] = yr[u];
            const int me = ssel[t];
            int rank = 0;
            for (int q = 0; q < j; ++q) rank += (ssel[q] < me);
            out_idx[rank] = me;
            out_val[rank] = yr[u];
        }
    }
    if (lane == 0) {
        *out_nnz = j;
        if (flag_out) *flag_out = st->done | (st->uncertain ? STOP_UNCERTAIN : 0);
    }
}


// ---- the back substitution for LARGE supports, over several CUs.  One workgroup cannot stream R faster than one CU reads memory
// (24 GB/s from HBM, ~60 GB/s from L2: tools/probes/finish_probe.hip -- 4 MiB of R at 1024 columns took 240 us in k_finish_b,
// 160 us of them the rows-above updates), so the triangle is cut into SUPER-BLOCKS of 256 columns:
//   k_trsv_blk   one workgroup solves the 256 x 256 triangle of a super-block (k_finish_b's scheme on 128 KiB);
//   k_trsv_upd   y[rows above] -= R[rows above, super-block] * x[super-block], 64 rows per workgroup;
//   k_trsv_emit  the sorted emission, 256 entries per workgroup.
// 2 ceil(j / 256) launches; y lives in `coef`.
constexpr int kTrsvBlk = 256;
__global__ __launch_bounds__(256) void k_trsv_blk(const double* __restrict__ R, const double* __restrict__ z, const DevState* st,
                                                  int kcap, double* __restrict__ y, int off, int init_from_z) {
    __shared__ double yl[kTrsvBlk];
    __shared__ double cb[64];
    const int tid = threadIdx.x, lane = tid & 63, j = st->nsel;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (init_from_z)
        for (int t = tid; t < j; t += 256)
            if (t < off || t >= off + kTrsvBlk) y[t] = z[t];  // (the super-block's own entries are written below)
    if (off >= j) return;
    const int len = min(kTrsvBlk, j - off);
    const int nb = (len + 63) / 64;
    // Wave b owns diagonal block b: it requests the block's rows NOW (row `lane` is D[lane][i] = R[g0 + lane, g0 + i], i > lane,
    // zero elsewhere) and runs the block's chain when its turn comes -- the four round trips through memory happen together, ahead
    // of the chain, instead of one at the head of every block.
    double drow[64];
    double rd = 0.0;
    {
        const int i0 = wave * 64, w = min(64, len - i0), g0 = off + i0;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            drow[i] = (wave < nb && i < w && i > lane && lane < w) ? R[(int64_t)(g0 + i) * kcap + g0 + lane] : 0.0;
        rd = (wave < nb && lane < w) ? 1.0 / R[(int64_t)(g0 + lane) * kcap + g0 + lane] : 0.0;
    }
    if (tid < len) yl[tid] = init_from_z ? z[off + tid] : y[off + tid];
    __syncthreads();
    for (int b = nb - 1; b >= 0; --b) {
        const int i0 = b * 64, w = min(64, len - i0), g0 = off + i0;  // block columns g0 .. g0 + w - 1
        if (wave == b) {
            double yv = lane < w ? yl[i0 + lane] : 0.0;
#pragma unroll
            for (int i = 63; i >= 0; --i) {
                const double c = readlane_f64(yv * rd, i);  // lanes >= w carry zeros
                yv = lane == i ? c : fma(-drow[i], c, yv);
            }
            if (lane < w) {
                yl[i0 + lane] = yv;
                cb[lane] = yv;
            } else {
                cb[lane] = 0.0;
            }
        }
        __syncthreads();
        if (tid < i0) {  // the rows of the super-block above this block (at most 192): all 64 loads of a row in flight at once
            const double* col = R + (int64_t)g0 * kcap + off + tid;
            double rv[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) rv[i] = i < w ? col[(int64_t)i * kcap] : 0.0;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int i = 0; i < 64; i += 4) {
                a0 = fma(rv[i], cb[i], a0);
                a1 = fma(rv[i + 1], cb[i + 1], a1);
                a2 = fma(rv[i + 2], cb[i + 2], a2);
                a3 = fma(rv[i + 3], cb[i + 3], a3);
            }
            yl[tid] -= (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
    if (tid < len) y[off + tid] = yl[tid];
}
// rows [64 blockIdx.x, +64) of y -= R[rows, off .. off + len) * y[off .. off + len): thread (row, column quarter), the four
// quarters meet in LDS in a fixed order
__global__ __launch_bounds__(256) void k_trsv_upd(const double* __restrict__ R, const DevState* st, int kcap, double* __restrict__ y,
                                                  int off) {
    __shared__ double xs[kTrsvBlk];
    __shared__ double part[256];
    const int tid = threadIdx.x, j = st->nsel;
    if (off >= j) return;
    const int len = min(kTrsvBlk, j - off);
    xs[tid] = tid < len ? y[off + tid] : 0.0;
    __syncthreads();
    const int row = blockIdx.x * 64 + (tid & 63), cq = tid >> 6;
    double a0 = 0.0, a1 = 0.0;
    if (row < off) {
        const double* col = R + (int64_t)(off + cq * 64) * kcap + row;
#pragma unroll 16
        for (int i = 0; i < 64; i += 2) {
            const double r0 = cq * 64 + i < len ? col[(int64_t)i * kcap] : 0.0;
            const double r1 = cq * 64 + i + 1 < len ? col[(int64_t)(i + 1) * kcap] : 0.0;
            a0 = fma(r0, xs[cq * 64 + i], a0);
            a1 = fma(r1, xs[cq * 64 + i + 1], a1);
        }
    }
    part[tid] = a0 + a1;
    __syncthreads();
    if (tid < 64 && row < off) y[row] -= (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
}
// (index, coefficient) pairs in ascending index order, 256 per workgroup (rank sort against the whole support in LDS)
__global__ __launch_bounds__(256) void k_trsv_emit(const double* __restrict__ y, const int* __restrict__ sel, const DevState* st,
                                                   int64_t* __restrict__ out_idx, double* __restrict__ out_val,
                                                   int64_t* __restrict__ out_nnz, int64_t* __restrict__ out_order, int outcap,
                                                   int* __restrict__ flag_out) {
    extern __shared__ __attribute__((aligned(16))) int ssel[];  // j rounded up to 4
    const int tid = threadIdx.x, j = st->nsel;
    for (int t = tid; t < ((j + 3) & ~3); t += 256) ssel[t] = t < j ? sel[t] : 0x7fffffff;
    __syncthreads();
    const int t = blockIdx.x * 256 + tid;
    if (t < outcap && t >= j) {
        out_idx[t] = -1;
        out_val[t] = 0.0;
    }
    if (t < outcap && out_order) out_order[t] = (t < j) ? ssel[t] : -1;
    if (t < j) {
        const int me = ssel[t];
        const int4* s4 = reinterpret_cast<const int4*>(ssel);
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (int u = 0; u < (j + 3) >> 2; ++u) {
            const int4 v = s4[u];
            r0 += (int)(v.x < me);
            r1 += (int)(v.y < me);
            r2 += (int)(v.z < me);
            r3 += (int)(v.w < me);
        }
        const int rank = r0 + r1 + r2 + r3;
        out_idx[rank] = me;
        out_val[rank] = y[t];
    }
    if (t == 0) {
        *out_nnz = j;
        if (flag_out) *flag_out = st->done | (st->uncertain ? STOP_UNCERTAIN : 0);
    }
}

// ---------------------------------------------------------------------------------------------
// b (any float type, host-staged or a column of a device matrix) -> Float64 b and r; state reset
template <typename TB>
__global__ __launch_bounds__(256) void k_init(const TB* __restrict__ src, int M, int Mpad, double* __restrict__ b,
                                              double* __restrict__ r, DevState* st) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Mpad) {
        const double v = (i < M) ? (double)src[i] : 0.0;
        b[i] = v;
        r[i] = v;
    }
    if (i == 0) {
        st->nsel = 0;
        st->j = 0;
        st->cand = -1;
        st->go = 0;
        st->done = 0;
        st->steps = 0;
        st->go2 = 0;
        st->pcount = 0;
        st->rnorm2 = 0.0;
        st->cval = 0.0;
        st->uncertain = 0;
    }
}

// residual!(r, A, x, b): r = b - A[:, idx] * val  (src/matchingpursuit.jl:158-161); thread <-> row
template <typename TA>
__global__ __launch_bounds__(256) void k_residual(const TA* __restrict__ A, int64_t ld, int M,
                                                  const int* __restrict__ idx, const double* __restrict__ val,
                                                  const int* __restrict__ nnzp, int nnz_arg,
                                                  const double* __restrict__ b, double* __restrict__ r) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const int nnz = nnzp ? *nnzp : nnz_arg;
    double acc = b[row];
    for (int t = 0; t < nnz; ++t) acc = fma(-(double)A[(int64_t)idx[t] * ld + row], val[t], acc);
    r[row] = acc;
}

// MP step: x[i] += <a_i, r>;  r -= <a_i, r> a_i   (src/matchingpursuit.jl:27-29, residual kept
// incrementally: unit-norm columns are NOT assumed, the update is exact for any column norm)
template <typename TA>
__global__ __launch_bounds__(256) void k_mp_update(const TA* __restrict__ A, int64_t ld, int M, double* __restrict__ r,
                                                   DevState* st, int* __restrict__ sel, double* __restrict__ z) {
    if (!st->go) return;
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int cand = st->cand;
    const double c = st->cval;
    if (row < M) r[row] = fma(-(double)A[(int64_t)cand * ld + row], c, r[row]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int j = st->j;
        sel[j] = cand;
        z[j] = c;
        st->nsel = j + 1;
        st->steps += 1;
    }
}

// out[t] = c[idx[t]] (OMPR reads the correlations of its support, src/twostage.jl:165)
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ c, const int* __restrict__ idx, int n,
                                                double* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = c[idx[t]];
}

// ||r||_2^2 by one workgroup (step-level API / SP loop control)
__global__ __launch_bounds__(256) void k_norm2(const double* __restrict__ r, int M, double* __restrict__ out) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) acc = fma(r[i], r[i], acc);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s[0];
}

}  // namespace csmp
