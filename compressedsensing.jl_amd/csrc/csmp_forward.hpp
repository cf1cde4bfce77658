// csmp_forward.hpp -- forward regression / orthogonal least squares (fr = ols = oomp = ormp) on gfx950.
//
// Reference primitives replaced (paths relative to the reference repository):
//   k_fr_sweep   forward_δ!(P, x):   mul!(δ², A', r); ols_rescaling!; δ² = δ²^2 / rescaling;
//                δ²[x.nzind] = 0                                   src/forward.jl:75-82
//                ols_rescaling!:  mul!(QA, Q', A); rescaling_j = |a_j|^2 - |(Q'A)[1:nnz, j]|^2
//                                                                  src/forward.jl:99-114
//                + norm(residual!) > max_ε                         src/forward.jl:59-61
//   (k_qr1 mode 3) findmax(δ²), min_δ^2 < max_δ² guard, addindex!   src/forward.jl:63-68
//
// The reference forms Q'A -- an (M x M) x (M x N) product, 2.2e12 flop at 4096 x 65536 -- at every
// step.  |Q_S' a_j|^2 = sum_i (q_i' a_j)^2 runs over the columns of ANY orthonormal basis of
// span(A_S), and this library's Q only ever grows by one column per step (csmp_kernels.hpp, k_qr2),
// so the same sum is evaluated one term per step: the sweep that streams the dictionary for
// c = A'r also forms g = A'q_new in the same pass and downdates rho2_j -= g_j^2.  A forward-regression
// step therefore costs exactly what an OMP step costs in HBM traffic: M*N*sizeof(T) bytes.
//
// Numerics as everywhere else: Float64 products of the exactly promoted dictionary, fixed summation
// order (no atomics), so equal columns score bit-identically and the first maximum is well defined.
#pragma once
#include "csmp_kernels.hpp"

namespace csmp {

// One column per wave at a time, software-pipelined across columns like sweep_body_pf.
//   U      16-byte row chunks per load block (U KiB in flight per wave, U..2U while a block is reduced)
//   FULL   Mv is a multiple of U*64*VEC rows; otherwise the tail chunks are predicated
//   NQ     directions whose squared projections correct rho2 in this pass:
//          -1  first step of a solve: no factorisation yet; the second accumulator forms |a_j|^2
//              (sum!(abs2, rescaling', A), src/forward.jl:108) and initialises rho2
//           0  rho2 is current: scores only
//           1  rho2_j += s1 * <a_j, q1>^2          (s1 = -1: q1 is the Q column appended last step)
//           2  ... + s2 * <a_j, q2>^2 as well      (stepwise regression with replacement: the column
//              appended by the forward step, s = -1, and the direction q_drop that the backward step
//              rotated out, s = +1; csmp_downdate.hpp)
//   q1 == nullptr with NQ >= 1: q1 = Q[:, nsel-1], looked up on the device
//   unmark (may be null): atom that LEFT the support; its rho2 is re-seeded with <a, q_last>^2, q_last being
//          the last direction of the pass (q_drop): the atom's squared distance from the new span.  The atom sel[nsel-1] is marked rho2 = +Inf (score 0 for good).
//   update_only: rho2 maintenance pass, no residual test and no scores
// dynamic LDS: r image | NQ direction images (each nblocks*U*64*VEC doubles) | 32 doubles of scratch
template <typename TA, int U, bool FULL, int NQ>
__device__ __forceinline__ void fr_sweep_body(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    const double* __restrict__ Q, int64_t ldq, const double* __restrict__ q1in, double s1,
    const double* __restrict__ q2in, double s2, const int* __restrict__ unmark, int update_only,
    double* __restrict__ rho2, double* __restrict__ dvec, double* __restrict__ pval, int* __restrict__ pidx,
    const int* __restrict__ sel, DevState* st, double max_eps, int skipmask, const int bid, const int nblk, double* lds) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    constexpr int NW = kSweepThreads / kWave;
    if (st->done & skipmask) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int nblocks = (nchunk + U - 1) / U;
    const int Mlds = nblocks * U * ROWS;
    constexpr bool FIRST = NQ < 0;
    constexpr int NIM = NQ > 0 ? NQ : 0;
    double* qim = lds + Mlds;
    double* qim2 = qim + (NIM > 1 ? Mlds : 0);
    double* red = lds + (size_t)(1 + NIM) * Mlds;
    double* redv = red + 8;
    int* redi = reinterpret_cast<int*>(redv + 4 * NW);

    const int nsel = st->nsel;
    const int lastsel = (!FIRST && nsel > 0) ? sel[nsel - 1] : -1;
    const int unsel = (NQ >= 1 && unmark) ? *unmark : -1;
    const double* q1 = nullptr;
    if constexpr (NQ >= 1) q1 = q1in ? q1in : (nsel > 0 ? Q + (int64_t)(nsel - 1) * ldq : nullptr);
    const int64_t stride = (int64_t)nblk * NW;
    auto load_block = [&](VT* dst, int64_t c, int blk) {
        const VT* p = reinterpret_cast<const VT*>(A + c * ld) + lane + (int64_t)blk * U * kWave;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (FULL) {
                dst[u] = __builtin_nontemporal_load(p + u * kWave);
            } else {
                dst[u] = (VT)0;
                if ((blk * U + u) * ROWS + lane * VEC < Mv) dst[u] = __builtin_nontemporal_load(p + u * kWave);
            }
        }
    };
    int64_t col = (int64_t)bid * NW + wave;
    if (col >= N) col = -1;
    VT cur[U], nxt[U];
    if (col >= 0) load_block(cur, col, 0);  // (ahead of the images: the first trip to HBM and the images' trips to the L2 overlap)
    // the images, eight rows per thread and image at a time (a rolled loop waits for every load in turn); ||r||^2 in the same
    // per-thread order as before (rows tid, tid + 256, ...)
    double n2 = 0.0;
    {
        constexpr int RP = 8;
        for (int m0 = tid; m0 < Mlds; m0 += RP * kSweepThreads) {
            double rv[RP], qv[NQ >= 1 ? RP : 1], qv2[NQ == 2 ? RP : 1];
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int m = m0 + q * kSweepThreads;
                const int mc = m < Mv ? m : Mv - 1;  // (no load under a branch)
                rv[q] = r[mc];
                if constexpr (NQ >= 1) qv[q] = q1 ? q1[mc] : 0.0;
                if constexpr (NQ == 2) qv2[q] = q2in ? q2in[mc] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int m = m0 + q * kSweepThreads;
                if (m < Mlds) {
                    const bool in = m < Mv;
                    const double v = in ? rv[q] : 0.0;
                    lds[r_slot<VEC>(m)] = v;
                    n2 = fma(v, v, n2);
                    if constexpr (NQ >= 1) qim[r_slot<VEC>(m)] = in ? qv[q] : 0.0;
                    if constexpr (NQ == 2) qim2[r_slot<VEC>(m)] = in ? qv2[q] : 0.0;
                }
            }
        }
    }
    n2 = block_sum256(n2, red);
    if (bid == 0 && tid == 0) st->rnorm2 = n2;
    if (!update_only && !(sqrt(n2) > max_eps)) {  // normr > max_ε || return false   (src/forward.jl:60-61)
        if (bid == 0 && tid == 0) st->done |= STOP_EPS;
        return;
    }
    const f64x2* rs = reinterpret_cast<const f64x2*>(lds);
    const f64x2* qs = reinterpret_cast<const f64x2*>(qim);
    const f64x2* qs2 = reinterpret_cast<const f64x2*>(qim2);
    double bestv = -1.0;  // a NaN score never wins a comparison
    int besti = 0x7fffffff;
    // rho2 / d2 are STAGED like the product sweep's c: lane s keeps the wave's s-th finished column, 64 columns to a store
    double srho = 0.0, sd2 = 0.0;
    int scol = -1, cslot = 0;
    auto flush = [&]() {
        if (scol >= 0) {
            if (NQ != 0 || scol == lastsel) rho2[scol] = srho;
            if (!update_only) dvec[scol] = sd2;
        }
        scol = -1;
        cslot = 0;
    };
    while (col >= 0) {
        double rho_old = 0.0;
        if constexpr (!FIRST) rho_old = rho2[col];  // requested before the column's loads are consumed
        double acc = 0.0, acg = 0.0, acg2 = 0.0;
        for (int blk = 0; blk < nblocks; ++blk) {
            const bool last = blk + 1 == nblocks;
            const int64_t ncol = last ? col + stride : col;
            if (!last || ncol < N) load_block(nxt, ncol, last ? 0 : blk + 1);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = blk * U + u;
                if constexpr (VEC == 4) {
                    const f64x2 r01 = rs[(t * 2 + 0) * kWave + lane];
                    const f64x2 r23 = rs[(t * 2 + 1) * kWave + lane];
                    const double a0 = (double)cur[u].x, a1 = (double)cur[u].y, a2 = (double)cur[u].z, a3 = (double)cur[u].w;
                    acc = fma(a0, r01.x, acc);
                    acc = fma(a1, r01.y, acc);
                    acc = fma(a2, r23.x, acc);
                    acc = fma(a3, r23.y, acc);
                    if constexpr (FIRST) {
                        acg = fma(a0, a0, acg);
                        acg = fma(a1, a1, acg);
                        acg = fma(a2, a2, acg);
                        acg = fma(a3, a3, acg);
                    }
                    if constexpr (NQ >= 1) {
                        const f64x2 q01 = qs[(t * 2 + 0) * kWave + lane];
                        const f64x2 q23 = qs[(t * 2 + 1) * kWave + lane];
                        acg = fma(a0, q01.x, acg);
                        acg = fma(a1, q01.y, acg);
                        acg = fma(a2, q23.x, acg);
                        acg = fma(a3, q23.y, acg);
                    }
                    if constexpr (NQ == 2) {
                        const f64x2 q01 = qs2[(t * 2 + 0) * kWave + lane];
                        const f64x2 q23 = qs2[(t * 2 + 1) * kWave + lane];
                        acg2 = fma(a0, q01.x, acg2);
                        acg2 = fma(a1, q01.y, acg2);
                        acg2 = fma(a2, q23.x, acg2);
                        acg2 = fma(a3, q23.y, acg2);
                    }
                } else {
                    const f64x2 r01 = rs[t * kWave + lane];
                    const double a0 = (double)cur[u].x, a1 = (double)cur[u].y;
                    acc = fma(a0, r01.x, acc);
                    acc = fma(a1, r01.y, acc);
                    if constexpr (FIRST) {
                        acg = fma(a0, a0, acg);
                        acg = fma(a1, a1, acg);
                    }
                    if constexpr (NQ >= 1) {
                        const f64x2 q01 = qs[t * kWave + lane];
                        acg = fma(a0, q01.x, acg);
                        acg = fma(a1, q01.y, acg);
                    }
                    if constexpr (NQ == 2) {
                        const f64x2 q01 = qs2[t * kWave + lane];
                        acg2 = fma(a0, q01.x, acg2);
                        acg2 = fma(a1, q01.y, acg2);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
        acc = wave_xsum(acc);  // (the butterfly without LDS trips: the same pairs in the same order as the shuffle form)
        if constexpr (NQ != 0) acg = wave_xsum(acg);
        if constexpr (NQ == 2) acg2 = wave_xsum(acg2);
        // rescaling_j (src/forward.jl:108-113), one more row of Q'A per step; atoms of the support get
        // +Inf once, which makes their score c^2 / Inf = 0 for good (δ²[x.nzind] = 0, :80)
        double rho = rho_old;
        if constexpr (FIRST) rho = acg;
        if constexpr (NQ >= 1) rho = fma(s1 * acg, acg, rho);
        if constexpr (NQ == 2) rho = fma(s2 * acg2, acg2, rho);
        if ((int)col == lastsel) rho = __builtin_inf();
        if constexpr (NQ >= 1)
            if ((int)col == unsel) rho = NQ == 2 ? acg2 * acg2 : acg * acg;
        const double d2 = acc * acc / rho;
        if (lane == cslot) {
            srho = rho;
            sd2 = d2;
            scol = (int)col;
        }
        if (++cslot == kWave) flush();
        if (d2 > bestv) {
            bestv = d2;
            besti = (int)col;
        }
        col += stride;
        if (col >= N) col = -1;
    }
    flush();
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

inline size_t fr_sweep_lds_bytes(int Mv, int vec, int U, int nq) {
    const int rows = kWave * vec;
    const int nchunk = (Mv + rows - 1) / rows;
    const int nblocks = (nchunk + U - 1) / U;
    const int images = nq == 4 ? 4 : 1 + (nq > 0 ? nq : 0);
    return ((size_t)images * nblocks * U * rows + 8 + 16 + 8) * sizeof(double);
}

template <typename TA, int U, bool FULL, int NQ>
__global__ __launch_bounds__(kSweepThreads) void k_fr_sweep(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    const double* __restrict__ Q, int64_t ldq, const double* __restrict__ q1in, double s1,
    const double* __restrict__ q2in, double s2, const int* __restrict__ unmark, int update_only,
    double* __restrict__ rho2, double* __restrict__ dvec, double* __restrict__ pval, int* __restrict__ pidx,
    const int* __restrict__ sel, DevState* st, double max_eps, int skipmask) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    fr_sweep_body<TA, U, FULL, NQ>(A, ld, Mv, N, r, Q, ldq, q1in, s1, q2in, s2, unmark, update_only, rho2, dvec, pval, pidx, sel, st,
                                   max_eps, skipmask, (int)blockIdx.x, (int)gridDim.x, lds);
}

// Tick kernel of the pipelined forward-regression batch (the k_tick of csmp_kernels.hpp with the OLS sweep):
// workgroups [0, G) run the k_qr2 stage of signal X, [G, 2G) the k_qr1 stage (mode 3) of signal Y,
// [2G, 2G + nblk) the forward-regression sweep of signal Z -- NQ = -1 on a signal's first step, 1 after.
template <typename TA>
struct TickFr {
    const TA* A; int64_t ld; int Mv; int64_t N;
    const double* r; const double* Q; int64_t ldq;
    double* rho2; double* dvec; double* pval; int* pidx; const int* sel; DevState* st;
    double max_eps; int skipmask; int nblk; int active;
};
template <typename TA, int U, int NQ>
__global__ __launch_bounds__(kSweepThreads) void k_tick_fr(const TickFr<TA> sw, const TickQr1<TA> q1, const TickQr2 q2,
                                                           const int G, const double min_d2) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int bid = (int)blockIdx.x;
    if (bid < G) {
        if (q2.active)
            qr2_body<8>(q2.Q, q2.ldq, q2.st, q2.avec, q2.r, q2.P1, q2.P1s, q2.G, q2.W1, q2.vvec, q2.P2, q2.P2s, q2.R, q2.z,
                        q2.sel, q2.kcap, q2.jpad, q2.force_reorth, q2.jh, q2.optimistic, bid, lds);
    } else if (bid < 2 * G) {
        if (q1.active)
            qr1_body<TA, 2>(q1.A, q1.ld, q1.M, q1.Q, q1.ldq, q1.st, q1.avec, q1.P1, q1.G, q1.kcap, q1.jpad, q1.mode, q1.pval,
                            q1.pidx, q1.nblk_sweep, q1.cands, q1.ncands, q1.which, q1.sel, q1.skipmask, q1.r, q1.P1s, q1.jh,
                            bid - G, lds, min_d2);
    } else {
        if (sw.active)
            fr_sweep_body<TA, U, true, NQ>(sw.A, sw.ld, sw.Mv, sw.N, sw.r, sw.Q, sw.ldq, nullptr, -1.0, nullptr, 1.0, nullptr, 0,
                                           sw.rho2, sw.dvec, sw.pval, sw.pidx, sw.sel, sw.st, sw.max_eps, sw.skipmask,
                                           bid - 2 * G, sw.nblk, lds);
    }
}


// Bulk form of the same maintenance on the Float64 matrix cores: rho2_j -= sum_{d < nd} <a_j, q_{d0+d}>^2 for up to
// 128 directions in ONE pass over the dictionary (k_fr_update4 needs nd/4 passes).  G = Q'A is a genuine GEMM here
// (directions x atoms, K = rows); it is never written: wave w of a workgroup owns 32 atoms and keeps one 16 x 16
// accumulator tile per 16 atoms x 16 directions (2 x 8 tiles = 128 VGPRs).  Operands are loaded straight from global
// memory with the K index permuted so that every lane walks CONTIGUOUS rows: within a 64-row block lane l holds
// rows 16 (l >> 4) .. + 16 of atom / direction l & 15 (64 B of an f32 column, 128 B of a Q column), and K-step kk
// multiplies row 16 (l >> 4) + kk of every quarter.  Q (nd x M doubles) is re-read by every atom tile from L2.
template <typename TA>
__global__ __launch_bounds__(256) void k_fr_rebuild(const TA* __restrict__ A, int64_t ld, int M, int64_t N,
                                                    const double* __restrict__ Q, int64_t ldq, int d0, int nd,
                                                    double* __restrict__ rho2) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int NA = 2;    // 16-atom tiles per wave: every Q fragment feeds NA MFMAs (Q is re-read from L2 by every wave)
    constexpr int MAXT = 8;  // 16-direction tiles per pass: NA * MAXT accumulator tiles = 128 VGPRs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t a0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * NA);
    if (a0 >= N) return;
    const TA* acol[NA];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        const int64_t atom = a0 + h * 16 + fr < N ? a0 + h * 16 + fr : N - 1;
        acol[h] = A + atom * ld + fq * 16;
    }
    const int nt = (nd + 15) / 16;
    d4 acc[NA][MAXT];
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[h][t] = d4{0.0, 0.0, 0.0, 0.0};
    const int nrb = (M + 63) / 64;
    for (int rb = 0; rb < nrb; ++rb) {
        double bv[NA][16];
        const int r0 = rb * 64 + fq * 16;
#pragma unroll
        for (int h = 0; h < NA; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[h][e] = (r0 + e < M) ? (double)acol[h][(int64_t)rb * 64 + e] : 0.0;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < nt) {  // uniform
                const int dir = d0 + t * 16 + fr;
                const double* qc = Q + (int64_t)(dir < d0 + nd ? dir : d0) * ldq + r0;
                double qv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) qv[e] = (dir < d0 + nd) ? qc[e] : 0.0;  // (rows beyond M are zero in Q)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                    for (int h = 0; h < NA; ++h) acc[h][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qv[kk], bv[h][kk], acc[h][t], 0, 0, 0);
            }
        }
    }
    // C/D layout: col = lane & 15 (atom), row = (lane >> 4) + 4 reg (direction within the tile)
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        double ssum = 0.0;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ssum = fma(acc[h][t][reg], acc[h][t][reg], ssum);
        ssum += shx(ssum, 16);
        ssum += shx(ssum, 32);
        if (fq == 0 && a0 + h * 16 + fr < N) rho2[a0 + h * 16 + fr] -= ssum;
    }
}

// The same product with the directions staged through the LDS: the 128 x 64 block of Q a row block needs is fetched ONCE per
// workgroup (coalesced 16-byte loads into registers while the matrix cores work on the block before, then written to the other
// LDS buffer), instead of once per wave from L2 behind a uniform branch per direction tile -- which is what k_fr_rebuild waits
// for: 16 loads, their latency, 32 MFMAs, and again.  Directions are LDS rows of 64 doubles + 16 bytes: the 16 lanes of a
// quarter read 16 different rows at a 528-byte stride, i.e. all 64 banks once per ds_read_b128.  Every steady row block is loaded
// unguarded (all its rows exist); the last one clamps its row indices and masks what lies beyond M.
constexpr int kRbDirs = 128, kRbRows = 64, kRbStride = kRbRows + 2;
__host__ __device__ constexpr size_t fr_rebuild_lds_bytes() { return (size_t)2 * kRbDirs * kRbStride * sizeof(double); }

template <typename TA, bool VEC>
__global__ __launch_bounds__(256) void k_fr_rebuild_lds(const TA* __restrict__ A, int64_t ld, int M, int64_t N,
                                                        const double* __restrict__ Q, int64_t ldq, int d0, int nd,
                                                        double* __restrict__ rho2) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef TA ta4 __attribute__((ext_vector_type(16 / sizeof(TA))));
    constexpr int NA = 2, NT = kRbDirs / 16, PERV = 16 / (int)sizeof(TA), NV = 16 / PERV;
    extern __shared__ double qlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t a0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * NA);
    const TA* acol[NA];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        const int64_t atom = a0 + h * 16 + fr < N ? a0 + h * 16 + fr : N - 1;
        acol[h] = A + atom * ld + fq * 16;
    }
    // loader: thread -> row pair tid % 32 of direction tid / 32 + 8 j (32 threads cover the 512 contiguous bytes of a direction)
    const int lrp = tid & 31, ldir = tid >> 5;
    const double* qsrc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int dir = ldir + 8 * j;
        qsrc[j] = Q + (int64_t)(d0 + (dir < nd ? dir : 0)) * ldq + lrp * 2;
    }
    d2 stage[16];
    TA raw[NA][16];
    d4 acc[NA][NT];
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[h][t] = d4{0.0, 0.0, 0.0, 0.0};
    const int nrb = (M + kRbRows - 1) / kRbRows;

    auto fetch_q = [&](int rb) {
#pragma unroll
        for (int j = 0; j < 16; ++j) stage[j] = *reinterpret_cast<const d2*>(qsrc[j] + (int64_t)rb * kRbRows);
    };
    auto store_q = [&](int buf) {
        double* dst = qlds + (size_t)buf * kRbDirs * kRbStride + lrp * 2;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int dir = ldir + 8 * j;
            *reinterpret_cast<d2*>(dst + dir * kRbStride) = dir < nd ? stage[j] : d2{0.0, 0.0};
        }
    };
    auto fetch_a_full = [&](int rb) {
#pragma unroll
        for (int h = 0; h < NA; ++h) {
            if (VEC) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const ta4 x = __builtin_nontemporal_load(reinterpret_cast<const ta4*>(acol[h] + (int64_t)rb * kRbRows) + v);
#pragma unroll
                    for (int c = 0; c < PERV; ++c) raw[h][v * PERV + c] = x[c];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) raw[h][e] = acol[h][(int64_t)rb * kRbRows + e];
            }
        }
    };
    auto fetch_a_last = [&](int rb) {  // row indices clamped into the column; to_operand masks them
        const int r0 = rb * kRbRows + fq * 16;
#pragma unroll
        for (int h = 0; h < NA; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + e < M ? r0 + e : M - 1;
                raw[h][e] = acol[h][row - fq * 16];
            }
    };
    auto compute = [&](int buf, const double (&bv)[NA][16]) {
        const double* src = qlds + (size_t)buf * kRbDirs * kRbStride + fq * 16;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double qv[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const d2 x = *reinterpret_cast<const d2*>(src + (t * 16 + fr) * kRbStride + 2 * e);
                qv[2 * e] = x[0];
                qv[2 * e + 1] = x[1];
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int h = 0; h < NA; ++h) acc[h][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qv[kk], bv[h][kk], acc[h][t], 0, 0, 0);
        }
    };

    fetch_q(0);
    if (nrb > 1) fetch_a_full(0); else fetch_a_last(0);
    store_q(0);
    __syncthreads();
    int buf = 0;
    for (int rb = 0; rb + 1 < nrb; ++rb) {
        double bv[NA][16];
#pragma unroll
        for (int h = 0; h < NA; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[h][e] = (double)raw[h][e];
        fetch_q(rb + 1);
        if (rb + 2 < nrb) fetch_a_full(rb + 1); else fetch_a_last(rb + 1);
        compute(buf, bv);
        store_q(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    {
        double bv[NA][16];
        const int r0 = (nrb - 1) * kRbRows + fq * 16;
#pragma unroll
        for (int h = 0; h < NA; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[h][e] = r0 + e < M ? (double)raw[h][e] : 0.0;
        compute(buf, bv);
    }
    if (a0 >= N) return;
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        double ssum = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ssum = fma(acc[h][t][reg], acc[h][t][reg], ssum);
        ssum += shx(ssum, 16);
        ssum += shx(ssum, 32);
        if (fq == 0 && a0 + h * 16 + fr < N) rho2[a0 + h * 16 + fr] -= ssum;
    }
}

// ---- dictionaries whose columns do not fit the LDS images of k_fr_sweep (8 (1 + NQ) M bytes: M beyond ~10 000 with one direction,
// ~6 800 with two): the same pass as separate launches -- c = A'r and g = A'q by the product sweep (k_sweep_gen: any M, the
// residual staged in phases), |a_j|^2 by k_fr_colnorm2 on a solve's first step, then k_fr_combine does what the fused kernel does
// per column: the rho2 correction, the score, the marks, the arg-max partials.  Two or three passes over the dictionary instead
// of one: the price of a shape the reference serves (src/forward.jl:99-114 forms Q'A whatever size(A) is) and the fused kernel cannot.
template <typename TA>
__global__ __launch_bounds__(256) void k_fr_colnorm2(const TA* __restrict__ A, int64_t ld, int M, int64_t N, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    const TA* a = A + col * ld;
    double acc = 0.0;
    for (int m = lane; m < M; m += 64) {
        const double v = (double)a[m];
        acc = fma(v, v, acc);
    }
    acc = wave_xsum(acc);
    if (lane == 0) out[col] = acc;
}
// the Q column appended last -> a vector of its own (a sweep takes it as its "residual"); zeros while the support is empty
__global__ __launch_bounds__(256) void k_fr_lastq(const double* __restrict__ Q, int64_t ldq, const DevState* st, int Mpad, int M,
                                                  double* __restrict__ out) {
    const int nsel = st->nsel;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < Mpad; m += gridDim.x * 256)
        out[m] = (nsel > 0 && m < M) ? Q[(int64_t)(nsel - 1) * ldq + m] : 0.0;
}
// per column what fr_sweep_body does after its reduction (same NQ meanings); c, g1, g2: the sweeps' outputs (g1 = |a_j|^2 for NQ = -1)
template <int NQ>
__global__ __launch_bounds__(256) void k_fr_combine(int64_t N, const double* __restrict__ c, const double* __restrict__ g1, double s1,
                                                    const double* __restrict__ g2, double s2, const int* __restrict__ unmark, int update_only,
                                                    double* __restrict__ rho2, double* __restrict__ dvec, double* __restrict__ pval,
                                                    int* __restrict__ pidx, const int* __restrict__ sel, DevState* st, double max_eps, int skipmask) {
    __shared__ double sv[256];
    __shared__ int si[256];
    if (st->done & skipmask) return;
    constexpr bool FIRST = NQ < 0;
    const int tid = threadIdx.x;
    if (!update_only && !(sqrt(st->rnorm2) > max_eps)) {  // normr > max_eps || return false   (src/forward.jl:60-61)
        __syncthreads();  // (every thread has read the flag word above before one of them changes it)
        if (blockIdx.x == 0 && tid == 0) st->done |= STOP_EPS;
        return;
    }
    const int nsel = st->nsel;
    const int lastsel = (!FIRST && nsel > 0) ? sel[nsel - 1] : -1;
    const int unsel = (NQ >= 1 && unmark) ? *unmark : -1;
    double bestv = -1.0;
    int besti = 0x7fffffff;
    for (int64_t j = (int64_t)blockIdx.x * 256 + tid; j < N; j += (int64_t)gridDim.x * 256) {
        double rho = FIRST ? g1[j] : rho2[j];
        if constexpr (NQ >= 1) rho = fma(s1 * g1[j], g1[j], rho);
        if constexpr (NQ == 2) rho = fma(s2 * g2[j], g2[j], rho);
        if ((int)j == lastsel) rho = __builtin_inf();
        if constexpr (NQ >= 1)
            if ((int)j == unsel) rho = NQ == 2 ? g2[j] * g2[j] : g1[j] * g1[j];
        const double d2 = c[j] * c[j] / rho;
        if constexpr (NQ != 0) rho2[j] = rho;
        else if ((int)j == lastsel) rho2[j] = rho;
        if (!update_only) dvec[j] = d2;
        if (d2 > bestv) {  // (ascending j per thread: '>' keeps the first maximum)
            bestv = d2;
            besti = (int)j;
        }
    }
    block_argmax(bestv, besti, sv, si);
    if (tid == 0) {
        pval[blockIdx.x] = bestv;
        pidx[blockIdx.x] = besti;
    }
}

// rho2 = +Inf for every atom of the support (bulk form of the per-step marking in k_fr_sweep)
__global__ __launch_bounds__(256) void k_mark_inf(double* __restrict__ rho2, const int* __restrict__ sel, const DevState* st) {
    for (int t = threadIdx.x; t < st->nsel; t += 256) rho2[sel[t]] = __builtin_inf();
}

}  // namespace csmp
