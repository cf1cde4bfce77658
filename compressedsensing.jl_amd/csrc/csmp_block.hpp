// csmp_block.hpp -- multi-column QR append (BASELINE config 5: "GOMP S=4 + Subspace Pursuit ...
// multi-column QR append").  A panel of P <= PB atoms joins the factorisation in one chain of five
// launches instead of 2-3 launches per atom:
//
//   k_blk1   guards (atoms already in the support are dropped: src/util.jl:119,129-134; capacity),
//            A_p = A[:, panel] (Float64), per-slab partial W1_g = Q_g' A_p,g
//   k_red    W1 = sum_g W1_g                      (fixed order: bitwise reproducible)
//   k_blk2   V = A_p - Q W1 per slab; per-slab partials of V'V, V'r and |a_p|^2
//   k_red    their sums
//   k_blk3   Cholesky V'V = Rp' Rp (every workgroup, redundantly), DGKS test on its diagonal,
//            Q_new = V Rp^-1, z_new = Rp^-T V'r, r -= Q_new z_new, R gets [W1; Rp], support += panel
//
// For panels of 32 the three products -- Q'A_p, A_p - Q W1, V'V -- run on the Float64 matrix cores
// (v_mfma_f64_16x16x4_f64, same peak as the vector FMA but without its LDS-operand bottleneck), and the
// 32 x 32 Cholesky lives in the registers of one wave with v_readlane broadcasts.
//
// i.e. block classical Gram-Schmidt against Q followed by CholeskyQR inside the panel.  The DGKS test
// (diag(Rp)^2 >= |a_p|^2 / 2) bounds the cancellation of both steps; a panel that fails it commits
// nothing and raises STOP_REORTH, and the host repeats the solve with the column-wise safe chain.
// SP re-factorises A[:, support] from scratch twice per iteration (src/twostage.jl:74,104-107):
// with PB = 32 that is 32 + 16 panels instead of 1536 single-column appends.
#pragma once
#include "csmp_kernels.hpp"

namespace csmp {

constexpr int kPanelMax = 32;

// dst[e] = sum_g src[g*stride + e], fixed order (threads <-> consecutive e: coalesced)
// With Rdst != NULL (the W1 reduction, e = c*PB + p) the sum is also stored as R[c, j+p]: the new
// columns of R are written by the whole grid instead of by one workgroup of k_blk3.
__global__ __launch_bounds__(256) void k_red(const double* __restrict__ src, double* __restrict__ dst, int n, int G,
                                             int64_t stride, const DevState* st, double* __restrict__ Rdst, int kcap,
                                             int PB) {
    if (st->pcount == 0) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int g = 0;
    for (; g + 8 <= G; g += 8) {
        double t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = src[(int64_t)(g + i) * stride + e];
        s0 += t[0]; s1 += t[1]; s2 += t[2]; s3 += t[3];
        s0 += t[4]; s1 += t[5]; s2 += t[6]; s3 += t[7];
    }
    for (; g < G; ++g) s0 += src[(int64_t)g * stride + e];
    const double sum = (s0 + s1) + (s2 + s3);
    dst[e] = sum;
    if (Rdst) {
        const int c = e / PB, p = e % PB, j = st->j;
        if (c < j && p < st->pcount) Rdst[(int64_t)(j + p) * kcap + c] = sum;
    }
}

// LDS of k_blk1: Aps[PB][64] | part[4][64][PB] | ints
template <int PB>
constexpr size_t blk1_lds_bytes() {
    return (size_t)(PB * kSlabRows + 4 * kWave * PB + kSlabRows * (PB + 1)) * sizeof(double) + 4 * PB * sizeof(int) + 64;
}

template <typename TA, int PB>
__global__ __launch_bounds__(kQrThreads) void k_blk1(const TA* __restrict__ A, int64_t ld, int M,
                                                     const double* __restrict__ Q, int64_t ldq, DevState* st,
                                                     const int* __restrict__ cands, const int* __restrict__ ncands, int base,
                                                     int want, const int* __restrict__ sel, int kcap, int skipmask,
                                                     double* __restrict__ Apan, double* __restrict__ PB1, int G,
                                                     int* __restrict__ pan_atoms) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Aps = lds;                              // [PB][64]
    double* part = Aps + PB * kSlabRows;            // [4][64][PB]
    double* ApT = part + 4 * kWave * PB;            // [64][PB + 1]: the slab transposed (B operand of the MFMA path)
    int* found = reinterpret_cast<int*>(ApT + kSlabRows * (PB + 1));  // [PB]
    int* pan = found + PB;                          // [PB]
    int* cnt = pan + PB;                            // [1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    if (st->done & skipmask) {
        if (g == 0 && tid == 0) st->pcount = 0;
        return;
    }
    const int nsel = st->nsel;
    int nreq = min(want, *ncands - base);
    if (nreq < 0) nreq = 0;
    if (nreq > PB) nreq = PB;
    if (tid < PB) found[tid] = 0;
    __syncthreads();
    for (int q = tid; q < nsel; q += kQrThreads) {
        const int v = sel[q];
        for (int p = 0; p < nreq; ++p)
            if (cands[base + p] == v) found[p] = 1;
    }
    __syncthreads();
    if (tid == 0) {
        const int room = min(kcap, M) - nsel;
        int c = 0;
        for (int p = 0; p < nreq && c < room; ++p)
            if (!found[p]) pan[c++] = cands[base + p];
        *cnt = c;
        if (g == 0) {
            st->j = nsel;
            st->pcount = c;
            if (room <= 0) st->done |= STOP_FULL;  // nnz(x) < size(A,1) guard (:117)
            for (int p = 0; p < c; ++p) pan_atoms[p] = pan[p];
        }
    }
    __syncthreads();
    const int P = *cnt;
    if (P == 0) return;
    // A_p slab -> LDS (Float64) and the global panel buffer
    for (int e = tid; e < PB * kSlabRows; e += kQrThreads) {
        const int p = e / kSlabRows, row = g * kSlabRows + (e % kSlabRows);
        const double a = (p < P && row < M) ? (double)A[(int64_t)pan[p] * ld + row] : 0.0;
        Aps[e] = a;
        ApT[(e % kSlabRows) * (PB + 1) + p] = a;
        if (p < P) Apan[(int64_t)p * ldq + row] = a;
    }
    __syncthreads();
    if constexpr (PB == 32) {
        // W1_g = Q_g' A_p,g on the Float64 matrix cores.  Wave w owns the 16 columns c0 + 16 w .. of a 64-column
        // chunk over ALL 64 rows of the slab: lane l holds 16 consecutive rows (16 (l >> 4) ..) of column l & 15 --
        // one 128-byte line -- and K-step kk multiplies row 16 (l >> 4) + kk of every quarter, so the 16 K-steps
        // consume exactly those registers.  B[k][p] comes from the transposed slab in LDS (row stride 33:
        // conflict-free).  No cross-wave reduction, no barrier inside the chunk loop.
        typedef double d4 __attribute__((ext_vector_type(4)));
        const int fr = lane & 15, fk = lane >> 4;
        const int cstep = kWave * (int)gridDim.y;
        double* out = PB1 + (int64_t)g * kcap * PB;
        const double* qb = Q + g * kSlabRows + fk * 16;
        const double* bt = ApT + (fk * 16) * (PB + 1) + fr;
        f64x2 qv[8], qn[8];
        auto loadq = [&](f64x2* dst, int cbase) {
            const int c = cbase + wave * 16 + fr;
            const f64x2* q = reinterpret_cast<const f64x2*>(qb + (int64_t)(c < nsel ? c : 0) * ldq);
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = (c < nsel) ? q[i] : (f64x2)0.0;
        };
        loadq(qv, (int)blockIdx.y * kWave);
        for (int c0 = (int)blockIdx.y * kWave; c0 < nsel; c0 += cstep) {
            loadq(qn, c0 + cstep);
            d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const double av = (kk & 1) ? qv[kk >> 1].y : qv[kk >> 1].x;
                const double b0 = bt[kk * (PB + 1)], b1 = bt[kk * (PB + 1) + 16];
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, a1, 0, 0, 0);
            }
            // C/D layout: col = lane & 15 (panel column), row = (lane >> 4) + 4 reg (Q column within the tile)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = c0 + wave * 16 + fk + 4 * reg;
                if (cc < nsel) {
                    if (fr < P) out[(int64_t)cc * PB + fr] = a0[reg];
                    if (fr + 16 < P) out[(int64_t)cc * PB + fr + 16] = a1[reg];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = qn[i];
        }
        return;
    }
    // W1_g = Q_g' A_p,g : lane <-> column of Q, wave <-> 16-row quarter, PB accumulators per thread;
    // the next 64-column chunk of Q is requested while the current one is multiplied
    const double* qbase = Q + g * kSlabRows + wave * 16;
    double* out = PB1 + (int64_t)g * kcap * PB;  // this slab's partials, [c][p]
    // (blockIdx.y splits the 64-column chunks of Q round-robin: the slabs alone are M/64 workgroups -- half
    //  the CUs at M = 8192 -- and the chunks of one slab are independent)
    const int cstep = kWave * (int)gridDim.y;
    f64x2 qv[8], qn[8];
    {
        const int c = (int)blockIdx.y * kWave + lane;
        const f64x2* q = reinterpret_cast<const f64x2*>(qbase + (int64_t)(c < nsel ? c : 0) * ldq);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = (c < nsel) ? q[i] : (f64x2)0.0;
    }
    for (int c0 = (int)blockIdx.y * kWave; c0 < nsel; c0 += cstep) {
        {
            const int c = c0 + cstep + lane;
            const f64x2* q = reinterpret_cast<const f64x2*>(qbase + (int64_t)(c < nsel ? c : 0) * ldq);
#pragma unroll
            for (int i = 0; i < 8; ++i) qn[i] = (c < nsel) ? q[i] : (f64x2)0.0;
        }
        double acc[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) acc[p] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const f64x2 ap = *reinterpret_cast<const f64x2*>(Aps + p * kSlabRows + wave * 16 + 2 * i);  // broadcast
                acc[p] = fma(qv[i].x, ap.x, acc[p]);
                acc[p] = fma(qv[i].y, ap.y, acc[p]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PB; ++p) part[(wave * PB + p) * kWave + lane] = acc[p];  // [wave][p][lane]: conflict-free
        __syncthreads();
        for (int e = tid; e < kWave * PB; e += kQrThreads) {
            const int p = e / kWave, cl = e % kWave, cc = c0 + cl;
            if (cc < nsel && p < P)
                out[(int64_t)cc * PB + p] = (part[(0 * PB + p) * kWave + cl] + part[(1 * PB + p) * kWave + cl]) +
                                            (part[(2 * PB + p) * kWave + cl] + part[(3 * PB + p) * kWave + cl]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = qn[i];
    }
}

// entries of the per-slab partial vector of k_blk2: V'V (PB*PB, row-major [p][q]) | V'r (PB) | |a_p|^2 (PB)
template <int PB>
constexpr int blk2_nent() { return PB * PB + 2 * PB; }
template <int PB>
constexpr size_t blk2_lds_bytes() {
    return (size_t)(PB * kSlabRows + kWave * PB + kSlabRows + PB * kSlabRows + kWave * kSlabRows) * sizeof(double) + 64;
}

template <int PB>
__global__ __launch_bounds__(kQrThreads) void k_blk2(const double* __restrict__ Q, int64_t ldq, const DevState* st,
                                                     const double* __restrict__ Apan, const double* __restrict__ W1b,
                                                     const double* __restrict__ r, double* __restrict__ Vpan,
                                                     double* __restrict__ PG, int G) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Vs = lds;                        // [PB][64]  (A_p slab, then V slab)
    double* Wt = Vs + PB * kSlabRows;        // [64][PB]  tile of W1
    double* rs = Wt + kWave * PB;            // [64]
    double* As = rs + kSlabRows;             // [PB][64]  copy of the A_p slab (for |a_p|^2)
    double* Qt = As + PB * kSlabRows;        // [64 cols][64 rows] tile of the Q slab
    const int P = st->pcount;
    if (P == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x, j = st->j;
    for (int e = tid; e < PB * kSlabRows; e += kQrThreads) {
        const int p = e / kSlabRows;
        const double a = (p < P) ? Apan[(int64_t)p * ldq + g * kSlabRows + (e % kSlabRows)] : 0.0;
        Vs[e] = a;
        As[e] = a;
    }
    if (tid < kSlabRows) rs[tid] = r[g * kSlabRows + tid];
    if constexpr (PB == 32) {
        // V = A_p - Q_g W1 on the Float64 matrix cores (v_mfma_f64_16x16x4_f64): wave w owns rows 16 w .. 16 w + 15
        // of the slab and all 32 panel columns (two 16 x 16 tiles).  Both operands come STRAIGHT from global
        // memory in fragment layout -- A[l&15][k = l>>4] is 16 consecutive rows of Q column c + k, B[k][l&15] is
        // 16 consecutive entries of W1 row c + k: 128-byte segments -- with 8 K-steps (32 columns of Q) of loads
        // in flight ahead of their MFMAs.  The LDS-staged scalar loop this replaces spent 9 LDS reads per 8 FMAs.
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c0v = {0.0, 0.0, 0.0, 0.0}, c1v = {0.0, 0.0, 0.0, 0.0};
        const int fr = lane & 15, fk = lane >> 4;
        const double* qa = Q + g * kSlabRows + wave * 16 + fr;  // + (c + fk) * ldq
        const double* wb = W1b + fr;                            // + (c + fk) * PB  (+ 16 for the second tile)
        constexpr int KU = 8;
        double av[KU], b0[KU], b1[KU], an[KU], n0[KU], n1[KU];
        auto fetchk = [&](double* a_, double* x0, double* x1, int cbase) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int c = cbase + 4 * u + fk;
                const bool ok = c < j;
                a_[u] = ok ? qa[(int64_t)c * ldq] : 0.0;
                x0[u] = ok ? wb[(int64_t)c * PB] : 0.0;
                x1[u] = ok ? wb[(int64_t)c * PB + 16] : 0.0;
            }
        };
        fetchk(av, b0, b1, 0);
        for (int cb = 0; cb < j; cb += 4 * KU) {
            fetchk(an, n0, n1, cb + 4 * KU);
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                c0v = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b0[u], c0v, 0, 0, 0);
                c1v = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b1[u], c1v, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                av[u] = an[u];
                b0[u] = n0[u];
                b1[u] = n1[u];
            }
        }
        __syncthreads();  // Vs / As staged above are complete
        // C/D layout: col = lane & 15 (panel column within the tile), row = (lane >> 4) + 4 reg
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = wave * 16 + fk + 4 * reg;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int p = nt * 16 + fr;
                const double v = Vs[p * kSlabRows + row] - (nt == 0 ? c0v[reg] : c1v[reg]);
                Vs[p * kSlabRows + row] = v;
                Qt[row * (PB + 1) + p] = v;  // transposed copy [row][33] for the Gram product below (Qt is free here)
                if (p < P) Vpan[(int64_t)p * ldq + g * kSlabRows + row] = v;
            }
        }
        __syncthreads();
    } else {
    // V = A_p - Q_g W1.  Q is staged in 64x64 tiles: every thread fetches 8 x 16 B of the NEXT tile
        // (thread <-> (column tid/32 + 8 i, row pair tid%32)) while the current tile is consumed from LDS with
        // lane <-> row and wave <-> PB/4 panel columns; W1 tiles come through LDS as well.
        constexpr int PW = PB / 4;
        double acc[PW];
#pragma unroll
        for (int t = 0; t < PW; ++t) acc[t] = 0.0;
        const int tc = tid >> 5, tr = (tid & 31) * 2;  // this thread's column offset (0..7) and row pair in a tile
        const double* qsrc = Q + g * kSlabRows + tr;
        f64x2 nx[8];
        constexpr int WPT = kWave * PB / kQrThreads;  // W1-tile entries per thread
        double nw[WPT];
        auto fetch = [&](int c0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + tc + 8 * i;
                nx[i] = (c < j) ? *reinterpret_cast<const f64x2*>(qsrc + (int64_t)c * ldq) : (f64x2)0.0;
            }
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int e = tid + i * kQrThreads;
                const int cc = c0 + e / PB;
                nw[i] = (cc < j) ? W1b[(int64_t)cc * PB + (e % PB)] : 0.0;
            }
        };
        fetch(0);
        for (int c0 = 0; c0 < j; c0 += kWave) {
            __syncthreads();  // previous tile fully consumed
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f64x2*>(Qt + (tc + 8 * i) * kSlabRows + tr) = nx[i];
#pragma unroll
            for (int i = 0; i < WPT; ++i) Wt[tid + i * kQrThreads] = nw[i];
            if (c0 + kWave < j) fetch(c0 + kWave);
            __syncthreads();
            const int nc = min(kWave, j - c0);
            for (int cl = 0; cl < nc; ++cl) {
                const double qv = Qt[cl * kSlabRows + lane];
#pragma unroll
                for (int t = 0; t < PW; ++t) acc[t] = fma(qv, Wt[cl * PB + wave * PW + t], acc[t]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PW; ++t) {
            const int p = wave * PW + t;
            const double v = Vs[p * kSlabRows + lane] - acc[t];
            Vs[p * kSlabRows + lane] = v;
            if (p < P) Vpan[(int64_t)p * ldq + g * kSlabRows + lane] = v;
        }
        __syncthreads();
    }
    // per-slab partials: V'V (upper triangle is enough, the full square is written), V'r, |a_p|^2
    double* out = PG + (int64_t)g * blk2_nent<PB>();
    if constexpr (PB == 32) {
        // V'V on the matrix cores as well: wave w owns the 16 x 16 tile (w >> 1, w & 1); both operands are read from
        // the transposed slab (row stride 33: conflict-free), K = the 64 rows in 16 steps
        typedef double d4 __attribute__((ext_vector_type(4)));
        const int fr = lane & 15, fk = lane >> 4, pt = wave >> 1, qt = wave & 1;
        const double* vt = Qt + (fk * 16) * (PB + 1);
        d4 gacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[kk * (PB + 1) + pt * 16 + fr], vt[kk * (PB + 1) + qt * 16 + fr], gacc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pp = pt * 16 + fk + 4 * reg, qq = qt * 16 + fr;
            out[pp * PB + qq] = (pp <= qq && qq < P) ? gacc[reg] : 0.0;
        }
        if (tid < 2 * PB) {
            const int pp = tid & (PB - 1);
            double sacc = 0.0;
            if (pp < P) {
                if (tid < PB)
                    for (int row = 0; row < kSlabRows; ++row) sacc = fma(Qt[row * (PB + 1) + pp], rs[row], sacc);
                else
                    for (int row = 0; row < kSlabRows; ++row) sacc = fma(As[pp * kSlabRows + row], As[pp * kSlabRows + row], sacc);
            }
            out[PB * PB + tid] = sacc;
        }
        return;
    }
    for (int e = tid; e < blk2_nent<PB>(); e += kQrThreads) {
        double s = 0.0;
        if (e < PB * PB) {
            const int p = e / PB, q = e % PB;
            if (p <= q && q < P)
                for (int row = 0; row < kSlabRows; ++row) s = fma(Vs[p * kSlabRows + row], Vs[q * kSlabRows + row], s);
        } else if (e < PB * PB + PB) {
            const int p = e - PB * PB;
            if (p < P)
                for (int row = 0; row < kSlabRows; ++row) s = fma(Vs[p * kSlabRows + row], rs[row], s);
        } else {
            const int p = e - PB * PB - PB;
            if (p < P)
                for (int row = 0; row < kSlabRows; ++row) s = fma(As[p * kSlabRows + row], As[p * kSlabRows + row], s);
        }
        out[e] = s;
    }
}

template <int PB>
constexpr size_t blk3_lds_bytes() {
    return (size_t)(blk2_nent<PB>() + PB * PB + 2 * PB + PB * kSlabRows) * sizeof(double) + 64;
}

template <int PB>
__global__ __launch_bounds__(kQrThreads) void k_blk3(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                     const double* __restrict__ Vpan, const double* __restrict__ Gsum,
                                                     const double* __restrict__ W1b, double* __restrict__ r,
                                                     double* __restrict__ R, double* __restrict__ z, int* __restrict__ sel,
                                                     const int* __restrict__ pan_atoms, int kcap) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Gs = lds;                          // PB*PB + 2 PB
    double* Rp = Gs + blk2_nent<PB>();         // [PB][PB] upper triangular, Rp[t*PB + p], t <= p
    double* zp = Rp + PB * PB;                 // [PB]
    double* rinv = zp + PB;                    // [PB]  1 / Rp[p][p]
    double* Vs = rinv + PB;                    // [PB][64]
    int& bad = *reinterpret_cast<int*>(Vs + PB * kSlabRows);
    const int P = st->pcount;
    if (P == 0) return;
    const int tid = threadIdx.x, g = blockIdx.x, j = st->j;
    for (int e = tid; e < blk2_nent<PB>(); e += kQrThreads) Gs[e] = Gsum[e];
    for (int e = tid; e < PB * kSlabRows; e += kQrThreads) {
        const int p = e / kSlabRows;
        Vs[e] = (p < P) ? Vpan[(int64_t)p * ldq + g * kSlabRows + (e % kSlabRows)] : 0.0;
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    // Cholesky V'V = Rp' Rp and zp = Rp^-T (V'r) in the registers of ONE wave: lane q owns column q
    // of the (padded to PB x PB, identity beyond P) Gram matrix; every dependent step is a register
    // broadcast (shuffle) + fma, no LDS round trips and no barriers.  Identical in every workgroup.
    if (tid < kWave) {
        const int q = tid;
        double gq[PB];
#pragma unroll
        for (int t = 0; t < PB; ++t)
            gq[t] = (q < PB && t <= q) ? ((q < P) ? Gs[t * PB + q] : (t == q ? 1.0 : 0.0)) : 0.0;
        double sv = (q < P) ? Gs[PB * PB + q] : 0.0;        // V'r
        const double na2 = (q < P) ? Gs[PB * PB + PB + q] : 0.0;
        double zmine = 0.0;
        int mybad = 0;
        double ri[PB];  // 1 / Rp[p][p]: divisions and square roots leave the dependent chain (v_rsq_f64 + two Newton steps)
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const double d = readlane_f64(gq[p], p);  // current pivot (lane p holds G[p][p]); p, s_ are unrolled constants: v_readlane, no LDS crossbar
            if (q == p && p < P && (!(d > 0.0) || !(d >= 0.5 * na2))) mybad = 1;  // DGKS: too much cancellation
            const bool okd = d > 0.0 && d < 1e300;
            double rs_ = __builtin_amdgcn_rsq(okd ? d : 1.0);
            rs_ = rs_ * fma(-0.5 * (okd ? d : 1.0) * rs_, rs_, 1.5);
            rs_ = rs_ * fma(-0.5 * (okd ? d : 1.0) * rs_, rs_, 1.5);
            ri[p] = okd ? rs_ : 1.0;
            const double rd = okd ? d * rs_ : 1.0;
            gq[p] = (q == p) ? rd : gq[p] * ri[p];  // row p of Rp (lanes q >= p)
#pragma unroll
            for (int s_ = p + 1; s_ < PB; ++s_) {
                const double rps = readlane_f64(gq[p], s_);  // Rp[p][s]
                if (q >= s_) gq[s_] = fma(-rps, gq[p], gq[s_]);
            }
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {  // forward substitution, column oriented
            const double zb = readlane_f64(sv * ri[p], p);
            if (q == p) zmine = zb;
            if (q > p) sv = fma(-gq[p], zb, sv);
        }
        if (__any(mybad) && tid == 0) bad = 1;
        if (q < PB) {
#pragma unroll
            for (int t = 0; t < PB; ++t)
                if (t <= q) Rp[t * PB + q] = gq[t];
            zp[q] = zmine;
        }
        if (q == 0) {
#pragma unroll
            for (int t = 0; t < PB; ++t) rinv[t] = ri[t];
        }
    }
    __syncthreads();
    if (bad) {
        if (g == 0 && tid == 0) st->done |= STOP_REORTH;  // nothing committed; the host falls back
        return;
    }
    // Q_new = V Rp^-1 row by row; r -= Q_new zp
    if (tid < kSlabRows) {
        const int row = g * kSlabRows + tid;
        double q[PB];
        double dr = 0.0;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            double s = Vs[p * kSlabRows + tid];
#pragma unroll
            for (int t = 0; t < PB; ++t)
                if (t < p) s = fma(-q[t], Rp[t * PB + p], s);
            q[p] = (p < P) ? s * rinv[p] : 0.0;
            if (p < P) {
                Q[(int64_t)(j + p) * ldq + row] = q[p];
                dr = fma(q[p], zp[p], dr);
            }
        }
        r[row] -= dr;
    }
    if (g == 0) {
        // (rows 0..j-1 of the new R columns = W1 were stored by the W1 reduction, k_red)
        for (int e = tid; e < P * P; e += kQrThreads) {
            const int t = e / P, p = e % P;
            if (t <= p) R[(int64_t)(j + p) * kcap + (j + t)] = Rp[t * PB + p];
        }
        if (tid < P) {
            z[j + tid] = zp[tid];
            sel[j + tid] = pan_atoms[tid];
        }
        __syncthreads();
        if (tid == 0) {
            st->nsel = j + P;
            st->steps += 1;
        }
    }
}

}  // namespace csmp
