// csmp_gram.hpp -- least squares on a whole column set at once, for the FROM-SCRATCH factorisations of Subspace
// Pursuit (factorize! + ldiv!, src/matchingpursuit.jl:219-227; solve!, src/twostage.jl:104-107: SP re-factorises
// A[:, support] twice per iteration, 2k and k columns) and for the lstsq primitive.
//
// The panel appends of csmp_block.hpp build Q column block by column block: 64 panels x 5 launches per SP solve at
// config 5, each a few microseconds of work.  Here the n columns are taken together:
//
//   k_gram        G = A_S' A_S (n x n, Float64 products of the exactly promoted dictionary values) on the Float64 matrix
//                 cores, upper 64 x 64 tiles only, the long dimension (M rows) split over workgroups;
//   k_gram_reduce the partials summed in a fixed order; the diagonal is kept aside for the DGKS test;
//   (k_gather_cols also sums a_j'b per row chunk: k_gram_reduce writes c = A_S'b as column n of G, the bordered matrix
//                 [G c; c' b'b])
//   k_chol_row / k_chol_step   right-looking blocked Cholesky G = R'R in place, 32 columns per step and ONE launch per step:
//                 the 32 x 32 diagonal block in the registers of one wave, the row panel by substitution (one thread per
//                 column), the trailing update of the previous panel on the matrix cores beside it.  The bordered column comes
//                 out as z = R^-T c = Q'b -- exactly what the append chain accumulates step by step;
//   k_gram_export R, z, support and count into the solver slot: from there on k_finish* (back substitution + sorted
//                 emission) and k_residual work as after any append chain.
//
// R is the triangular factor the appends would have produced (the Cholesky factor of A_S'A_S IS the R of A_S = QR), so
// the same DGKS-type test applies, column by column: R_jj^2 >= |a_j|^2 / 2.  A set that fails it anywhere (coherent
// atoms) raises STOP_REORTH, nothing is exported, and the host repeats the factorisation with the panel appends
// (which orthogonalise explicitly and fall back further to the column-wise chain).  For the supports SP meets on
// incoherent dictionaries cond(A_S) is a small constant and the normal-equation error cond^2 * eps stays at 1e-15.
// No Q is formed: callers that go on appending or removing columns (ompr, srr, br) keep using the panel appends.
#pragma once
#include "csmp_kernels.hpp"

namespace csmp {

using d4g = __attribute__((ext_vector_type(4))) double;
constexpr int kGramTile = 64;   // G tile edge per workgroup
constexpr int kCholNB = 32;     // columns per Cholesky step.  (64 was tried twice: a straight 64-step elimination in one wave, 203 us per
                                // step; two levels of 32 with an MFMA update in between, 63 us -- against 2 x 24 us for two steps of 32:
                                // the per-thread work of a step grows with NB^2 and Float64 FMAs are what a step consists of.)

// The n columns of the set, copied into one contiguous block: column j at j * ldo, ldo = M rounded up to 16 rows, zero rows
// beyond M and zero columns from n to np.  k_gram touches 96 columns per wave and block of rows, 16 bytes of each: from the
// compact copy every one of those is an unconditional, aligned 16-byte load (no bounds logic in the loop), and the 32 MiB
// of a 1024-column set at config 5 sit on a handful of pages instead of one page per access.
// The same pass delivers the right-hand side: workgroup (row chunk, column j) sums a_j[r] * b[r] over its 256 rows (column n:
// b[r]^2, the corner), k_gram_reduce adds the chunks in a fixed order into column n of the bordered matrix.
template <typename TA>
__global__ __launch_bounds__(256) void k_gather_cols(const TA* __restrict__ A, int64_t ld, int M, const int* __restrict__ cols, int n,
                                                     TA* __restrict__ out, int64_t ldo, const double* __restrict__ b, int np,
                                                     double* __restrict__ rhs_part) {
    __shared__ double red[8];
    const int j = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double prod = 0.0;
    if (r < ldo) {
        const TA v = (j < n && r < M) ? A[(int64_t)cols[j] * ld + r] : (TA)0;
        out[(int64_t)j * ldo + r] = v;
        if (r < M && j <= n) {
            const double br = b[r];
            prod = (j < n ? (double)v : br) * br;
        }
    }
    if (j <= n) {  // (uniform per workgroup)
        prod = block_sum256(prod, red);
        if (threadIdx.x == 0) rhs_part[(int64_t)blockIdx.x * np + j] = prod;
    }
}

constexpr int kGramRpl = 4;    // rows of a column a lane holds per block of rows (16 bytes of f32)
constexpr int kGramWgI = 128;  // G rows per workgroup (k_gram): 2 x 2 waves, each 64 rows x 32 columns
constexpr int kGramWgJ = 64;   // G columns per workgroup

// One (128-row block I, 64-column block J) piece of the upper triangle x one slice of the rows, on the compact copy Ac.
// 4 waves as 2 x 2, each a 64 x 32 piece = 4 x 2 MFMA tiles (64 accumulator registers).  Lane (fr = l & 15, fq = l >> 4)
// loads 4 consecutive rows (block base + 4 fq) of column fr of each 16-column group; MFMA step kk multiplies row
// (base + 4 fq + kk) of both operands: a permutation of the summation index, the same on both sides.  The fragments of the
// NEXT block of rows are requested before the current block's 32 MFMAs are issued (two register sets), so the matrix
// cores do not wait for L2: operands come straight from global memory, no LDS, no barriers.  Three workgroups per CU (f32):
// one wave per SIMD cannot issue Float64 MFMAs back to back (measured: 35 TFLOP/s with one, 47 with two waves per SIMD).
template <typename TA>
__global__ __launch_bounds__(256, (sizeof(TA) == 4 ? 3 : 2)) void k_gram(const TA* __restrict__ Ac, int64_t ldo, int np,
                                                                         int rows_per_split, double* __restrict__ Gpart, int jtile0 = 0,
                                                                         int jtile1 = 1 << 30) {
    constexpr int RPL = kGramRpl, BLK = 4 * RPL;
    struct alignas(sizeof(TA) * RPL) Frag { TA v[RPL]; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const int I = blockIdx.y, J = blockIdx.x, ks = blockIdx.z;
    if (I * kGramWgI > J * kGramWgJ + kGramWgJ - 1) return;  // entirely below the diagonal
    if (J < jtile0) return;  // (bordered extension: only the new columns' tiles are needed, k_ext_reduce)
    if (J >= jtile1) return;  // (column tiles that hold nothing but the bordered column and padding: the reduce kernels take that column
                              // from the gather pass's sums -- at n = 512, np = 576 they were 5 of the 25 pieces)
    const int k0 = ks * rows_per_split, k1 = (int)min((int64_t)ldo, (int64_t)k0 + rows_per_split);
    const int i0 = I * kGramWgI + wi * 64, j0 = J * kGramWgJ + wj * 32;
    const TA *ci[4], *cj[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) ci[t] = Ac + (int64_t)min(i0 + t * 16 + fr, np - 1) * ldo + fq * RPL;  // (rows >= np: clamped, never stored)
#pragma unroll
    for (int u = 0; u < 2; ++u) cj[u] = Ac + (int64_t)min(j0 + u * 16 + fr, np - 1) * ldo + fq * RPL;
    d4g acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = d4g{0.0, 0.0, 0.0, 0.0};
    Frag ca[4], cb[2], na[4], nb[2];
    auto fetch = [&](Frag (&fa)[4], Frag (&fb)[2], int rb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const Frag*>(ci[t] + rb);
#pragma unroll
        for (int u = 0; u < 2; ++u) fb[u] = *reinterpret_cast<const Frag*>(cj[u] + rb);
    };
    if (k0 < k1) fetch(ca, cb, k0);
    for (int rb = k0; rb < k1; rb += BLK) {
        if (rb + BLK < k1) fetch(na, nb, rb + BLK);
#pragma unroll
        for (int kk = 0; kk < RPL; ++kk)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double b = (double)cb[u].v[kk];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ca[t].v[kk], b, acc[t][u], 0, 0, 0);
            }
#pragma unroll
        for (int t = 0; t < 4; ++t) ca[t] = na[t];
#pragma unroll
        for (int u = 0; u < 2; ++u) cb[u] = nb[u];
    }
    // C/D layout: column = lane & 15, row = (lane >> 4) + 4 reg
    double* out = Gpart + (int64_t)ks * np * np;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = i0 + t * 16 + fq + 4 * reg, col = j0 + u * 16 + fr;
                if (row < np && col < np) out[row + (int64_t)col * np] = acc[t][u][reg];
            }
}

// G = sum of the row-slice partials (fixed order) on the upper tiles; identity on the padding diagonal (columns beyond the
// bordered one), zeros elsewhere in the padding; the original diagonal goes to gdiag (DGKS reference, |a_j|^2)
// AUGMENTED form (npa > np): G has leading dimension npa and npa - np further columns, the first n of them the unit vectors
// e_0 .. e_{n-1} (zeros otherwise).  The Cholesky kernels treat them like the bordered column: their row panels come out as
// R^-T e_j, i.e. the block G[0:n, np:np+n] ends up holding (R^-1)' -- the explicit inverse a LATER extension of this set needs
// (ls_gram_extend_t: W = R_F^-T G_FN as one tiled product), for the price of wider row panels in launches whose duration is
// set by their dependent chain, not by their width.
__device__ __forceinline__ bool gram_aug_entry(int64_t e, int np, int npa, int n, double* __restrict__ G) {
    // entries of the augmented columns, one per thread e in [np * npa, npa * npa): true when e was one of them
    if (e < (int64_t)np * npa) return false;
    if (e < (int64_t)npa * npa) {
        const int row = (int)(e % npa), col = (int)(e / npa);
        G[e] = (row < n && row == col - np) ? 1.0 : 0.0;
    }
    return true;
}
__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ Gpart, int nsplit, int n, int np,
                                                     double* __restrict__ G, double* __restrict__ gdiag,
                                                     const double* __restrict__ rhs_part, int nchunk,
                                                     double* __restrict__ Gkeep, double* __restrict__ gdkeep, int npa) {
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gram_aug_entry(e0 - (int64_t)np * np + (int64_t)np * npa, np, npa, n, G)) return;  // (threads beyond np * np take the augmented columns)
    const int64_t e = e0;
    if (e >= (int64_t)np * np) return;
    const int row = (int)(e % np), col = (int)(e / np);
    if ((row / kGramTile) > (col / kGramTile)) return;  // lower tiles are never read
    double s = 0.0;
    if (row < n && col < n) {
        for (int k = 0; k < nsplit; ++k) s += Gpart[(int64_t)k * np * np + e];
    } else if (row <= n && col == n) {  // the bordered column c = A_S'b and the corner b'b: the gather pass's row-chunk sums
        for (int k = 0; k < nchunk; ++k) s += rhs_part[(int64_t)k * np + row];
    } else if (row == col && row > n) {
        s = 1.0;
    }
    G[row + (int64_t)col * npa] = s;
    Gkeep[e] = s;  // the copy that survives the factorisation (a later subset of this set gathers its matrix from it)
    if (row == col && row < n) {
        gdiag[row] = s;
        gdkeep[row] = s;
    }
}

// The bordered Gram matrix of a set that lies inside the set of the kept matrix K (kn columns, leading dimension knp): entry
// (i, j) is K's entry (pos[i], pos[j]) -- K holds the upper tiles, so the pair is ordered first -- the right-hand side and
// the corner come from K's bordered column, the diagonal reference from K's.  Same layout rules as k_gram_reduce.
__global__ __launch_bounds__(256) void k_gram_subset(const double* __restrict__ K, int knp, int kn, const double* __restrict__ kdiag,
                                                     const int* __restrict__ pos, int n, int np, double* __restrict__ G,
                                                     double* __restrict__ gdiag, int npa) {
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gram_aug_entry(e0 - (int64_t)np * np + (int64_t)np * npa, np, npa, n, G)) return;
    const int64_t e = e0;
    if (e >= (int64_t)np * np) return;
    const int row = (int)(e % np), col = (int)(e / np);
    if ((row / kGramTile) > (col / kGramTile)) return;  // lower tiles are never read
    double s = 0.0;
    if (row < n && col < n) {
        const int a = pos[row], b = pos[col];
        s = K[min(a, b) + (int64_t)max(a, b) * knp];
    } else if (row < n && col == n) {
        s = K[pos[row] + (int64_t)kn * knp];
    } else if (row == n && col == n) {
        s = K[kn + (int64_t)kn * knp];
    } else if (row == col && row > n) {
        s = 1.0;
    }
    G[row + (int64_t)col * npa] = s;
    if (row == col && row < n) gdiag[row] = kdiag[pos[row]];
}

// Block row kb of the factor, by workgroups of 5 waves: wave 0 factorises the 32 x 32 diagonal block in its registers (every
// workgroup, redundantly; lane q holds column q, lanes 32..63 mirror 0..31); waves 1..4 own one column c >= c0 + 32 of the row
// panel each, R[c0 .. c0+31, c] = U^-T G[c0 .. c0+31, c] (kCholRowCols of them per workgroup).  UPD: the trailing update of the
// PREVIOUS panel (rows c0-32 .. c0-1, already final) has not been applied to this block row yet -- every thread applies it to its
// own column first (k_chol_step below).
//   * The row of the factor an elimination step produces goes from lane to lanes through LDS (one 8-byte write per lane, then
//     wave-uniform 16-byte reads: the LDS serves a wave's instructions in order, so the reads see the row) -- not through 62
//     v_readlane_b32 per step.
//   * UPD: the multipliers R[c0 - 32 + e, c0 + p] are the same for every lane: they come through the SCALAR unit (s_load of
//     64 B from the finished panel in global memory, constant address space) and enter the FMAs as scalar operands.
// What a step costs (tools/probes/chol_probe.hip, n = 1024): 17.6 us (24.1 with 256 columns per workgroup and scalar updates;
// 22.4 with 64 columns; 18.9 with the diagonal block's update on the matrix cores; 17.6 with the column wave's too), of which the 24.1 were 2.6 us launch, 4.4 us the dependent trip
// through memory between two launches, 5 us the update, 3 + 2 us elimination and its rsqrt chain, 3.4 us substitution; the
// trailing update runs beside it for free.  Tried and slower: rolled loops on a shifting register window (38 us: twice the
// FMAs -- code size was NOT the limit), the update over LDS operands (24.7 us).
constexpr int kCholRowCols = 64;    // panel columns per workgroup: ONE column wave (waves 2..4 of a row workgroup only join the barrier).  A
                                    // workgroup's 2 x 32 rows of its columns come through one CU's memory port: 256 columns per workgroup
                                    // cost 24.5 us per step, 128: 23.2, 64: 22.4, 32: 22.4 (tools/probes/chol_probe.hip)
constexpr int kCholThreads = 320;   // + the wave of the diagonal block

template <bool UPD>
__device__ __forceinline__ void chol_row_body(double* __restrict__ G, int np, int n, int kb, const double* __restrict__ gdiag,
                                              DevState* st, int wg, double* __restrict__ Dfac) {
    constexpr int NB = kCholNB;
    __shared__ __attribute__((aligned(16))) double U[NB * NB];  // U[t * NB + q]: row t of the factored diagonal block
    __shared__ double rinv[NB];
    const int tid = threadIdx.x, c0 = kb * NB;
    const bool diag = tid < kWave;
    const int q = tid & (NB - 1);
    const int c = diag ? c0 + q : c0 + NB + wg * kCholRowCols + (tid - kWave);
    const bool have = diag ? tid < NB : (c < np && (tid - kWave) < kCholRowCols);
    double* gcol = G + c0 + (int64_t)(c < np ? c : c0) * np;  // (np is a multiple of 64: 16-byte aligned)
    double x[NB];
    {
        const f64x2* g2 = reinterpret_cast<const f64x2*>(gcol);
#pragma unroll
        for (int p = 0; p < NB; p += 2) {
            const f64x2 v = g2[p / 2];
            x[p] = v.x;
            x[p + 1] = v.y;
        }
    }
    if (diag) {
#pragma unroll
        for (int p = 0; p < NB; ++p) x[p] = (p <= q) ? x[p] : 0.0;  // (below the diagonal: not part of the stored triangle)
    }
    if constexpr (UPD) {
        if (diag) {
            // The diagonal block's update is on the step's critical path (its factorisation waits for it): wave 0 forms
            // Xd'Xd (Xd = the 32 x 32 piece of the previous panel above the block) on the matrix cores -- 24 MFMAs for the three
            // upper 16 x 16 tiles instead of 1024 dependent scalar FMAs per lane -- and redistributes it through LDS from the
            // MFMA layout (column = lane & 15, row = (lane >> 4) + 4 reg) to "lane q holds column q".
            __shared__ __attribute__((aligned(16))) double Hp[NB * NB];
            const int lane = tid, fr = lane & 15, fq = lane >> 4;
            const double* xb = G + (c0 - NB) + fq * 8;  // lane fq supplies rows 8 fq .. 8 fq + 7 of Xd (a permutation of the summation index)
            double a0[8], a1[8];
            {
                const f64x2* p0 = reinterpret_cast<const f64x2*>(xb + (int64_t)(c0 + fr) * np);
                const f64x2* p1 = reinterpret_cast<const f64x2*>(xb + (int64_t)(c0 + 16 + fr) * np);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f64x2 u = p0[e / 2], v = p1[e / 2];
                    a0[e] = u.x;
                    a0[e + 1] = u.y;
                    a1[e] = v.x;
                    a1[e + 1] = v.y;
                }
            }
            d4g h00 = {0.0, 0.0, 0.0, 0.0}, h01 = h00, h11 = h00;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                h00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], a0[kk], h00, 0, 0, 0);
                h01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], a1[kk], h01, 0, 0, 0);
                h11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], a1[kk], h11, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = fq + 4 * reg;
                Hp[r * NB + fr] = h00[reg];              // rows 0..15,  columns 0..15
                Hp[r * NB + 16 + fr] = h01[reg];         // rows 0..15,  columns 16..31
                Hp[(16 + r) * NB + 16 + fr] = h11[reg];  // rows 16..31, columns 16..31
            }
            asm volatile("" ::: "memory");  // (same wave, in-order LDS: the reads below see the writes)
#pragma unroll
            for (int p = 0; p < NB; ++p)
                if (p <= q) x[p] -= Hp[p * NB + q];  // (the upper triangle is all the factorisation reads)
        } else if (tid < 2 * kWave) {
            // The column wave (kCholRowCols = 64 columns: wave 1) does the same on its 32 x 64 piece of the block row:
            // D = Xd' Xc, 2 x 4 tiles x 8 MFMA steps, handed through LDS to "thread holds its column".  (The scalar form -- 1024 FMAs
            // with s_load operands per thread -- took 5 us and crowded the scalar pipe wave 0's elimination lives on.)
            static_assert(kCholRowCols == kWave, "one column wave per row workgroup");
            __shared__ __attribute__((aligned(16))) double Hc[NB * kCholRowCols];  // Hc[row * 64 + column]
            const int lane = tid - kWave, fr = lane & 15, fq = lane >> 4;
            const int cb = c0 + NB + wg * kCholRowCols;  // first column of this workgroup's piece
            const double* xb = G + (c0 - NB) + fq * 8;
            double a0[8], a1[8], bq[4][8];
            {
                const f64x2* p0 = reinterpret_cast<const f64x2*>(xb + (int64_t)(c0 + fr) * np);
                const f64x2* p1 = reinterpret_cast<const f64x2*>(xb + (int64_t)(c0 + 16 + fr) * np);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f64x2 u = p0[e / 2], v = p1[e / 2];
                    a0[e] = u.x;
                    a0[e + 1] = u.y;
                    a1[e] = v.x;
                    a1[e + 1] = v.y;
                }
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const int cj = cb + 16 * tj + fr;
                    const f64x2* pb = reinterpret_cast<const f64x2*>(xb + (int64_t)(cj < np ? cj : c0) * np);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f64x2 v = pb[e / 2];
                        bq[tj][e] = cj < np ? v.x : 0.0;
                        bq[tj][e + 1] = cj < np ? v.y : 0.0;
                    }
                }
            }
            d4g h0[4], h1[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) h0[tj] = h1[tj] = d4g{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    h0[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], bq[tj][kk], h0[tj], 0, 0, 0);
                    h1[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], bq[tj][kk], h1[tj], 0, 0, 0);
                }
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = fq + 4 * reg;
                    Hc[r * kCholRowCols + 16 * tj + fr] = h0[tj][reg];
                    Hc[(16 + r) * kCholRowCols + 16 * tj + fr] = h1[tj][reg];
                }
            asm volatile("" ::: "memory");  // (same wave, in-order LDS)
#pragma unroll
            for (int p = 0; p < NB; ++p) x[p] -= Hc[p * kCholRowCols + lane];
        }
    }
    if (diag) {
        const double ref = (c0 + q < n) ? gdiag[c0 + q] : 0.0;
        int mybad = 0;
        double ri[NB];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const double d = readlane_f64(x[p], p);
            if (q == p && c0 + p < n && (!(d > 0.0) || !(d >= 0.5 * ref))) mybad = 1;  // DGKS: too much cancellation
            const bool okd = d > 0.0 && d < 1e300;
            const double dd = okd ? d : 1.0;
            double rs_ = __builtin_amdgcn_rsq(dd);
            rs_ = rs_ * fma(-0.5 * dd * rs_, rs_, 1.5);
            rs_ = rs_ * fma(-0.5 * dd * rs_, rs_, 1.5);
            ri[p] = okd ? rs_ : 1.0;
            const double top = (q == p) ? (okd ? d * rs_ : 1.0) : x[p] * ri[p];  // R[p][q] (q >= p; lanes q < p: dead values)
            x[p] = top;
            U[p * NB + q] = top;
            asm volatile("" ::: "memory");  // (the reads below follow the write in program order: same wave, in-order LDS)
#pragma unroll
            for (int s_ = p + 1; s_ < NB; ++s_) x[s_] = fma(-U[p * NB + s_], top, x[s_]);
        }
        if (__any(mybad) && wg == 0 && tid == 0) st->done |= STOP_REORTH;
        if (q == 0) {
#pragma unroll
            for (int t = 0; t < NB; ++t) rinv[t] = ri[t];
        }
        // The factored diagonal block goes to a SIDE buffer (block kb: Dfac[kb * NB * NB + row + NB * column]), not back into
        // G: every row workgroup of this launch reads the unfactored block from G at its entry and factorises it for itself, and
        // nothing orders workgroup 0's store against a late workgroup's load (a GPU shared with another stream or process
        // starts them far apart) -- the input stays read-only for the whole launch.  k_gram_export reads the diagonal blocks here.
        if (wg == 0 && tid < NB) {
            double* dcol = Dfac + (int64_t)kb * NB * NB + (int64_t)q * NB;
#pragma unroll
            for (int t = 0; t < NB; ++t)
                if (t <= q) dcol[t] = x[t];
        }
    }
    __syncthreads();
    if (diag || !have) return;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const double xt = x[t] * rinv[t];
        x[t] = xt;
#pragma unroll
        for (int p = t + 1; p < NB; ++p) x[p] = fma(-U[t * NB + p], xt, x[p]);
    }
    f64x2* o2 = reinterpret_cast<f64x2*>(gcol);
#pragma unroll
    for (int p = 0; p < NB; p += 2) o2[p / 2] = f64x2{x[p], x[p + 1]};
}

// The first block row (nothing to apply before it).
__global__ __launch_bounds__(kCholThreads) void k_chol_row(double* __restrict__ G, int np, int n, int kb, const double* __restrict__ gdiag,
                                                           DevState* st, double* __restrict__ Dfac) {
    chol_row_body<false>(G, np, n, kb, gdiag, st, (int)blockIdx.x, Dfac);
}

// The trailing update of step kb: G[i][j] -= sum_p X[p][i] X[p][j] over the NB rows X = G[c0 .. c0+NB-1, :] just finished, for
// the upper 64 x 64 tile `p` of the trailing matrix (columns >= c0 + NB).  Matrix cores; lane fq takes 8 consecutive rows of X
// per pass (NB / 32 passes); the accumulators start from G itself and one operand enters negated, so the loads of a pass are
// all issued before its first MFMA.  skip_first: rows c0+NB .. c0+2NB-1 (the next block row) are left to chol_row_body<true>.
__device__ __forceinline__ void chol_trail_body(double* __restrict__ G, int np, int kb, int p, bool skip_first, int rowlim) {
    constexpr int NB = kCholNB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const int c0 = kb * NB, t0 = c0 + NB;  // trailing matrix starts at column t0
    int J = 0;
    while ((J + 1) * (J + 2) / 2 <= p) ++J;
    const int I = p - J * (J + 1) / 2;
    if (t0 + I * kGramTile >= rowlim) return;  // (augmented factorisation: rows beyond the real matrix are never factorised)
    const int tlo = (skip_first && I == 0) ? NB / 16 : 0;  // 16-row sub-tiles of the tile that belong to the next block row
    const int cj = t0 + J * kGramTile + wave * 16 + fr;
    d4g acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = t0 + I * kGramTile + t * 16 + fq + 4 * reg;
            acc[t][reg] = (t >= tlo && row < np && cj < np && row <= cj) ? G[row + (int64_t)cj * np] : 0.0;
        }
#pragma unroll
    for (int pass = 0; pass < NB / 32; ++pass) {
        const double* X = G + c0 + pass * 32 + fq * 8;
        double bj[8], ai[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bj[e] = (cj < np) ? X[e + (int64_t)cj * np] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ci = t0 + I * kGramTile + t * 16 + fr;
#pragma unroll
            for (int e = 0; e < 8; ++e) ai[t][e] = (t >= tlo && ci < np) ? -X[e + (int64_t)ci * np] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[t][kk], bj[kk], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = t0 + I * kGramTile + t * 16 + fq + 4 * reg;
            if (t >= tlo && row < np && cj < np && row <= cj) G[row + (int64_t)cj * np] = acc[t][reg];
        }
}

// Cholesky step kb -> kb + 1 in ONE launch (the factorisation is a chain of ~n/32 dependent steps, a few microseconds of work
// each: what it costs is the number of launches on that chain).  Workgroups [0, nrow) produce block row kb + 1: they apply
// panel kb to their own 32 x 256 part of it, factorise its diagonal block, solve.  Workgroups [nrow, ..) apply panel kb to the
// rest of the trailing matrix meanwhile.  The two groups touch disjoint rows; panel kb itself is only read.
__global__ __launch_bounds__(kCholThreads) void k_chol_step(double* __restrict__ G, int np, int n, int kb,
                                                            const double* __restrict__ gdiag, DevState* st, int nrow,
                                                            double* __restrict__ Dfac, int rowlim) {
    if ((int)blockIdx.x < nrow) {
        chol_row_body<true>(G, np, n, kb + 1, gdiag, st, (int)blockIdx.x, Dfac);
    } else {
        if (threadIdx.x >= 256) return;
        chol_trail_body(G, np, kb, (int)blockIdx.x - nrow, true, rowlim);
    }
}

// ---- the host link of a whole-set solve: ONE kernel up, ONE kernel down (a small hipMemcpyAsync is a blit kernel of its own,
// 2-5 us on the chain each, and there were three and four of them per phase).  Page-locked host memory is mapped into the
// device's address space under the same pointer: the kernels read / write it directly over the link.
// up: [cols (n) | n | positions (npos)] from the staging buffer into the slot's column list, its count and the position list
__global__ __launch_bounds__(256) void k_put_lists(const int* __restrict__ src, int n, int npos, int* __restrict__ cands, int* __restrict__ ncands,
                                                   int* __restrict__ kpos) {
    const int t = (int)blockIdx.x * 256 + threadIdx.x;
    if (t < n) cands[t] = src[t];
    if (t == 0) *ncands = src[n];
    if (t < npos) kpos[t] = src[n + 1 + t];
}
// down: [idx (n) | val (n) | control block | shares of ||r||^2 (nshare)] into the landing area
__global__ __launch_bounds__(256) void k_land_ls(const int64_t* __restrict__ out_idx, const double* __restrict__ out_val, int n,
                                                 const DevState* __restrict__ st, const double* __restrict__ rn2part, int nshare,
                                                 int64_t* __restrict__ hi, double* __restrict__ hv, int* __restrict__ hst, double* __restrict__ hn2) {
    const int t = (int)blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        hi[t] = out_idx[t];
        hv[t] = out_val[t];
    }
    if (t < (int)(sizeof(DevState) / sizeof(int))) hst[t] = reinterpret_cast<const int*>(st)[t];
    if (t < nshare) hn2[t] = rn2part[t];
}
// down: the k selected atoms, their count and (screened selection) the certificate flag
__global__ __launch_bounds__(256) void k_land_sel(const int* __restrict__ cands, const int* __restrict__ ncands, int k, const int* __restrict__ flag,
                                                  int* __restrict__ dst) {
    const int t = (int)blockIdx.x * 256 + threadIdx.x;
    if (t < k) dst[t] = cands[t];
    if (t == 0) {
        dst[k] = *ncands;
        dst[k + 1] = flag ? *flag : 0;
    }
}

// R (n x n upper, leading dimension kcap), z = column n, support = cols, count = n -> the solver slot.  Nothing is
// exported when the DGKS test failed anywhere (the host sees STOP_REORTH and falls back).
__global__ __launch_bounds__(256) void k_gram_export(const double* __restrict__ G, int np, int n, const int* __restrict__ cols,
                                                     double* __restrict__ R, int kcap, double* __restrict__ z,
                                                     int* __restrict__ sel, DevState* st, const double* __restrict__ Dfac) {
    if (st->done & STOP_REORTH) return;
    constexpr int NB = kCholNB;
    // entry (row, col) of the factor: the diagonal blocks live in the side buffer (chol_row_body), the row panels in G
    auto fac = [&](int row, int col) {
        return (row / NB == col / NB) ? Dfac[(int64_t)(row / NB) * NB * NB + (row % NB) + (int64_t)(col % NB) * NB] : G[row + (int64_t)col * np];
    };
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < (int64_t)n * n) {
        const int row = (int)(e % n), col = (int)(e / n);
        if (row <= col) R[row + (int64_t)col * kcap] = fac(row, col);
    }
    if (e < n) {
        z[e] = fac((int)e, n);
        sel[e] = cols[e];
    }
    if (e == 0) {
        st->nsel = n;
        st->j = n;
        st->steps += 1;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Bordered extension of a factorised set (Subspace Pursuit's update!: the 2k-set T = F + N holds the k atoms F whose factor
// the previous solve left in the solver slot -- src/twostage.jl:74 "IDEA: could add efficient qr updating"; this is that idea on
// the normal equations).  With G_TT = [G_FF G_FN; . G_NN] and R_F'R_F = G_FF already known,
//     R_T = [R_F W; 0 R_C],   W = R_F^-T G_FN,   R_C'R_C = G_NN - W'W,   z_T = [z_F; R_C^-T (c_N - W'z_F)]
// -- the block Cholesky of G_TT in the column order [F | N], of which only the N part is computed: n_N/32 dependent steps instead
// of (n_F + n_N)/32, the substitution W as n_N independent columns in ONE launch (k_trsm_rt), the Schur complement on the
// Float64 matrix cores (k_gram<double> on W, bordered by z_F).
//
// k_ext_reduce: the kept bordered Gram matrix of T (Knew, np x np, upper tiles; the next subset solve gathers from it) from
//   * the old kept matrix (G_FF: entry (posF[i], posF[j])), * k_gram's partials (columns >= nF), * k_gather_cols' right-hand side;
// G_FN into Gin (ldw x nN: k_wgemm's input), and Wb's (ldw x np2) zero padding and border column z_F (k_wgemm fills the rest).
__global__ __launch_bounds__(256) void k_ext_reduce(const double* __restrict__ Gpart, int nsplit, int nF, int n, int np,
                                                    const double* __restrict__ rhs_part, int nchunk, const double* __restrict__ Kold,
                                                    int knp, const int* __restrict__ posF, const double* __restrict__ zF,
                                                    double* __restrict__ Knew, double* __restrict__ kdnew, double* __restrict__ Wb,
                                                    int ldw, int np2, double* __restrict__ Gin) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < (int64_t)ldw * np2) {  // Wb's padding and border first (entries of G_FN are written below, by their owners)
        const int row = (int)(e % ldw), col = (int)(e / ldw), nN = n - nF;
        if (row >= nF || col > nN) Wb[e] = 0.0;
        else if (col == nN) Wb[e] = zF[row];
    }
    if (e >= (int64_t)np * np) return;
    const int row = (int)(e % np), col = (int)(e / np);
    if ((row / kGramTile) > (col / kGramTile)) return;  // lower tiles are never read
    double s = 0.0;
    if (row < nF && col < nF) {
        const int a = posF[row], b = posF[col];
        s = Kold[min(a, b) + (int64_t)max(a, b) * knp];
    } else if (row < n && col < n) {  // (col >= nF: the tiles k_gram computed)
        for (int k = 0; k < nsplit; ++k) s += Gpart[(int64_t)k * np * np + e];
        if (row < nF) Gin[row + (int64_t)(col - nF) * ldw] = s;  // (k_wgemm's input; its output goes to Wb)
    } else if (row <= n && col == n) {
        for (int k = 0; k < nchunk; ++k) s += rhs_part[(int64_t)k * np + row];
    } else if (row == col && row > n) {
        s = 1.0;
    }
    Knew[e] = s;
    if (row == col && row < n) kdnew[row] = s;
}

// x = R^-1 z as a product with the explicit inverse the augmented factorisation left: x_j = sum_{t >= j} Tt[t, j] z_t (column j of
// Tt = (R^-1)', contiguous), one wave per coefficient -- instead of the back substitution's chain of 256-column super-blocks.
__global__ __launch_bounds__(256) void k_tt_gemv(const double* __restrict__ Tt, int ldt, const double* __restrict__ z, const DevState* st,
                                                 int n, double* __restrict__ x, int nsel) {
    if (st->nsel != nsel) return;  // (nothing was exported: the set failed its DGKS test)
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double* col = Tt + (int64_t)j * ldt;
    double a0 = 0.0, a1 = 0.0;
    int t = j + lane;
    for (; t + 64 < n; t += 128) {
        a0 = fma(col[t], z[t], a0);
        a1 = fma(col[t + 64], z[t + 64], a1);
    }
    if (t < n) a0 = fma(col[t], z[t], a0);
    double a = a0 + a1;
    for (int s_ = 32; s_ >= 1; s_ >>= 1) a += shx(a, s_);
    if (lane == 0) x[j] = a;
}

// W = R_F^-T G_FN = Tt G_FN as a tiled product (no chain): W[i][c] = sum_{t <= i} Tt[i, t] G[t, c], Tt = (R_F^-1)' from the
// augmented factorisation of F (lower triangular: zeros above its diagonal).  32 x 32 output tiles (256 workgroups at n_F = n_N = 512),
// 256 threads with 2 x 2 outputs each, K-tiles of 32 through LDS, the next K-tile's operands already on their way (registers)
// while the current one is multiplied; K-tiles beyond the output tile's rows are skipped.
__global__ __launch_bounds__(256) void k_wgemm(const double* __restrict__ Tt, int ldt, const double* __restrict__ Gin, int ldg, int nF, int nN,
                                               double* __restrict__ Wb, int ldw) {
    constexpr int TM = 32, TN = 32, TK = 32;
    __shared__ double As[TK][TM + 1];  // As[k][i] = Tt[i0 + i, k0 + k]
    __shared__ double Bs[TK][TN + 1];  // Bs[k][c] = Gin[k0 + k, c0 + c]
    const int i0 = blockIdx.x * TM, c0 = blockIdx.y * TN;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // outputs rows i0 + tx, i0 + tx + 16; columns c0 + ty, c0 + ty + 16
    const int li = threadIdx.x & 31, lk = threadIdx.x >> 5;  // loads: element li of the contiguous dimension, 4 of the 32 others (lk + 8 j)
    double acc00 = 0.0, acc01 = 0.0, acc10 = 0.0, acc11 = 0.0;
    const int kend = min(nF, i0 + TM);
    double ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = lk + 8 * j;
            const int gi = i0 + li, gk = k0 + kk;  // Tt: contiguous along i
            ra[j] = (gi < nF && gk < nF) ? Tt[gi + (int64_t)gk * ldt] : 0.0;
            const int gkb = k0 + li, gc = c0 + kk;  // Gin: contiguous along t = k
            rb[j] = (gkb < nF && gc < nN) ? Gin[gkb + (int64_t)gc * ldg] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kend; k0 += TK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[lk + 8 * j][li] = ra[j];
            Bs[li][lk + 8 * j] = rb[j];
        }
        __syncthreads();
        if (k0 + TK < kend) fetch(k0 + TK);
#pragma unroll 8
        for (int k = 0; k < TK; ++k) {
            const double a0 = As[k][tx], a1 = As[k][tx + 16], b0 = Bs[k][ty], b1 = Bs[k][ty + 16];
            acc00 = fma(a0, b0, acc00);
            acc01 = fma(a0, b1, acc01);
            acc10 = fma(a1, b0, acc10);
            acc11 = fma(a1, b1, acc11);
        }
        __syncthreads();
    }
    const int gi0 = i0 + tx, gi1 = i0 + tx + 16, gc0 = c0 + ty, gc1 = c0 + ty + 16;
    if (gi0 < nF && gc0 < nN) Wb[gi0 + (int64_t)gc0 * ldw] = acc00;
    if (gi0 < nF && gc1 < nN) Wb[gi0 + (int64_t)gc1 * ldw] = acc01;
    if (gi1 < nF && gc0 < nN) Wb[gi1 + (int64_t)gc0 * ldw] = acc10;
    if (gi1 < nF && gc1 < nN) Wb[gi1 + (int64_t)gc1 * ldw] = acc11;
}

// The Schur complement, bordered: G2 (np2 x np2, the layout k_chol_* expects) = [G_NN c_N; . .] - sum of k_gram<double>'s
// partials of [W z_F]'[W z_F]; identity on the padding diagonal; gdiag2 = diag(G_NN) = |a_j|^2 (the DGKS reference).
// npa2 > np2: augmented by the unit vectors like k_gram_reduce's matrix -- (R_C^-1)' comes out beside R_C.
__global__ __launch_bounds__(256) void k_schur_reduce(const double* __restrict__ Knew, int np, int nF, int nN, int np2,
                                                      const double* __restrict__ Wpart, int nsplit, double* __restrict__ G2,
                                                      double* __restrict__ gdiag2, int npa2) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gram_aug_entry(e - (int64_t)np2 * np2 + (int64_t)np2 * npa2, np2, npa2, nN, G2)) return;
    if (e >= (int64_t)np2 * np2) return;
    const int row = (int)(e % np2), col = (int)(e / np2);
    if ((row / kGramTile) > (col / kGramTile)) return;
    double s = 0.0;
    if (row <= nN && col <= nN) {
        s = Knew[(nF + row) + (int64_t)(nF + col) * np];
        double w = 0.0;
        for (int k = 0; k < nsplit; ++k) w += Wpart[(int64_t)k * np2 * np2 + e];
        if (row == col && row < nN) gdiag2[row] = s;
        s -= w;
    } else if (row == col) {
        s = 1.0;
    }
    G2[row + (int64_t)col * npa2] = s;
}

// y = z_F - W x_N (the coupling of the bordered extension's back substitution: x_F = R_F^-1 y); 16 rows per workgroup (128-byte
// segments of W's columns), sixteen column phases per row summed through LDS in a fixed order
__global__ __launch_bounds__(256) void k_wx(const double* __restrict__ Wb, int ldw, int nF, int nN, const double* __restrict__ z,
                                            const double* __restrict__ xN, const DevState* st, int nsel, double* __restrict__ y) {
    if (st->nsel != nsel) return;
    __shared__ double part[16][17];
    const int r = threadIdx.x & 15, g = threadIdx.x >> 4, i = blockIdx.x * 16 + r;
    double a0 = 0.0, a1 = 0.0;
    if (i < nF) {
        int j = g;
        for (; j + 16 < nN; j += 32) {
            a0 = fma(Wb[i + (int64_t)j * ldw], xN[j], a0);
            a1 = fma(Wb[i + (int64_t)(j + 16) * ldw], xN[j + 16], a1);
        }
        if (j < nN) a0 = fma(Wb[i + (int64_t)j * ldw], xN[j], a0);
    }
    part[g][r] = a0 + a1;
    __syncthreads();
    if (g == 0 && i < nF) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][r];
        y[i] = z[i] - t;
    }
}

// R_T's new blocks into the solver slot: W (rows of F, columns of N), R_C, z_N, the new atoms; count = nF + nN.
__global__ __launch_bounds__(256) void k_gram_export_b(const double* __restrict__ G2, int np2, int nF, int nN, const int* __restrict__ cols,
                                                       const double* __restrict__ Wb, int ldw, double* __restrict__ R, int kcap,
                                                       double* __restrict__ z, int* __restrict__ sel, DevState* st,
                                                       const double* __restrict__ Dfac) {
    if (st->done & STOP_REORTH) return;
    constexpr int NB = kCholNB;
    auto fac = [&](int row, int col) {
        return (row / NB == col / NB) ? Dfac[(int64_t)(row / NB) * NB * NB + (row % NB) + (int64_t)(col % NB) * NB] : G2[row + (int64_t)col * np2];
    };
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = nF + nN;
    if (e < (int64_t)n * nN) {
        const int row = (int)(e % n), j = (int)(e / n);
        double* dst = R + row + (int64_t)(nF + j) * kcap;
        if (row < nF) *dst = Wb[row + (int64_t)j * ldw];
        else if (row - nF <= j) *dst = fac(row - nF, j);
    }
    if (e < nN) {
        z[nF + e] = fac((int)e, nN);
        sel[nF + e] = cols[nF + e];
    }
    if (e == 0) {
        st->nsel = n;
        st->j = n;
        st->steps += 1;
    }
}

// r = b - A[:, cols] x for a LARGE support (residual!, src/matchingpursuit.jl:158-161): the columns are cut into chunks of
// 64, workgroup (row block, chunk) sums its chunk for 256 rows (8 columns of loads in flight), and k_residual_sum
// subtracts the chunk sums from b in a fixed order.
constexpr int kResChunk = 64;
template <typename TA>
__global__ __launch_bounds__(256) void k_residual_part(const TA* __restrict__ A, int64_t ld, int M, const int* __restrict__ idx,
                                                       const double* __restrict__ val, int n, double* __restrict__ part) {
    const int row = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    if (row >= M) return;
    const int t0 = ch * kResChunk, t1 = min(n, t0 + kResChunk);
    double a0 = 0.0, a1 = 0.0;
    int t = t0;
    for (; t + 8 <= t1; t += 8) {
        TA v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = A[(int64_t)idx[t + u] * ld + row];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            a0 = fma((double)v[u], val[t + u], a0);
            a1 = fma((double)v[u + 1], val[t + u + 1], a1);
        }
    }
    for (; t < t1; ++t) a0 = fma((double)A[(int64_t)idx[t] * ld + row], val[t], a0);
    part[(int64_t)ch * M + row] = a0 + a1;
}
// (+ the workgroup's share of |r|^2 into n2part[blockIdx.x]: the host adds the M/256 shares in order -- no norm kernel)
__global__ __launch_bounds__(256) void k_residual_sum(const double* __restrict__ part, int nch, int M, const double* __restrict__ b,
                                                      double* __restrict__ r, const DevState* st, double* __restrict__ n2part) {
    __shared__ double red[8];
    if (st->done & STOP_REORTH) return;
    const int row = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (row < M) {
        double s = 0.0;
        for (int c = 0; c < nch; ++c) s += part[(int64_t)c * M + row];
        v = b[row] - s;
        r[row] = v;
    }
    v = block_sum256(v * v, red);
    if (threadIdx.x == 0) n2part[blockIdx.x] = v;
}

}  // namespace csmp
