// csmp_swap.hpp -- OMP with replacement's exchange of ONE atom (update!(P::OMPR, x), src/twostage.jl:158-178: add_column! +
// remove_column! + ldiv!!) on the INVERSE GRAM MATRIX H = (A_S' A_S)^-1 of the support instead of on a QR factorisation.
//
// The reference keeps an UpdatableQR and pays a Givens sweep per removal; rounds 1-4 did the same on the device (k_tdel_apply: the
// rotations walk Q, 8 MB at k = 256, as a dependent chain -- 30-55 us per exchange, a third of an OMPR iteration beside the
// 159-us sweep).  H has no triangular shape to restore: an exchange is two rank-one corrections with NO dependent chain,
//     remove slot p:   H' = H - h_p h_p' / H_pp                      (rows / columns != p)
//     add atom a:      g = A_S' a,  u = H' g,  sigma = a'a - g'u     (the Schur complement: squared distance of a from span A_S')
//                      H'' = [H' + u u' / sigma, -u / sigma; -u' / sigma, 1 / sigma]     (a takes slot p)
// and the coefficients follow by vector updates (x'_i = x_i - H_ip x_p / H_pp;  x_a = (a'b - g'x') / sigma;  x''_i = x'_i - u_i x_a).
// The residual is recomputed from the coefficients (one pass over the k columns: 4 MB), not carried.  Conditioning: sigma is
// tested against a'a (the DGKS-style guard of the append chain); an exchange that fails it is NOT applied -- the host rebuilds
// the QR state from the current support and goes on with the rotation path (host/twostage.hpp, OmprJob).
// Launches per exchange: k_swap_dots -> k_swap_ufin -> k_swap_commit_res -> k_swap_rsum.  WHICH atoms are exchanged comes from
// device memory (meta, written by k_ompr_pick -- csmp_tinv.hpp -- or by the host): meta[0] = 1 exchange, anything else: every kernel
// of the chain returns at once; meta[1] = the atom that leaves, meta[2] = the atom that joins, meta[3] = the leaving atom's slot.
// The chain is queued right behind the sweep's pick, and the host reads the outcome of both in one landing.
#pragma once
#include "csmp_kernels.hpp"

namespace csmp {

constexpr int kSwapChunk = 16;  // columns of H per partial product (k_swap_ufin)

// H = T T' from the explicit inverse factor T = R^-1 (csmp_tinv.hpp layout: T[m * kcap + r], column m, rows r <= m): entry
// (i, j) = sum over m >= max(i, j) of T[m][i] T[m][j].  Once per solve.
__global__ __launch_bounds__(256) void k_swap_init(const double* __restrict__ T, int kcap, int n, double* __restrict__ H) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int i = e % n, j = e / n;
    double s = 0.0;
    for (int m = (i > j ? i : j); m < n; ++m) s = fma(T[(int64_t)m * kcap + i], T[(int64_t)m * kcap + j], s);
    H[(int64_t)j * kcap + i] = s;
}

// One wave per slot: out[j] = <a_sel[j], v> for the k slots (v = the joining atom's column, or b for the right-hand side
// c = A_S'b); with TV == TA two more waves: out[k] = <v, v> and out[k + 1] = <v, b>.  Float64 products of the promoted values.
template <typename TA, typename TV>
__global__ __launch_bounds__(256) void k_swap_dots(const TA* __restrict__ A, int64_t ld, int M, const int* __restrict__ sel, int k,
                                                   const TV* __restrict__ v, const double* __restrict__ b, int extras,
                                                   double* __restrict__ out, const int* __restrict__ meta) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= k + extras) return;
    if (meta) {  // (v: the joining atom's column)
        if (meta[0] != 1) return;
        if constexpr (sizeof(TA) == sizeof(TV)) v = reinterpret_cast<const TV*>(A + (int64_t)meta[2] * ld);
    }
    double acc = 0.0;
    if (j < k) {
        const TA* a = A + (int64_t)sel[j] * ld;
        if constexpr (sizeof(TA) == sizeof(TV)) {
            // both vectors are dictionary columns (16-byte aligned, ld padded to 16 bytes): 16 bytes per lane and load, eight loads of
            // each in flight (scalar 4-byte loads made this kernel 27 us for 4 MB)
            using VT = typename Vec<TA>::type;
            constexpr int VEC = Vec<TA>::n;
            const VT* a4 = reinterpret_cast<const VT*>(a);
            const VT* v4 = reinterpret_cast<const VT*>(v);
            const int nv = M / VEC;
            constexpr int NL = 16;  // loads of each vector in flight per lane (a 16-KiB column: one batch)
            for (int m0 = lane; m0 < nv; m0 += 64 * NL) {
                VT x[NL], y[NL];
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    const int m = m0 + u * 64;
                    x[u] = a4[m < nv ? m : nv - 1];
                    y[u] = v4[m < nv ? m : nv - 1];
                }
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    if (m0 + u * 64 < nv) {
                        acc = fma((double)x[u].x, (double)y[u].x, acc);
                        acc = fma((double)x[u].y, (double)y[u].y, acc);
                        if constexpr (VEC == 4) {
                            acc = fma((double)x[u].z, (double)y[u].z, acc);
                            acc = fma((double)x[u].w, (double)y[u].w, acc);
                        }
                    }
                }
            }
            for (int m = nv * VEC + lane; m < M; m += 64) acc = fma((double)a[m], (double)v[m], acc);  // (M not a multiple of the vector)
        } else {
            for (int m0 = lane; m0 < M; m0 += 64 * 8) {
                double x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = m0 + u * 64;
                    x[u] = (double)a[m < M ? m : M - 1];
                    y[u] = (double)v[m < M ? m : M - 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (m0 + u * 64 < M) acc = fma(x[u], y[u], acc);
            }
        }
    } else {
        for (int m = lane; m < M; m += 64) acc = fma((double)v[m], j == k ? (double)v[m] : b[m], acc);
    }
    acc = wave_xsum(acc);
    if (lane == 0) out[j] = acc;
}

// u = H' g, then everything that hangs on it, in ONE launch of nch workgroups.  Workgroup cb: the partial sums of u over the
// kSwapChunk columns of chunk cb for every row (H'_rj = H_rj - H_rp H_pj / H_pp, row p and column p excluded).  The workgroup that
// finishes LAST (a counter in device memory; it leaves it at zero for the next exchange) goes on alone: u from the partials,
// sigma and the guard, the coefficient updates, the slot's new owner, column p of the old H saved for the commit, and x emitted
// in index order (the list the next k_ompr_pick gathers with).
// info[0] = sigma, info[1] = H_pp (old), info[2] = 1 when the exchange was refused (nothing changed).
__global__ __launch_bounds__(256) void k_swap_ufin(const double* __restrict__ H, int ldh, int k, const int* __restrict__ meta,
                                                   const double* __restrict__ g, double* __restrict__ part, int nch,
                                                   unsigned* __restrict__ counter, double* __restrict__ u, double* __restrict__ x,
                                                   double* __restrict__ c, int* __restrict__ sel, double* __restrict__ hp,
                                                   double* __restrict__ info, int64_t* __restrict__ out_idx, double* __restrict__ out_val,
                                                   int64_t* __restrict__ out_nnz, double guard) {
    __shared__ double red[8];
    __shared__ double sh_xa;
    __shared__ int last;
    if (meta[0] != 1) return;
    const int tid = threadIdx.x, p = meta[3], anew = meta[2];
    const double hpp = H[(int64_t)p * ldh + p];
    {
        const int c0 = blockIdx.x * kSwapChunk;
        for (int r = tid; r < k; r += 256) {
            const double hrp = H[(int64_t)p * ldh + r] / hpp;  // (ONE division per row: H_rp / H_pp)
            double v[kSwapChunk], w[kSwapChunk];
#pragma unroll
            for (int q = 0; q < kSwapChunk; ++q) {
                const int jc = c0 + q < k ? c0 + q : k - 1;
                v[q] = H[(int64_t)jc * ldh + r];
                w[q] = H[(int64_t)jc * ldh + p];  // H_pj (symmetry): the same address in every lane
            }
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < kSwapChunk; ++q) {
                const int j = c0 + q;
                if (j < k && j != p) s = fma(fma(-hrp, w[q], v[q]), g[j], s);
            }
            part[(int64_t)blockIdx.x * k + r] = (r == p) ? 0.0 : s;
        }
    }
    __threadfence();  // (the partials are out before the count says so)
    __syncthreads();
    if (tid == 0) last = atomicAdd(counter, 1u) == (unsigned)(nch - 1);
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (tid == 0) *counter = 0u;
    const double gamma = g[k], beta = g[k + 1];
    const double xp = x[p];
    double gu = 0.0;
    for (int r = tid; r < k; r += 256) {
        double s = 0.0;
        for (int q = 0; q < nch; ++q) s += __hip_atomic_load(part + (int64_t)q * k + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u[r] = s;
        hp[r] = H[(int64_t)p * ldh + r];
        if (r != p) gu = fma(g[r], s, gu);
    }
    gu = block_sum256(gu, red);
    const double sigma = gamma - gu;
    // the guard: the joining atom must keep a real component outside the span of the others (sigma = its squared distance);
    // the append chain's DGKS test re-orthogonalises below 1/2, here there is no second pass: a small ratio hands the solve back
    const bool bad = !(sigma > guard * gamma) || !(hpp > 0.0);  // (guard: 1e-6; a test hook passes 2 -- sigma <= gamma always -- to walk the fallback)
    if (tid == 0) {
        info[0] = sigma;
        info[1] = hpp;
        info[2] = bad ? 1.0 : 0.0;
    }
    if (bad) return;
    __syncthreads();
    double gx = 0.0;
    for (int r = tid; r < k; r += 256) {
        if (r == p) continue;
        const double xr = x[r] - hp[r] * xp / hpp;  // (hp written by this thread above: same r)
        x[r] = xr;
        gx = fma(g[r], xr, gx);
    }
    gx = block_sum256(gx, red);
    if (tid == 0) sh_xa = (beta - gx) / sigma;
    __syncthreads();
    const double xa = sh_xa;
    for (int r = tid; r < k; r += 256) x[r] = (r == p) ? xa : x[r] - u[r] * xa;
    if (tid == 0) {
        c[p] = beta;
        sel[p] = anew;
        *out_nnz = k;
    }
    __syncthreads();
    extern __shared__ int ssel[];  // k atoms: the rank sort reads every one of them k times
    for (int t = tid; t < k; t += 256) ssel[t] = sel[t];
    __syncthreads();
    for (int t = tid; t < k; t += 256) {  // rank sort by atom index (k_emit_sorted's)
        const int me = ssel[t];
        int rank = 0;
        for (int q = 0; q < k; ++q) rank += (ssel[q] < me);
        out_idx[rank] = me;
        out_val[rank] = x[t];
    }
}

// One launch for the two things that follow and do not depend on each other.  Workgroups [0, ncommit): the new H, entry by entry
// (no dependent chain: u, the saved column hp and sigma are complete).  The others: k_residual_part's share (row block, chunk) of
// A_S x for the new coefficients.
template <typename TA>
__global__ __launch_bounds__(256) void k_swap_commit_res(double* __restrict__ H, int ldh, int k, const int* __restrict__ meta,
                                                         const double* __restrict__ u, const double* __restrict__ hp,
                                                         const double* __restrict__ info, int ncommit, const TA* __restrict__ A,
                                                         int64_t ld, int M, const int* __restrict__ sel, const double* __restrict__ x,
                                                         int nrb, double* __restrict__ rpart) {
    if (meta[0] != 1 || info[2] != 0.0) return;
    if ((int)blockIdx.x < ncommit) {
        const int p = meta[3];
        const int e = blockIdx.x * 256 + threadIdx.x;
        if (e >= k * k) return;
        const int i = e % k, j = e / k;
        const double sigma = info[0], hpp = info[1];
        double v;
        if (i == p && j == p) v = 1.0 / sigma;
        else if (i == p) v = -u[j] / sigma;
        else if (j == p) v = -u[i] / sigma;
        else v = H[(int64_t)j * ldh + i] - hp[i] * hp[j] / hpp + u[i] * u[j] / sigma;
        H[(int64_t)j * ldh + i] = v;
        return;
    }
    const int b = (int)blockIdx.x - ncommit;
    const int row = (b % nrb) * 256 + threadIdx.x, ch = b / nrb;
    if (row >= M) return;
    const int t0 = ch * kResChunk, t1 = min(k, t0 + kResChunk);
    double a0 = 0.0, a1 = 0.0;
    int t = t0;
    for (; t + 8 <= t1; t += 8) {
        TA v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = A[(int64_t)sel[t + q] * ld + row];
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            a0 = fma((double)v[q], x[t + q], a0);
            a1 = fma((double)v[q + 1], x[t + q + 1], a1);
        }
    }
    for (; t < t1; ++t) a0 = fma((double)A[(int64_t)sel[t] * ld + row], x[t], a0);
    rpart[(int64_t)ch * M + row] = a0 + a1;
}

// r = b - (chunk sums of the residual shares), and the workgroup's share of |r|^2 (the host adds the shares in order)
__global__ __launch_bounds__(256) void k_swap_rsum(const double* __restrict__ part, int nch, int M, const double* __restrict__ b,
                                                   double* __restrict__ r, const double* __restrict__ info, double* __restrict__ n2part,
                                                   const int* __restrict__ meta) {
    __shared__ double red[8];
    if (meta[0] != 1 || info[2] != 0.0) return;
    const int row = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (row < M) {
        double s = 0.0;
        for (int q = 0; q < nch; ++q) s += part[(int64_t)q * M + row];
        v = b[row] - s;
        r[row] = v;
    }
    v = block_sum256(v * v, red);
    if (threadIdx.x == 0) n2part[blockIdx.x] = v;
}

}  // namespace csmp
