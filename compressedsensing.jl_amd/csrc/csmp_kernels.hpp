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
#include <stdint.h>

namespace csmp {

constexpr int kWave = 64;
constexpr int kSweepThreads = 256;  // 4 waves
constexpr int kCPW = 4;             // dictionary columns a wave reduces together
constexpr int kSlabRows = 64;       // rows of Q owned by one QR workgroup
constexpr int kQrThreads = 256;

enum : int { STOP_EPS = 1, STOP_STAG = 2, STOP_FULL = 4 };

// Control block of one solve, in device memory.  Written only by single-workgroup control
// kernels (k_select / k_ctl / k_init) and by workgroup 0 of k_qr3 / k_mp_update at their very
// end (fields no workgroup of the same launch reads), so no launch races with itself.
struct DevState {
    int nsel;       // columns in the QR / atoms in the support (MP: steps taken)
    int j;          // nsel frozen for the current step's QR kernels
    int cand;       // atom chosen for the current step
    int go;         // 1: the current step's append kernels run
    int done;       // STOP_* bits
    int steps;      // update! calls that changed x
    int pad0, pad1;
    double rnorm2;  // ||r||^2 seen by the last sweep prologue
    double cval;    // signed <a_cand, r> (MP coefficient, src/matchingpursuit.jl:29)
};

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
// Sweep: c = A' r (Float64), fused |.| + arg-max partials.  One wave owns kCPW whole columns at a
// time (16 KiB contiguous each at M = 4096 f32), lanes stride the rows with 16-B loads, r lives
// in LDS.  Grid-stride over column groups; one (max |c|, first index) pair per workgroup.
//   U     row chunks (64 lanes x 16 B) whose loads are issued together: U*kCPW loads in flight
//   FULL  Mv is a multiple of U*64*VEC rows (no row predicate in the hot loop)
//   NT    non-temporal dictionary loads (A is streamed once per sweep and exceeds every cache)
//   TACC  double = product; float exists only as a bandwidth probe (csmp_bench_sweep)
// dynamic LDS: r image (nchunk*64*VEC doubles) + 64 doubles of reduction scratch
template <typename TA, typename TACC, int U, bool FULL, bool NT>
__global__ __launch_bounds__(kSweepThreads) void k_sweep(
    const TA* __restrict__ A, int64_t ld, int Mv, int64_t N, const double* __restrict__ r,
    double* __restrict__ cvec, double* __restrict__ pval, int* __restrict__ pidx, DevState* st,
    double eps, int check_eps, int skipmask) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;  // rows per chunk
    constexpr int NW = kSweepThreads / kWave;
    extern __shared__ __attribute__((aligned(16))) double lds[];

    if (st->done & skipmask) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    double* red = lds + Mlds;                           // [4]
    double* redv = red + 8;                             // [16]
    int* redi = reinterpret_cast<int*>(redv + 4 * NW);  // [16]

    // prologue: r -> LDS (zero beyond Mv), ||r||^2 in a fixed order (identical in every workgroup)
    double n2 = 0.0;
    for (int m = tid; m < Mlds; m += kSweepThreads) {
        const double v = (m < Mv) ? r[m] : 0.0;
        lds[r_slot<VEC>(m)] = v;
        n2 = fma(v, v, n2);
    }
    n2 = block_sum256(n2, red);
    if (blockIdx.x == 0 && tid == 0) st->rnorm2 = n2;
    if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break  (:79,:132)
        if (blockIdx.x == 0 && tid == 0) st->done |= STOP_EPS;
        return;
    }

    const f64x2* rs = reinterpret_cast<const f64x2*>(lds);
    double bestv = -1.0;
    int besti = 0x7fffffff;
    const int64_t stride = (int64_t)gridDim.x * NW * kCPW;
    for (int64_t cg = ((int64_t)blockIdx.x * NW + wave) * kCPW; cg < N; cg += stride) {
        const VT* p[kCPW];
#pragma unroll
        for (int c = 0; c < kCPW; ++c) {
            const int64_t col = (cg + c < N) ? cg + c : N - 1;
            p[c] = reinterpret_cast<const VT*>(A + col * ld) + lane;
        }
        TACC acc[kCPW];
#pragma unroll
        for (int c = 0; c < kCPW; ++c) acc[c] = (TACC)0;

        for (int t = 0; t < nchunk; t += U) {
            VT a[U][kCPW];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < kCPW; ++c) {
                    if constexpr (FULL) {
                        if constexpr (NT)
                            a[u][c] = __builtin_nontemporal_load(p[c] + (t + u) * kWave);
                        else
                            a[u][c] = p[c][(t + u) * kWave];
                    } else {
                        const int row = (t + u) * ROWS + lane * VEC;
                        a[u][c] = (VT)0;
                        if (row < Mv) a[u][c] = p[c][(t + u) * kWave];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (VEC == 4) {
                    const f64x2 r01 = rs[((t + u) * 2 + 0) * kWave + lane];
                    const f64x2 r23 = rs[((t + u) * 2 + 1) * kWave + lane];
#pragma unroll
                    for (int c = 0; c < kCPW; ++c) {
                        acc[c] = fma((TACC)a[u][c].x, (TACC)r01.x, acc[c]);
                        acc[c] = fma((TACC)a[u][c].y, (TACC)r01.y, acc[c]);
                        acc[c] = fma((TACC)a[u][c].z, (TACC)r23.x, acc[c]);
                        acc[c] = fma((TACC)a[u][c].w, (TACC)r23.y, acc[c]);
                    }
                } else {
                    const f64x2 r01 = rs[(t + u) * kWave + lane];
#pragma unroll
                    for (int c = 0; c < kCPW; ++c) {
                        acc[c] = fma((TACC)a[u][c].x, (TACC)r01.x, acc[c]);
                        acc[c] = fma((TACC)a[u][c].y, (TACC)r01.y, acc[c]);
                    }
                }
            }
        }
        // transposing butterfly: 4 accumulators x 64 lanes -> 16-lane group g holds column cg+g
        double s0, s1;
        {
            const bool hi = lane & 32;
            const double k0 = hi ? (double)acc[2] : (double)acc[0], k1 = hi ? (double)acc[3] : (double)acc[1];
            const double g0 = hi ? (double)acc[0] : (double)acc[2], g1 = hi ? (double)acc[1] : (double)acc[3];
            s0 = k0 + shx(g0, 32);
            s1 = k1 + shx(g1, 32);
        }
        {
            const bool hi = lane & 16;
            const double k = hi ? s1 : s0, g = hi ? s0 : s1;
            s0 = k + shx(g, 16);
        }
        s0 += shx(s0, 8);
        s0 += shx(s0, 4);
        s0 += shx(s0, 2);
        s0 += shx(s0, 1);
        const int64_t col = cg + (lane >> 4);
        if (col < N) {
            if ((lane & 15) == 0) cvec[col] = s0;
            const double av = fabs(s0);
            if (av > bestv) {  // columns arrive in increasing order: '>' keeps the first maximum
                bestv = av;
                besti = (int)col;
            }
        }
    }
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
        pval[blockIdx.x] = bv;
        pidx[blockIdx.x] = bi;
    }
}
inline size_t sweep_lds_bytes(int Mv, int vec) {
    const int rows = kWave * vec;
    const int nchunk = (Mv + rows - 1) / rows;
    return (size_t)(nchunk * rows + 8 + 16 + 8) * sizeof(double);
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

// Control kernel for one entry of a candidate list (GOMP: src/util.jl:129-134 walks the l best
// atoms and skips those already in the support, util.jl:119).
__global__ __launch_bounds__(256) void k_ctl(const int* __restrict__ cands, int which, const int* __restrict__ ncands,
                                             const int* __restrict__ sel, DevState* st, int M, int kcap, int skipmask) {
    __shared__ int found;
    const int tid = threadIdx.x;
    if (st->done & skipmask) {
        if (tid == 0) st->go = 0;
        return;
    }
    const int nsel = st->nsel;
    const int valid = which < *ncands;
    const int cand = valid ? cands[which] : -1;
    if (tid == 0) found = 0;
    __syncthreads();
    for (int q = tid; q < nsel; q += 256)
        if (sel[q] == cand) found = 1;
    __syncthreads();
    if (tid == 0) {
        int go = valid && !found;
        if (nsel >= M || nsel >= kcap) {  // :117 guard / QR capacity
            st->done |= STOP_FULL;
            go = 0;
        }
        st->cand = cand;
        st->j = nsel;
        st->go = go;
    }
}

// ---------------------------------------------------------------------------------------------
// On-device QR append (classical Gram-Schmidt with one re-orthogonalisation, "CGS2", in its
// two-reduction form): Q is M x kcap Float64 column-major, split into slabs of 64 rows, one
// workgroup per slab.  A grid-wide sum is a kernel boundary (cheaper on MI355X than an in-kernel
// grid barrier): partial sums are written per slab and re-summed in a fixed order by every
// workgroup of the next kernel, so results are bitwise reproducible.
//   k_qr1:  a = A[:,cand];                 P1[g] = Q_g' a_g
//   k_qr2:  w1 = sum_g P1[g];  v = a - Q w1;  P2[g] = Q_g' v_g,  |v_g|^2,  v_g' r_g
//   k_qr3:  w2 = sum_g P2[g];  rho^2 = |v|^2 - |w2|^2;  q = (v - Q w2)/rho;  z_j = v'r/rho;
//           r -= q z_j;  R[:,j] = [w1 + w2; rho];  support += cand
// (r is orthogonal to Q, so q'b == q'r up to rounding; z accumulates Q'b for the final solve.)

// partial Q_g' x for this workgroup's slab: wave w covers rows 16w..16w+15, lane <-> column
__device__ __forceinline__ void slab_qt_x(const double* __restrict__ Q, int64_t ldq, int g, int j,
                                          const double* xs /*LDS, 64*/, double* part /*LDS 4*jpad*/, int jpad,
                                          double* __restrict__ out /*global, this slab's row of partials*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double xr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xr[i] = xs[wave * 16 + i];
    for (int c = lane; c < j; c += kWave) {
        const f64x2* q = reinterpret_cast<const f64x2*>(Q + (int64_t)c * ldq + g * kSlabRows + wave * 16);
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f64x2 v = q[i];
            acc = fma(v.x, xr[2 * i], acc);
            acc = fma(v.y, xr[2 * i + 1], acc);
        }
        part[wave * jpad + c] = acc;
    }
    __syncthreads();
    for (int c = tid; c < j; c += kQrThreads)
        out[c] = (part[c] + part[jpad + c]) + (part[2 * jpad + c] + part[3 * jpad + c]);
    __syncthreads();
}

// x_g -= Q_g w for this slab: lane <-> row, wave w takes columns c = w (mod 4); result in xs (LDS)
__device__ __forceinline__ void slab_x_minus_qw(const double* __restrict__ Q, int64_t ldq, int g, int j,
                                                const double* ws /*LDS, j*/, double* xs /*LDS 64, in/out*/,
                                                double* tmp /*LDS 4*64*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* q = Q + g * kSlabRows + lane;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = wave;
    for (; c + 12 < j; c += 16) {
        const double q0 = q[(int64_t)c * ldq], q1 = q[(int64_t)(c + 4) * ldq];
        const double q2 = q[(int64_t)(c + 8) * ldq], q3 = q[(int64_t)(c + 12) * ldq];
        a0 = fma(q0, ws[c], a0);
        a1 = fma(q1, ws[c + 4], a1);
        a2 = fma(q2, ws[c + 8], a2);
        a3 = fma(q3, ws[c + 12], a3);
    }
    for (; c < j; c += 4) a0 = fma(q[(int64_t)c * ldq], ws[c], a0);
    tmp[wave * kSlabRows + lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < kSlabRows)
        xs[tid] -= (tmp[tid] + tmp[kSlabRows + tid]) + (tmp[2 * kSlabRows + tid] + tmp[3 * kSlabRows + tid]);
    __syncthreads();
}

// LDS carve shared by the three QR kernels (dynamic): part[4*jpad] | ws[jpad] | xs[64] | tmp[256]
__device__ __forceinline__ void qr_carve(double* base, int jpad, double*& part, double*& ws, double*& xs, double*& tmp) {
    part = base;
    ws = part + 4 * jpad;
    xs = ws + jpad;
    tmp = xs + kSlabRows;
}
inline size_t qr_lds_bytes(int kcap) {
    const int jpad = ((kcap + 63) / 64) * 64 + 2;
    return (size_t)(5 * jpad + kSlabRows + 4 * kSlabRows + 8) * sizeof(double);
}
inline int qr_jpad(int kcap) { return ((kcap + 63) / 64) * 64 + 2; }

template <typename TA>
__global__ __launch_bounds__(kQrThreads) void k_qr1(const TA* __restrict__ A, int64_t ld, int M,
                                                    const double* __restrict__ Q, int64_t ldq, const DevState* st,
                                                    double* __restrict__ avec, double* __restrict__ P1, int kcap, int jpad) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!st->go) return;
    double *part, *ws, *xs, *tmp;
    qr_carve(lds, jpad, part, ws, xs, tmp);
    const int tid = threadIdx.x, g = blockIdx.x, j = st->j;
    if (tid < kSlabRows) {
        const int row = g * kSlabRows + tid;
        const double a = (row < M) ? (double)A[(int64_t)st->cand * ld + row] : 0.0;
        xs[tid] = a;
        avec[row] = a;
    }
    __syncthreads();
    slab_qt_x(Q, ldq, g, j, xs, part, jpad, P1 + (int64_t)g * kcap);
}

__global__ __launch_bounds__(kQrThreads) void k_qr2(const double* __restrict__ Q, int64_t ldq, const DevState* st,
                                                    const double* __restrict__ avec, const double* __restrict__ r,
                                                    const double* __restrict__ P1, int G, double* __restrict__ W1,
                                                    double* __restrict__ vvec, double* __restrict__ P2,
                                                    double* __restrict__ P2s, int kcap, int jpad) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!st->go) return;
    double *part, *ws, *xs, *tmp;
    qr_carve(lds, jpad, part, ws, xs, tmp);
    const int tid = threadIdx.x, g = blockIdx.x, j = st->j;
    for (int c = tid; c < j; c += kQrThreads) {  // w1 = sum over slabs, fixed order
        double s = 0.0;
        for (int gg = 0; gg < G; ++gg) s += P1[(int64_t)gg * kcap + c];
        ws[c] = s;
        if (g == 0) W1[c] = s;
    }
    if (tid < kSlabRows) xs[tid] = avec[g * kSlabRows + tid];
    __syncthreads();
    slab_x_minus_qw(Q, ldq, g, j, ws, xs, tmp);  // v_g
    if (tid < kSlabRows) {
        const double v = xs[tid];
        vvec[g * kSlabRows + tid] = v;
        double n2 = v * v, vr = v * r[g * kSlabRows + tid];
        for (int s = 32; s >= 1; s >>= 1) {
            n2 += shx(n2, s);
            vr += shx(vr, s);
        }
        if (tid == 0) {
            P2s[2 * g] = n2;
            P2s[2 * g + 1] = vr;
        }
    }
    slab_qt_x(Q, ldq, g, j, xs, part, jpad, P2 + (int64_t)g * kcap);
}

__global__ __launch_bounds__(kQrThreads) void k_qr3(double* __restrict__ Q, int64_t ldq, DevState* st,
                                                    const double* __restrict__ vvec, double* __restrict__ r,
                                                    const double* __restrict__ P2, const double* __restrict__ P2s,
                                                    int G, const double* __restrict__ W1, double* __restrict__ R,
                                                    double* __restrict__ z, int* __restrict__ sel, int kcap, int jpad) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!st->go) return;
    double *part, *ws, *xs, *tmp;
    qr_carve(lds, jpad, part, ws, xs, tmp);
    double* sc = tmp + 4 * kSlabRows;  // 8 spare doubles behind tmp
    const int tid = threadIdx.x, g = blockIdx.x, j = st->j;
    double w2sq = 0.0;
    for (int c = tid; c < j; c += kQrThreads) {
        double s = 0.0;
        for (int gg = 0; gg < G; ++gg) s += P2[(int64_t)gg * kcap + c];
        ws[c] = s;
        w2sq = fma(s, s, w2sq);
    }
    if (tid < kSlabRows) xs[tid] = vvec[g * kSlabRows + tid];
    w2sq = block_sum256(w2sq, sc);  // |w2|^2, fixed order
    double n2 = 0.0, vr = 0.0;
    for (int gg = tid; gg < G; gg += kQrThreads) {
        n2 += P2s[2 * gg];
        vr += P2s[2 * gg + 1];
    }
    n2 = block_sum256(n2, sc);
    vr = block_sum256(vr, sc);
    __syncthreads();
    if (tid == 0) {
        const double rho2 = n2 - w2sq;
        const double rho_ = (rho2 > 0.0) ? sqrt(rho2) : 0.0;
        sc[0] = rho_;
        sc[1] = (rho_ > 0.0) ? vr / rho_ : 0.0;  // z_j = q_j' r
    }
    __syncthreads();
    const double rho = sc[0], zj = sc[1];
    slab_x_minus_qw(Q, ldq, g, j, ws, xs, tmp);  // v_g - Q_g w2
    if (tid < kSlabRows) {
        const int row = g * kSlabRows + tid;
        const double q = (rho > 0.0) ? xs[tid] / rho : 0.0;
        Q[(int64_t)j * ldq + row] = q;
        r[row] = fma(-q, zj, r[row]);
    }
    if (g == 0) {
        for (int c = tid; c < j; c += kQrThreads) R[(int64_t)j * kcap + c] = W1[c] + ws[c];
        if (tid == 0) {
            R[(int64_t)j * kcap + j] = (rho > 0.0) ? rho : 1.0;  // degenerate column: coefficient 0
            z[j] = zj;
            sel[j] = st->cand;
            st->nsel = j + 1;
            st->steps += 1;
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
                                                int64_t* __restrict__ out_order, int outcap) {
    extern __shared__ __attribute__((aligned(16))) double y[];  // kcap + 2
    double& ci = y[kcap];
    const int tid = threadIdx.x, j = st->nsel;
    for (int t = tid; t < j; t += 256) y[t] = z[t];
    __syncthreads();
    for (int i = j - 1; i >= 0; --i) {
        if (tid == 0) {
            ci = y[i] / R[(int64_t)i * kcap + i];
            y[i] = ci;
        }
        __syncthreads();
        const double c = ci;
        for (int t = tid; t < i; t += 256) y[t] = fma(-R[(int64_t)i * kcap + t], c, y[t]);
        __syncthreads();
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
    if (tid == 0) *out_nnz = j;
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
        st->rnorm2 = 0.0;
        st->cval = 0.0;
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
