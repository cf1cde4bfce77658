// csmp_batched.hpp -- batched OMP: many signals sharing one dictionary (BASELINE configs 3/4).
//
// Not in the reference (it solves one b at a time, src/matchingpursuit.jl:73-82); this is the loop
// `[omp(A, B[:,s], eps, k) for s in axes(B,2)]` restructured so that the dominant cost, the
// residual-correlation sweep, becomes ONE dense GEMM  C = A' [r_1 ... r_B]  on the matrix cores:
//
//   k_b_screen256p (csmp_screen.hip)  bf16 MFMA, f32 accumulate, 256 atoms x 256 signals per workgroup, with a fused
//               epilogue that keeps the 4 largest |c| per (signal, 128-atom tile): the N x B product (256 MiB at C3)
//               is never written.  Option CSMP_OPT_BATCH_SCREEN = 1: the same kernel on int8 images (k_b_convert_i8: one
//               step for the dictionary; k_b_init / k_b_append: one step per residual) with v_mfma_i32_16x16x64_i8.
//   k_b_pick    one workgroup per signal (argmaxinner!, src/matchingpursuit.jl:181-185, + update!'s guards :63,66):
//               takes the tile candidates whose screened value could still be the exact maximum (the WINDOW), RESCORES
//               them exactly (f32/f64 master dictionary, Float64 products and sums against the Float64 residual), picks
//               the arg-max by the exact value (first index on ties) and certifies it against everything not rescored.
//   k_b_append  one workgroup per signal (addindex! + residual!, src/util.jl:118-126, src/matchingpursuit.jl:152-161):
//               appends the atom to the signal's factorisation and updates the residual and its bf16 image.
//
// The bf16 product only SCREENS; every value that decides or enters the result is Float64 on the
// exactly promoted dictionary, so supports match the Float64 oracle.  The certificate (exact best > an upper bound on
// the exact value of every atom that was not rescored) guards the screen; a signal that ever fails it (or whose
// support becomes ill-conditioned) is re-solved by the exact single-signal path.
//
// Per-signal factorisation: no Q is stored (it would be M x k Float64 per signal and read twice a
// step).  With A_S = Q R:  w = Q'a = R^-T (A_S' a),  v = a - Q w = a - A_S (R^-1 w); the inverse
// T = R^-1 is kept explicitly (and its transpose), so both "triangular solves" are coalesced
// mat-vecs and the only large traffic is two streams over the support's f32 columns of A -- or ONE, when the
// caller has asked for the resident Gram matrix G = A'A (csmp_set_option CSMP_OPT_BATCH_GRAM): A_S'a is then a
// gather of j numbers.
#pragma once
#include "csmp_kernels.hpp"
#include "csmp_screen.hpp"

namespace csmp {

using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// workgroups of the per-signal kernels a CU holds (register budget 512 / (4 x this) per lane): rows per thread = 4 NI, and
// an f64 dictionary doubles the registers of every column in flight
template <typename TA, int NI> constexpr int b_wgs() { return sizeof(TA) == 4 ? (NI <= 4 ? 4 : 2) : (NI <= 2 ? 4 : NI <= 4 ? 2 : 1); }
constexpr int kWinMax = 128;      // capacity of the rescoring window (candidates rescored per signal and step)

struct BState {
    int nsel, done, uncertain, illcond;
    double rnorm2;
    // the first failed certificate of the signal (diagnostics: csmp_batch_stats' callers see only the counts)
    int unc_step, unc_nall;
    double unc_cb, unc_best, unc_s1;
    float rstep;  // int8 screen (CSMP_OPT_BATCH_SCREEN = 1): the quantisation step of the signal's residual image
    int pad_;
};
// hand-off k_b_pick -> k_b_append, one per signal
struct BPick {
    int atom;       // the step's atom; -1: nothing to append (stopped, stagnated)
    int nwin;       // candidates rescored (statistics)
    double cexact;  // <a_atom, r>, Float64
};

// dictionary (f32/f64, column-major M x N) -> bf16 [Npad][Mk], zero padded (RNE: v_cvt_pk_bf16_f32)
template <typename TA>
__global__ __launch_bounds__(256) void k_b_convert(const TA* __restrict__ A, int64_t ld, int M, int64_t N,
                                                   __bf16* __restrict__ out, int Mk, int64_t Npad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kc = Mk / 8;
    const int64_t n = idx / kc;
    const int c = (int)(idx % kc);
    if (n >= Npad) return;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = c * 8 + e;
        v[e] = (__bf16)((n < N && k < M) ? (float)A[n * ld + k] : 0.0f);
    }
    *reinterpret_cast<bf16x8*>(out + n * Mk + c * 8) = v;
}

// dictionary (f32/f64, column-major M x N) -> binary16 [Npad][Mk] of scale * A, zero padded.  scale is a power of two that puts
// max|A| in [2^14, 2^15): exact, and nothing the narrow exponent range of binary16 could flush (entries below 2^-28 max|A|
// become subnormal: an ABSOLUTE error of at most 2^-39 max|A|, part of the certificate's bound).
using f16x8v = __attribute__((ext_vector_type(8))) _Float16;
template <typename TA>
__global__ __launch_bounds__(256) void k_b_convert_f16(const TA* __restrict__ A, int64_t ld, int M, int64_t N,
                                                       _Float16* __restrict__ out, int Mk, int64_t Npad, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kc = Mk / 8;
    const int64_t n = idx / kc;
    const int c = (int)(idx % kc);
    if (n >= Npad) return;
    f16x8v v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = c * 8 + e;
        v[e] = (_Float16)((n < N && k < M) ? (float)A[n * ld + k] * scale : 0.0f);
    }
    *reinterpret_cast<f16x8v*>(out + n * Mk + c * 8) = v;
}
// power of two that scales a largest magnitude vmax into [2^14, 2^15) (1 for vmax == 0).  The exponent is clamped to +-100 so that
// the scale AND its reciprocal stay finite normal floats whatever vmax is: a residual of 1e-35 (a Float32 subnormal, or a Float64
// far below FLT_MIN) used to give 2^127+ = inf and an image of NaNs; an infinite vmax overflowed the integer exponent.  A
// clamped image underflows to zeros instead, which the certificate charges as a flush to zero (the signal is then re-solved exactly).
__host__ __device__ __forceinline__ float f16_scale(float vmax) {
    if (!(vmax > 0.0f) || !(vmax <= 3.4028234e38f)) return 1.0f;
    int e = 14 - ilogbf(vmax);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.0f, e);
}

// dictionary -> int8 [Npad][Mk8] with ONE step for the whole dictionary (astep = max|A| / 127; inv = 1 / astep), zero padded:
// the int8 screen's operand (k_b_screen256p<true>).  A common step makes the absolute rounding error of every entry the
// same, which is what the certificate's absolute term assumes.
template <typename TA>
__global__ __launch_bounds__(256) void k_b_convert_i8(const TA* __restrict__ A, int64_t ld, int M, int64_t N,
                                                      signed char* __restrict__ out, int Mk8, int64_t Npad, float inv) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kc = Mk8 / 16;
    const int64_t n = idx / kc;
    const int c = (int)(idx % kc);
    if (n >= Npad) return;
    using c16 = signed char __attribute__((ext_vector_type(16)));
    c16 v;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = c * 16 + e;
        const float a = (n < N && k < M) ? (float)A[n * ld + k] : 0.0f;
        v[e] = (signed char)max(-127, min(127, __float2int_rn(a * inv)));
    }
    *reinterpret_cast<c16*>(out + n * Mk8 + c * 16) = v;
}

// largest |v| over the 256 threads of a workgroup (every thread gets it); scratch: 4 floats
__device__ __forceinline__ float block_absmax256(float v, float* scratch4) {
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, kWave));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scratch4[0], scratch4[1]), fmaxf(scratch4[2], scratch4[3]));
}
// the step of a residual's int8 image and what multiplies the integer dot products of the screen: astep * rstep, nudged up so
// that the candidate values stay upper bounds
__device__ __forceinline__ float i8_step(float rmax) { return rmax > 0.0f ? rmax * (1.0f / 127.0f) : 1.0f; }

// ---------------------------------------------------------------------------------------------
// helpers of the per-signal kernels: thread t owns the 4-row groups g = t + 256 i, i < NI
template <typename TA>
__device__ __forceinline__ void load4(const TA* p, int avail, double (&o)[4]) {
    if constexpr (sizeof(TA) == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        const f64x2 v0 = reinterpret_cast<const f64x2*>(p)[0];
        const f64x2 v1 = (avail >= 4) ? reinterpret_cast<const f64x2*>(p)[1] : (f64x2)0.0;
        o[0] = v0.x; o[1] = v0.y; o[2] = v1.x; o[3] = v1.y;
    }
}
// 4 consecutive rows kept in the dictionary's own type (f32: one 16-B register quad) until used
template <typename TA> struct Raw4;
template <> struct Raw4<float> {
    f32x4 v;
    // avail = rows left in the column from p (a multiple of 4 for f32 storage)
    __device__ __forceinline__ void load(const float* p, int avail) { (void)avail; v = *reinterpret_cast<const f32x4*>(p); }
    __device__ __forceinline__ void zero() { v = (f32x4)0.0f; }
    __device__ __forceinline__ double get(int e) const { return (double)v[e]; }
};
template <> struct Raw4<double> {
    f64x2 v0, v1;
    // f64 columns are padded to 2 rows only: the second pair may lie beyond the column
    __device__ __forceinline__ void load(const double* p, int avail) {
        v0 = reinterpret_cast<const f64x2*>(p)[0];
        v1 = (avail >= 4) ? reinterpret_cast<const f64x2*>(p)[1] : (f64x2)0.0;
    }
    __device__ __forceinline__ void zero() { v0 = v1 = (f64x2)0.0; }
    __device__ __forceinline__ double get(int e) const { return e < 2 ? v0[e] : v1[e - 2]; }
};

// b -> r (Float64) and its bf16 image; state reset.  One workgroup per signal (pad signals: zeros).
template <typename TB>
__global__ __launch_bounds__(256) void k_b_init(const TB* __restrict__ Bsig, int64_t ldB, int M, int nsig,
                                                double* __restrict__ r_all, double* __restrict__ b_all, int Mr,
                                                __bf16* __restrict__ rb_all, int Mk, BState* __restrict__ bs,
                                                signed char* __restrict__ r8_all, int Mk8, float* __restrict__ sigscale, float astep, int img) {
    __shared__ float smax[4];
    const int s = blockIdx.x;
    float amax = 0.0f;
    for (int m = threadIdx.x; m < Mk || m < Mr; m += 256) {
        const double v = (s < nsig && m < M) ? (double)Bsig[(int64_t)s * ldB + m] : 0.0;
        if (m < Mr) {
            r_all[(int64_t)s * Mr + m] = v;
            b_all[(int64_t)s * Mr + m] = v;
        }
        if (m < Mk && img == kOpBf16) rb_all[(int64_t)s * Mk + m] = (__bf16)(float)v;
        amax = fmaxf(amax, fabsf((float)v));
    }
    float rstep = 0.0f;
    if (img == kOpF16) {  // binary16 image of the signal under its own power-of-two scale (astep = 1 / the dictionary's scale)
        const float sc = f16_scale(block_absmax256(amax, smax));
        _Float16* rh = reinterpret_cast<_Float16*>(rb_all) + (int64_t)s * Mk;
        for (int m = threadIdx.x; m < Mk; m += 256) {
            const double v = (s < nsig && m < M) ? (double)Bsig[(int64_t)s * ldB + m] : 0.0;
            rh[m] = (_Float16)(float)(v * (double)sc);
        }
        if (threadIdx.x == 0) sigscale[s] = astep / sc;
    }
    if (img == kOpI8) {  // int8 image of the signal, one step per signal
        rstep = i8_step(block_absmax256(amax, smax));
        const float inv = 1.0f / rstep;
        for (int m = threadIdx.x; m < Mk8; m += 256) {
            const float v = (s < nsig && m < M) ? (float)(double)Bsig[(int64_t)s * ldB + m] : 0.0f;
            r8_all[(int64_t)s * Mk8 + m] = (signed char)max(-127, min(127, __float2int_rn(v * inv)));
        }
        if (threadIdx.x == 0) sigscale[s] = astep * rstep * (1.0f + 0x1p-20f);
    }
    if (threadIdx.x == 0) {
        bs[s].rstep = rstep;
        bs[s].nsel = 0;
        bs[s].done = (s < nsig) ? 0 : STOP_FULL;  // padding signals never run
        bs[s].uncertain = 0;
        bs[s].illcond = 0;
        bs[s].rnorm2 = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// Selection of one OMP step of one signal, by one workgroup.
//
// The screening launch left, per (signal, 128-atom tile), its kTileCand largest screened values s_n ~ |<a_n, r>| (bf16
// operands, f32 accumulation).  With an error bound  | |<a_n, r>| - s_n | <= d(s_n) = cert_abs |r| + cert_rel s_n :
//   * window: every candidate whose upper bound s + d(s) reaches the lower bound of the largest screened value,
//     s_1 - d(s_1), could be the exact arg-max and is rescored exactly; the others cannot (their exact value is below
//     the exact value of the top candidate);
//   * certificate: every atom that was NOT rescored -- a candidate outside the window, or an atom hidden behind the
//     last kept candidate of its tile -- has an exact value <= cb = the largest upper bound among them; the exact best
//     of the window must exceed cb, else the signal is flagged `uncertain` (and re-solved exactly by the caller).
// A passed certificate therefore proves the pick, given the bound.  cert_abs / cert_rel: csmp.hip (statistical model of
// independent roundings + the coherent "whole operand scaled" term, or the deterministic bound).
// A window larger than kwin entries is a failed certificate.
// <A[:, c], r> in Float64 by ONE wave: the sweep's inner loop (csmp_kernels.hpp) on one column -- lane l takes 16 bytes of every
// 64-lane chunk, U chunks' loads in flight, the residual from its LDS image (r_slot layout: conflict-free 16-byte reads).
// Every lane returns the sum.
template <typename TA, int U>
__device__ __forceinline__ double wave_col_dot(const TA* __restrict__ col, int Mv, int nchunk, const double* rimg, int lane) {
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const VT* p = reinterpret_cast<const VT*>(col) + lane;
    double acc0 = 0.0, acc1 = 0.0;
    for (int t0 = 0; t0 < nchunk; t0 += U) {
        VT v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            v[u] = (VT)0;
            if (t < nchunk && t * ROWS + lane * VEC < Mv) v[u] = p[t * kWave];  // (default policy: the winner's column is read again by k_b_append)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            if (t < nchunk) {
                if constexpr (VEC == 4) {
                    const f64x2 r01 = *reinterpret_cast<const f64x2*>(rimg + (((2 * t) << 6) + lane) * 2);
                    const f64x2 r23 = *reinterpret_cast<const f64x2*>(rimg + (((2 * t + 1) << 6) + lane) * 2);
                    acc0 = fma((double)v[u].x, r01.x, acc0);
                    acc1 = fma((double)v[u].y, r01.y, acc1);
                    acc0 = fma((double)v[u].z, r23.x, acc0);
                    acc1 = fma((double)v[u].w, r23.y, acc1);
                } else {
                    const f64x2 r01 = *reinterpret_cast<const f64x2*>(rimg + t * ROWS + lane * 2);
                    acc0 = fma((double)v[u].x, r01.x, acc0);
                    acc1 = fma((double)v[u].y, r01.y, acc1);
                }
            }
        }
    }
    double acc = acc0 + acc1;
    return wave_xsum(acc);  // (the shuffle butterfly's pairs in its order, without the LDS trips)
}

// dynamic LDS: the residual image, nchunk * 64 * VEC Float64 (32 KiB at M = 4096: four workgroups per CU)
template <typename TA, int U>
__global__ __launch_bounds__(256, 4) void k_b_pick(
    const TA* __restrict__ A, int64_t ld, int Mv, const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int ncand,
    const int* __restrict__ sel_all, BState* __restrict__ bs, BPick* __restrict__ pick, const double* __restrict__ r_all, int Mr,
    int kcap, int Mrows, double eps, int check_eps, double cert_abs, double cert_rel, int kwin, int sig0, double cert_abs2) {
    extern __shared__ __attribute__((aligned(16))) double rimg[];
    __shared__ double sc[8];
    __shared__ double red[kWinMax];
    __shared__ int wi_[kWinMax];
    __shared__ float fsc[4];
    __shared__ int cnt;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = sig0 + (int)blockIdx.x;
    BState& st = bs[s];
    if (tid == 0) {
        pick[s].atom = -1;
        cnt = 0;
    }
    if (st.done) return;
    // the tile candidates are requested FIRST: they depend on nothing, and their round trip overlaps the residual's
    constexpr int EPL = 8;
    const float* cvs = cand_val + (int64_t)s * ncand;
    const int* cis = cand_idx + (int64_t)s * ncand;
    const bool inreg = ncand <= EPL * 256;
    float ev[EPL];
    int ei[EPL];
    if (inreg) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int t = tid + 256 * e;
            ev[e] = t < ncand ? cvs[t] : -1.0f;
            ei[e] = t < ncand ? cis[t] : 0x7fffffff;
        }
    }
    // residual -> LDS image (rows beyond M are stored zeros; beyond Mr: zeros here), ||r||^2, eps-stop
    const double* r = r_all + (int64_t)s * Mr;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    double n2 = 0.0;
    for (int m0 = 4 * tid; m0 < Mlds; m0 += 4 * 256) {
        f64x2 lo = (f64x2)0.0, hi = (f64x2)0.0;
        if (m0 < Mr) {
            lo = reinterpret_cast<const f64x2*>(r + m0)[0];
            hi = reinterpret_cast<const f64x2*>(r + m0)[1];
        }
        n2 = fma(lo.x, lo.x, n2);
        n2 = fma(lo.y, lo.y, n2);
        n2 = fma(hi.x, hi.x, n2);
        n2 = fma(hi.y, hi.y, n2);
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0)) = lo;
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0 + 2)) = hi;
    }
    n2 = block_sum256(n2, sc);  // (its barriers also complete the image)
    const int j = st.nsel;
    if (tid == 0) st.rnorm2 = n2;
    if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break (src/matchingpursuit.jl:79)
        if (tid == 0) st.done |= STOP_EPS;
        return;
    }
    if (j >= Mrows || j >= kcap) {  // nnz(x) < size(A,1) guard (:63)
        if (tid == 0) st.done |= STOP_FULL;
        return;
    }
    // ---- the largest screened value
    float m1 = -1.0f;
    if (inreg) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) m1 = fmaxf(m1, ev[e]);
    } else {
        for (int t = tid; t < ncand; t += 256) m1 = fmaxf(m1, cvs[t]);
    }
    for (int sft = 32; sft >= 1; sft >>= 1) m1 = fmaxf(m1, __shfl_xor(m1, sft, kWave));
    if (lane == 0) fsc[wave] = m1;
    __syncthreads();
    m1 = fmaxf(fmaxf(fsc[0], fsc[1]), fmaxf(fsc[2], fsc[3]));
    if (!(m1 >= 0.0f)) {  // no candidate at all (N == 0)
        if (tid == 0) st.done |= STOP_FULL;
        return;
    }
    // ---- window and certificate bound
    // (int8 screen: the residual image's own rounding adds cert_abs2 * its step, independent of the dictionary's)
    const double dabs = cert_abs2 > 0.0 ? sqrt(cert_abs * cert_abs * n2 + cert_abs2 * cert_abs2 * (double)st.rstep * (double)st.rstep) : cert_abs * sqrt(n2);
    const double lb1 = (double)m1 - dabs - cert_rel * (double)m1;
    double cb = -1.0;
    auto visit = [&](float v, int i, int slot) {
        if (!(v >= 0.0f)) return;
        const double ub = (double)v + dabs + cert_rel * (double)v;
        if (ub >= lb1) {
            const int pos = atomicAdd(&cnt, 1);
            if (pos < kwin) wi_[pos] = i;
            if (slot == kTileCand - 1) cb = fmax(cb, ub);  // atoms hidden behind a tile's last candidate
        } else {
            cb = fmax(cb, ub);
        }
    };
    if (inreg) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) visit(ev[e], ei[e], (tid + 256 * e) & (kTileCand - 1));
    } else {
        for (int t = tid; t < ncand; t += 256) visit(cvs[t], cis[t], t & (kTileCand - 1));
    }
    for (int sft = 32; sft >= 1; sft >>= 1) cb = fmax(cb, shx(cb, sft));
    __syncthreads();  // (sc is free again; the window list is complete)
    if (lane == 0) sc[wave] = cb;
    __syncthreads();
    cb = fmax(fmax(sc[0], sc[1]), fmax(sc[2], sc[3]));
    const int nall = cnt;
    const int nw = min(nall, kwin);
    // ---- exact rescoring of the window: <a_c, r> in Float64, one wave per column (four columns of the signal in flight)
    for (int q = wave; q < nw; q += 4) {
        const double exq = wave_col_dot<TA, U>(A + (int64_t)wi_[q] * ld, Mv, nchunk, rimg, lane);
        if (lane == 0) red[q] = exq;
    }
    __syncthreads();
    // arg-max by the exact value, first index on ties (Julia argmax); the window list's order does not matter
    int besti = 0x7fffffff;
    double bestv = -1.0, cexact = 0.0;
#pragma unroll 1
    for (int q = 0; q < nw; ++q) {
        const int c = wi_[q];
        const double exq = red[q];
        const double v = fabs(exq);
        if (v > bestv || (v == bestv && c < besti)) {
            bestv = v;
            besti = c;
            cexact = exq;
        }
    }
    const bool certified = nall <= kwin && (cb < 0.0 || bestv > cb);
    // "i not in x.nzind" (:66): a re-selected atom makes every later step the same no-op
    const int* sel = sel_all + (int64_t)s * kcap;
    int found = 0;
    for (int t = tid; t < j; t += 256) found |= (sel[t] == besti);
    found = __syncthreads_or(found);
    if (tid == 0) {
        if (!certified) {
            if (st.uncertain == 0) {
                st.unc_step = j;
                st.unc_nall = nall;
                st.unc_cb = cb;
                st.unc_best = bestv;
                st.unc_s1 = (double)m1;
            }
            st.uncertain += 1;
        }
        if (found) {
            st.done |= STOP_STAG;
        } else {
            pick[s].atom = besti;
            pick[s].nwin = nw;
            pick[s].cexact = cexact;
        }
    }
}
inline size_t b_pick_lds_bytes(int Mv, int vec) {
    const int rows = kWave * vec;
    return (size_t)((Mv + rows - 1) / rows) * rows * sizeof(double);
}

// ---------------------------------------------------------------------------------------------
// The append of one OMP step of one signal, by one workgroup.  See the file header for the algebra.
// LDS: vectors g,w,y (3 x kcap Float64) | reduction scratch | the support | the new atom's column in the dictionary's own
// type (f32: 16 KiB at M = 4096: four workgroups per CU).
// DEPTH: columns (pass 2) / row chunks (pass 1) whose loads are issued together.  GRAM: pass 1 is a gather from G = A'A.
template <typename TA, int NI, int DEPTH, bool GRAM>
__global__ __launch_bounds__(256, (b_wgs<TA, NI>())) void k_b_append(
    const TA* __restrict__ A, int64_t ld, int Mv, const double* __restrict__ Gm, int64_t Ng, const BPick* __restrict__ pick,
    double* __restrict__ T_all, double* __restrict__ Tt_all, double* __restrict__ z_all, int* __restrict__ sel_all,
    BState* __restrict__ bs, double* __restrict__ r_all, int Mr, __bf16* __restrict__ rb_all, int Mk, int kcap, int Mrows, int sig0,
    signed char* __restrict__ r8_all, int Mk8, float* __restrict__ sigscale, float astep, int img) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ float smax[4];
    using VT = typename Vec<TA>::type;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = sig0 + (int)blockIdx.x;
    BState& st = bs[s];
    if (st.done) return;
    const int besti = pick[s].atom;
    if (besti < 0) return;
    const double cexact = pick[s].cexact;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    double* gv = lds;                   // kcap
    double* wv = gv + kcap;             // kcap
    double* yv = wv + kcap;             // kcap
    double* sc = yv + kcap;             // 8
    double* part = sc + 8;              // 256: partial sums of the T mat-vecs
    int* selL = reinterpret_cast<int*>(part + 256);  // kcap: the support, staged once (the passes index it per column)
    TA* aimg = reinterpret_cast<TA*>((reinterpret_cast<uintptr_t>(selL + kcap) + 15) & ~(uintptr_t)15);  // Mlds entries in natural row order: lane l of chunk t reads its
                                                            // 16 bytes at (t * 64 + l) * 16 -- consecutive lanes, conflict-free
    double* r = r_all + (int64_t)s * Mr;
    const int j = st.nsel;
    int* sel = sel_all + (int64_t)s * kcap;
    for (int t = tid; t < j; t += 256) selL[t] = sel[t];  // (visible after the barrier inside block_sum256 below)

    // ---- a = A[:, besti] -> LDS image (natural row order, the dictionary's own type: exact); ||a||^2.  Nothing of a stays
    // in registers across pass 1 and the T mat-vecs: pass 2 takes it from the image again.
    double na2 = 0.0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = 4 * (tid + 256 * i);
        double a0[4] = {0.0, 0.0, 0.0, 0.0};
        if (row < Mv) load4(A + (int64_t)besti * ld + row, Mv - row, a0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            na2 = fma(a0[e], a0[e], na2);
            if (row + e < Mlds) aimg[row + e] = (TA)a0[e];
        }
    }
    for (int m = 4 * 256 * NI + tid; m < Mlds; m += 256) aimg[m] = (TA)0;
    na2 = block_sum256(na2, sc);  // (barrier inside: aimg is complete afterwards)

    if constexpr (GRAM) {
        // ---- pass 1 from the resident Gram matrix: g_i = G[s_i, atom], one 8-byte gather per support atom (upper triangle:
        // row <= column) instead of a 16-KiB column read
        for (int t = tid; t < j; t += 256) {
            const int a_ = selL[t];
            const int lo = min(a_, besti), hi = max(a_, besti);
            gv[t] = Gm[lo + (int64_t)hi * Ng];
        }
    } else {
    // ---- pass 1: g_i = <a_{s_i}, a>, one wave per 4 support columns (the sweep's inner loop)
    {
        const VT* as = reinterpret_cast<const VT*>(aimg);
        const int jp1 = j;
        // Software-pipelined: the wave's work is the flat list of (4-column group, row chunk) items; DEPTH items (4 loads
        // each) are always in flight -- an item's registers are refilled for item q + DEPTH right after item q is consumed,
        // across group boundaries too, so the memory pipe never drains between the groups.
        const int ngrp = (jp1 > wave * 4) ? (jp1 - wave * 4 + 15) / 16 : 0;
        const int total = ngrp * nchunk;
        VT a[DEPTH][4];
        const VT* p[4] = {nullptr, nullptr, nullptr, nullptr};
        int ql = 0, tl = 0, c0l = wave * 4;
        auto issue = [&](VT(&dst)[4]) {
            if (ql < total) {
                if (tl == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) p[c] = reinterpret_cast<const VT*>(A + (int64_t)selL[min(c0l + c, j - 1)] * ld) + lane;
                }
                const bool ok = tl * ROWS + lane * VEC < Mv;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dst[c] = (VT)0;
                    if (ok) dst[c] = p[c][tl * kWave];
                }
                if (++tl == nchunk) {
                    tl = 0;
                    c0l += 16;
                }
            }
            ++ql;
        };
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) issue(a[u]);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        int tc = 0, c0c = wave * 4;
        for (int q0 = 0; q0 < total; q0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                if (q0 + u < total) {
                    const VT av = as[tc * kWave + lane];
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[c] = fma((double)a[u][c].x, (double)av.x, acc[c]);
                            acc[c] = fma((double)a[u][c].y, (double)av.y, acc[c]);
                            acc[c] = fma((double)a[u][c].z, (double)av.z, acc[c]);
                            acc[c] = fma((double)a[u][c].w, (double)av.w, acc[c]);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[c] = fma((double)a[u][c].x, (double)av.x, acc[c]);
                            acc[c] = fma((double)a[u][c].y, (double)av.y, acc[c]);
                        }
                    }
                    if (++tc == nchunk) {
                        {
                            // ONE transposing butterfly for the four columns (csmp_kernels.hpp, k_sweep_short): after the xor-32 and
                            // xor-16 steps row q of the wave holds column q's partial sums, the in-row steps finish all four -- per
                            // column the pairs and their order are the plain butterfly's, so g has the same bits
                            double v = xs16(xs32(acc[0], acc[2]), xs32(acc[1], acc[3]));
                            v = row_xsum(v);
                            const int c = lane >> 4;
                            if ((lane & 15) == 0 && c0c + c < j) gv[c0c + c] = v;
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) acc[cc] = 0.0;
                        }
                        tc = 0;
                        c0c += 16;
                    }
                }
                issue(a[u]);
            }
        }
    }
    }
    __syncthreads();

    // ---- w = T' g (w_i = sum_{t<=i} T[t,i] g_t): thread i walks column i of T (rows of Tt contiguous)
    const double* T = T_all + (int64_t)s * kcap * kcap;    // column-major: T[t + i*kcap]
    double* Tw = T_all + (int64_t)s * kcap * kcap;
    const double* Tt = Tt_all + (int64_t)s * kcap * kcap;  // Tt[i + t*kcap] = T[t,i]
    double* Ttw = Tt_all + (int64_t)s * kcap * kcap;
    // Both mat-vecs are split over ALL 256 threads: thread (i, grp) of nI x G sums the terms t = grp, grp + G, ... of its
    // output, 16 loads in flight, and the G partial sums meet in LDS -- a fraction of the dependent round trips of
    // "one thread per output" (j / 8 of them at G = 1), which is what these two short phases consist of.
    int nI = 16;
    while (nI < j) nI <<= 1;
    const int G = nI <= 256 ? 256 / nI : 1;
    double w2 = 0.0;
    if (nI <= 256) {
        const int i = tid & (nI - 1), grp = tid / nI;
        double acc = 0.0, acc1 = 0.0;
        if (i < j) {
            int t = grp;
            for (; t + 15 * G <= i; t += 16 * G) {
                double tv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) tv[u] = Tt[i + (int64_t)(t + u * G) * kcap];
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc = fma(tv[u], gv[t + u * G], acc);
                    acc1 = fma(tv[u + 1], gv[t + (u + 1) * G], acc1);
                }
            }
            double tv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) tv[u] = (t + u * G <= i) ? Tt[i + (int64_t)(t + u * G) * kcap] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (t + u * G <= i) acc = fma(tv[u], gv[t + u * G], acc);
        }
        part[tid] = acc + acc1;
        __syncthreads();
        if (tid < j) {
            double sum = part[tid];
            for (int g2 = 1; g2 < G; ++g2) sum += part[g2 * nI + tid];
            wv[tid] = sum;
            w2 = sum * sum;
        }
    } else {
    for (int i = tid; i < j; i += 256) {
        double acc = 0.0, acc1 = 0.0;
        int t = 0;
        for (; t + 8 <= i + 1; t += 8) {
            double tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = Tt[i + (int64_t)(t + u) * kcap];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = fma(tv[u], gv[t + u], acc);
                acc1 = fma(tv[u + 1], gv[t + u + 1], acc1);
            }
        }
        for (; t <= i; ++t) acc = fma(Tt[i + (int64_t)t * kcap], gv[t], acc);
        acc += acc1;
        wv[i] = acc;
        w2 = fma(acc, acc, w2);
    }
    }
    w2 = block_sum256(w2, sc);
    const double rho2 = na2 - w2;
    // DGKS-style guard: a badly cancelling first pass means an ill-conditioned support; this
    // light-weight factorisation is then not trusted and the signal goes to the exact path
    if (!(rho2 >= 0.5 * na2) || !(rho2 > 0.0)) {
        if (tid == 0) {
            st.illcond += 1;
            st.done |= STOP_FULL;
        }
        return;
    }
    const double rho = sqrt(rho2);
    // ---- y = T w (y_t = sum_{i>=t} T[t,i] w_i): thread (t, grp) walks row t of T (column-major: coalesced over t)
    if (nI <= 256) {
        const int t = tid & (nI - 1), grp = tid / nI;
        double acc = 0.0, acc1 = 0.0;
        if (t < j) {
            int i = t + grp;
            for (; i + 15 * G < j; i += 16 * G) {
                double tv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) tv[u] = T[t + (int64_t)(i + u * G) * kcap];
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc = fma(tv[u], wv[i + u * G], acc);
                    acc1 = fma(tv[u + 1], wv[i + (u + 1) * G], acc1);
                }
            }
            double tv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) tv[u] = (i + u * G < j) ? T[t + (int64_t)(i + u * G) * kcap] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (i + u * G < j) acc = fma(tv[u], wv[i + u * G], acc);
        }
        part[tid] = acc + acc1;  // (the partial sums of w were consumed before the barriers of block_sum256)
        __syncthreads();
        if (tid < j) {
            double sum = part[tid];
            for (int g2 = 1; g2 < G; ++g2) sum += part[g2 * nI + tid];
            yv[tid] = sum;
        }
    } else {
    for (int t = tid; t < j; t += 256) {
        double acc = 0.0, acc1 = 0.0;
        int i = t;
        for (; i + 8 <= j; i += 8) {
            double tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = T[t + (int64_t)(i + u) * kcap];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = fma(tv[u], wv[i + u], acc);
                acc1 = fma(tv[u + 1], wv[i + u + 1], acc1);
            }
        }
        for (; i < j; ++i) acc = fma(T[t + (int64_t)i * kcap], wv[i], acc);
        yv[t] = acc + acc1;
    }
    }
    __syncthreads();

    // ---- pass 2: v = a - sum_i y_i a_{s_i}  (row-owner form, DEPTH columns x NI loads in flight)
    double areg[NI][4];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int row = 4 * (tid + 256 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) areg[u][e] = (row + e < Mlds) ? (double)aimg[row + e] : 0.0;
    }
    // Columns are visited LAST TO FIRST: pass 1 has just streamed them first to last, and the other resident workgroups have
    // pushed this signal's early columns out of the L2 / Infinity Cache in the meantime -- the late ones are still there.
    // A ring of DEPTH columns: column i - DEPTH is requested into the registers column i has just been consumed from.
    {
        const int jtop = j;
        Raw4<TA> cv4[DEPTH][NI];
        auto issue = [&](Raw4<TA>(&dst)[NI], int i) {
            if (i >= 0) {
                const int col = selL[i];
#pragma unroll
                for (int u = 0; u < NI; ++u) {
                    const int row = 4 * (tid + 256 * u);
                    dst[u].zero();
                    if (row < Mv) dst[u].load(A + (int64_t)col * ld + row, Mv - row);
                }
            }
        };
#pragma unroll
        for (int c = 0; c < DEPTH; ++c) issue(cv4[c], jtop - 1 - c);
        for (int i0 = jtop - 1; i0 >= 0; i0 -= DEPTH) {
#pragma unroll
            for (int c = 0; c < DEPTH; ++c) {
                const int i = i0 - c;
                if (i >= 0) {
                    const double yc = yv[i];
#pragma unroll
                    for (int u = 0; u < NI; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) areg[u][e] = fma(-yc, cv4[c][u].get(e), areg[u][e]);
                }
                issue(cv4[c], i - DEPTH);
            }
        }
    }
    // ---- q = v / rho, z_j = <a, r> / rho, r -= q z_j; new column of T = [-y / rho; 1 / rho]
    const double zj = cexact / rho;
    const double f = zj / rho;  // r -= v * (z_j / rho)
    if (img == kOpBf16) {
        __bf16* rb = rb_all + (int64_t)s * Mk;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int row = 4 * (tid + 256 * u);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double rold = (row + e < Mrows) ? r[row + e] : 0.0;  // (reloaded: the residual is not held in registers across the step)
                const double nr = fma(-areg[u][e], f, rold);
                if (row + e < Mrows) r[row + e] = nr;
                o[e] = (__bf16)(float)((row + e < Mrows) ? nr : 0.0);
            }
            if (row < Mk) *reinterpret_cast<bf16x4*>(rb + row) = o;
        }
    } else {  // int8 / binary16 image: the new residual's largest magnitude first (one step / scale per signal), then the image
        float amax = 0.0f;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int row = 4 * (tid + 256 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double rold = (row + e < Mrows) ? r[row + e] : 0.0;
                const double nr = (row + e < Mrows) ? fma(-areg[u][e], f, rold) : 0.0;
                if (row + e < Mrows) r[row + e] = nr;
                areg[u][e] = nr;  // (the register's old content is spent)
                amax = fmaxf(amax, fabsf((float)nr));
            }
        }
        if (img == kOpF16) {  // binary16 image under the new residual's own power-of-two scale
            const float sc = f16_scale(block_absmax256(amax, smax));
            _Float16* rh = reinterpret_cast<_Float16*>(rb_all) + (int64_t)s * Mk;
            using f16x4v = __attribute__((ext_vector_type(4))) _Float16;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int row = 4 * (tid + 256 * u);
                f16x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)(float)(areg[u][e] * (double)sc);
                if (row < Mk) *reinterpret_cast<f16x4v*>(rh + row) = o;
            }
            if (tid == 0) sigscale[s] = astep / sc;
        } else {
        const float rstep = i8_step(block_absmax256(amax, smax));
        const float inv = 1.0f / rstep;
        signed char* r8 = r8_all + (int64_t)s * Mk8;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int row = 4 * (tid + 256 * u);
            if (row < Mk8) {
                int pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk |= (max(-127, min(127, __float2int_rn((float)areg[u][e] * inv))) & 0xff) << (8 * e);
                *reinterpret_cast<int*>(r8 + row) = pk;
            }
        }
        if (tid == 0) {
            st.rstep = rstep;
            sigscale[s] = astep * rstep * (1.0f + 0x1p-20f);
        }
        }
    }
    for (int t = tid; t < j; t += 256) {
        const double v = -yv[t] / rho;
        Tw[t + (int64_t)j * kcap] = v;
        Ttw[j + (int64_t)t * kcap] = v;
    }
    if (tid == 0) {
        Tw[j + (int64_t)j * kcap] = 1.0 / rho;
        Ttw[j + (int64_t)j * kcap] = 1.0 / rho;
        z_all[(int64_t)s * kcap + j] = zj;
        sel[j] = besti;
        st.nsel = j + 1;
    }
}
inline size_t b_append_lds_bytes(int Mv, int vec, int kcap) {
    const int rows = kWave * vec;
    const int nchunk = (Mv + rows - 1) / rows;
    return (size_t)(3 * kcap + 8 + 256) * sizeof(double) + (size_t)kcap * 4 + (size_t)nchunk * rows * (16 / vec) + 64;
}

// x = T z (ldiv!), sorted-index assembly; one workgroup per signal
__global__ __launch_bounds__(256) void k_b_finish(const double* __restrict__ T_all, const double* __restrict__ z_all,
                                                  const int* __restrict__ sel_all, const BState* __restrict__ bs, int kcap,
                                                  int outk, int64_t* __restrict__ out_idx, double* __restrict__ out_val,
                                                  int64_t* __restrict__ out_nnz) {
    extern __shared__ __attribute__((aligned(16))) double x[];  // kcap
    const int s = blockIdx.x, tid = threadIdx.x;
    const int j = bs[s].nsel;
    const double* T = T_all + (int64_t)s * kcap * kcap;
    const double* z = z_all + (int64_t)s * kcap;
    const int* sel = sel_all + (int64_t)s * kcap;
    for (int t = tid; t < j; t += 256) {
        double acc = 0.0;
        for (int i = t; i < j; ++i) acc = fma(T[t + (int64_t)i * kcap], z[i], acc);
        x[t] = acc;
    }
    for (int t = tid; t < outk; t += 256) {
        out_idx[(int64_t)s * outk + t] = -1;
        out_val[(int64_t)s * outk + t] = 0.0;
    }
    __syncthreads();
    for (int t = tid; t < j; t += 256) {
        const int me = sel[t];
        int rank = 0;
        for (int u = 0; u < j; ++u) rank += (sel[u] < me);
        out_idx[(int64_t)s * outk + rank] = me;
        out_val[(int64_t)s * outk + rank] = x[t];
    }
    if (tid == 0) out_nnz[s] = j;
}

}  // namespace csmp
