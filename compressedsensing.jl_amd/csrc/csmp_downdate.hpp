// csmp_downdate.hpp -- removing a column from the on-device thin QR.
//
// Reference primitives replaced (paths relative to the reference repository):
//   k_qrdel_r + k_qrdel_q   remove_column!(AiQR, i) via _dropindex!(x, AiQR, i)   src/util.jl:137-161
//                           (UpdatableQRFactorizations.jl: Givens down-date of the updatable QR)
//                           call sites: backward_step! src/backward.jl:58-62, OMPR update!
//                           src/twostage.jl:171-176
//   (the backward scores backward_δ!, src/backward.jl:70-83, come from the explicit inverse: csmp_tinv.hpp)
//
// The factorisation lives in insertion order (column t of Q/R belongs to atom sel[t]); deleting
// position p leaves R upper Hessenberg from column p on.  Rotations G_p .. G_{n-2} on row pairs
// (i, i+1) restore the triangle; the same rotations act on the column pairs (i, i+1) of Q and on
// z = Q'b.  What is rotated out of the last position is the unit vector q_drop of the old column
// space orthogonal to the new one, and zeta = q_drop'b: the residual grows by exactly zeta*q_drop.
// Cost: O(n^2) on one workgroup for R (a chain of n-p dependent rotations, each one barrier) and
// one pass over the trailing columns of Q -- against a fresh O(M n^2) factorisation.
#pragma once
#include "csmp_kernels.hpp"

namespace csmp {

constexpr int kDelMaxCols = 1023;  // one thread per column plus one for z, in a single workgroup
constexpr int kDelPre = 8;         // rows of look-ahead in the rotation chain

// R side.  Thread c owns OLD column c (it becomes new column c-1 for c > p); thread n owns z.
// Writes the new factor into Rnew (Rold is left intact: the host swaps the two buffers), z and sel
// in place, the rotations into G[2i], G[2i+1] (i = p .. n-2), zeta into scal[0] and the (p, n)
// pair the Q kernel needs into meta[0..1], the leaving atom into meta[2].  delpos < 0 or >= n: nothing is removed (Rnew = Rold).
__global__ __launch_bounds__(1024) void k_qrdel_r(const double* __restrict__ Rold, double* __restrict__ Rnew, int kcap,
                                                  double* __restrict__ z, int* __restrict__ sel, DevState* st,
                                                  const int* __restrict__ delpos, double* __restrict__ G,
                                                  double* __restrict__ scal, int* __restrict__ meta) {
    __shared__ double gcs[1024], gsn[1024];
    const int c = threadIdx.x;
    const int n = st->nsel;
    int p = *delpos;
    if (p < 0 || p >= n) p = -1;
    const bool iscol = c < n, isz = c == n;
    const int64_t co = (int64_t)c * kcap;
    if (p < 0) {
        if (iscol)
            for (int t = 0; t <= c; ++t) Rnew[co + t] = Rold[co + t];
        if (c == 0) {
            meta[0] = -1;
            meta[2] = -1;
        }
        return;
    }
    const int mysel = iscol ? sel[c] : -1;
    const int64_t cn = (int64_t)(c - 1) * kcap;  // destination column of a shifted column
    if (iscol && c < p)
        for (int t = 0; t <= c; ++t) Rnew[co + t] = Rold[co + t];
    if (iscol && c > p)
        for (int t = 0; t < p; ++t) Rnew[cn + t] = Rold[co + t];
    const bool chain = (iscol && c > p) || isz;
    const int lastrow = isz ? n - 1 : c;  // last existing row of this thread's column
    double carry = 0.0;
    if (chain) carry = isz ? z[p] : Rold[co + p];
    auto fetch = [&](double* dst, int i0) {  // rows i0+1 .. i0+kDelPre of this thread's column
#pragma unroll
        for (int u = 0; u < kDelPre; ++u) {
            const int row = i0 + 1 + u;
            dst[u] = (chain && row <= lastrow) ? (isz ? z[row] : Rold[co + row]) : 0.0;
        }
    };
    double pre[kDelPre], nxt[kDelPre];
    fetch(pre, p);
    __syncthreads();  // every z / sel / Rold value that is overwritten below at rows <= p has been read
    if (iscol && c > p) sel[c - 1] = mysel;
    for (int ib = p; ib <= n - 2; ib += kDelPre) {
        fetch(nxt, ib + kDelPre);
#pragma unroll
        for (int u = 0; u < kDelPre; ++u) {
            const int i = ib + u;
            if (i <= n - 2) {  // uniform
                if (iscol && c == i + 1) {  // old column i+1 closes rotation i: rows (i, i+1) -> (rr, 0)
                    // (cs, sn) = (carry, d) / sqrt(carry^2 + d^2) from v_rsq_f64 + two Newton steps: this is the
                    // serial chain of the kernel, and hypot() + a division cost ~4x as much.  Entries of R
                    // are O(1) for unit-norm atoms, so the squares neither overflow nor underflow.
                    const double d = pre[u];
                    const double ss = fma(carry, carry, d * d), hs = 0.5 * ss;
                    double inv = __builtin_amdgcn_rsq(ss);
                    inv = inv * fma(-hs * inv, inv, 1.5);
                    inv = inv * fma(-hs * inv, inv, 1.5);
                    const bool okr = ss > 0.0 && ss < 1e300;
                    const double rr = okr ? ss * inv : hypot(carry, d);
                    if (!okr) inv = rr > 0.0 ? 1.0 / rr : 0.0;
                    const double cs = rr > 0.0 ? carry * inv : 1.0, sn = d * inv;
                    gcs[i] = cs;
                    gsn[i] = sn;
                    G[2 * i] = cs;
                    G[2 * i + 1] = sn;
                    Rnew[(int64_t)i * kcap + i] = rr;
                }
                lds_barrier();
                if ((iscol && c > i + 1) || isz) {
                    const double cs = gcs[i], sn = gsn[i], x = pre[u];
                    const double top = fma(cs, carry, sn * x);
                    carry = fma(cs, x, -sn * carry);
                    if (isz)
                        z[i] = top;
                    else
                        Rnew[cn + i] = top;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kDelPre; ++u) pre[u] = nxt[u];
    }
    if (isz) {
        scal[0] = carry;  // zeta = q_drop' b
        z[n - 1] = 0.0;
    }
    if (c == p) meta[2] = mysel;  // the atom that leaves
    if (c == 0) {
        meta[0] = p;
        meta[1] = n;
        st->nsel = n - 1;
        st->done &= ~(STOP_FULL | STOP_STAG);
    }
}

// Q side: one thread per row applies the rotation chain to the trailing columns in place, keeps the
// rotated-out column q_drop (needed by the forward-regression rescaling, csmp_forward.hpp) and
// restores the residual r += zeta * q_drop.  One wave per 64-row slab, kQPre columns in flight.
constexpr int kQPre = 32;
constexpr int kRotBlk = 16, kRotBuf = 3;  // rot_chain (csmp_tinv.hpp): a ring of three 16-element blocks
__global__ __launch_bounds__(64) void k_qrdel_q(double* __restrict__ Q, int64_t ldq, const double* __restrict__ G,
                                                const double* __restrict__ scal, const int* __restrict__ meta,
                                                double* __restrict__ r, double* __restrict__ qdrop,
                                                double* __restrict__ qsave) {
    const int p = meta[0], n = meta[1];
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (p < 0) {
        qdrop[row] = 0.0;
        return;
    }
    double* q = Q + row;
    qsave[row] = q[(int64_t)(n - 1) * ldq];  // the last column as it was before the rotations
    double carry = q[(int64_t)p * ldq];
    double pre[kQPre], nxt[kQPre];
    auto fetch = [&](double* dst, int i0) {
#pragma unroll
        for (int u = 0; u < kQPre; ++u) {
            const int col = i0 + 1 + u;
            dst[u] = col <= n - 1 ? q[(int64_t)col * ldq] : 0.0;
        }
    };
    fetch(pre, p);
    for (int ib = p; ib <= n - 2; ib += kQPre) {
        fetch(nxt, ib + kQPre);
#pragma unroll
        for (int u = 0; u < kQPre; ++u) {
            const int i = ib + u;
            if (i <= n - 2) {
                const double cs = G[2 * i], sn = G[2 * i + 1], x = pre[u];
                q[(int64_t)i * ldq] = fma(cs, carry, sn * x);
                carry = fma(cs, x, -sn * carry);
            }
        }
#pragma unroll
        for (int u = 0; u < kQPre; ++u) pre[u] = nxt[u];
    }
    q[(int64_t)(n - 1) * ldq] = 0.0;
    qdrop[row] = carry;
    r[row] = fma(scal[0], carry, r[row]);
}

// insertion position of an atom (dropindex!(x, i): findfirst(==(i), x.nzind), src/util.jl:138-146); -1 if absent
__global__ __launch_bounds__(256) void k_find_pos(const int* __restrict__ sel, const DevState* st, int atom,
                                                  int* __restrict__ delpos) {
    __shared__ int pos;
    if (threadIdx.x == 0) pos = -1;
    __syncthreads();
    for (int t = threadIdx.x; t < st->nsel; t += 256)
        if (sel[t] == atom) pos = t;
    __syncthreads();
    if (threadIdx.x == 0) *delpos = pos;
}

// argmin over the backward scores, first minimum in SORTED-INDEX order (findmin over x.nzval order,
// src/backward.jl:57): ties go to the smaller atom index.  One workgroup.  Decision of backward_step!
// (:58-66): the atom is dropped iff sqrt(min + |r|^2) < max_eps and min < max_delta^2; delpos
// receives its insertion position, or -1.  |r|^2 is taken from r itself.
// coef != nullptr selects LACE's rule (src/backward.jl:247-270): the candidate is the atom of least |x_i|
// (argmin(abs, x.nzval), first minimum) and ITS delta2 is what the two thresholds see.
__global__ __launch_bounds__(256) void k_bwd_pick(const double* __restrict__ sc, const int* __restrict__ sel,
                                                  const DevState* st, const double* __restrict__ r, int M,
                                                  double max_eps, double max_d2, int* __restrict__ delpos,
                                                  double* __restrict__ info /* [0]=min δ², [1]=|r|^2 */,
                                                  const double* __restrict__ coef = nullptr, int skipmask = 0, int need_n = -1) {
    __shared__ double sv[256];
    __shared__ int si[256], sp[256];
    __shared__ double red[4];
    const int tid = threadIdx.x, n = st->nsel;
    double n2 = 0.0;
    for (int m = tid; m < M; m += 256) n2 = fma(r[m], r[m], n2);
    n2 = block_sum256(n2, red);
    double bv = __builtin_inf();
    int bi = 0x7fffffff, bp = -1;
    for (int t = tid; t < n; t += 256) {
        const double v = coef ? fabs(coef[t]) : sc[t];
        const int a = sel[t];
        if (v < bv || (v == bv && a < bi)) {
            bv = v;
            bi = a;
            bp = t;
        }
    }
    sv[tid] = bv;
    si[tid] = bi;
    sp[tid] = bp;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
            const double v = sv[tid + s];
            const int a = si[tid + s];
            if (v < sv[tid] || (v == sv[tid] && a < si[tid])) {
                sv[tid] = v;
                si[tid] = a;
                sp[tid] = sp[tid + s];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double mn = (coef && sp[0] >= 0) ? sc[sp[0]] : sv[0];
        // (skipmask, need_n: a backward step queued behind a forward step the host has not seen yet drops nothing unless that step
        // went through -- no stop flag, the support one atom larger)
        const bool gated = (st->done & skipmask) || (need_n >= 0 && n != need_n);
        const bool drop = !gated && n > 0 && sp[0] >= 0 && sqrt(mn + n2) < max_eps && mn < max_d2;
        *delpos = drop ? sp[0] : -1;
        info[0] = mn;
        info[1] = n2;
    }
}

}  // namespace csmp
