// csmp_tinv.hpp -- the two-stage solvers' factorisation with an explicit T = R^-1 (gfx950).
//
// OMPR and SRR alternate one column in, one column out, and need the coefficients (OMPR) or the
// backward scores (SRR) after every exchange (src/twostage.jl:19-31,158-178; src/backward.jl:51-83).
// With R alone each of those is a chain of n dependent steps -- back substitution, n forward
// substitutions for gamma = diag((R'R)^-1), Givens elimination of a Hessenberg R -- and on a GPU a
// chain costs a memory round trip per step (~0.4 us measured, 50-200 us per operation at n = 256).
// Keeping T = R^-1 next to R turns every one of them into independent dot products:
//     coefficients    x = T z                                  (ldiv!, src/matchingpursuit.jl:170-176)
//     backward scores gamma_p = |T[p,:]|^2, delta2_p = x_p^2 / gamma_p     (src/backward.jl:70-83)
//     append          T[:,j] = (-T w / rho, 1 / rho)           (add_column!)
//     removal of p    u = T[p,:] is, in Q coordinates, the direction that leaves the column space.
//                     The plane rotations (i, i+1), i = p..n-2, that take u to sigma e_{n-1} are the
//                     Givens rotations of the down-date, and they follow from PREFIX SUMS of u^2:
//                     sigma_i^2 = u_p^2 + ... + u_i^2, cs_i = u_{i+1}/sigma_{i+1}, sn_i = -sigma_i/sigma_{i+1}
//                     -- no elimination chain.  Rows of Q and T, columns of R and z then take the
//                     rotations independently of one another (one thread each, loads known ahead).
//                     (remove_column!, src/util.jl:152-161)
// Accuracy: T carries cond(R) eps like any explicit inverse; the supports these solvers visit are
// those of compressed-sensing dictionaries (cond(R) ~ 1-10^2), and the C oracle, which refactorises
// from scratch at every change, is the referee in tests/.
#pragma once
#include "csmp_downdate.hpp"

namespace csmp {

constexpr int kTChunk = 32;  // columns per partial dot product
constexpr int kTMaxCols = 4095;  // k_tinv_build_big<64>: 64 entries per lane; k_tdel_prep<4>: four columns per thread

// tmeta[0] = number of columns T currently holds

// Builds all of T from R: workgroup p (one wave) solves R' y = e_p (y = row p of T) right-looking:
// lane l owns columns i = p+1+l+64u with a partial sum each, the owner of column t+1 closes y_{t+1}
// and v_readlane broadcasts it.  Run once per solve, after the initial support has been factorised.
template <int NU, int D>
__global__ __launch_bounds__(64) void k_tinv_build(const double* __restrict__ R, int kcap, const DevState* st,
                                                   double* __restrict__ T, int* __restrict__ tmeta) {
    const int n = st->nsel, p = blockIdx.x, lane = threadIdx.x;
    if (p == 0 && lane == 0) tmeta[0] = n;
    if (p >= n) return;
    const double* colp[NU];
    double acc[NU], rdg[NU];
    bool own[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = p + 1 + lane + 64 * u;
        own[u] = i < n;
        colp[u] = R + (int64_t)(own[u] ? i : p) * kcap;
        acc[u] = 0.0;
        rdg[u] = own[u] ? 1.0 / colp[u][i] : 0.0;
    }
    double yt = 1.0 / R[(int64_t)p * kcap + p];
    if (lane == 0) T[(int64_t)p * kcap + p] = yt;
    double cur[D][NU], nxt[D][NU];
    auto fetch = [&](double (*dst)[NU], int t0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int t = t0 + d, i = p + 1 + lane + 64 * u;
                dst[d][u] = (own[u] && t < i) ? colp[u][t] : 0.0;
            }
    };
    fetch(cur, p);
    for (int tb = p; tb <= n - 2; tb += D) {
        fetch(nxt, tb + D);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tb + d;
            if (t <= n - 2) {
                const int rel = t - p, su = rel >> 6, sl = rel & 63;
                double mine = 0.0;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    acc[u] = fma(cur[d][u], yt, acc[u]);
                    if (u == su) mine = -acc[u] * rdg[u];
                }
                yt = readlane_f64(mine, sl);
                if (lane == sl) T[(int64_t)(t + 1) * kcap + p] = yt;
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int u = 0; u < NU; ++u) cur[d][u] = nxt[d][u];
    }
}

// The same for supports beyond 1024 columns (NU = 32: up to 2049, 64: up to 4097): column pointers are formed on the fly and the
// closing division takes the diagonal entry by a wave-uniform load, so that a lane's state is its NU partial sums and one row of
// prefetched entries -- no second and third per-lane array (they would not fit the register file).  Run once per solve.
template <int NU>
__global__ __launch_bounds__(64) void k_tinv_build_big(const double* __restrict__ R, int kcap, const DevState* st,
                                                       double* __restrict__ T, int* __restrict__ tmeta) {
    const int n = st->nsel, p = blockIdx.x, lane = threadIdx.x;
    if (p == 0 && lane == 0) tmeta[0] = n;
    if (p >= n) return;
    double acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = 0.0;
    double yt = 1.0 / R[(int64_t)p * kcap + p];
    if (lane == 0) T[(int64_t)p * kcap + p] = yt;
    double cur[NU], nxt[NU];
    auto fetch = [&](double* dst, int t) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = p + 1 + lane + 64 * u;
            dst[u] = (i < n && t < i) ? R[(int64_t)i * kcap + t] : 0.0;
        }
    };
    fetch(cur, p);
    for (int t = p; t <= n - 2; ++t) {
        fetch(nxt, t + 1);
        const int rel = t - p, su = rel >> 6, sl = rel & 63;
        const double rd = 1.0 / R[(int64_t)(t + 1) * kcap + (t + 1)];  // (uniform)
        double mine = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            acc[u] = fma(cur[u], yt, acc[u]);
            if (u == su) mine = -acc[u] * rd;
        }
        yt = readlane_f64(mine, sl);
        if (lane == sl) T[(int64_t)(t + 1) * kcap + p] = yt;
#pragma unroll
        for (int u = 0; u < NU; ++u) cur[u] = nxt[u];
    }
}

// Partial products of T with a vector: workgroup (rb, cb) covers rows rb*64 + lane and the kTChunk
// columns of chunk cb; all kTChunk loads of a lane are independent.  mode 0: vec = z over the nsel
// columns (+ squared row norms); mode 1 (append): the new column j = tmeta[0] of R holds w above its
// diagonal, columns 0..j-1 of T take part -- nothing happens unless nsel > j.
__global__ __launch_bounds__(64) void k_tinv_matvec(const double* __restrict__ T, int kcap, const DevState* st,
                                                    const int* __restrict__ tmeta, const double* __restrict__ z,
                                                    const double* __restrict__ R, int mode,
                                                    double* __restrict__ pd, double* __restrict__ pn) {
    const int n = st->nsel, j = tmeta[0];
    int ncols;
    const double* vec;
    if (mode == 0) {
        ncols = n;
        vec = z;
    } else {
        if (!(n > j)) return;
        ncols = j;
        vec = R + (int64_t)j * kcap;
    }
    const int r = blockIdx.x * 64 + threadIdx.x, c0 = blockIdx.y * kTChunk;
    if (c0 >= ncols) return;
    double v[kTChunk];
#pragma unroll
    for (int q = 0; q < kTChunk; ++q) {
        const int i = c0 + q;
        v[q] = (r < ncols && i >= r && i < ncols) ? T[(int64_t)i * kcap + r] : 0.0;
    }
    double d = 0.0, s = 0.0;
#pragma unroll
    for (int q = 0; q < kTChunk; ++q) {
        const int i = c0 + q;
        const double x = i < ncols ? vec[i] : 0.0;
        d = fma(v[q], x, d);
        s = fma(v[q], v[q], s);
    }
    if (r < kcap) {
        pd[(int64_t)blockIdx.y * kcap + r] = d;
        pn[(int64_t)blockIdx.y * kcap + r] = s;
    }
}

// Closes k_tinv_matvec.  mode 0: x = T z, gamma, delta2 = x^2 / gamma (src/backward.jl:81).
// mode 1: column j of T from the partials with w: T[0:j, j] = -(T w) / rho, T[j, j] = 1 / rho.
__global__ __launch_bounds__(256) void k_tinv_fin(double* __restrict__ T, int kcap, const DevState* st,
                                                  int* __restrict__ tmeta, const double* __restrict__ R, int mode,
                                                  const double* __restrict__ pd, const double* __restrict__ pn,
                                                  double* __restrict__ coef, double* __restrict__ d2) {
    const int n = st->nsel, j = tmeta[0];
    if (mode == 0) {
        const int nch = (n + kTChunk - 1) / kTChunk;
        for (int r = threadIdx.x; r < n; r += 256) {
            double x = 0.0, g = 0.0;
            for (int c = r / kTChunk; c < nch; ++c) {
                x += pd[(int64_t)c * kcap + r];
                g += pn[(int64_t)c * kcap + r];
            }
            coef[r] = x;
            d2[r] = x * x / g;
        }
        return;
    }
    if (!(n > j)) return;
    const double rho = R[(int64_t)j * kcap + j];
    const int nch = (j + kTChunk - 1) / kTChunk;
    for (int r = threadIdx.x; r <= j; r += 256) {
        if (r == j) {
            T[(int64_t)j * kcap + j] = 1.0 / rho;
        } else {
            double x = 0.0;
            for (int c = r / kTChunk; c < nch; ++c) x += pd[(int64_t)c * kcap + r];
            T[(int64_t)j * kcap + r] = -x / rho;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) tmeta[0] = j + 1;
}

// (index, coefficient) pairs in ascending index order from coefficients in insertion order (the tail of k_finish).
// r != NULL: ||r||^2 into *n2out as well (k_norm2's arithmetic: the two used to be two launches on a latency chain).
__global__ __launch_bounds__(256) void k_emit_sorted(const double* __restrict__ coef, const int* __restrict__ sel,
                                                     const DevState* st, int64_t* __restrict__ out_idx,
                                                     double* __restrict__ out_val, int64_t* __restrict__ out_nnz,
                                                     int64_t* __restrict__ out_order, int outcap,
                                                     const double* __restrict__ r = nullptr, int M = 0, double* __restrict__ n2out = nullptr) {
    const int tid = threadIdx.x, j = st->nsel;
    if (r) {
        __shared__ double sred[256];
        double acc = 0.0;
        for (int i = tid; i < M; i += 256) acc = fma(r[i], r[i], acc);
        sred[tid] = acc;
        __syncthreads();
        for (int k = 128; k >= 1; k >>= 1) {
            if (tid < k) sred[tid] += sred[tid + k];
            __syncthreads();
        }
        if (tid == 0) *n2out = sred[0];
    }
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
        out_val[rank] = coef[t];
    }
    if (tid == 0) *out_nnz = j;
}

// OMPR's update!, between the sweep and the host's decision (one workgroup): the arg-max over the sweep's partials (k_select, mode 0:
// no guards) and c[i] for the atoms of the support in INDEX order (the sorted list the last k_emit_sorted left on the device) --
// the numbers "x.nzval = P.Ar[x.nzind]" needs (src/twostage.jl:166).  It was three launches: the support going up, k_select, k_gather.
__global__ __launch_bounds__(256) void k_ompr_pick(const double* __restrict__ pval, const int* __restrict__ pidx, int nblk,
                                                   const double* __restrict__ cvec, DevState* st, const int64_t* __restrict__ sorted_idx,
                                                   int k, double* __restrict__ cs, const double* __restrict__ sorted_val,
                                                   const int* __restrict__ sel, int* __restrict__ meta) {
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ int cnt[2];
    const int tid = threadIdx.x;
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int q = tid; q < nblk; q += 256)
        if (better(pval[q], pidx[q], bv, bi)) {
            bv = pval[q];
            bi = pidx[q];
        }
    block_argmax(bv, bi, sv, si);
    for (int t = tid; t < k; t += 256) cs[t] = cvec[sorted_idx[t]];
    if (tid == 0) {
        st->cand = bi;
        st->cval = cvec[bi];
        st->j = st->nsel;
        st->go = 1;
    }
    if (!meta) return;
    // ... and the decision itself (src/twostage.jl:158-171), so that the exchange can be queued behind this kernel without a trip
    // to the host: x[i] = NaN; x.nzval = Ar[x.nzind]; the FIRST entry of smallest magnitude leaves.
    //   meta[0] = 1 exchange (meta[1] leaves, meta[2] joins, meta[3] = the leaving atom's slot in sel), 0 nothing changes (no
    //   candidate, or the candidate itself is the smallest), 2 the arg-max lies inside the support (the host scans: :139-155)
    const double ccand = cvec[bi];
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    int below = 0, inside = 0;
    for (int t = tid; t < k; t += 256) {
        const int64_t a = sorted_idx[t];
        below += a < bi;
        inside += a == bi;
    }
    if (below) atomicAdd(&cnt[0], below);
    if (inside) atomicAdd(&cnt[1], inside);
    __syncthreads();
    const int pos = cnt[0];
    if (cnt[1] || !(fabs(ccand) > 0.0)) {
        if (tid == 0) meta[0] = cnt[1] ? 2 : 0;
        return;
    }
    // merged order: support entry t sits at t + (t >= pos), the candidate at pos; ties go to the lower merged index
    double mv = -__builtin_inf();
    int mi = 0x7fffffff;
    for (int t = tid; t <= k; t += 256) {
        double v;
        int at;
        if (t == k) {
            v = ccand;
            at = pos;
        } else {
            v = sorted_val[t] + cvec[sorted_idx[t]];
            at = t + (t >= pos);
        }
        if (better(-fabs(v), at, mv, mi)) {
            mv = -fabs(v);
            mi = at;
        }
    }
    block_argmax(mv, mi, sv, si);
    // (the host loop starts from entry 0 whatever it is: a NaN there is never displaced)
    const double v0 = pos == 0 ? ccand : sorted_val[0] + cvec[sorted_idx[0]];
    if (v0 != v0) mi = 0;
    if (mi == pos) {
        if (tid == 0) meta[0] = 0;
        return;
    }
    const int leaving = (int)sorted_idx[mi < pos ? mi : mi - 1];
    for (int t = tid; t < k; t += 256)
        if (sel[t] == leaving) meta[3] = t;
    if (tid == 0) {
        meta[0] = 1;
        meta[1] = leaving;
        meta[2] = bi;
    }
}
// ... and after it: the atom that leaves -> its insertion position (k_find_pos), the atom that joins -> the append's candidate list
__global__ __launch_bounds__(256) void k_swap_prep(const int* __restrict__ sel, const DevState* st, int leaving, int joining,
                                                   int* __restrict__ delpos, int* __restrict__ cands, int* __restrict__ ncands) {
    __shared__ int pos;
    if (threadIdx.x == 0) pos = -1;
    __syncthreads();
    for (int t = threadIdx.x; t < st->nsel; t += 256)
        if (sel[t] == leaving) pos = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        *delpos = pos;
        cands[0] = joining;
        *ncands = 1;
    }
}

// Removal, step 1 (one workgroup of 1024 threads; thread c <-> the CPT consecutive columns c CPT .. c CPT + CPT - 1, CPT = 1 up to
// 1024 columns, 4 up to 4096): the rotations from row p of T.
// G[2i], G[2i+1] = (cs_i, sn_i), i = p..n-2; scal[0] = zeta = u'z / sigma_{n-1} (the component of b along
// the direction that leaves); meta = (p, n, leaving atom); sel shifted; nsel, tmeta[0] decremented.
template <int CPT>
__global__ __launch_bounds__(1024) void k_tdel_prep(const double* __restrict__ T, int kcap, const double* __restrict__ z,
                                                    int* __restrict__ sel, DevState* st, const int* __restrict__ delpos,
                                                    double* __restrict__ G, double* __restrict__ scal,
                                                    int* __restrict__ meta, int* __restrict__ tmeta) {
    __shared__ double sa[1024], sb[1024];
    __shared__ double red[16];
    const int c = threadIdx.x, n = st->nsel;
    int p = *delpos;
    if (p < 0 || p >= n) {
        if (c == 0) {
            meta[0] = -1;
            meta[2] = -1;
        }
        return;
    }
    double u[CPT], S[CPT];
    int mysel[CPT];
    double loc = 0.0, uz = 0.0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int i = c * CPT + q;
        const bool in = i >= p && i < n;
        u[q] = in ? T[(int64_t)i * kcap + p] : 0.0;
        mysel[q] = i < n ? sel[i] : -1;
        loc = fma(u[q], u[q], loc);
        S[q] = loc;  // inclusive prefix inside the thread
        uz = in ? fma(u[q], z[i], uz) : uz;
    }
    // inclusive scan of the threads' totals (Hillis-Steele, ping-pong in LDS)
    double* a = sa;
    double* b = sb;
    a[c] = loc;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) {
        b[c] = a[c] + (c >= off ? a[c - off] : 0.0);
        __syncthreads();
        double* t = a;
        a = b;
        b = t;
    }
    const double before = c > 0 ? a[c - 1] : 0.0;  // sum of u^2 over the columns of the threads before this one
    const double total = a[blockDim.x - 1];
    __syncthreads();
    // sigma_i: sigma_p keeps the sign of u_p; positive from there on.  b[c] = sigma of the thread's LAST column (its right
    // neighbour's first rotation needs it)
    double sig[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int i = c * CPT + q;
        sig[q] = i == p ? u[q] : sqrt(before + S[q]);
    }
    b[c] = sig[CPT - 1];
    // zeta = u'z / sigma_{n-1}
    for (int s_ = 32; s_ >= 1; s_ >>= 1) uz += shx(uz, s_);
    if ((c & 63) == 0) red[c >> 6] = uz;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int i = c * CPT + q;
        if (i > p && i < n) {
            const double inv = 1.0 / sig[q];
            const double sprev = q > 0 ? sig[q - 1] : b[c - 1];  // sigma_{i-1}
            G[2 * (i - 1)] = u[q] * inv;
            G[2 * (i - 1) + 1] = -sprev * inv;
            sel[i - 1] = mysel[q];  // (every thread took its own values before the barriers above)
        }
        if (i == p) meta[2] = mysel[q];
    }
    if (c == 0) {
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
        const double slast = (n - 1 == p) ? T[(int64_t)p * kcap + p] : sqrt(total);  // sigma_{n-1}
        scal[0] = tot / slast;
        meta[0] = p;
        meta[1] = n;
        st->nsel = n - 1;
        st->done &= ~(STOP_FULL | STOP_STAG);
        tmeta[0] = n - 1;
    }
}

// Removal, step 2: every row of Q and of T, every column of R, and z take the rotations -- one thread
// per vector, a ring of blocks requested ahead (rot_chain).  Workgroups (64 threads): [0, GQ) rows of Q (in place;
// also q_drop, the pre-rotation last column, and r += zeta q_drop); then NB blocks of T rows (into
// T2, row p dropped: the part from column p on); NB blocks of R columns (into R2, column p dropped; skipR: not at all); one
// block for z; then kcap / 4 copy blocks (T's columns left of p, unchanged).  With meta[0] < 0 nothing is removed and T2 / R2
// receive copies.
// ld(i) is a PLAIN load from a clamped address -- no branch around it and no select on its value (either one makes the compiler wait
// for the load on the spot, and the chain then pays a memory latency per element instead of one per block of kQPre).  Elements past
// `last` + 1 may therefore be anything, NaN included: they meet the identity rotation, their outputs are not stored and the carry
// is not advanced past `last`.
template <typename LD, typename ST>
__device__ __forceinline__ double rot_chain(int p, int last, const double* __restrict__ G, LD ld, ST stv) {
    // A ring of kRotBuf blocks of kRotBlk elements: a block is consumed, then refilled at once with the block kRotBuf ahead, so
    // 2-3 blocks of loads are in flight all the time (one pre-fetched block drained once per block: a memory latency per 32
    // elements was most of the chain).  The rotations (c_i, s_i) -- the same address in every lane -- travel with the data.
    constexpr int RB = kRotBlk, NBUF = kRotBuf;
    double carry = ld(p);
    double x[NBUF][RB];
    f64x2 g[NBUF][RB];
    const f64x2* G2 = reinterpret_cast<const f64x2*>(G);
    const int glast = last > p ? last : p;
    auto fill = [&](double(&xb)[RB], f64x2(&gb)[RB], int base) {  // elements base+1 .. base+RB, rotations base .. base+RB-1
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            xb[u] = ld(base + 1 + u);
            const f64x2 gv = G2[base + u <= glast ? base + u : glast];
            gb[u] = (base + u <= last) ? gv : f64x2{1.0, 0.0};
        }
    };
    auto eat_full = [&](const double(&xb)[RB], const f64x2(&gb)[RB], int base) {  // a whole block: no test per element
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const double cs = gb[u].x, sn = gb[u].y, xv = xb[u];
            stv(base + u, fma(cs, carry, sn * xv));
            carry = fma(-sn, carry, cs * xv);  // (cs * x is off the chain: ONE dependent fma per rotation)
        }
    };
    auto eat_part = [&](const double(&xb)[RB], const f64x2(&gb)[RB], int base) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const double cs = gb[u].x, sn = gb[u].y, xv = xb[u];
            const double out = fma(cs, carry, sn * xv);
            if (base + u <= last) {
                stv(base + u, out);
                carry = fma(-sn, carry, cs * xv);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < NBUF; ++d) fill(x[d], g[d], p + d * RB);
    int ib = p;
    // whole rounds of the ring: nothing conditional between a block's loads and their use, so the compiler waits for that block's
    // loads only (a load under a branch costs a wait for everything in flight)
    for (; ib + NBUF * RB - 1 <= last; ib += NBUF * RB) {
#pragma unroll
        for (int d = 0; d < NBUF; ++d) {
            eat_full(x[d], g[d], ib + d * RB);
            fill(x[d], g[d], ib + d * RB + NBUF * RB);
        }
    }
#pragma unroll
    for (int d = 0; d < NBUF; ++d)  // the last, partial round (the ring is aligned at block 0 again)
        if (ib + d * RB <= last) eat_part(x[d], g[d], ib + d * RB);
    return carry;
}

template <int PART = 0>  // 0: everything; 1 / 2 / 3: the Q / T / R-and-z blocks only (a diagnostic split: one row each in a kernel trace)
__global__ __launch_bounds__(64) void k_tdel_apply(double* __restrict__ Q, int64_t ldq, int GQ, const double* __restrict__ T,
                                                   double* __restrict__ T2, const double* __restrict__ R,
                                                   double* __restrict__ R2, int kcap, int NB, double* __restrict__ z,
                                                   const double* __restrict__ G, const double* __restrict__ scal,
                                                   const int* __restrict__ meta, double* __restrict__ r,
                                                   double* __restrict__ qdrop, double* __restrict__ qsave, int skipR) {
    const int p = meta[0], n = meta[1];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (skipR && b >= GQ + NB && b < GQ + 2 * NB) return;  // (the two-stage solvers never read R's old columns: see launch_delete_t)
    if (PART != 0 && (PART == 1) != (b < GQ)) return;
    if (PART != 0 && b >= GQ && (PART == 2) != (b < GQ + NB || b > GQ + 2 * NB)) return;
    if (b < GQ) {  // rows of Q
        const int64_t row = (int64_t)b * 64 + lane;
        if (p < 0) {
            qdrop[row] = 0.0;
            return;
        }
        double* q = Q + row;
        qsave[row] = q[(int64_t)(n - 1) * ldq];
        const double carry = rot_chain(
            p, n - 2, G, [&](int i) { return q[(int64_t)(i < kcap ? i : kcap - 1) * ldq]; },
            [&](int i, double v) { q[(int64_t)i * ldq] = v; });
        q[(int64_t)(n - 1) * ldq] = 0.0;
        qdrop[row] = carry;
        r[row] = fma(scal[0], carry, r[row]);
        return;
    }
    if (b < GQ + NB) {  // rows of T
        const int t = (b - GQ) * 64 + lane;
        if (p < 0) {  // plain copy, row t of the upper triangle
            for (int i = t; i < kcap; ++i) T2[(int64_t)i * kcap + t] = T[(int64_t)i * kcap + t];
            return;
        }
        if (t >= n || t == p) return;
        const int tn = t - (t > p ? 1 : 0);
        rot_chain(
            p, n - 2, G, [&](int i) { return T[(int64_t)(i < kcap ? i : kcap - 1) * kcap + t]; },  // (entries i < t: the zero lower triangle, tinv_ensure)
            [&](int i, double v) {
                if (i >= tn) T2[(int64_t)i * kcap + tn] = v;
            });
        return;
    }
    if (b < GQ + 2 * NB) {  // columns of R
        const int c = (b - GQ - NB) * 64 + lane;
        if (p < 0) {
            if (c < kcap)
                for (int i = 0; i <= c; ++i) R2[(int64_t)c * kcap + i] = R[(int64_t)c * kcap + i];
            return;
        }
        if (c >= n || c == p) return;
        const double* src = R + (int64_t)c * kcap;
        if (c < p) {
            for (int i = 0; i <= c; ++i) R2[(int64_t)c * kcap + i] = src[i];
            return;
        }
        double* dst = R2 + (int64_t)(c - 1) * kcap;
        for (int i = 0; i < p; ++i) dst[i] = src[i];
        rot_chain(
            p, c - 1, G, [&](int i) { return i <= c ? src[i] : 0.0; }, [&](int i, double v) { dst[i] = v; });  // (the functor's path only: conditional loads)
        return;
    }
    if (b > GQ + 2 * NB) {
        // copy blocks: the columns left of p keep their entries (rows <= column < p, unchanged position) -- four columns per block,
        // rows along the lanes, all loads of a block of rows requested together (walking them one row per lane inside the T blocks
        // was as long as the rotation chain itself)
        if (p <= 0) return;
        const int c0 = (b - GQ - 2 * NB - 1) * 4;
        if (c0 >= p) return;
        for (int r0 = 0; r0 < p; r0 += 4 * 64) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int i = c0 + u, rr = r0 + w * 64 + lane;
                    v[u][w] = T[(int64_t)(i < kcap ? i : kcap - 1) * kcap + (rr < kcap ? rr : kcap - 1)];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int i = c0 + u, rr = r0 + w * 64 + lane;
                    if (i < p && rr <= i) T2[(int64_t)i * kcap + rr] = v[u][w];
                }
        }
        return;
    }
    if (lane == 0 && p >= 0) {  // z
        rot_chain(
            p, n - 2, G, [&](int i) { return z[i < kcap ? i : kcap - 1]; }, [&](int i, double v) { z[i] = v; });
        z[n - 1] = 0.0;
    }
}

}  // namespace csmp
