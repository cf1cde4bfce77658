// csmp_shard.hpp -- column-sharded single-signal OMP (SURVEY.md section 8e "not required but natural",
// section 8f rank 4): ONE signal, the dictionary's columns split over several GPUs (or over several
// contexts of one GPU), so that one omp(A, b, k) call (src/matchingpursuit.jl:73-82) is served by G sweeps
// of N/G columns each instead of one sweep of N.
//
// Every rank keeps the FULL solver state (residual, Q, R, support: a few MiB) and a slice of A.  One step:
//   1. local sweep over the rank's columns (k_sweep*, unchanged) -> per-workgroup arg-max partials;
//   2. k_shard_pack: the rank's best atom as one RECORD = { |c|, global column index, c, pad, the column itself };
//   3. the host's collective (RCCL all_gather of one record per rank: 32 B + M * sizeof(T_A), 16 KiB at C2);
//   4. k_shard_pick on every rank: arg-max over the G records -- larger |c|, ties to the LOWER global index, which
//      is Julia's argmax over the whole dictionary (src/matchingpursuit.jl:181-185) -- and the winner's column
//      copied into a one-column scratch dictionary;
//   5. the ordinary append chain (k_qr1 mode 4 / k_qr2 / k_qr3) on that column: guards of update!(::OMP)
//      (src/matchingpursuit.jl:63,66), add_column!, residual update -- replicated, bit-identical on every rank
//      (same kernels, same inputs, fixed summation orders), so no further exchange is needed.
// The column travels inside the record so that step 3 is the ONLY collective of a step (no owner broadcast
// that would have to wait for the arg-max).
#pragma once
#include "csmp_kernels.hpp"

namespace csmp {

struct ShardRecHdr {
    double absval;   // |<a, r>| of the rank's best atom; -1: the rank has nothing to offer (solve already stopped)
    int64_t label;   // global column index, -1: none
    double cval;     // signed <a, r>
    int64_t pad;
};
static_assert(sizeof(ShardRecHdr) == 32, "record header is 32 bytes");

inline size_t shard_record_bytes(int Mv, size_t elem) { return (sizeof(ShardRecHdr) + (size_t)Mv * elem + 15) / 16 * 16; }

// ONE workgroup: final arg-max over the sweep's workgroup partials (first index on ties), then the record.
template <typename TA>
__global__ __launch_bounds__(256) void k_shard_pack(const double* __restrict__ pval, const int* __restrict__ pidx, int nblk,
                                                    const double* __restrict__ cvec, const TA* __restrict__ A, int64_t ld,
                                                    int Mv, int64_t col_offset, const DevState* __restrict__ st, int skipmask,
                                                    char* __restrict__ rec) {
    __shared__ double sv[256];
    __shared__ int si[256];
    const int tid = threadIdx.x;
    ShardRecHdr* h = reinterpret_cast<ShardRecHdr*>(rec);
    TA* col = reinterpret_cast<TA*>(rec + sizeof(ShardRecHdr));
    if (st->done & skipmask) {  // (the sweep returned early and left stale partials behind)
        if (tid == 0) {
            h->absval = -1.0;
            h->label = -1;
            h->cval = 0.0;
            h->pad = 0;
        }
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
    const bool have = bi != 0x7fffffff && bv >= 0.0;
    if (tid == 0) {
        h->absval = have ? bv : -1.0;
        h->label = have ? col_offset + bi : -1;
        h->cval = have ? cvec[bi] : 0.0;
        h->pad = 0;
    }
    if (have)
        for (int m = tid; m < Mv; m += 256) col[m] = A[(int64_t)bi * ld + m];
}

// ONE workgroup: the winner among nrec records (all ranks, same order everywhere) -> cands[0] = its global
// index, its column -> extcol (a one-column dictionary for k_qr1 mode 4).
template <typename TA>
__global__ __launch_bounds__(256) void k_shard_pick(const char* __restrict__ recs, int nrec, int64_t rec_bytes, int Mv,
                                                    TA* __restrict__ extcol, int* __restrict__ cands,
                                                    int* __restrict__ ncands, DevState* st) {
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ int swin;
    const int tid = threadIdx.x;
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int q = tid; q < nrec; q += 256) {
        const ShardRecHdr* h = reinterpret_cast<const ShardRecHdr*>(recs + (int64_t)q * rec_bytes);
        const int64_t lab = h->label;
        if (lab >= 0 && lab < 0x7fffffff && better(h->absval, (int)lab, bv, bi)) {
            bv = h->absval;
            bi = (int)lab;
        }
    }
    block_argmax(bv, bi, sv, si);
    if (tid == 0) swin = -0x7fffffff;
    __syncthreads();
    for (int q = tid; q < nrec; q += 256) {
        const ShardRecHdr* h = reinterpret_cast<const ShardRecHdr*>(recs + (int64_t)q * rec_bytes);
        if (bi != 0x7fffffff && h->label == (int64_t)bi) atomicMax(&swin, -q);  // the lowest record holding the label
    }
    __syncthreads();
    const int w = swin == -0x7fffffff ? -1 : -swin;
    if (tid == 0) {
        cands[0] = w >= 0 ? bi : -1;
        ncands[0] = 1;
        if (w >= 0) st->cval = reinterpret_cast<const ShardRecHdr*>(recs + (int64_t)w * rec_bytes)->cval;
    }
    if (w >= 0) {
        const TA* col = reinterpret_cast<const TA*>(recs + (int64_t)w * rec_bytes + sizeof(ShardRecHdr));
        for (int m = tid; m < Mv; m += 256) extcol[m] = col[m];
    }
}

}  // namespace csmp
