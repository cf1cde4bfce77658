// csmp_screened.hpp -- single-signal OMP with a SCREENED sweep (option CSMP_OPT_SCREENED_SWEEP).
//
// The exact sweep streams the f32 dictionary once per atom (1 GiB at config 2) and is bound by HBM, not by arithmetic.
// The batched path already keeps a bf16 image of the dictionary ([Npad][Mk], half the bytes) and a machinery that turns
// approximate correlations into EXACT selections: rescore the candidates that could still be the maximum in Float64 from
// the master dictionary, certify the pick against an error bound, and fall back to the exact path when the certificate
// fails (csmp_batched.hpp: k_b_pick).  SURVEY.md section 7 (hard part 1) proposes the same for one signal; this is it:
//
//   k_sweep_bf16   c~ = Ab' r  as a GEMV over the bf16 image: four neighbouring columns per wave side by side, 16 bytes per
//                  lane and 64-lane chunk, f32 accumulation against the f32 image of the residual in LDS, a ring of items in
//                  flight, column groups handed out by partition ticket counters; every workgroup keeps its 4 largest |c~|
//                  (ties: lower index) -- its 4th bounds every atom it did not list.  Prologue as the exact sweep: ||r||^2
//                  from the Float64 residual and the driver's residual test (src/matchingpursuit.jl:79).  HBM-bound:
//                  M N 2 bytes per atom.
//   k_pick1        ONE workgroup: window + exact rescoring + certificate, exactly k_b_pick's logic on the <= 4 x 256
//                  candidates; publishes the pick as the single "sweep partial" (pval[0], pidx[0]) that k_qr1 (mode 1:
//                  arg-max + update!'s guards) consumes -- the append chain is the exact path's, unchanged.
//   k_pickS        the same around the S-th largest value (GOMP's partialsortperm(abs(A'r), 1:S)): the whole top-S set and
//                  its order certified, handed to the (panel) append kernels as k_top_merge would.
//
// A failed certificate raises DevState::uncertain; the driver repeats that solve with the exact sweep, so results equal
// csmp_omp's.  Only the dictionary is rounded here (the residual enters in f32, 2^-24): the error model is the batched
// path's with one rounded operand -- the same (conservative) bounds are used.
#pragma once
#include "csmp_batched.hpp"
#include <type_traits>

namespace csmp {

constexpr int kScrCand = 8;  // candidates kept per sweep workgroup (its LAST one bounds every atom it did not list: with 4, a workgroup
                             // that happened to hold four atoms of a 60-atom GOMP window failed the certificate once per few solves)

// LDS image of the residual as f32 for the bf16 sweep: lane l of chunk t (512 rows) multiplies rows 8 (64 t + l) .. + 7;
// two planes of four floats so that consecutive lanes read consecutive 16-byte slots (conflict-free ds_read_b128)
__device__ __forceinline__ int rf_slot(int m) {  // index in floats
    const int t = m >> 9, l = (m & 511) >> 3, e = m & 7;
    return ((((t << 1) + (e >> 2)) << 6) + l) * 4 + (e & 3);
}

constexpr int kScrPartWgs = 8;        // workgroups that share a ticket counter
constexpr int kScrTicketStride = 64;  // words between two counters (a 256-byte line each)
constexpr int kScrCols = 4;  // columns a wave multiplies side by side (they share the residual registers and one reduction)

// sum over the 16 lanes of a row, left in every lane of it: four DPP adds (row_mirror, row_half_mirror, quad_perm) -- no LDS trip
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // lane i <- 15 - i
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // i <- 7 - i (halves)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad [2,3,0,1]
    return v;
}

// The wave's work is a stream of ITEMS: U chunks (U x 1 KiB) of each of FOUR neighbouring columns; a column group is `nblocks`
// items.  D - 1 items are in flight beyond the one being multiplied (a ring of D register sets) and the first loads are issued
// BEFORE the residual prologue -- they do not depend on it.  The four columns share the residual's registers (a quarter of the
// LDS reads) and ONE transposing butterfly that leaves column c's sum in lanes 16 c .. 16 c + 15 (two LDS round trips per
// group instead of six per column: with two or three waves per SIMD that latency chain is not hidden).  Each 16-lane row keeps
// the 4 largest of ITS columns; the workgroup's 4 largest are the top of the 16 lists.
// Groups are handed out DYNAMICALLY.  With a static grid-stride split the workgroups of one launch finished between 60 and
// 90 us at configs[1] (tools/probes/sweep_probe.hip) and the slowest set the kernel's time.  ONE ticket counter for the chip
// is no answer: agent-scope atomics on one address execute at the memory side, ~11 ns apiece one after the other -- 18 000
// of them made the kernel 210 us.  So the workgroups form PARTITIONS of eight consecutive block ids (one per XCD under the
// round-robin dispatch, so a slow XCD is diluted), partition p owns the groups p, p + NP, p + 2 NP, ... (all partitions walk
// the image at the same pace) and has its own counter on its own 256-byte line: a wave's first group is its number inside
// the partition, every further one a ticket, asked for a whole group (32 KiB) before the answer is needed.
// k_pick1 zeroes the counters after every sweep (kScrTicketStride words apart).
// FULL: Mk is a multiple of 512 U (every item is U whole chunks): no guard anywhere in the stream.  Otherwise the loads of a
// partial item are clamped into the column and meet zeros in the residual image (rows >= Mk >= M), chunks beyond the column
// are skipped (wave-uniform).  The image has Npad >= N columns (zeros): a group never leaves it; columns >= N are not listed.
__device__ __forceinline__ int row16_sum(int v) {  // the same for the int8 sweep's exact integer sums
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    return v;
}

// I8 = true: the same stream over the INT8 image (csmp_batched.hpp: k_b_convert_i8, one step `astep` for the dictionary): a
// 16-byte load is 16 elements, a chunk 1024 rows; the residual is quantised in the prologue under its own step (max|r_i| / 127,
// one workgroup maximum), kept as int8 in LDS (4 KiB at M = 4096), multiplied with v_dot4c_i32_i8 -- four exact integer
// multiply-adds per instruction -- and the candidate values are |integer sum| * astep * rstep.  Mk counts the image's
// elements per row (bf16: 2 bytes each, int8: 1).
// IMG = kOpF16: the same stream as bf16 over the BINARY16 image (csmp_batched.hpp: k_b_convert_f16, the dictionary under one
// power-of-two scale): eleven significand bits instead of eight -- the rigorous certificate's bound shrinks from 2^-8 to 2^-11 of
// |a||r| (one rounded operand: the residual enters in f32) -- at the same 2 bytes per element; the candidate values are multiplied
// by `astep` = 1 / scale (exact).
template <int U, int D, bool FULL, int C, int IMG, int LC = kScrCand>
__device__ __forceinline__ void sweep_img_body(const char* __restrict__ Ab, int Mk, int64_t N, const double* __restrict__ r, int Mr,
                                               float* __restrict__ cand_val, int* __restrict__ cand_idx, DevState* st, double eps,
                                               int check_eps_flags, int skipmask, unsigned* __restrict__ tickets, float astep, char* smem) {
    constexpr bool I8 = IMG == kOpI8;
    const int check_eps = check_eps_flags & 1;
    const bool stat = (check_eps_flags & 2) != 0;  // (csmp_tune screen_static: the groups dealt out statically, no tickets)
    constexpr int NW = kSweepThreads / kWave;
    static_assert(C == 2 || C == 4, "two or four columns side by side");
    constexpr int NL = NW * (kWave / 16);  // lists per workgroup (one per 16-lane row)
    constexpr int EV = I8 ? 16 : 8;        // elements per 16-byte load
    constexpr int CH = kWave * EV;         // rows per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
    if (st->done & skipmask) return;
    const int nchunk = (Mk + CH - 1) / CH;
    const int Ml = nchunk * CH;
    const int64_t RB = (int64_t)Mk * (I8 ? 1 : 2);  // bytes per image row
    using V = bf16x8;
    using i32x4 = int __attribute__((ext_vector_type(4)));
    const int nblocks = (nchunk + U - 1) / U;
    const int64_t ngroups = (N + C - 1) / C;
    // partition (see above): NP of them, WPP workgroups each (the host launches a multiple of 8 workgroups, or fewer than 8)
    const int NP = nblk >= kScrPartWgs ? nblk / kScrPartWgs : 1;
    const int WPP = nblk >= kScrPartWgs ? kScrPartWgs : nblk;
    const int part = nblk >= kScrPartWgs ? bid / kScrPartWgs : 0;
    const int wl = (bid - part * WPP) * NW + wave;  // the wave's number inside its partition
    unsigned* const ticket = tickets + part * kScrTicketStride;
    const int64_t g0 = stat ? (int64_t)bid * NW + wave : (int64_t)wl * NP + part;
    const int64_t gstride = (int64_t)nblk * NW;
    // the residual's loads go out FIRST: loads return in order, and behind the dictionary prefetch they would wait for 16 KiB
    // per wave of HBM traffic before the prologue could start
    constexpr int PR = 4;  // row quads per thread and pass (one pass at M <= 4096)
    f64x2 rlo[PR], rhi[PR];
    auto r_issue = [&](int base) {
#pragma unroll
        for (int p = 0; p < PR; ++p) {
            const int m0 = base + (p * kSweepThreads + tid) * 4;
            rlo[p] = rhi[p] = (f64x2)0.0;
            if (m0 < Mr) {
                rlo[p] = reinterpret_cast<const f64x2*>(r + m0)[0];
                rhi[p] = reinterpret_cast<const f64x2*>(r + m0)[1];
            }
        }
    };
    r_issue(0);
    V buf[D][U][C];
    int sg[D];                       // group of the item in ring slot d (-1: none -- the stream has ended)
    int lg = g0 < ngroups ? (int)g0 : -1;  // group of the next item to load
    const char* lp = Ab + g0 * C * RB;
    int lb = 0;                      // its block
    unsigned tk = 0;                 // lane 0: the ticket for the group after `lg`
    if (!stat && lg >= 0 && lane == 0) tk = atomicAdd(ticket, 1u);
    auto load_next = [&](V (&dst)[U][C], int& slot_g) {
        slot_g = lg;
        if (lg >= 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = lb * U + u;
                if (FULL || t < nchunk) {
                    const int off = FULL ? t * 64 + lane : min(t * 64 + lane, Mk / EV - 1);
#pragma unroll
                    for (int c = 0; c < C; ++c) dst[u][c] = __builtin_nontemporal_load(reinterpret_cast<const V*>(lp + (int64_t)c * RB) + off);
                }
            }
            if (++lb == nblocks) {  // the following group: the ticket asked for a group ago, and the next request
                lb = 0;
                const int64_t g = stat ? (int64_t)lg + gstride : ((int64_t)WPP * NW + (int64_t)__builtin_amdgcn_readfirstlane((int)tk)) * NP + part;
                lg = g < ngroups ? (int)g : -1;
                if (lg >= 0) {
                    lp = Ab + g * C * RB;
                    if (!stat && lane == 0) tk = atomicAdd(ticket, 1u);
                }
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load_next(buf[d], sg[d]);
    float* rimgf = reinterpret_cast<float*>(smem);                    // bf16 sweep: Ml floats
    signed char* rimg8 = reinterpret_cast<signed char*>(smem);        // int8 sweep: Ml bytes
    double* red = reinterpret_cast<double*>(smem + (size_t)Ml * (I8 ? 1 : 4));  // 8 doubles (the last four: the maxima, as floats)
    float* wlv = reinterpret_cast<float*>(red + 8);                   // [NL][4]
    int* wli = reinterpret_cast<int*>(wlv + NL * LC);
    // residual: Float64 norm (fixed order), and its image -- f32, or (int8 sweep) the largest magnitude first
    double n2 = 0.0;
    float amax = 0.0f;
    const bool onepass = Ml <= 4 * kSweepThreads * PR;
    for (int base = 0; base < Ml; base += 4 * kSweepThreads * PR) {
        if (base > 0) r_issue(base);
#pragma unroll
        for (int p = 0; p < PR; ++p) {
            const int m0 = base + (p * kSweepThreads + tid) * 4;
            if (m0 < Ml) {
                if constexpr (!I8) {
                    f32x4 f;
                    f.x = (float)rlo[p].x;
                    f.y = (float)rlo[p].y;
                    f.z = (float)rhi[p].x;
                    f.w = (float)rhi[p].y;
                    *reinterpret_cast<f32x4*>(rimgf + rf_slot(m0)) = f;
                } else {
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf((float)rlo[p].x), fabsf((float)rlo[p].y))), fmaxf(fabsf((float)rhi[p].x), fabsf((float)rhi[p].y)));
                }
                n2 = fma(rlo[p].x, rlo[p].x, n2);
                n2 = fma(rlo[p].y, rlo[p].y, n2);
                n2 = fma(rhi[p].x, rhi[p].x, n2);
                n2 = fma(rhi[p].y, rhi[p].y, n2);
            }
        }
    }
    // sum over the workgroup in a fixed order; barriers that order LDS only (a __syncthreads would drain the prefetch)
    for (int sft = 32; sft >= 1; sft >>= 1) n2 += __shfl_xor(n2, sft, kWave);
    if constexpr (I8)
        for (int sft = 32; sft >= 1; sft >>= 1) amax = fmaxf(amax, __shfl_xor(amax, sft, kWave));
    if (lane == 0) {
        red[wave] = n2;
        if constexpr (I8) reinterpret_cast<float*>(red + 4)[wave] = amax;
    }
    lds_barrier();
    n2 = (red[0] + red[1]) + (red[2] + red[3]);
    float scale = 1.0f;
    if constexpr (I8) {
        const float* fm = reinterpret_cast<const float*>(red + 4);
        const float rstep = i8_step(fmaxf(fmaxf(fm[0], fm[1]), fmaxf(fm[2], fm[3])));
        const float inv = 1.0f / rstep;
        scale = astep * rstep * (1.0f + 0x1p-20f);
        if (bid == 0 && tid == 0) st->rstep = rstep;
        for (int base = 0; base < Ml; base += 4 * kSweepThreads * PR) {
            if (!onepass) r_issue(base);  // (one pass: the registers still hold it)
#pragma unroll
            for (int p = 0; p < PR; ++p) {
                const int m0 = base + (p * kSweepThreads + tid) * 4;
                if (m0 < Ml) {
                    const int q0 = max(-127, min(127, __float2int_rn((float)rlo[p].x * inv))), q1 = max(-127, min(127, __float2int_rn((float)rlo[p].y * inv)));
                    const int q2 = max(-127, min(127, __float2int_rn((float)rhi[p].x * inv))), q3 = max(-127, min(127, __float2int_rn((float)rhi[p].y * inv)));
                    *reinterpret_cast<int*>(rimg8 + m0) = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
                }
            }
        }
        lds_barrier();
    }
    if (bid == 0 && tid == 0) st->rnorm2 = n2;
    if (check_eps && !(sqrt(n2) >= eps)) {  // norm(residual!) >= eps || break (src/matchingpursuit.jl:79)
        if (bid == 0 && tid == 0) st->done |= STOP_EPS;
        return;
    }
    const f32x4* rs = reinterpret_cast<const f32x4*>(rimgf);
    const i32x4* rs8 = reinterpret_cast<const i32x4*>(rimg8);
    float tv[LC];
    int ti[LC];
#pragma unroll
    for (int q = 0; q < LC; ++q) {
        tv[q] = -1.0f;
        ti[q] = 0x7fffffff;
    }
    float acc0[C], acc1[C];
    int acci[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc0[c] = acc1[c] = 0.0f;
        acci[c] = 0;
    }
    int cb = 0;  // block of the item being multiplied
    bool alive = true;
    // transposing butterfly over the C column sums of the lanes: row = lane / 16 ends with the sum of column row (C = 4) / row / 2
    auto butterfly = [&](auto (&a)[C]) {
        auto s0 = a[0];
        if constexpr (C == 4) {
            decltype(s0) s1;
            {
                const bool hi = lane & 32;
                const auto k0 = hi ? a[2] : a[0], k1 = hi ? a[3] : a[1];
                const auto h0 = hi ? a[0] : a[2], h1 = hi ? a[1] : a[3];
                s0 = k0 + __shfl_xor(h0, 32, kWave);
                s1 = k1 + __shfl_xor(h1, 32, kWave);
            }
            {
                const bool hi = lane & 16;
                const auto k = hi ? s1 : s0, h = hi ? s0 : s1;
                s0 = k + __shfl_xor(h, 16, kWave);
            }
        } else {  // two columns: a 32-lane half each (its second row lists nothing)
            const bool hi = lane & 32;
            const auto k = hi ? a[1] : a[0], h = hi ? a[0] : a[1];
            s0 = k + __shfl_xor(h, 32, kWave);
            s0 += __shfl_xor(s0, 16, kWave);
        }
        return row16_sum(s0);
    };
    while (alive) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (alive && sg[d] >= 0) {
                load_next(buf[(d + D - 1) % D], sg[(d + D - 1) % D]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = cb * U + u;
                    if (FULL || t < nchunk) {
                        if constexpr (I8) {
                            const i32x4 rr = rs8[t * kWave + lane];
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                const i32x4 a = __builtin_bit_cast(i32x4, buf[d][u][c]);
                                acci[c] = __builtin_amdgcn_sdot4(a.x, rr.x, acci[c], false);
                                acci[c] = __builtin_amdgcn_sdot4(a.y, rr.y, acci[c], false);
                                acci[c] = __builtin_amdgcn_sdot4(a.z, rr.z, acci[c], false);
                                acci[c] = __builtin_amdgcn_sdot4(a.w, rr.w, acci[c], false);
                            }
                        } else {
                            const f32x4 r0 = rs[(t * 2 + 0) * kWave + lane], r1 = rs[(t * 2 + 1) * kWave + lane];
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                using VE = typename std::conditional<IMG == kOpF16, f16x8v, V>::type;
                                const VE a = __builtin_bit_cast(VE, buf[d][u][c]);
                                acc0[c] = fmaf((float)a[0], r0.x, acc0[c]);
                                acc1[c] = fmaf((float)a[1], r0.y, acc1[c]);
                                acc0[c] = fmaf((float)a[2], r0.z, acc0[c]);
                                acc1[c] = fmaf((float)a[3], r0.w, acc1[c]);
                                acc0[c] = fmaf((float)a[4], r1.x, acc0[c]);
                                acc1[c] = fmaf((float)a[5], r1.y, acc1[c]);
                                acc0[c] = fmaf((float)a[6], r1.z, acc0[c]);
                                acc1[c] = fmaf((float)a[7], r1.w, acc1[c]);
                            }
                        }
                    }
                }
                if (++cb == nblocks) {  // the group is complete
                    cb = 0;
                    float vabs;
                    if constexpr (I8) {
                        const int si = butterfly(acci);
#pragma unroll
                        for (int c = 0; c < C; ++c) acci[c] = 0;
                        vabs = (float)abs(si) * scale;
                    } else {
                        float a[C];
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            a[c] = acc0[c] + acc1[c];
                            acc0[c] = acc1[c] = 0.0f;
                        }
                        vabs = fabsf(butterfly(a));
                        if constexpr (IMG == kOpF16) vabs *= astep;  // (1 / the image's power-of-two scale: exact)
                    }
                    const int64_t col = (int64_t)sg[d] * C + (C == 4 ? (lane >> 4) : (lane >> 5));
                    float v = col < N && (C == 4 || !(lane & 16)) ? vabs : -1.0f;
                    int i = (int)col;
                    // the row's running 4 largest (every lane of the row holds the same list); an equal value: the lower index
#pragma unroll
                    for (int q = 0; q < LC; ++q) {
                        const bool up = v > tv[q] || (v == tv[q] && i < ti[q]);
                        const float ov = up ? tv[q] : v;
                        const int oi = up ? ti[q] : i;
                        tv[q] = up ? v : tv[q];
                        ti[q] = up ? i : ti[q];
                        v = ov;
                        i = oi;
                    }
                }
            } else {
                alive = false;
            }
        }
    }
    if ((lane & 15) == 0) {
        const int l = wave * (kWave / 16) + (lane >> 4);
#pragma unroll
        for (int q = 0; q < LC; ++q) {
            wlv[l * LC + q] = tv[q];
            wli[l * LC + q] = ti[q];
        }
    }
    __syncthreads();
    if (tid < NL * LC) {  // one entry per thread: an entry's rank in (value desc, index asc, slot asc) is its place
        static_assert(NL * LC <= kSweepThreads && LC % 4 == 0, "the workgroup ranks its lists in one go");
        const float v = wlv[tid];
        const int i = wli[tid];
        int rank = 0;
#pragma unroll
        for (int e4 = 0; e4 < NL * LC / 4; ++e4) {  // (broadcast reads, all issued before the first compare: the rolled loop cost 3.5 us)
            const f32x4 ve = reinterpret_cast<const f32x4*>(wlv)[e4];
            const i32x4 ie = reinterpret_cast<const i32x4*>(wli)[e4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e4 * 4 + q;
                rank += (ve[q] > v || (ve[q] == v && (ie[q] < i || (ie[q] == i && e < tid)))) ? 1 : 0;
            }
        }
        if (rank < LC) {
            cand_val[bid * LC + rank] = v;
            cand_idx[bid * LC + rank] = i;
        }
    }
}
template <int U, int D, bool FULL, int C = kScrCols, int LC = kScrCand>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_bf16(const __bf16* __restrict__ Ab, int Mk, int64_t N,
                                                              const double* __restrict__ r, int Mr, float* __restrict__ cand_val,
                                                              int* __restrict__ cand_idx, DevState* st, double eps, int check_eps,
                                                              int skipmask, unsigned* __restrict__ tickets) {
    extern __shared__ __attribute__((aligned(16))) char smem_sweep[];  // the residual image | reduction scratch | the lists
    sweep_img_body<U, D, FULL, C, kOpBf16, LC>(reinterpret_cast<const char*>(Ab), Mk, N, r, Mr, cand_val, cand_idx, st, eps, check_eps, skipmask, tickets,
                                           0.0f, smem_sweep);
}
template <int U, int D, bool FULL, int C = kScrCols, int LC = kScrCand>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_f16(const unsigned short* __restrict__ Ah /* binary16 bit patterns */, int Mk, int64_t N,
                                                             const double* __restrict__ r, int Mr, float* __restrict__ cand_val,
                                                             int* __restrict__ cand_idx, DevState* st, double eps, int check_eps,
                                                             int skipmask, unsigned* __restrict__ tickets, float inv_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_sweep[];
    sweep_img_body<U, D, FULL, C, kOpF16, LC>(reinterpret_cast<const char*>(Ah), Mk, N, r, Mr, cand_val, cand_idx, st, eps, check_eps, skipmask, tickets,
                                          inv_scale, smem_sweep);
}
template <int U, int D, bool FULL, int C = kScrCols, int LC = kScrCand>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_i8(const signed char* __restrict__ A8, int Mk8, int64_t N,
                                                            const double* __restrict__ r, int Mr, float* __restrict__ cand_val,
                                                            int* __restrict__ cand_idx, DevState* st, double eps, int check_eps,
                                                            int skipmask, unsigned* __restrict__ tickets, float astep) {
    extern __shared__ __attribute__((aligned(16))) char smem_sweep[];
    sweep_img_body<U, D, FULL, C, kOpI8, LC>(reinterpret_cast<const char*>(A8), Mk8, N, r, Mr, cand_val, cand_idx, st, eps, check_eps, skipmask, tickets,
                                         astep, smem_sweep);
}
inline size_t sweep_i8_lds_bytes(int Mk8, int lc = kScrCand) {
    const int nchunk = (Mk8 + 1023) / 1024;
    return (size_t)nchunk * 1024 + 8 * sizeof(double) + (kSweepThreads / 16) * lc * 8 + 64;
}
inline size_t sweep_bf16_lds_bytes(int Mk, int lc = kScrCand) {
    const int nchunk = (Mk + 511) / 512;
    return (size_t)nchunk * 512 * sizeof(float) + 8 * sizeof(double) + (kSweepThreads / 16) * lc * 8 + 64;
}

// the sweep's candidates of one pick workgroup: thread t holds entries t, t + NT, ... (2048 / NT of them) in registers when all
// fit, loaded by one unrolled batch; `each` visits (value, atom, entry number) from the registers or, for a larger sweep grid,
// from memory
template <int NT = 256>
struct PickCands {
    static constexpr int kPickEpl = 2048 / NT;
    float v[kPickEpl];
    int i[kPickEpl];
    bool inreg;
    __device__ __forceinline__ void load(const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int ncand, int tid) {
        inreg = ncand <= NT * kPickEpl;
#pragma unroll
        for (int e = 0; e < kPickEpl; ++e) {
            const int t = tid + NT * e;
            const bool ok = inreg && t < ncand;
            v[e] = ok ? cand_val[t] : -1.0f;
            i[e] = ok ? cand_idx[t] : 0x7fffffff;
        }
    }
    template <typename F>
    __device__ __forceinline__ void each(const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int ncand, int tid, F&& f) const {
        if (inreg) {
#pragma unroll
            for (int e = 0; e < kPickEpl; ++e) f(v[e], i[e], tid + NT * e);
        } else {
            for (int t = tid; t < ncand; t += NT) f(cand_val[t], cand_idx[t], t);
        }
    }
};

// The pick of one signal by ONE workgroup: k_b_pick's window / rescoring / certificate (see there) on the sweep workgroups'
// candidates; ||r||^2 comes from the sweep's prologue (st->rnorm2).  Publishes (|<a, r>| exact, atom) as pval[0] / pidx[0]: the
// one "partial" k_qr1 (mode 1, nblk = 1) takes its arg-max from -- its guards (already selected, full support) apply as in the
// exact path.  dynamic LDS: the Float64 residual image (r_slot layout).
// NT threads: every wave rescores one window column at a time, so 16 waves finish a 10-15 column window (the int8 image's) in one
// round where 4 waves needed three or four.
template <typename TA, int U, int NT = 1024>
__global__ __launch_bounds__(NT) void k_pick1(const TA* __restrict__ A, int64_t ld, int Mv, const float* __restrict__ cand_val,
                                               const int* __restrict__ cand_idx, int ncand, DevState* st, const double* __restrict__ r,
                                               int Mr, double* __restrict__ pval, int* __restrict__ pidx, double cert_abs, double cert_rel,
                                               int kwin, int skipmask, unsigned* __restrict__ tickets, int nparts, double cert_abs2,
                                               int mp_select) {
    extern __shared__ __attribute__((aligned(16))) double rimg[];
    constexpr int NWV = NT / kWave;
    __shared__ double sc[NWV];
    __shared__ double red[kWinMax];
    __shared__ int wi_[kWinMax];
    __shared__ float fsc[NWV];
    __shared__ int cnt;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = tid; p < nparts; p += NT) tickets[p * kScrTicketStride] = 0u;  // (the sweep has ended: kernel boundary) ready for the next one
    if (st->done & skipmask) return;
    if (tid == 0) cnt = 0;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    for (int m0 = 4 * tid; m0 < Mlds; m0 += 4 * NT) {
        f64x2 lo = (f64x2)0.0, hi = (f64x2)0.0;
        if (m0 < Mr) {
            lo = reinterpret_cast<const f64x2*>(r + m0)[0];
            hi = reinterpret_cast<const f64x2*>(r + m0)[1];
        }
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0)) = lo;
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0 + 2)) = hi;
    }
    const double n2 = st->rnorm2;
    // the candidates: one unrolled batch of loads into registers (a rolled loop waits for every load in turn) when they fit
    PickCands<NT> pc;
    pc.load(cand_val, cand_idx, ncand, tid);
    float m1 = -1.0f;
    pc.each(cand_val, cand_idx, ncand, tid, [&](float v, int, int) { m1 = fmaxf(m1, v); });
    for (int sft = 32; sft >= 1; sft >>= 1) m1 = fmaxf(m1, __shfl_xor(m1, sft, kWave));
    if (lane == 0) fsc[wave] = m1;
    __syncthreads();  // (also: cnt = 0 and the residual image are visible)
    m1 = fsc[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m1 = fmaxf(m1, fsc[w]);
    if (!(m1 >= 0.0f)) {  // no candidate at all: k_qr1 sees an invalid atom and stops the solve as the exact path would
        if (tid == 0) {
            pval[0] = -1.0;
            pidx[0] = 0x7fffffff;
        }
        return;
    }
    // (int8 sweep: the residual image's own rounding adds cert_abs2 * its step)
    const double dabs = cert_abs2 > 0.0 ? sqrt(cert_abs * cert_abs * n2 + cert_abs2 * cert_abs2 * (double)st->rstep * (double)st->rstep) : cert_abs * sqrt(n2);
    const double lb1 = (double)m1 - dabs - cert_rel * (double)m1;
    double cb = -1.0;
    pc.each(cand_val, cand_idx, ncand, tid, [&](float v, int i, int t) {
        if (!(v >= 0.0f)) return;
        const double ub = (double)v + dabs + cert_rel * (double)v;
        if (ub >= lb1) {
            const int pos = atomicAdd(&cnt, 1);
            if (pos < kwin) wi_[pos] = i;
            if ((t & (kScrCand - 1)) == kScrCand - 1) cb = fmax(cb, ub);  // atoms hidden behind a workgroup's last candidate
        } else {
            cb = fmax(cb, ub);
        }
    });
    for (int sft = 32; sft >= 1; sft >>= 1) cb = fmax(cb, shx(cb, sft));
    __syncthreads();
    if (lane == 0) sc[wave] = cb;
    __syncthreads();
    cb = sc[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) cb = fmax(cb, sc[w]);
    const int nall = cnt;
    const int nw = min(nall, kwin);
    for (int q = wave; q < nw; q += NWV) {
        const double exq = wave_col_dot<TA, U>(A + (int64_t)wi_[q] * ld, Mv, nchunk, rimg, lane);
        if (lane == 0) red[q] = exq;
    }
    __syncthreads();
    if (tid == 0) {
        int besti = 0x7fffffff;
        double bestv = -1.0;
        for (int q = 0; q < nw; ++q) {
            const int c = wi_[q];
            const double v = fabs(red[q]);
            if (v > bestv || (v == bestv && c < besti)) {
                bestv = v;
                besti = c;
            }
        }
        if (!(nall <= kwin && (cb < 0.0 || bestv > cb))) st->uncertain += 1;
        pval[0] = bestv;
        pidx[0] = besti;
        if (mp_select) {  // Matching Pursuit: the selection kernel's job too (k_select, mode 0: no guards, atoms may repeat)
            double signedv = 0.0;
            for (int q = 0; q < nw; ++q)
                if (wi_[q] == besti) signedv = red[q];
            st->cand = besti;
            st->cval = signedv;
            st->j = st->nsel;
            st->go = 1;
        }
    }
}

// The TOP-S pick of one signal by ONE workgroup (GOMP: partialsortperm(abs(A'r), 1:S, rev=true), src/matchingpursuit.jl:192 --
// descending |c|, ties by ascending index): k_pick1's scheme around the S-th largest screened value instead of the largest.
// Window = every candidate whose exact value could reach the exact value of the S-th best; all of it is rescored in Float64
// and ranked; certified when the S-th exact value beats the bound of every atom that was not rescored (a sweep workgroup's 4th
// entry inside the window hides the atoms behind it: its bound counts).  Output: cands[0..S), cvals, ncands -- what k_top_merge
// hands to the append kernels of the exact path.  S <= kTopSmall.
template <typename TA, int U, int NT = 256>
__global__ __launch_bounds__(NT) void k_pickS(const TA* __restrict__ A, int64_t ld, int Mv, const float* __restrict__ cand_val,
                                               const int* __restrict__ cand_idx, int ncand, DevState* st, const double* __restrict__ r,
                                               int Mr, int S, int* __restrict__ cands, double* __restrict__ cvals,
                                               int* __restrict__ ncands, double cert_abs, double cert_rel, int kwin, int skipmask,
                                               unsigned* __restrict__ tickets, int nparts, double cert_abs2) {
    extern __shared__ __attribute__((aligned(16))) double rimg[];
    constexpr int NWV = NT / kWave;
    __shared__ double sc[NWV];
    __shared__ double red[kWinMax];
    __shared__ int wi_[kWinMax];
    __shared__ float fsv[NWV];
    __shared__ int fsi[NWV];
    __shared__ int cnt;
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = tid; p < nparts; p += NT) tickets[p * kScrTicketStride] = 0u;
    if (st->done & skipmask) return;
    if (tid == 0) cnt = 0;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    for (int m0 = 4 * tid; m0 < Mlds; m0 += 4 * NT) {
        f64x2 lo = (f64x2)0.0, hi = (f64x2)0.0;
        if (m0 < Mr) {
            lo = reinterpret_cast<const f64x2*>(r + m0)[0];
            hi = reinterpret_cast<const f64x2*>(r + m0)[1];
        }
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0)) = lo;
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0 + 2)) = hi;
    }
    const double n2 = st->rnorm2;
    // the S-th largest screened value: S rounds of "the best entry that is worse than the previous round's" in the total
    // order (value desc, index asc) -- indices are distinct, so nothing has to be marked
    float pv = __builtin_inff(), mS = -1.0f;
    int pi = -1;
    PickCands<NT> pc;
    pc.load(cand_val, cand_idx, ncand, tid);
    for (int sidx = 0; sidx < S; ++sidx) {
        float bv = -1.0f;
        int bi = 0x7fffffff;
        pc.each(cand_val, cand_idx, ncand, tid, [&](float v, int i, int) {
            if (!(v >= 0.0f)) return;
            const bool after_prev = v < pv || (v == pv && i > pi);
            if (after_prev && (v > bv || (v == bv && i < bi))) {
                bv = v;
                bi = i;
            }
        });
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft, kWave);
            const int oi = __shfl_xor(bi, sft, kWave);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        __syncthreads();  // (round 0: also cnt = 0 and the residual image)
        if (lane == 0) {
            fsv[wave] = bv;
            fsi[wave] = bi;
        }
        __syncthreads();
        bv = fsv[0];
        bi = fsi[0];
        for (int w = 1; w < NWV; ++w)
            if (fsv[w] > bv || (fsv[w] == bv && fsi[w] < bi)) {
                bv = fsv[w];
                bi = fsi[w];
            }
        if (!(bv >= 0.0f)) break;  // fewer than S candidates (uniform)
        pv = bv;
        pi = bi;
        mS = bv;
        if (sidx + 1 < S) mS = -1.0f;  // (only the S-th counts; with fewer candidates everything is in the window)
    }
    __syncthreads();
    // (int8 sweep: the residual image's own rounding adds cert_abs2 * its step)
    const double dabs = cert_abs2 > 0.0 ? sqrt(cert_abs * cert_abs * n2 + cert_abs2 * cert_abs2 * (double)st->rstep * (double)st->rstep) : cert_abs * sqrt(n2);
    const double lbS = mS >= 0.0f ? (double)mS - dabs - cert_rel * (double)mS : -1.0;
    double cb = -1.0;
    pc.each(cand_val, cand_idx, ncand, tid, [&](float v, int i, int t) {
        if (!(v >= 0.0f)) return;
        const double ub = (double)v + dabs + cert_rel * (double)v;
        if (ub >= lbS) {
            const int pos = atomicAdd(&cnt, 1);
            if (pos < kwin) wi_[pos] = i;
            if ((t & (kScrCand - 1)) == kScrCand - 1) cb = fmax(cb, ub);  // atoms hidden behind a workgroup's last candidate
        } else {
            cb = fmax(cb, ub);
        }
    });
    for (int sft = 32; sft >= 1; sft >>= 1) cb = fmax(cb, shx(cb, sft));
    __syncthreads();
    if (lane == 0) sc[wave] = cb;
    __syncthreads();
    cb = sc[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) cb = fmax(cb, sc[w]);
    const int nall = cnt;
    const int nw = min(nall, kwin);
    for (int q = wave; q < nw; q += NWV) {
        const double exq = wave_col_dot<TA, U>(A + (int64_t)wi_[q] * ld, Mv, nchunk, rimg, lane);
        if (lane == 0) red[q] = fabs(exq);
    }
    __syncthreads();
    // rank of every rescored entry (value desc, index asc); the first S are the step's atoms, in that order
    const int Seff = min(S, nw);
    if (tid < nw) {
        const double v = red[tid];
        const int c = wi_[tid];
        int rank = 0;
        for (int q = 0; q < nw; ++q) rank += (red[q] > v || (red[q] == v && wi_[q] < c)) ? 1 : 0;
        if (rank < Seff) {
            cands[rank] = c;
            cvals[rank] = v;
        }
        if (rank == Seff - 1 && !(nall <= kwin && (cb < 0.0 || (nw >= S && v > cb)))) {
            st->uncertain += 1;
        }
    }
    if (tid == 0) {
        *ncands = Seff;
        if (nw == 0) st->uncertain += 1;  // (cannot happen with a non-empty dictionary: the best candidate is always in the window)
    }
}


// ---------------------------------------------------------------------------------------------
// Subspace Pursuit's acquisition on the screened sweep: the k largest |<a_j, r>| as a SET (sp_acquisition!, src/twostage.jl:67-72;
// argmaxinner!(P, k), src/matchingpursuit.jl:187-193), k in the hundreds.  The sweep lists kScrCandK candidates per workgroup;
// the selection reuses the exact path's top-S machinery (radix select on an N-vector of values) twice:
//   k_spk_scatter   cvec := 0, cvec[candidate] := its screened value           -> launch_topS: the k-th largest SCREENED value
//   k_spk_rescore   every candidate whose upper bound reaches the k-th value's lower bound is rescored exactly (one wave per
//                   column, Float64): cvec[candidate] := |exact|; the others := 0 and their bound goes into cb (as does the
//                   bound of a workgroup's last listed candidate when it is inside the window: atoms hidden behind it)
//                                                                               -> launch_topS: the k largest EXACT values
//   k_spk_cert      certified when the k-th exact value beats cb (and there are k of them); else the acquisition is repeated
//                   with the exact sweep (the flag is read with the selection).
constexpr int kScrCandK = 16;

__global__ __launch_bounds__(256) void k_spk_scatter(const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int ncand,
                                                     double* __restrict__ cvec, unsigned long long* __restrict__ cb_bits, int* __restrict__ flag) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) {
        *cb_bits = 0ull;
        *flag = 0;
    }
    if (t < ncand) {
        const float v = cand_val[t];
        if (v >= 0.0f) cvec[cand_idx[t]] = (double)v;
    }
}

// grid: workgroups of 4 waves, wave w of workgroup b takes candidates b * 4 + w, + 4 * gridDim.x, ...
template <typename TA, int U>
__global__ __launch_bounds__(256) void k_spk_rescore(const TA* __restrict__ A, int64_t ld, int Mv, const float* __restrict__ cand_val,
                                                     const int* __restrict__ cand_idx, int ncand, const DevState* __restrict__ st,
                                                     const double* __restrict__ r, int Mr, const double* __restrict__ cvals,
                                                     const int* __restrict__ ncands, int k, double* __restrict__ cvec,
                                                     unsigned long long* __restrict__ cb_bits, double cert_abs, double cert_rel,
                                                     double cert_abs2) {
    extern __shared__ __attribute__((aligned(16))) double rimg[];
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    for (int m0 = 4 * tid; m0 < Mlds; m0 += 4 * 256) {
        f64x2 lo = (f64x2)0.0, hi = (f64x2)0.0;
        if (m0 < Mr) {
            lo = reinterpret_cast<const f64x2*>(r + m0)[0];
            hi = reinterpret_cast<const f64x2*>(r + m0)[1];
        }
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0)) = lo;
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0 + 2)) = hi;
    }
    const double n2 = st->rnorm2;
    const double dabs = cert_abs2 > 0.0 ? sqrt(cert_abs * cert_abs * n2 + cert_abs2 * cert_abs2 * (double)st->rstep * (double)st->rstep) : cert_abs * sqrt(n2);
    // the k-th largest screened value (fewer than k positive candidates: everything is in the window)
    const double mk = (*ncands >= k) ? cvals[k - 1] : 0.0;
    const double lbk = mk - dabs - cert_rel * mk;
    __syncthreads();
    double cb = 0.0;
    for (int t = blockIdx.x * 4 + wave; t < ncand; t += 4 * gridDim.x) {
        const float v = cand_val[t];
        if (!(v >= 0.0f)) continue;  // (wave-uniform)
        const double ub = (double)v + dabs + cert_rel * (double)v;
        const int col = cand_idx[t];
        if (ub >= lbk) {
            const double ex = wave_col_dot<TA, U>(A + (int64_t)col * ld, Mv, nchunk, rimg, lane);
            if (lane == 0) cvec[col] = fabs(ex);
            if ((t & (kScrCandK - 1)) == kScrCandK - 1) cb = fmax(cb, ub);
        } else {
            if (lane == 0) cvec[col] = 0.0;
            cb = fmax(cb, ub);
        }
    }
    if (lane == 0 && cb > 0.0) atomicMax(cb_bits, (unsigned long long)__double_as_longlong(cb));  // (non-negative doubles: their bits order like the values)
}

__global__ void k_spk_cert(const double* __restrict__ cvals, const int* __restrict__ ncands, int k, const unsigned long long* __restrict__ cb_bits,
                           int* __restrict__ flag) {
    const double cb = __longlong_as_double((long long)*cb_bits);
    const bool ok = *ncands >= k && cvals[k - 1] > cb;
    *flag = ok ? 0 : 1;
}


// <a_j, r> (signed, Float64) for a short list of atoms -- the correlations ompr needs on its SUPPORT (x.nzval = Ar[x.nzind],
// src/twostage.jl:158-160) when the sweep ran on an image and left no exact correlation vector: one wave per column.
template <typename TA, int U>
__global__ __launch_bounds__(256) void k_cols_dot(const TA* __restrict__ A, int64_t ld, int Mv, const int* __restrict__ cols, int n,
                                                  const double* __restrict__ r, int Mr, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double rimg[];
    constexpr int VEC = Vec<TA>::n;
    constexpr int ROWS = kWave * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = (Mv + ROWS - 1) / ROWS;
    const int Mlds = nchunk * ROWS;
    for (int m0 = 4 * tid; m0 < Mlds; m0 += 4 * 256) {
        f64x2 lo = (f64x2)0.0, hi = (f64x2)0.0;
        if (m0 < Mr) {
            lo = reinterpret_cast<const f64x2*>(r + m0)[0];
            hi = reinterpret_cast<const f64x2*>(r + m0)[1];
        }
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0)) = lo;
        *reinterpret_cast<f64x2*>(rimg + r_slot<VEC>(m0 + 2)) = hi;
    }
    __syncthreads();
    for (int t = blockIdx.x * 4 + wave; t < n; t += 4 * gridDim.x) {
        const double ex = wave_col_dot<TA, U>(A + (int64_t)cols[t] * ld, Mv, nchunk, rimg, lane);
        if (lane == 0) out[t] = ex;
    }
}


}  // namespace csmp
