// csmp_screen.hip -- screening GEMM kernels of the batched path + their launcher (see csmp_screen.hpp).
#include "csmp_screen.hpp"

namespace csmp {

// ---------------------------------------------------------------------------------------------
// 256 atoms x 256 signals per workgroup, 8 waves as 2 (atoms) x 4 (signals), each 128 x 64 = 8 x 4 MFMA
// tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16): 64 MFMAs per 24 ds_read_b128 and K-tile of 64, twice the
// reuse of the 128^2 kernels.  The operand tiles go global -> LDS directly (global_load_lds_dwordx4: no
// staging registers, no ds_write pass); an LDS-DMA writes wave-uniform base + lane*16, i.e. 1 KiB = 8
// rows of 128 B per wave-instruction, so an XOR swizzle (chunk ^= (row>>1)&7, which makes every 16-lane group of a ds_read_b128 of the 16x16x32
// fragments hit 16 distinct 16-byte slots of the unpadded 128-byte rows) is applied
// to the per-lane SOURCE address.  LDS: 2 buffers x (A 32 KiB + R 32 KiB) = 128 KiB, one workgroup per
// CU.  (K-loop schedule: see k_b_screen256p below.)  Each (wave row, signal) pair lives in ONE wave, so the top-4 epilogue needs
// no LDS: it is written per 128-atom half tile, the granularity k_b_pick expects.
constexpr int kBT2 = 256;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
// ---------------------------------------------------------------------------------------------
// k_b_screen256p: the 256 x 256 tile in the eight-phase schedule (two K-tiles of 64 per loop iteration, four phases each).
// A phase is one quarter of a wave's 128 x 64 output (64 atoms x 32 signals = 4 x 2 MFMA tiles) over one K-tile:
//     fragment reads of the quarter's new operand sub-tile | one LDS-DMA unit issued | counted s_waitcnt vmcnt(8) |
//     s_barrier | 16 MFMAs | s_barrier
// Quarters of a K-tile run (a0,b0) -> (a0,b1) -> (a1,b1) -> (a1,b0): phase 1 reads a0 and b0 (12 ds_read_b128), phase 2 b1
// (4), phase 3 a1 (8), phase 4 b0 again (4: keeping it would cost 16 registers the kernel does not have).  The two wave rows (waves 0-3 / 4-7: the two waves of every SIMD) run
// ONE barrier apart, so that on each SIMD one wave issues MFMAs while its partner reads fragments and issues DMAs.
// Staging: the operand tiles are cut into four 16-KiB units in the order they are consumed -- UA0 (the a0 rows of both
// wave rows), UB0, UB1, UA1 -- one unit = 2 global_load_lds_dwordx4 per thread, one unit issued per phase, FOUR phases
// ahead of the phase that reads it (two LDS stages of four units each).  After the issue every phase waits vmcnt(4): all
// but the two youngest units have landed, which includes the unit the NEXT phase reads (the wait precedes the barrier,
// the read follows it).  A unit's LDS region is rewritten at least two phases after its last read (UB0, which phase 4
// reads a second time, sets the lead: six phases ahead would rewrite it in the very phase that still reads it).
// No vmcnt(0), no __syncthreads() in the loop: the DMAs stay in flight across the barriers.
//
// OP selects the operands (kOpBf16 / kOpI8 / kOpF16).  kOpF16: v_mfma_f32_16x16x32_f16 on binary16 images -- the same instruction
// shape, rate and data movement as bf16 with eleven significand bits instead of eight: the rigorous certificate's bound is
// 2^-10 instead of 2^-7 of |a||r| (csmp_omp_batch_mfma, host/batched.hpp).  Both images carry power-of-two scales (the dictionary
// one, every residual its own) that put their largest entry in [2^14, 2^15): binary16's narrow exponent range loses nothing, and the
// epilogue multiplies the candidate values by scale[signal] = 1 / (dictionary scale x residual scale), an exact operation.
// kOpI8: the SAME data movement on an int8 image (a row of Mk "bf16 slots" is 2 Mk int8 values; an LDS row of 128 bytes is a
// K-tile of 128 instead of 64; a fragment of 16 bytes is 16 k-values instead of 8) with v_mfma_i32_16x16x64_i8 -- the same
// cycles per instruction for twice the k, i.e. half the K-loop.  The accumulators are exact integers; the epilogue ranks
// |acc| (one scale per signal: a rank inside a signal does not depend on it) and hands on |acc| * scale[signal], scale =
// (dictionary step) x (the signal's residual step), see csmp_batched.hpp.
using i32x4s = __attribute__((ext_vector_type(4))) int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
// DIAG (tools/probes/screen_epilogue_probe.hip only; the library instantiates 0): 1 = no epilogue (one accumulator sum per lane is
// stored, so that the K-loop stays alive): the epilogue's share of the launch; 2 = the full kernel + clock stamps around the K-loop
// (s_memtime / s_memrealtime per workgroup into cand_idx's tail): the clock the chip holds under this loop.
template <int OP, int DIAG = 0>
__global__ __launch_bounds__(512) void k_b_screen256p(const __bf16* __restrict__ Ab, const __bf16* __restrict__ Rb, int Mk,
                                                      int n_at2, int n_st2, int64_t N, int n_atiles128,
                                                      float* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                      const float* __restrict__ sigscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
    int atile, stile;
    {
        const int bid = blockIdx.x;
        if ((n_at2 & 7) == 0) {
            const int xcd = bid & 7, local = bid >> 3;
            stile = local % n_st2;
            atile = (local / n_st2) * 8 + xcd;
        } else {
            stile = bid % n_st2;
            atile = bid / n_st2;
        }
    }
    const __bf16* gA = Ab + (int64_t)atile * kBT2 * Mk;
    const __bf16* gR = Rb + (int64_t)stile * kBT2 * Mk;
    f32x4s acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4s)0.0f;
    // DMA maps.  Unit kinds: 0 = UA0, 1 = UB0, 2 = UB1, 3 = UA1.  A unit is 16 pieces of 8 rows; wave w issues pieces 2w, 2w+1.
    // urow(kind, piece): first tile row of the piece.  Lane l lands at row + (l >> 3), slot l & 7, and fetches chunk slot ^ key(row).
    auto urow = [](int kind, int pc) -> int {
        switch (kind) {
            case 0: return (pc < 8 ? 0 : 128) + (pc & 7) * 8;
            case 3: return (pc < 8 ? 64 : 192) + (pc & 7) * 8;
            case 1: return (pc >> 2) * 64 + (pc & 3) * 8;
            default: return (pc >> 2) * 64 + 32 + (pc & 3) * 8;
        }
    };
    // Source offset of lane l for a piece starting at row r0 (a multiple of 8): (r0 + (l >> 3)) * Mk + 8 * chunk with
    // chunk = (l & 7) ^ (((r0 + (l >> 3)) >> 1) & 7) = (l & 7) ^ (l >> 4) ^ ((r0 >> 1) & 4): the lane part has only two
    // variants (bit 3 of r0), everything else is wave-uniform and lives in scalar registers.
    int lanepart[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) lanepart[v] = (lane >> 3) * Mk + (((lane & 7) ^ (lane >> 4) ^ (4 * v)) << 3);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nkb = Mk / kBK, U = 4 * nkb;  // (nkb even, >= 2: the launcher guarantees it)
    auto issue = [&](int u) {
        const int t = u >> 2, kind = u & 3;
        const bool isA = kind == 0 || kind == 3;
        const __bf16* g = (isA ? gA : gR) + t * kBK;
        char* l = smem + (size_t)(t & 1) * 65536 + (isA ? 0 : 32768);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r0 = urow(kind, 2 * wv + i);  // (scalar)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (int64_t)r0 * Mk + lanepart[(r0 >> 3) & 1]), (lds_void_t*)(l + r0 * 128), 16, 0, 0);
        }
    };
    const int key = (fr >> 1) & 7;
    const int co0 = ((0 * 4 + fq) ^ key) << 4, co1 = ((1 * 4 + fq) ^ key) << 4;
    const char* laW = smem + (wr * 128 + fr) * 128;           // + stage * 65536 + m * 2048 + co
    const char* lrW = smem + 32768 + (wc * 64 + fr) * 128;    // + stage * 65536 + n * 2048 + co
    bf16x8 a[4][2], b[2][2];  // the current A sub-tile (4 m-tiles x 2 k-halves) and B sub-tile (2 n-tiles x 2 k-halves)
    constexpr bool I8 = OP == kOpI8;
    auto mma = [](const bf16x8& x, const bf16x8& y, const f32x4s& c) -> f32x4s {
        if constexpr (OP == kOpF16)
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
        else if constexpr (I8)
            return __builtin_bit_cast(f32x4s, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4s, x), __builtin_bit_cast(i32x4s, y),
                                                                                    __builtin_bit_cast(i32x4s, c), 0, 0, 0));
        else
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
    };
#define CSMP_PH(STAGE, QA, QB, LOADA, LOADB, UNIT, WAITN)                                                        \
    {                                                                                                            \
        if (LOADB) {                                                                                             \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                      \
                b[n][0] = *reinterpret_cast<const bf16x8*>(lrW + (STAGE) * 65536 + ((QB) * 2 + n) * 2048 + co0); \
                b[n][1] = *reinterpret_cast<const bf16x8*>(lrW + (STAGE) * 65536 + ((QB) * 2 + n) * 2048 + co1); \
            }                                                                                                    \
        }                                                                                                        \
        if (LOADA) {                                                                                             \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                      \
                a[m][0] = *reinterpret_cast<const bf16x8*>(laW + (STAGE) * 65536 + ((QA) * 4 + m) * 2048 + co0); \
                a[m][1] = *reinterpret_cast<const bf16x8*>(laW + (STAGE) * 65536 + ((QA) * 4 + m) * 2048 + co1); \
            }                                                                                                    \
        }                                                                                                        \
        if ((UNIT) < U) issue(UNIT);                                                                             \
        asm volatile("s_waitcnt vmcnt(" #WAITN ")" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                        \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                    \
                    acc[(QA) * 4 + m][(QB) * 2 + n] = mma(a[m][kk], b[n][kk], acc[(QA) * 4 + m][(QB) * 2 + n]);         \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                            \
    }
    unsigned long long diag_t0 = 0, diag_r0 = 0;
    if constexpr (DIAG == 2) {
        diag_t0 = __builtin_amdgcn_s_memtime();
        diag_r0 = __builtin_amdgcn_s_memrealtime();
    }
    // prologue: units 0 .. 3 (K-tile 0); units 0, 1 must have landed
#pragma unroll
    for (int u = 0; u < 4; ++u) issue(u);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first
    int t = 0;
    for (; t + 2 < nkb; t += 2) {
        const int u0 = 4 * t + 4;
        CSMP_PH(0, 0, 0, true, true, u0 + 0, 4)
        CSMP_PH(0, 0, 1, false, true, u0 + 1, 4)
        CSMP_PH(0, 1, 1, true, false, u0 + 2, 4)
        CSMP_PH(0, 1, 0, false, true, u0 + 3, 4)
        CSMP_PH(1, 0, 0, true, true, u0 + 4, 4)
        CSMP_PH(1, 0, 1, false, true, u0 + 5, 4)
        CSMP_PH(1, 1, 1, true, false, u0 + 6, 4)
        CSMP_PH(1, 1, 0, false, true, u0 + 7, 4)
    }
    {   // the last two K-tiles: unit U - 1 is the last to issue (fourth phase); afterwards the waits count down
        const int u0 = 4 * t + 4;
        CSMP_PH(0, 0, 0, true, true, u0 + 0, 4)
        CSMP_PH(0, 0, 1, false, true, u0 + 1, 4)
        CSMP_PH(0, 1, 1, true, false, u0 + 2, 4)
        CSMP_PH(0, 1, 0, false, true, u0 + 3, 4)
        CSMP_PH(1, 0, 0, true, true, u0 + 4, 2)
        CSMP_PH(1, 0, 1, false, true, u0 + 5, 0)
        CSMP_PH(1, 1, 1, true, false, u0 + 6, 0)
        CSMP_PH(1, 1, 0, false, true, u0 + 7, 0)
    }
#undef CSMP_PH
    if (wr == 0) __builtin_amdgcn_s_barrier();  // (barrier counts of the two wave rows match again)
    if constexpr (DIAG == 2) {
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(cand_idx + (int64_t)n_st2 * kBT2 * n_atiles128 * kTileCand) + 2 * (int64_t)blockIdx.x;
            o[0] = __builtin_amdgcn_s_memtime() - diag_t0;
            o[1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
        }
    }
    if constexpr (DIAG == 1) {
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
        cand_val[(int64_t)blockIdx.x * 512 + tid] = sum;
        return;
    }
    // epilogue: per signal the 4 largest |c| over this wave's 128 atoms, on packed keys -- the upper 23 bits of |c| (f32, sign
    // cleared, low 8 mantissa bits dropped) over 8 bits of (128 - atom-in-tile), so that one unsigned max orders by value and then by
    // LOWER atom index; the value handed on is the key with its low byte set: an upper bound of |c| that is 2^-15 relative wide
    const int at128 = atile * 2 + wr;
    const bool ragged = (int64_t)(at128 + 1) * 128 > N;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        unsigned keyv[32];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float cv = acc[m][n][j];
                if constexpr (I8) cv = (float)abs(__float_as_int(cv));  // (exact below 2^24, 2^-24 relative above: far inside the key's 2^-15)
                const unsigned bits = __float_as_uint(cv) & 0x7fffff00u;
                keyv[m * 4 + j] = bits | (unsigned)(128 - (m * 16 + j)) - (unsigned)(fq * 4);
            }
        if (ragged) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((int64_t)at128 * 128 + m * 16 + fq * 4 + j >= N) keyv[m * 4 + j] = 0u;
        }
        unsigned w[4];
        unsigned prev = 0xffffffffu;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned tt[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) tt[e] = (p == 0 || keyv[e] < prev) ? keyv[e] : 0u;
#pragma unroll
            for (int w2 = 16; w2 >= 1; w2 >>= 1)
#pragma unroll
                for (int e = 0; e < w2; ++e) tt[e] = tt[e] > tt[e + w2] ? tt[e] : tt[e + w2];
            w[p] = tt[0];
            prev = tt[0];
        }
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            unsigned o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (unsigned)__shfl_xor((int)w[q], sh, kSWave);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = w[q] > o[3 - q] ? w[q] : o[3 - q];
            auto cx = [&](int x, int y) {
                const unsigned hi = w[x] > w[y] ? w[x] : w[y], lo = w[x] > w[y] ? w[y] : w[x];
                w[x] = hi;
                w[y] = lo;
            };
            cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
        }
        if (fq == 0) {
            float ov[4];
            int oi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = w[q] != 0u;
                ov[q] = ok ? __uint_as_float(w[q] | 0xffu) : -1.0f;
                oi[q] = ok ? at128 * 128 + (128 - (int)(w[q] & 0xffu)) : 0x7fffffff;
            }
            const int64_t sig = (int64_t)stile * kBT2 + wc * 64 + n * 16 + fr;
            const int64_t base = (sig * n_atiles128 + at128) * kTileCand;
            if constexpr (OP != kOpBf16) {
                const float sc = sigscale[sig];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ov[q] >= 0.0f) ov[q] *= sc;
            }
            *reinterpret_cast<f32x4s*>(cand_val + base) = f32x4s{ov[0], ov[1], ov[2], ov[3]};
            *reinterpret_cast<int4*>(cand_idx + base) = make_int4(oi[0], oi[1], oi[2], oi[3]);
        }
    }
}


hipError_t launch_screen(hipStream_t stream, int mode, const __bf16* Ab, const __bf16* Rb, int Mk, int n_atiles, int n_stiles,
                         int64_t N, float* cand_val, int* cand_idx, const float* sigscale) {
    static bool attr_done = false;
    if (!attr_done) {
        for (const void* k : {(const void*)k_b_screen256p<kOpBf16>, (const void*)k_b_screen256p<kOpI8>, (const void*)k_b_screen256p<kOpF16>}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScreenLds256);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    if ((n_atiles & 1) || (n_stiles & 1) || Mk % (2 * kBK) != 0 || Mk < 4 * kBK) return hipErrorInvalidValue;  // (batch_dict / batch_ensure pad to these)
    if (mode != kScreen256p && !sigscale) return hipErrorInvalidValue;
    const dim3 grid((n_atiles / 2) * (n_stiles / 2)), block(512);
    // (int8: Ab / Rb are int8 images, Mk counts 2-byte slots -- rows of 2 Mk int8 values)
    auto kern = mode == kScreen256i8 ? k_b_screen256p<kOpI8> : mode == kScreen256f16 ? k_b_screen256p<kOpF16> : k_b_screen256p<kOpBf16>;
    hipLaunchKernelGGL(kern, grid, block, kScreenLds256, stream, Ab, Rb, Mk, n_atiles / 2, n_stiles / 2, N, n_atiles, cand_val, cand_idx, sigscale);
    return hipGetLastError();
}
const char* screen_kernel_name(int mode) {
    if (mode == 0) return "none (M > 8192 or a support capacity beyond the per-signal kernels' LDS: csmp_omp_batch's exact sweeps)";
    if (mode == kScreen256i8)
        return "csmp::k_b_screen256p<i8> (v_mfma_i32_16x16x64_i8 on int8 images, 256x256 tiles, eight-phase schedule, exact integer "
               "accumulation; fused top-4 epilogue)";
    if (mode == kScreen256f16)
        return "csmp::k_b_screen256p<f16> (v_mfma_f32_16x16x32_f16 on binary16 images, 256x256 tiles, eight-phase schedule: LDS-DMA units "
               "four phases ahead, counted vmcnt, the two waves of a SIMD one barrier apart; fused top-4 epilogue)";
    return "csmp::k_b_screen256p<bf16> (v_mfma_f32_16x16x32_bf16, 256x256 tiles, eight-phase schedule: LDS-DMA units four phases ahead, "
           "counted vmcnt, the two waves of a SIMD one barrier apart; fused top-4 epilogue)";
}

}  // namespace csmp
