// csmp_screen.hip -- screening GEMM kernels of the batched path + their launcher (see csmp_screen.hpp).
#include "csmp_screen.hpp"

namespace csmp {

// ---------------------------------------------------------------------------------------------
// Screening GEMM.  D[atom][signal] = sum_k A[atom][k] R[signal][k]; A-operand rows = atoms,
// B-operand columns = signals, so a lane's 16 accumulator registers are 16 atoms of ONE signal
// (C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8(reg>>2) + 4(lane>>5)).
// 4 waves as 2 (atoms) x 2 (signals), each 64 x 64 = 2 x 2 MFMA tiles; BK = 64 staged through LDS
// (register staging, padded rows), double buffered, one barrier per k-step.
// Block map: the 8 XCDs each take whole atom tiles and walk all signal tiles of it back to back, so
// an atom tile's 1 MiB of bf16 is fetched into one L2 once.
struct top4 {
    float v[4];
    int i[4];
};
// insert (v, i) into the descending list (ties: lower atom index first)
__device__ __forceinline__ void top4_push(top4& t, float v, int i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool up = (v > t.v[q]) || (v == t.v[q] && i < t.i[q]);
        const float tv = up ? t.v[q] : v;
        const int ti = up ? t.i[q] : i;
        t.v[q] = up ? v : t.v[q];
        t.i[q] = up ? i : t.i[q];
        v = tv;
        i = ti;
    }
}

__global__ __launch_bounds__(256) void k_b_screen(const __bf16* __restrict__ Ab, const __bf16* __restrict__ Rb, int Mk,
                                                  int n_atiles, int n_stiles, int64_t N,
                                                  float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
    int atile, stile;
    {
        const int bid = blockIdx.x;
        if ((n_atiles & 7) == 0) {
            const int xcd = bid & 7, local = bid >> 3;
            stile = local % n_stiles;
            atile = (local / n_stiles) * 8 + xcd;
        } else {
            stile = bid % n_stiles;
            atile = bid / n_stiles;
        }
    }
    const __bf16* gA = Ab + (int64_t)atile * kBT * Mk;
    const __bf16* gR = Rb + (int64_t)stile * kBT * Mk;
    f32x16s acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x16s)0.0f;

    // staging map: 1024 16-B pieces per operand tile, 4 per thread
    int srow[4], skc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        srow[i] = p >> 3;
        skc[i] = p & 7;
    }
    // Two register sets: the tile for k-step kb+1 is written to LDS while the loads for kb+2 and kb+3 are
    // already in flight, so a global load has two full MFMA phases (and a barrier) to land.
    bf16x8 ra0[4], rr0[4], ra1[4], rr1[4];
    const int nkb = Mk / kBK;
    auto gload = [&](bf16x8 (&ra)[4], bf16x8 (&rr)[4], int kb) {
        if (kb >= nkb) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const bf16x8*>(gA + (int64_t)srow[i] * Mk + kb * kBK + skc[i] * 8);
            rr[i] = *reinterpret_cast<const bf16x8*>(gR + (int64_t)srow[i] * Mk + kb * kBK + skc[i] * 8);
        }
    };
    auto lstore = [&](const bf16x8 (&ra)[4], const bf16x8 (&rr)[4], int buf) {
        char* la = smem + (size_t)(buf * 2 + 0) * kBT * kBRow;
        char* lr = smem + (size_t)(buf * 2 + 1) * kBT * kBRow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<bf16x8*>(la + srow[i] * kBRow + skc[i] * 16) = ra[i];
            *reinterpret_cast<bf16x8*>(lr + srow[i] * kBRow + skc[i] * 16) = rr[i];
        }
    };
    auto compute = [&](int buf) {
        const char* la = smem + (size_t)(buf * 2 + 0) * kBT * kBRow + (wr * 64 + r) * kBRow + h * 16;
        const char* lr = smem + (size_t)(buf * 2 + 1) * kBT * kBRow + (wc * 64 + r) * kBRow + h * 16;
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(la + kk * 32);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(la + 32 * kBRow + kk * 32);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(lr + kk * 32);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(lr + 32 * kBRow + kk * 32);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    gload(ra0, rr0, 0);
    lstore(ra0, rr0, 0);
    gload(ra0, rr0, 1);
    gload(ra1, rr1, 2);
    __syncthreads();
    for (int kb = 0; kb < nkb; kb += 2) {
        compute(0);                                   // tile kb
        if (kb + 1 < nkb) lstore(ra0, rr0, 1);        // tile kb+1
        gload(ra0, rr0, kb + 3);
        __syncthreads();
        if (kb + 1 < nkb) {
            compute(1);                               // tile kb+1
            if (kb + 2 < nkb) lstore(ra1, rr1, 0);    // tile kb+2
            gload(ra1, rr1, kb + 4);
            __syncthreads();
        }
    }

    // epilogue: the 4 largest |c| per signal over this tile's 128 atoms
    top4* sc = reinterpret_cast<top4*>(smem);  // [2 (wr)][128 signals]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        top4 t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t.v[q] = -1.0f;
            t.i[q] = 0x7fffffff;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int atom = atile * kBT + wr * 64 + m * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                const float v = (atom < N) ? fabsf(acc[m][n][q]) : -1.0f;
                top4_push(t, v, atom);
            }
        // merge the two lane halves (same signal, interleaved atoms)
        top4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o.v[q] = __shfl_xor(t.v[q], 32, kSWave);
            o.i[q] = __shfl_xor(t.i[q], 32, kSWave);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) top4_push(t, o.v[q], o.i[q]);
        if (h == 0) sc[wr * kBT + wc * 64 + n * 32 + r] = t;
    }
    __syncthreads();
    if (tid < kBT) {
        top4 t = sc[tid];
        const top4 o = sc[kBT + tid];
#pragma unroll
        for (int q = 0; q < 4; ++q) top4_push(t, o.v[q], o.i[q]);
        const int64_t sig = (int64_t)stile * kBT + tid;
        const int64_t base = (sig * n_atiles + atile) * kTileCand;
        *reinterpret_cast<f32x4s*>(cand_val + base) = f32x4s{t.v[0], t.v[1], t.v[2], t.v[3]};
        *reinterpret_cast<int4*>(cand_idx + base) = make_int4(t.i[0], t.i[1], t.i[2], t.i[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// 256 atoms x 256 signals per workgroup, 8 waves as 2 (atoms) x 4 (signals), each 128 x 64 = 8 x 4 MFMA
// tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16): 64 MFMAs per 24 ds_read_b128 and K-tile of 64, twice the
// reuse of the 128^2 kernels.  The operand tiles go global -> LDS directly (global_load_lds_dwordx4: no
// staging registers, no ds_write pass); an LDS-DMA writes wave-uniform base + lane*16, i.e. 1 KiB = 8
// rows of 128 B per wave-instruction, so an XOR swizzle (chunk ^= (row>>1)&7, which makes every 16-lane group of a ds_read_b128 of the 16x16x32
// fragments hit 16 distinct 16-byte slots of the unpadded 128-byte rows) is applied
// to the per-lane SOURCE address.  LDS: 2 buffers x (A 32 KiB + R 32 KiB) = 128 KiB, one workgroup per
// CU.  K-loop: the DMAs of tile t+1 are issued before the MFMAs of tile t and retired (vmcnt(0)) at the
// barrier that ends it.  Each (wave row, signal) pair lives in ONE wave, so the top-4 epilogue needs
// no LDS: it is written per 128-atom half tile, the granularity k_b_step expects.
constexpr int kBT2 = 256;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
__global__ __launch_bounds__(512) void k_b_screen256(const __bf16* __restrict__ Ab, const __bf16* __restrict__ Rb, int Mk,
                                                     int n_at2, int n_st2, int64_t N, int n_atiles128,
                                                     float* __restrict__ cand_val, int* __restrict__ cand_idx) {
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
    // DMA map: wave w, instruction i (0..3) fills the 1 KiB region g = 4 w + i of an operand tile = rows
    // 8 g .. 8 g + 7; lane l lands at row 8 g + (l >> 3), slot l & 7, and fetches chunk slot ^ key(row)
    int64_t goff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        goff[i] = (int64_t)row * Mk + chunk * 8;
    }
    const int nkb = Mk / kBK;
    auto issue = [&](int buf, int kb) {
        char* la = smem + (size_t)(buf * 2 + 0) * kBT2 * 128 + (size_t)(4 * wave) * 1024;
        char* lr = smem + (size_t)(buf * 2 + 1) * kBT2 * 128 + (size_t)(4 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gA + goff[i] + kb * kBK), (lds_void_t*)(la + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gR + goff[i] + kb * kBK), (lds_void_t*)(lr + i * 1024), 16, 0, 0);
        }
    };
    const int key = (fr >> 1) & 7;
    auto compute = [&](int buf) {
        const char* la = smem + (size_t)(buf * 2 + 0) * kBT2 * 128 + (wr * 128 + fr) * 128;
        const char* lr = smem + (size_t)(buf * 2 + 1) * kBT2 * 128 + (wc * 64 + fr) * 128;
#pragma unroll
        for (int kk = 0; kk < kBK / 32; ++kk) {
            const int co = ((kk * 4 + fq) ^ key) << 4;
            bf16x8 b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const bf16x8*>(lr + t * 16 * 128 + co);
            bf16x8 a[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const bf16x8*>(la + m * 16 * 128 + co);
            __builtin_amdgcn_s_setprio(1);  // the MFMA cluster of this wave ahead of the other wave's address / DMA issue
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) issue(buf ^ 1, kb + 1);
        compute(buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue: per signal the 4 largest |c| over this wave's 128 atoms (= one 128-atom tile of the candidate
    // arrays), on packed 32-bit keys: bits 31..8 = the leading bits of |c| (sign cleared: monotone as
    // unsigned), bits 7..0 = 128 - (atom offset in the tile), so that ONE v_max_u32 orders by value and then
    // by ascending atom, and 0 is "no atom".  Four passes of a max tree over the lane's 32 keys (each pass
    // masks what the previous ones took), then two sorted-list merges across the lane quarters.  Values are
    // reported rounded UP to the key granularity (2^-15 relative), so they still bound every atom they
    // displaced -- what the certificate of k_b_step needs (delta there is ~1e-2 of the values in play).
    const int at128 = atile * 2 + wr;
    const bool ragged = (int64_t)(at128 + 1) * 128 > N;  // only the last tile can hold atoms >= N
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        unsigned key[32];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned bits = __float_as_uint(acc[m][n][j]) & 0x7fffff00u;
                key[m * 4 + j] = bits | (unsigned)(128 - (m * 16 + j)) - (unsigned)(fq * 4);
            }
        if (ragged) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((int64_t)at128 * 128 + m * 16 + fq * 4 + j >= N) key[m * 4 + j] = 0u;
        }
        unsigned w[4];
        unsigned prev = 0xffffffffu;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned t[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) t[e] = (p == 0 || key[e] < prev) ? key[e] : 0u;
#pragma unroll
            for (int w2 = 16; w2 >= 1; w2 >>= 1)
#pragma unroll
                for (int e = 0; e < w2; ++e) t[e] = t[e] > t[e + w2] ? t[e] : t[e + w2];
            w[p] = t[0];
            prev = t[0];
        }
        // merge with the other lane quarters: the 4 largest of two descending 4-lists are max(a_i, b_{3-i}); re-sort
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            unsigned o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (unsigned)__shfl_xor((int)w[q], sh, kSWave);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = w[q] > o[3 - q] ? w[q] : o[3 - q];
            auto cx = [&](int x, int y) {  // descending compare-exchange
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
            *reinterpret_cast<f32x4s*>(cand_val + base) = f32x4s{ov[0], ov[1], ov[2], ov[3]};
            *reinterpret_cast<int4*>(cand_idx + base) = make_int4(oi[0], oi[1], oi[2], oi[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_b_screen256c: the 256 x 256 kernel in the form that SHARES a CU with a rescoring / append workgroup (k_b_step_co)
// of the other half-batch: at most 168 registers per lane (three waves per SIMD: two of this kernel, one of the other),
// 128 KiB of LDS, and a PERSISTENT grid (one workgroup per CU walks the tiles), so that a CU never holds more than one
// of these.  Same tiles, same staging (LDS-DMA with the source-side XOR swizzle), same packed-key epilogue result as
// k_b_screen256; the differences are in how registers are spent:
//   * operand fragments are streamed: per 32-deep k-step the four signal fragments stay (16 registers), the eight atom
//     fragments pass through a two-deep window (8 registers) -- 24 instead of 48;
//   * the epilogue never materialises its 32 keys per signal: each of the four max passes recomputes them from the
//     accumulators (3 VALU per key) and keeps only a running maximum.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_b_screen256c(const __bf16* __restrict__ Ab, const __bf16* __restrict__ Rb, int Mk,
                    int n_at2, int n_st2, int64_t N, int n_atiles128,
                    float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int nkb = Mk / kBK;
    const int key = (fr >> 1) & 7;
    // DMA map (as k_b_screen256): lane offsets within an operand tile, 32-bit
    int goff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        goff[i] = row * Mk + chunk * 8;
    }
    const int ntiles = n_at2 * n_st2;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // consecutive workgroups (= the CUs of one XCD, round-robin over XCDs) share an atom tile across the signal tiles
        int atile, stile;
        if ((n_at2 & 7) == 0) {
            const int xcd = tile & 7, local = tile >> 3;
            stile = local % n_st2;
            atile = (local / n_st2) * 8 + xcd;
        } else {
            stile = tile % n_st2;
            atile = tile / n_st2;
        }
        const __bf16* gA = Ab + (int64_t)atile * kBT2 * Mk;
        const __bf16* gR = Rb + (int64_t)stile * kBT2 * Mk;
        f32x4s acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4s)0.0f;
        auto issue = [&](int buf, int kb) {
            char* la = smem + (size_t)(buf * 2 + 0) * kBT2 * 128 + (size_t)(4 * wave) * 1024;
            char* lr = smem + (size_t)(buf * 2 + 1) * kBT2 * 128 + (size_t)(4 * wave) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gA + goff[i] + kb * kBK), (lds_void_t*)(la + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gR + goff[i] + kb * kBK), (lds_void_t*)(lr + i * 1024), 16, 0, 0);
            }
        };
        __syncthreads();  // (the previous tile's last reads of buffer 0 are done)
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb) issue(buf ^ 1, kb + 1);
            const char* la = smem + (size_t)(buf * 2 + 0) * kBT2 * 128 + (wr * 128 + fr) * 128;
            const char* lr = smem + (size_t)(buf * 2 + 1) * kBT2 * 128 + (wc * 64 + fr) * 128;
#pragma unroll
            for (int kk = 0; kk < kBK / 32; ++kk) {
                const int co = ((kk * 4 + fq) ^ key) << 4;
                bf16x8 b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const bf16x8*>(lr + t * 16 * 128 + co);
                bf16x8 a0 = *reinterpret_cast<const bf16x8*>(la + co);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    bf16x8 a1 = a0;
                    if (m + 1 < 8) a1 = *reinterpret_cast<const bf16x8*>(la + (m + 1) * 16 * 128 + co);
                    __builtin_amdgcn_sched_barrier(0);  // keep the window two deep: the next fragment's read stays here
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[n], acc[m][n], 0, 0, 0);
                    a0 = a1;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // epilogue: per signal the 4 largest |c| over this wave's 128 atoms on packed keys (see k_b_screen256)
        const int at128 = atile * 2 + wr;
        const bool ragged = (int64_t)(at128 + 1) * 128 > N;
        unsigned lo0 = 128u - (unsigned)(fq * 4);  // low key byte of this lane's first atom; the others follow by constants
        asm volatile("" : "+v"(lo0));              // (opaque per tile: 32 precomputed low bytes per lane would cost 32 registers)
        const int nleft = (int)(N - (int64_t)at128 * 128) - fq * 4;  // atoms of this tile from this lane's first one on (ragged tile only)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            unsigned w[4];
            unsigned prev = 0xffffffffu;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned best = 0u;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned kv = (__float_as_uint(acc[m][n][j]) & 0x7fffff00u) | (lo0 - (unsigned)(m * 16 + j));
                        if (ragged && m * 16 + j >= nleft) kv = 0u;
                        kv = kv < prev ? kv : 0u;
                        best = best > kv ? best : kv;
                    }
                w[p] = best;
                prev = best;  // (0 stays 0: nothing is left below)
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
                *reinterpret_cast<f32x4s*>(cand_val + base) = f32x4s{ov[0], ov[1], ov[2], ov[3]};
                *reinterpret_cast<int4*>(cand_idx + base) = make_int4(oi[0], oi[1], oi[2], oi[3]);
            }
        }
    }
}

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
__global__ __launch_bounds__(512) void k_b_screen256p(const __bf16* __restrict__ Ab, const __bf16* __restrict__ Rb, int Mk,
                                                      int n_at2, int n_st2, int64_t N, int n_atiles128,
                                                      float* __restrict__ cand_val, int* __restrict__ cand_idx) {
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
                    acc[(QA) * 4 + m][(QB) * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][kk], b[n][kk], acc[(QA) * 4 + m][(QB) * 2 + n], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                            \
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
    // epilogue: identical to k_b_screen256 (per signal the 4 largest |c| over this wave's 128 atoms, packed keys)
    const int at128 = atile * 2 + wr;
    const bool ragged = (int64_t)(at128 + 1) * 128 > N;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        unsigned keyv[32];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned bits = __float_as_uint(acc[m][n][j]) & 0x7fffff00u;
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
            *reinterpret_cast<f32x4s*>(cand_val + base) = f32x4s{ov[0], ov[1], ov[2], ov[3]};
            *reinterpret_cast<int4*>(cand_idx + base) = make_int4(oi[0], oi[1], oi[2], oi[3]);
        }
    }
}

hipError_t launch_screen(hipStream_t stream, int mode, const __bf16* Ab, const __bf16* Rb, int Mk, int n_atiles, int n_stiles,
                         int64_t N, float* cand_val, int* cand_idx, int ncu) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_b_screen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScreenLds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_b_screen256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScreenLds256);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_b_screen256c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScreenLds256);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_b_screen256p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScreenLds256);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (mode == kScreenCo) {
        const int ntiles = (n_atiles / 2) * (n_stiles / 2);
        hipLaunchKernelGGL(k_b_screen256c, dim3(ntiles < ncu ? ntiles : ncu), dim3(512), kScreenLds256, stream, Ab, Rb, Mk,
                           n_atiles / 2, n_stiles / 2, N, n_atiles, cand_val, cand_idx);
    } else if (mode == kScreen256p)
        hipLaunchKernelGGL(k_b_screen256p, dim3((n_atiles / 2) * (n_stiles / 2)), dim3(512), kScreenLds256, stream, Ab, Rb, Mk,
                           n_atiles / 2, n_stiles / 2, N, n_atiles, cand_val, cand_idx);
    else if (mode == kScreen256)
        hipLaunchKernelGGL(k_b_screen256, dim3((n_atiles / 2) * (n_stiles / 2)), dim3(512), kScreenLds256, stream, Ab, Rb, Mk,
                           n_atiles / 2, n_stiles / 2, N, n_atiles, cand_val, cand_idx);
    else
        hipLaunchKernelGGL(k_b_screen, dim3(n_atiles * n_stiles), dim3(256), kScreenLds, stream, Ab, Rb, Mk, n_atiles, n_stiles, N,
                           cand_val, cand_idx);
    return hipGetLastError();
}
const char* screen_kernel_name(int mode) {
    return mode == kScreenCo ? "csmp::k_b_screen256c (v_mfma_f32_16x16x32_bf16, 256x256 tiles, LDS-DMA staging, fused top-4 epilogue; persistent, "
                               "168 registers: shares each CU with a k_b_step_co workgroup of the other half-batch)"
           : mode == kScreen256p ? "csmp::k_b_screen256p (v_mfma_f32_16x16x32_bf16, 256x256 tiles, eight-phase schedule: LDS-DMA units four phases ahead, "
                                   "counted vmcnt, the two waves of a SIMD one barrier apart; fused top-4 epilogue)"
           : mode == kScreen256 ? "csmp::k_b_screen256 (v_mfma_f32_16x16x32_bf16, 256x256 tiles, LDS-DMA staging, fused top-4 epilogue)"
                                : "csmp::k_b_screen (v_mfma_f32_32x32x16_bf16, 128x128 tiles, fused top-4 epilogue)";
}

}  // namespace csmp
