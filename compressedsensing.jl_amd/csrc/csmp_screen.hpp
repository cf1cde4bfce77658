// csmp_screen.hpp -- the screening GEMM of the batched path (kernels only; compiled in csmp_screen.hip,
// a translation unit of its own: these MFMA kernels dominate the build time of the library).
// See csmp_batched.hpp for the algorithm they serve.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csmp {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16s = __attribute__((ext_vector_type(16))) float;
using f32x4s = __attribute__((ext_vector_type(4))) float;
constexpr int kSWave = 64;
constexpr int kBT = 128;         // tile edge: atoms x signals
constexpr int kBK = 64;          // k elements staged per step
constexpr int kTileCand = 4;     // candidates kept per (signal, atom tile)
constexpr int kBatchMaxRows = 8192;  // rows of the dictionary the batched path's per-signal kernels hold (registers + LDS)
constexpr size_t kScreenLds256 = 2 * 2 * 256 * 128;  // 2 buffers x (A 32 KiB + R 32 KiB) = 131,072 B

// one screening launch: D = Ab Rb' in 256 x 256 tiles with the fused top-4-per-(signal, 128-atom tile) epilogue.
// Needs n_atiles and n_stiles (counted in 128s) even and Mk an even number (>= 4) of 64-deep K-tiles.
enum : int { kScreen256p = 3, kScreen256i8 = 5, kScreen256f16 = 6 };  // operands: bf16 / int8 / binary16 images
enum : int { kOpBf16 = 0, kOpI8 = 1, kOpF16 = 2 };
hipError_t launch_screen(hipStream_t stream, int mode, const __bf16* Ab, const __bf16* Rb, int Mk, int n_atiles, int n_stiles,
                         int64_t N, float* cand_val, int* cand_idx, const float* sigscale = nullptr);
const char* screen_kernel_name(int mode);

}  // namespace csmp
