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
constexpr int kBRow = 144;       // LDS bytes per staged row: 128 + 16 pad -> conflict-free ds_read_b128
constexpr int kTileCand = 4;     // candidates kept per (signal, atom tile)
constexpr size_t kScreenLds = 2 * 2 * kBT * kBRow;  // [buffer][A|R][row] = 73,728 B
constexpr size_t kScreenLds256 = 2 * 2 * 256 * 128;  // the 256^2 kernels: 2 buffers x (A 32 KiB + R 32 KiB) = 131,072 B

// one screening launch: D = Ab Rb' tile by tile with the fused top-4-per-(signal, 128-atom tile) epilogue.
// mode: kScreen128 = the 128^2 kernel; kScreen256 = 256^2 tiles with LDS-DMA staging (needs n_atiles and n_stiles even);
// kScreenCo = the same tiles by the persistent, 168-register kernel that shares CUs with k_b_step_co (ncu workgroups).
enum : int { kScreen128 = 0, kScreen256 = 1, kScreenCo = 2, kScreen256p = 3 };  // 256p: the eight-phase schedule (needs Mk % 128 == 0)
hipError_t launch_screen(hipStream_t stream, int mode, const __bf16* Ab, const __bf16* Rb, int Mk, int n_atiles, int n_stiles,
                         int64_t N, float* cand_val, int* cand_idx, int ncu);
const char* screen_kernel_name(int mode);

}  // namespace csmp
