// overlap_probe.hip -- can a matrix-core kernel and an HBM-streaming kernel share the CUs of an MI355X?
// The premise of a co-resident screening GEMM for the batched path ("Overlapping the screen ..."): a
// workgroup of FOUR waves (one per SIMD) with <= 256 registers and 96 KiB of LDS leaves room on every CU for two
// 256-thread workgroups of <= 128 registers and ~22 KiB of LDS each (k_b_append's footprint).
//   k_mfma    4 waves per workgroup, 32 independent 16x16x32 bf16 accumulators per wave, register operands only:
//             the matrix pipe's ceiling for one wave per SIMD.  96 KiB of dynamic LDS pins it to one workgroup per CU.
//   k_stream  256 threads, 8 x 16-byte non-temporal loads in flight per lane over a 2 GiB buffer, 22 KiB of LDS.
// Measured alone and together (two streams): TFLOP/s and GB/s.
// Build: hipcc -O3 --offload-arch=gfx950 -o overlap_probe overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mfma(float* out, int iters) {
    extern __shared__ char smem[];
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4)0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(float)(threadIdx.x + i + e);
            b[i][e] = (__bf16)(float)(threadIdx.x * 3 + i - e);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    f32x4 s = (f32x4)0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    if (s.x == 12345.f) out[threadIdx.x] = s.x + smem[0];
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
// the same with v_mfma_f32_32x32x16_bf16: 8 independent 32 x 32 accumulators (128 registers), 32768 flop per instruction
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mfma32(float* out, int iters) {
    extern __shared__ char smem[];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)0.f;
    bf16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(float)(threadIdx.x + i + e);
            b[i & 1][e] = (__bf16)(float)(threadIdx.x * 3 + i - e);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    f32x16 s = (f32x16)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s[0] == 12345.f) out[threadIdx.x] = s[0] + smem[0];
}

__global__ __launch_bounds__(256, 4) void k_stream(const f32x4* __restrict__ p, size_t n, float* sink) {
    extern __shared__ char smem2[];
    f32x4 acc = (f32x4)0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x + smem2[0];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const size_t bytes = (size_t)2 << 30;
    f32x4* buf;
    float *sink, *out;
    CK(hipMalloc((void**)&buf, bytes));
    CK(hipMalloc((void**)&sink, 4));
    CK(hipMalloc((void**)&out, 4096));
    CK(hipMemset(buf, 0, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int lds_m = 96 * 1024, lds_s = 22 * 1024;
    CK(hipFuncSetAttribute((const void*)k_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_m));
    CK(hipFuncSetAttribute((const void*)k_mfma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int iters = 20000;  // 32 MFMAs x 16*16*32*2 flop x iters per wave
    auto flops = [&](int waves, int wgs) { return 2.0 * 16 * 16 * 32 * 32.0 * iters * waves * wgs; };
    auto run_m = [&](int waves, hipStream_t s, int wgs) {
        if (waves == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(wgs), dim3(256), lds_m, s, out, iters);
        else hipLaunchKernelGGL(k_mfma<8>, dim3(wgs), dim3(512), 128 * 1024, s, out, iters);
    };
    auto run_s = [&](hipStream_t s, int reps, int wg_per_cu) {
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream, dim3(ncu * wg_per_cu), dim3(256), lds_s, s, buf, bytes / 16, sink);
    };
    float ms;
    // alone
    for (int waves : {4, 8}) {
        run_m(waves, s1, ncu);
        CK(hipEventRecord(a0, s1)); run_m(waves, s1, ncu); CK(hipEventRecord(a1, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, a0, a1));
        printf("k_mfma alone, %d waves per workgroup (1 workgroup per CU): %.2f ms, %.0f TFLOP/s\n", waves, ms, flops(waves, ncu) / ms * 1e-9);
    }
    for (int w : {1, 2, 4}) {
        run_s(s2, 1, w);
        CK(hipEventRecord(b0, s2)); run_s(s2, 4, w); CK(hipEventRecord(b1, s2)); CK(hipStreamSynchronize(s2));
        CK(hipEventElapsedTime(&ms, b0, b1));
        printf("k_stream alone, %d workgroups per CU: %.2f ms per 2 GiB, %.0f GB/s\n", w, ms / 4, 4.0 * bytes / ms * 1e-6);
    }
    CK(hipFuncSetAttribute((const void*)k_mfma32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_m));
    {   // 32x32x16 instructions, one wave per SIMD: 16 per iteration x 32768 flop
        const double fl = 2.0 * 32 * 32 * 16 * 16.0 * iters * 4 * ncu;
        hipLaunchKernelGGL(k_mfma32<4>, dim3(ncu), dim3(256), lds_m, s1, out, iters);
        CK(hipEventRecord(a0, s1)); hipLaunchKernelGGL(k_mfma32<4>, dim3(ncu), dim3(256), lds_m, s1, out, iters); CK(hipEventRecord(a1, s1));
        CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, a0, a1));
        printf("k_mfma32 (32x32x16) alone, 4 waves per workgroup: %.2f ms, %.0f TFLOP/s\n", ms, fl / ms * 1e-9);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a0, s1)); hipLaunchKernelGGL(k_mfma32<4>, dim3(ncu), dim3(256), lds_m, s1, out, iters); CK(hipEventRecord(a1, s1));
        CK(hipEventRecord(b0, s2)); run_s(s2, 8, 2); CK(hipEventRecord(b1, s2));
        CK(hipDeviceSynchronize());
        float mm, sm;
        CK(hipEventElapsedTime(&mm, a0, a1));
        CK(hipEventElapsedTime(&sm, b0, b1));
        printf("together (32x32x16, stream 2 workgroups per CU): k_mfma32 %.2f ms = %.0f TFLOP/s; k_stream %.2f ms per 2 GiB = %.0f GB/s\n", mm,
               fl / mm * 1e-9, sm / 8, 8.0 * bytes / sm * 1e-6);
    }
    // together: the MFMA kernel (4 waves) for ~T ms, the stream repeated under it
    for (int w : {1, 2}) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a0, s1)); run_m(4, s1, ncu); CK(hipEventRecord(a1, s1));
        CK(hipEventRecord(b0, s2)); run_s(s2, 8, w); CK(hipEventRecord(b1, s2));
        CK(hipDeviceSynchronize());
        float mm, sm;
        CK(hipEventElapsedTime(&mm, a0, a1));
        CK(hipEventElapsedTime(&sm, b0, b1));
        printf("together (stream %d workgroups per CU): k_mfma %.2f ms = %.0f TFLOP/s; k_stream %.2f ms per 2 GiB = %.0f GB/s\n", w, mm,
               flops(4, ncu) / mm * 1e-9, sm / 8, 8.0 * bytes / sm * 1e-6);
    }
    // the 8-wave form (what the product screen looks like to the dispatcher: 2 x ~250 registers per SIMD) beside the stream
    {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a0, s1)); run_m(8, s1, ncu); CK(hipEventRecord(a1, s1));
        CK(hipEventRecord(b0, s2)); run_s(s2, 8, 2); CK(hipEventRecord(b1, s2));
        CK(hipDeviceSynchronize());
        float mm, sm;
        CK(hipEventElapsedTime(&mm, a0, a1));
        CK(hipEventElapsedTime(&sm, b0, b1));
        printf("together, 8-wave MFMA workgroups: k_mfma %.2f ms = %.0f TFLOP/s; k_stream %.2f ms per 2 GiB = %.0f GB/s\n", mm,
               flops(8, ncu) / mm * 1e-9, sm / 8, 8.0 * bytes / sm * 1e-6);
    }
    printf("done\n");
    return 0;
}
