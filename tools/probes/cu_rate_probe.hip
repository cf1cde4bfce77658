// Round 6 probe: how much of the HBM stream can ONE CU pull?  The phased sweep (M = 32768) stages a 128-KiB residual image per
// workgroup -- one workgroup of four waves per CU, 206 of them -- and the image traffic (workgroups x 256 KiB per sweep) is what is
// left of its loss.  Fewer, fatter workgroups (eight waves sharing one image) would halve it, if ~100 CUs can carry the stream.
// A register-only streaming kernel shaped like the sweep: every wave reads 128-KiB "columns" (wave w: columns w, w + waves, ...),
// 32 non-temporal 16-byte loads per lane in flight, 130 KiB of LDS requested (one workgroup per CU).  1 GiB per launch.
// hipcc -O3 --offload-arch=gfx950 -o cu_rate_probe cu_rate_probe.hip && ./cu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using f4 = __attribute__((ext_vector_type(4))) float;
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_stream(const f4* __restrict__ A, int64_t ncol, int colvec, float* out) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = THREADS / 64;
    const int64_t w = (int64_t)blockIdx.x * NW + wave, nw = (int64_t)gridDim.x * NW;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 buf[32];
    const int units = colvec / (64 * 32);  // 32 loads of 64 lanes per unit
    for (int64_t col = w; col < ncol; col += nw) {
        const f4* pc = A + col * colvec;
        for (int u = 0; u < units; ++u) {
#pragma unroll
            for (int q = 0; q < 32; ++q) buf[q] = __builtin_nontemporal_load(pc + (u * 32 + q) * 64 + lane);
#pragma unroll
            for (int q = 0; q < 32; ++q) acc += buf[q];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[threadIdx.x] = acc.x + lds[0];
}
template <int THREADS>
static int run(const f4* A, float* out, int grid, size_t bytes) {
    const int colvec = 128 * 1024 / 16;
    const int64_t ncol = bytes / (128 * 1024);
    auto kern = k_stream<THREADS>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), 130 * 1024, 0, A, ncol, colvec, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double us = best / 20 * 1e3;
    printf("%3d workgroups x %d waves: %7.1f us per GiB = %5.2f TB/s = %.3f of 8 TB/s, %5.1f GB/s per CU\n", grid, THREADS / 64, us, bytes / us / 1e6,
           bytes / us / 1e6 / 8.0, bytes / us / 1e3 / grid);
    return 0;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    f4* A;
    float* out;
    CK(hipMalloc(&A, bytes));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(A, 0, bytes));
    for (int g : {256, 206, 192, 160, 128, 103, 96, 64}) if (run<256>(A, out, g, bytes)) return 1;
    for (int g : {256, 206, 160, 128, 103, 96, 64}) if (run<512>(A, out, g, bytes)) return 1;
    for (int g : {128, 103, 64}) if (run<1024>(A, out, g, bytes)) return 1;
    return 0;
}
