// Round 6 probe: where a sweep workgroup's time goes when the residual is staged in two phases (M = 32768, Float32, 1 GiB) --
// wall-clock stamps (100 MHz) per wave at: 0 start, 1 image + norm done, 2 arrival at the stage barrier, 3 barrier passed,
// 4 next image staged, 5 last column done.  The one-image body at M = 18432 beside it (stamps 0, 1, 5).
// hipcc -O3 --offload-arch=gfx950 -DCSMP_PH_TRACE -o ph_trace ph_trace.hip && ./ph_trace
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../compressedsensing.jl_amd/csrc/csmp_kernels.hpp"
using namespace csmp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(float* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
        a[i] = (float)((int)(h & 0xffff) - 32768) * 1e-6f;
    }
}
static void report(const char* what, const std::vector<unsigned long long>& t, int waves, std::initializer_list<int> slots) {
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < waves; ++w) t0 = std::min(t0, t[(size_t)w * 8]);
    printf("%s (%d waves; us after the first wave's start: min / median / max)\n", what, waves);
    for (int s : slots) {
        std::vector<double> v;
        for (int w = 0; w < waves; ++w) v.push_back((double)(t[(size_t)w * 8 + s] - t0) * 0.01);
        std::sort(v.begin(), v.end());
        printf("  stamp %d: %8.2f %8.2f %8.2f\n", s, v.front(), v[v.size() / 2], v.back());
    }
}
static void diffs(const char* what, const std::vector<unsigned long long>& t, int waves, int a, int b) {
    std::vector<double> v;
    for (int w = 0; w < waves; ++w) v.push_back((double)(t[(size_t)w * 8 + b] - t[(size_t)w * 8 + a]) * 0.01);
    std::sort(v.begin(), v.end());
    printf("  %s: min %7.2f  median %7.2f  p90 %7.2f  max %7.2f us\n", what, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    float* A;
    double *r, *c, *pval;
    int* pidx;
    DevState* st;
    unsigned long long* tr;
    const int maxw = 8192;
    CK(hipMalloc(&A, bytes));
    CK(hipMalloc(&r, 65536 * 8));
    CK(hipMalloc(&c, 1 << 20));
    CK(hipMalloc(&pval, 8192 * 8));
    CK(hipMalloc(&pidx, 8192 * 4));
    CK(hipMalloc(&st, sizeof(DevState)));
    CK(hipMalloc(&tr, (size_t)maxw * 8 * 8));
    CK(hipMemset(st, 0, sizeof(DevState)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ph_trace), &tr, sizeof(tr)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, bytes / 4);
    std::vector<double> hr(65536);
    for (int i = 0; i < 65536; ++i) hr[i] = ((i * 7919) % 1000 - 500) * 1e-3;
    CK(hipMemcpy(r, hr.data(), 65536 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> t((size_t)maxw * 8);
    {
        const int M = 32768, N = 8192, grid = 206, KP = 16384, pcap = (N + grid * 4 - 1) / (grid * 4);
        auto kern = k_sweep_ph<float, 8, 4>;
        const size_t lds = sweep_ph_lds_bytes(KP, pcap);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float ms = 0;
        for (int it = 0; it < 6; ++it) {
            CK(hipMemset(tr, 0, (size_t)maxw * 64));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, (int64_t)M, M, (int64_t)N, r, c, pval, pidx, st, 0.0, 0, 0, KP, pcap);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(t.data(), tr, (size_t)maxw * 64, hipMemcpyDeviceToHost));
        printf("k_sweep_ph<float, 8, 4> M = 32768, N = 8192, 206 workgroups, two stages of 16384 rows: %.1f us (events)\n", ms * 1e3);
        report("two stages", t, grid * 4, {0, 1, 2, 3, 4, 5});
        diffs("image + norm (0 -> 1)", t, grid * 4, 0, 1);
        diffs("stage 0 stream (1 -> 2)", t, grid * 4, 1, 2);
        diffs("barrier wait (2 -> 3)", t, grid * 4, 2, 3);
        diffs("next image (3 -> 4)", t, grid * 4, 3, 4);
        diffs("stage 1 stream (4 -> 5)", t, grid * 4, 4, 5);
    }
    {
        const int M = 18432, N = 14560, grid = 206, KP = 18432;
        auto kern = k_sweep_gen<float, 16, 2>;
        const size_t lds = sweep_gen_lds_bytes(KP);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float ms = 0;
        for (int it = 0; it < 6; ++it) {
            CK(hipMemset(tr, 0, (size_t)maxw * 64));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, (int64_t)M, M, (int64_t)N, r, c, pval, pidx, st, 0.0, 0, 0, KP);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(t.data(), tr, (size_t)maxw * 64, hipMemcpyDeviceToHost));
        printf("k_sweep_gen<float, 16, 2> M = 18432, N = 14560, 206 workgroups, one image: %.1f us (events)\n", ms * 1e3);
        report("one image", t, grid * 4, {0, 1, 5});
        diffs("image + norm (0 -> 1)", t, grid * 4, 0, 1);
        diffs("stream (1 -> 5)", t, grid * 4, 1, 5);
    }
    {
        const int M = 256, N = 1048576, grid = 768, KP = 256;
        auto kern = k_sweep_short<float, 1, 4>;
        const size_t lds = sweep_gen_lds_bytes(KP);
        float ms = 0;
        double* cbig;
        CK(hipMalloc(&cbig, (size_t)N * 8));
        for (int it = 0; it < 6; ++it) {
            CK(hipMemset(tr, 0, (size_t)maxw * 64));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, (int64_t)M, M, (int64_t)N, r, cbig, pval, pidx, st, 0.0, 0, 0, KP);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(t.data(), tr, (size_t)maxw * 64, hipMemcpyDeviceToHost));
        printf("k_sweep_short<float, 1, 4> M = 256, N = 1048576, 768 workgroups: %.1f us (events)\n", ms * 1e3);
        report("short columns", t, grid * 4, {0, 1, 5});
        diffs("image + norm (0 -> 1)", t, grid * 4, 0, 1);
        diffs("stream (1 -> 5)", t, grid * 4, 1, 5);
    }
    for (int grid : {192, 176, 256}) {
        const int M = 4096, N = 65536, KP = 4096;
        auto kern = k_sweep_gen<float, 16, 2>;
        const size_t lds = sweep_gen_lds_bytes(KP);
        float ms = 0;
        for (int it = 0; it < 6; ++it) {
            CK(hipMemset(tr, 0, (size_t)maxw * 64));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, (int64_t)M, M, (int64_t)N, r, c, pval, pidx, st, 0.0, 0, 0, KP);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(t.data(), tr, (size_t)maxw * 64, hipMemcpyDeviceToHost));
        printf("k_sweep_gen<float, 16, 2> M = 4096, N = 65536 (the headline's sweep), %d workgroups: %.1f us (events)\n", grid, ms * 1e3);
        report("headline shape", t, grid * 4, {0, 1, 5});
        diffs("image + norm (0 -> 1)", t, grid * 4, 0, 1);
        diffs("stream (1 -> 5)", t, grid * 4, 1, 5);
    }
    return 0;
}
