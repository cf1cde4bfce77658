// sweep_probe.hip -- where does the fixed cost of the bf16 sweep (k_sweep_bf16) go?  Wall-clock stamps (s_memrealtime, 100 MHz)
// of every workgroup's phases: start, after the stop-flag load, after the residual prologue, after the stream, end.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../compressedsensing.jl_amd/csrc -I../../include -o sweep_probe sweep_probe.hip
#define CSMP_SWEEP_TRACE 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "csmp_kernels.hpp"
#include "csmp_screened.hpp"
using namespace csmp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(unsigned short* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 20 & 0xff));
}

int main(int argc, char** argv) {
    const int Mk = 4096;
    const int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    __bf16* Ab; double* r; float* cv; int* ci; DevState* st; unsigned* tk;
    CK(hipMalloc((void**)&Ab, (size_t)N * Mk * 2));
    CK(hipMalloc((void**)&r, Mk * 8)); CK(hipMalloc((void**)&cv, 4096 * 4 * 4)); CK(hipMalloc((void**)&ci, 4096 * 4 * 4)); CK(hipMalloc((void**)&st, sizeof(DevState)));
    CK(hipMemset(st, 0, sizeof(DevState)));
    CK(hipMalloc((void**)&tk, 4096 * 4 * 16));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned short*)Ab, (size_t)N * Mk);
    std::vector<double> hr(Mk);
    for (int i = 0; i < Mk; ++i) hr[i] = (i % 7) - 3.0;
    CK(hipMemcpy(r, hr.data(), Mk * 8, hipMemcpyHostToDevice));
    const size_t lds = sweep_bf16_lds_bytes(Mk);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(st, 0, sizeof(DevState)));
        CK(hipMemset(tk, 0, 4096 * 4 * 16));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        switch (variant) {
            case 0: hipLaunchKernelGGL((k_sweep_bf16<2, 3, true, 4>), dim3(grid), dim3(kSweepThreads), lds, 0, Ab, Mk, N, r, Mk, cv, ci, st, 0.0, 0, 0, tk); break;
            case 1: hipLaunchKernelGGL((k_sweep_bf16<2, 5, true, 4>), dim3(grid), dim3(kSweepThreads), lds, 0, Ab, Mk, N, r, Mk, cv, ci, st, 0.0, 0, 0, tk); break;
            case 2: hipLaunchKernelGGL((k_sweep_bf16<4, 5, true, 2>), dim3(grid), dim3(kSweepThreads), lds, 0, Ab, Mk, N, r, Mk, cv, ci, st, 0.0, 0, 0, tk); break;
            case 3: hipLaunchKernelGGL((k_sweep_bf16<4, 3, true, 2>), dim3(grid), dim3(kSweepThreads), lds, 0, Ab, Mk, N, r, Mk, cv, ci, st, 0.0, 0, 0, tk); break;
            case 4: hipLaunchKernelGGL((k_sweep_bf16<2, 7, true, 2>), dim3(grid), dim3(kSweepThreads), lds, 0, Ab, Mk, N, r, Mk, cv, ci, st, 0.0, 0, 0, tk); break;
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> tr((size_t)grid * 8);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_sweep_trace), tr.size() * 8));
        unsigned long long t0 = ~0ull, tend = 0;
        for (int b = 0; b < grid; ++b) { t0 = std::min(t0, tr[b * 8]); tend = std::max(tend, tr[b * 8 + 4]); }
        double mn[7], mx[7], av[7];
        for (int k = 0; k < 7; ++k) { mn[k] = 1e30; mx[k] = 0; av[k] = 0; }
        for (int b = 0; b < grid; ++b)
            for (int k = 0; k < 7; ++k) {
                const double v = (double)(tr[b * 8 + k] - t0) * 0.01;  // us
                mn[k] = std::min(mn[k], v); mx[k] = std::max(mx[k], v); av[k] += v / grid;
            }
        printf("variant %d N %lld grid %d: events %.1f us, first start -> last end %.1f us\n", variant, (long long)N, grid, ms * 1e3, (double)(tend - t0) * 0.01);
        const char* names[7] = {"start", "after stop-flag load", "after prologue", "after stream", "end", "r landed, image written", "wave sum done"};
        if (rep == 4)
            for (int k = 0; k < 7; ++k) printf("   %-22s min %7.2f  avg %7.2f  max %7.2f us\n", names[k], mn[k], av[k], mx[k]);
    }
    return 0;
}
