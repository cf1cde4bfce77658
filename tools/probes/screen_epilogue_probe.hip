// screen_epilogue_probe.hip -- the screening GEMM of the batched path at configs[2] size (65536 atoms x 1024 signals, M = 4096, binary16
// operands): (1) the launch as the library runs it, (2) the same without its top-4 epilogue (DIAG = 1): the epilogue's share,
// (3) the clock the chip holds inside the K-loop (DIAG = 2: s_memtime / s_memrealtime around the loop, after two seconds of
// back-to-back launches on random data -- MI355X_MICROARCH.md, "DVFS give-back", item 6).  VERDICT round 4, item 9.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../compressedsensing.jl_amd/csrc -o screen_epilogue_probe screen_epilogue_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "csmp_screen.hip"
using namespace csmp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill_f16(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * 16384.0f);  // spread over the image's range, as a scaled dictionary is
    }
}

template <int DIAG>
static float run(const __bf16* A, const __bf16* R, int Mk, int n_at, int n_st, int64_t N, float* cv, int* ci, const float* sc, int reps) {
    auto kern = k_b_screen256p<kOpF16, DIAG>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScreenLds256));
    const dim3 grid((n_at / 2) * (n_st / 2)), block(512);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, block, kScreenLds256, 0, A, R, Mk, n_at / 2, n_st / 2, N, n_at, cv, ci, sc);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, kScreenLds256, 0, A, R, Mk, n_at / 2, n_st / 2, N, n_at, cv, ci, sc);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const int M = 4096, B = 1024;
    const int64_t N = 65536;
    _Float16 *A, *R; float *cv, *sc; int* ci;
    const int n_at = (int)(N / 128), n_st = B / 128;
    const size_t ncand = (size_t)B * n_at * 4, nwg = (size_t)(n_at / 2) * (n_st / 2);
    CK(hipMalloc((void**)&A, (size_t)N * M * 2));
    CK(hipMalloc((void**)&R, (size_t)B * M * 2));
    CK(hipMalloc((void**)&cv, std::max(ncand * 4, nwg * 512 * 4)));
    CK(hipMalloc((void**)&ci, ncand * 4 + nwg * 16));
    CK(hipMalloc((void**)&sc, B * 4));
    std::vector<float> hs(B, 1.0f);
    CK(hipMemcpy(sc, hs.data(), B * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill_f16, dim3(4096), dim3(256), 0, 0, A, (size_t)N * M, 17u);
    hipLaunchKernelGGL(k_fill_f16, dim3(512), dim3(256), 0, 0, R, (size_t)B * M, 99u);
    CK(hipDeviceSynchronize());
    const double flop = 2.0 * M * (double)N * B;
    const __bf16 *Ab = (const __bf16*)A, *Rb = (const __bf16*)R;
    (void)run<0>(Ab, Rb, M, n_at, n_st, N, cv, ci, sc, 4000);  // ~2 s of back-to-back launches: the clock settles
    const float full = run<0>(Ab, Rb, M, n_at, n_st, N, cv, ci, sc, 200);
    const float noepi = run<1>(Ab, Rb, M, n_at, n_st, N, cv, ci, sc, 200);
    const float stamped = run<2>(Ab, Rb, M, n_at, n_st, N, cv, ci, sc, 200);
    std::vector<unsigned long long> st(nwg * 2);
    CK(hipMemcpy(st.data(), ci + ncand, nwg * 16, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (size_t w = 0; w < nwg; ++w)
        if (st[2 * w + 1] > 0) mhz.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    printf("k_b_screen256p<f16>  full: %.1f us = %.3f PFLOP/s (%.3f of 2.5)\n", full, flop / full * 1e-9, flop / full * 1e-9 / 2.5);
    printf("                     no epilogue: %.1f us -> the epilogue is %.1f us = %.1f %% of the launch\n", noepi, full - noepi, 100.0 * (full - noepi) / full);
    printf("                     with clock stamps: %.1f us; in-kernel clock (K-loop, median over %zu workgroups): %.0f MHz (min %.0f, max %.0f)\n",
           stamped, mhz.size(), mhz.empty() ? 0.0 : mhz[mhz.size() / 2], mhz.empty() ? 0.0 : mhz.front(), mhz.empty() ? 0.0 : mhz.back());
    if (!mhz.empty()) {
        const double clk = mhz[mhz.size() / 2] * 1e6;
        // 16x16x32 f16: 8 passes of 4 cycles = 16 cycles per instruction per SIMD?  priced from the dense peak instead: 2.5 PF at 2.4 GHz
        const double peak_at_clk = 2.5e15 * clk / 2.4e9;
        printf("                     dense 16-bit peak at that clock: %.3f PFLOP/s -> the launch runs at %.3f of it\n", peak_at_clk * 1e-15, flop / (full * 1e-6) / peak_at_clk);
    }
    return 0;
}
