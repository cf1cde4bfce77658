// mfma_f64_probe.hip -- issue rate of v_mfma_f64_16x16x4_f64 on gfx950: bare loops, 1 or 2 waves per SIMD, 1..8 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(int wgs_per_cu, int iters) {
    double* out;
    hipMalloc(&out, 256 * 256 * 8 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * NACC * 2048.0;
    printf("NACC %d, %d WG/CU (x4 waves): %.1f us, %.1f TFLOP/s, %.1f cycles@2.4GHz per MFMA per SIMD\n", NACC, wgs_per_cu, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wgs_per_cu));
    hipFree(out);
}
int main() {
    run<1>(1, 20000);
    run<4>(1, 5000);
    run<8>(1, 2500);
    run<8>(2, 2500);
    run<16>(1, 1250);
    return 0;
}
