// screen_i8_probe.hip -- the screening GEMM of the batched path at configs[2] size (65536 atoms x 1024 signals, M = 4096) with
// bf16 operands (v_mfma_f32_16x16x32_bf16) and with int8 operands (v_mfma_i32_16x16x64_i8: half the K-loop): kernel time, and the
// int8 result checked against a host dot product on a few (atom tile, signal) candidates.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../compressedsensing.jl_amd/csrc -o screen_i8_probe screen_i8_probe.hip ../../compressedsensing.jl_amd/csrc/csmp_screen.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "csmp_screen.hpp"
using namespace csmp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill_i8(signed char* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = (signed char)((int)(h % 255u) - 127);
    }
}

int main() {
    const int M = 4096, B = 1024;
    const int64_t N = 65536;
    signed char *A8, *R8; float *cv, *sc; int* ci;
    CK(hipMalloc((void**)&A8, (size_t)N * M * 2));  // (large enough for the bf16 run too)
    CK(hipMalloc((void**)&R8, (size_t)B * M * 2));
    CK(hipMalloc((void**)&cv, (size_t)B * (N / 128) * 4 * 4));
    CK(hipMalloc((void**)&ci, (size_t)B * (N / 128) * 4 * 4));
    CK(hipMalloc((void**)&sc, B * 4));
    std::vector<float> hs(B, 1.0f);
    CK(hipMemcpy(sc, hs.data(), B * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill_i8, dim3(4096), dim3(256), 0, 0, A8, (size_t)N * M * 2, 17u);
    hipLaunchKernelGGL(k_fill_i8, dim3(512), dim3(256), 0, 0, R8, (size_t)B * M * 2, 99u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_at = (int)(N / 128), n_st = B / 128;
    for (int mode : {kScreen256p, kScreen256i8}) {
        const int Mk = mode == kScreen256i8 ? M / 2 : M;  // 2-byte slots per row
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            CK(launch_screen(0, mode, (const __bf16*)A8, (const __bf16*)R8, Mk, n_at, n_st, N, cv, ci, sc));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = 2.0 * M * (double)N * B;
            if (rep) printf("%s: %.1f us  = %.2f P(FL)OP/s\n", mode == kScreen256i8 ? "int8  (i32_16x16x64_i8) " : "bf16 (f32_16x16x32_bf16)", ms * 1e3, ops / ms * 1e-12);
        }
        if (mode == kScreen256i8) {  // check: the largest |dot| over the first 128 atoms for signal 3 (rows of M int8)
            std::vector<signed char> ha((size_t)128 * M), hr(M);
            CK(hipMemcpy(ha.data(), A8, ha.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), R8 + (size_t)3 * M, M, hipMemcpyDeviceToHost));
            long best = -1; int bi = -1;
            for (int a = 0; a < 128; ++a) {
                long d = 0;
                for (int m = 0; m < M; ++m) d += (long)ha[(size_t)a * M + m] * hr[m];
                if (labs(d) > best) { best = labs(d); bi = a; }
            }
            float gv[4]; int gi[4];
            CK(hipMemcpy(gv, cv + ((size_t)3 * n_at + 0) * 4, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gi, ci + ((size_t)3 * n_at + 0) * 4, 16, hipMemcpyDeviceToHost));
            printf("check signal 3, atom tile 0: host best |dot| %ld at atom %d; kernel candidates (%d: %.0f) (%d: %.0f) -> %s\n", best, bi, gi[0], gv[0], gi[1],
                   gv[1], (gi[0] == bi && fabs(gv[0] - (double)best) <= 1e-4 * best + 256) ? "OK" : "MISMATCH");
        }
    }
    return 0;
}
