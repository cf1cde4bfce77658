// Round 6 probe (verdict item 6): what a Chebyshev / CG solve of G_PP x = c_P could cost in place of Subspace Pursuit's third
// factorisation.  An iteration is one 512 x 512 Float64 matrix-vector product whose input is the previous iteration's output: a chain
// of dependent launches, each the product's rows split over many workgroups (the matrix, 2 MiB, stays in L2).  Timed: 24 and 28
// iterations (kappa = 2.8 with exact / 20 % widened spectral bounds to 1e-13, below) as ONE stream of launches, the way the
// library would enqueue them; beside it the time of the chain it would replace (profiles/r05_bench_sp_single_kernel_stats.csv:
// 16 k_chol_step of 17.8 us + k_chol_row 15.3 us + two k_tt_gemv of 4.7 us = 310 us).
// hipcc -O3 --offload-arch=gfx950 -o cheb_probe cheb_probe.hip && ./cheb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

constexpr int N = 512;
// x_new = x + alpha (c - G x) + beta (x - x_old): ROWS rows per workgroup, one wave per row pair
template <int ROWS>
__global__ __launch_bounds__(256) void k_cheb(const double* __restrict__ G, const double* __restrict__ c, const double* __restrict__ x,
                                              const double* __restrict__ xo, double* __restrict__ xn, double alpha, double beta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * ROWS + wave; r < (blockIdx.x + 1) * ROWS; r += 4) {
        double s = 0.0;
        for (int j = lane; j < N; j += 64) s = fma(G[(size_t)r * N + j], x[j], s);
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) xn[r] = x[r] + alpha * (c[r] - s) + beta * (x[r] - xo[r]);
    }
}
int main() {
    std::vector<double> G((size_t)N * N), c(N), x0(N, 0.0);
    // G = I + E with the spectrum of a Gaussian sub-dictionary at k / M = 1 / 16: E = (B'B - I) for B (8192 x 512) Gaussian / sqrt(8192)
    unsigned long long sd = 1234567;
    auto rnd = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return ((double)(sd >> 11) / 9007199254740992.0) - 0.5; };
    {
        const int M = 2048;  // (k / M = 1/4 here: kappa ~ 9, harsher than the benchmark's 2.8 -- the timing does not depend on it)
        std::vector<double> B((size_t)M * N);
        for (auto& v : B) v = rnd() * std::sqrt(12.0 / M);
        for (int i = 0; i < N; ++i)
            for (int j = i; j < N; ++j) {
                double s = 0.0;
                for (int m = 0; m < M; ++m) s += B[(size_t)m * N + i] * B[(size_t)m * N + j];
                G[(size_t)i * N + j] = G[(size_t)j * N + i] = s;
            }
    }
    for (auto& v : c) v = rnd();
    double *dG, *dc, *dx[3];
    hipMalloc(&dG, G.size() * 8); hipMalloc(&dc, N * 8);
    for (auto& p : dx) { hipMalloc(&p, N * 8); hipMemcpy(p, x0.data(), N * 8, hipMemcpyHostToDevice); }
    hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), N * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rows : {8, 4, 16})
        for (int iters : {24, 28}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < iters; ++it) {
                    double* xc = dx[it % 3]; double* xo = dx[(it + 2) % 3]; double* xn = dx[(it + 1) % 3];
                    if (rows == 8) k_cheb<8><<<N / 8, 256>>>(dG, dc, xc, xo, xn, 0.9, 0.05);
                    else if (rows == 4) k_cheb<4><<<N / 4, 256>>>(dG, dc, xc, xo, xn, 0.9, 0.05);
                    else k_cheb<16><<<N / 16, 256>>>(dG, dc, xc, xo, xn, 0.9, 0.05);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("%2d rows per workgroup (%3d workgroups), %d dependent iterations: %.1f us = %.2f us per iteration\n", rows, N / rows, iters, best * 1e3, best * 1e3 / iters);
        }
    printf("the chain it would replace: 16 x 17.8 + 15.3 + 2 x 4.7 = 309.5 us (profiles/r05_bench_sp_single_kernel_stats.csv)\n");
    return 0;
}
