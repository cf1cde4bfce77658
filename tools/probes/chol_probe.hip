// chol_probe.hip -- microseconds per step of the blocked Cholesky chain (csrc/csmp_gram.hpp) on a synthetic SPD matrix,
// n = 1024 (+ the bordered column), the launch sequence of ls_gram_t.  Checks the factor against a host Cholesky.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../compressedsensing.jl_amd/csrc -o chol_probe chol_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "csmp_gram.hpp"
using namespace csmp;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int np = (n + 1 + 63) / 64 * 64;
    std::vector<double> G((size_t)np * np, 0.0), ref;
    srand(7);
    for (int j = 0; j < np; ++j)
        for (int i = 0; i <= j; ++i) {
            double v = (i == j) ? 1.0 : 0.02 * ((double)rand() / RAND_MAX - 0.5);
            if (i > n || j > n) v = (i == j) ? 1.0 : 0.0;  // identity padding
            G[i + (size_t)j * np] = v;
            G[j + (size_t)i * np] = v;
        }
    G[n + (size_t)n * np] = 1.0e3;  // the corner b'b
    std::vector<double> gd(np);
    for (int i = 0; i < np; ++i) gd[i] = G[i + (size_t)i * np];
    // host reference: upper R with G = R'R (columns 0..n)
    ref = G;
    const int nn = n + 1;
    for (int p = 0; p < nn; ++p) {
        double d = ref[p + (size_t)p * np];
        for (int t = 0; t < p; ++t) d -= ref[t + (size_t)p * np] * ref[t + (size_t)p * np];
        d = sqrt(d);
        ref[p + (size_t)p * np] = d;
        for (int c = p + 1; c < nn; ++c) {
            double s = ref[p + (size_t)c * np];
            for (int t = 0; t < p; ++t) s -= ref[t + (size_t)p * np] * ref[t + (size_t)c * np];
            ref[p + (size_t)c * np] = s / d;
        }
    }
    double *dG, *dG0, *dgd, *dfac;
    DevState* st;
    CK(hipMalloc(&dG, G.size() * 8));
    CK(hipMalloc(&dG0, G.size() * 8));
    CK(hipMalloc(&dgd, np * 8));
    CK(hipMalloc(&dfac, (size_t)np * kCholNB * 8));  // the factored diagonal blocks (side buffer of chol_row_body)
    CK(hipMalloc(&st, sizeof(DevState)));
    CK(hipMemset(st, 0, sizeof(DevState)));
    CK(hipMemcpy(dG0, G.data(), G.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgd, gd.data(), np * 8, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int nsteps = (n + kCholNB - 1) / kCholNB;  // (as ls_gram_t: the corner of the bordered matrix is never needed)
    auto chain = [&]() {
        const int left0 = np - kCholNB;
        hipLaunchKernelGGL(k_chol_row, dim3(std::max(1, (left0 + kCholRowCols - 1) / kCholRowCols)), dim3(kCholThreads), 0, s, dG, np, n, 0,
                           (const double*)dgd, st, dfac);
        for (int kb = 0; kb + 1 < nsteps; ++kb) {
            const int left = np - (kb + 1) * kCholNB, left2 = left - kCholNB;
            const int Tt = (left + kGramTile - 1) / kGramTile;
            const int ntrail = left > kCholNB ? Tt * (Tt + 1) / 2 : 0;
            const int nrow = std::max(1, (left2 + kCholRowCols - 1) / kCholRowCols);
            hipLaunchKernelGGL(k_chol_step, dim3(nrow + ntrail), dim3(kCholThreads), 0, s, dG, np, n, kb, (const double*)dgd, st,
                               nrow, dfac);
        }
    };
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemcpyAsync(dG, dG0, G.size() * 8, hipMemcpyDeviceToDevice, s));
        CK(hipEventRecord(e0, s));
        chain();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    std::vector<double> out(G.size());
    CK(hipMemcpy(out.data(), dG, G.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> fac((size_t)np * kCholNB);
    CK(hipMemcpy(fac.data(), dfac, fac.size() * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < nn; ++j)  // diagonal blocks: from the side buffer
        for (int i = (j / kCholNB) * kCholNB; i <= j; ++i)
            out[i + (size_t)j * np] = fac[(size_t)(j / kCholNB) * kCholNB * kCholNB + (i % kCholNB) + (size_t)(j % kCholNB) * kCholNB];
    DevState hs;
    CK(hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost));
    double err = 0.0;
    for (int j = 0; j < nn; ++j)
        for (int i = 0; i <= j && i < n; ++i) err = fmax(err, fabs(out[i + (size_t)j * np] - ref[i + (size_t)j * np]));  // rows < n
    printf("n %d np %d steps %d: %.1f us per factorisation, %.2f us per step; max |R - R_ref| = %.3e; done flags 0x%x\n", n, np, nsteps,
           best * 1e3, best * 1e3 / nsteps, err, hs.done);
    return err < 1e-10 ? 0 : 2;
}
