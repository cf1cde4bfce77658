// finish_probe.hip -- microseconds of the blocked back substitution k_finish_b (csrc/csmp_kernels.hpp) on a random upper
// triangular R (n = 1024 and 512, leading dimension 1024), checked against a host solve (solution and sorted emission), R either just
// written by a many-workgroup kernel (default) or cold (argv[1] = cold).  -DFIN_VARIANT=2: the super-block form k_trsv_*.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../compressedsensing.jl_amd/csrc -o finish_probe finish_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "csmp_kernels.hpp"
using namespace csmp;
#define KERN k_finish_b
#define NTHR 256
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int kcap = 1024;
    const bool cold = argc > 1 && argv[1][0] == 'c';
    for (int n : {1024, 512}) {
        std::vector<double> R((size_t)kcap * kcap, 0.0), z(kcap), x(n);
        srand(5);
        for (int j = 0; j < n; ++j) {
            for (int i = 0; i < j; ++i) R[i + (size_t)j * kcap] = 0.05 * ((double)rand() / RAND_MAX - 0.5);
            R[j + (size_t)j * kcap] = 1.0 + 0.1 * (double)rand() / RAND_MAX;
            z[j] = (double)rand() / RAND_MAX - 0.5;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = z[i];
            for (int j = i + 1; j < n; ++j) s -= R[i + (size_t)j * kcap] * x[j];
            x[i] = s / R[i + (size_t)i * kcap];
        }
        std::vector<int> sel(kcap);
        for (int i = 0; i < kcap; ++i) sel[i] = (i * 7919) % 100003;  // distinct pseudo atoms
        double *dR, *dz, *dcoef, *dval;
        int* dsel;
        int64_t *didx, *dnnz, *dord;
        DevState hs{};
        hs.nsel = n;
        DevState* dst;
        CK(hipMalloc(&dR, R.size() * 8)); CK(hipMalloc(&dz, kcap * 8)); CK(hipMalloc(&dcoef, kcap * 8)); CK(hipMalloc(&dval, kcap * 8));
        CK(hipMalloc(&dsel, kcap * 4)); CK(hipMalloc(&didx, kcap * 8)); CK(hipMalloc(&dnnz, 8)); CK(hipMalloc(&dord, kcap * 8));
        CK(hipMalloc(&dst, sizeof hs));
        double* dR0;
        CK(hipMalloc(&dR0, R.size() * 8));
        CK(hipMemcpy(dR0, R.data(), R.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dz, z.data(), kcap * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsel, sel.data(), kcap * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dst, &hs, sizeof hs, hipMemcpyHostToDevice));
        const size_t lds = (size_t)(kcap + 64) * 8 + (size_t)kcap * 4 + 8192;
        CK(hipFuncSetAttribute((const void*)k_finish_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 8; ++rep) {
            // R as the solver finds it: just written by a many-workgroup kernel (here: a device-to-device copy), or cold (argv[1] = cold)
            if (cold) CK(hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice));
            else CK(hipMemcpy(dR, dR0, R.size() * 8, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(e0, 0));
#if defined(FIN_VARIANT) && FIN_VARIANT == 2
            {
                const int nsb = (n + kTrsvBlk - 1) / kTrsvBlk;
                for (int sb = nsb - 1; sb >= 0; --sb) {
                    const int off = sb * kTrsvBlk;
                    hipLaunchKernelGGL(k_trsv_blk, dim3(1), dim3(256), 0, 0, (const double*)dR, (const double*)dz, (const DevState*)dst, kcap, dcoef,
                                       off, sb == nsb - 1 ? 1 : 0);
                    if (sb > 0) hipLaunchKernelGGL(k_trsv_upd, dim3(off / 64), dim3(256), 0, 0, (const double*)dR, (const DevState*)dst, kcap, dcoef, off);
                }
                hipLaunchKernelGGL(k_trsv_emit, dim3((n + 255) / 256), dim3(256), (size_t)(n + 4) * 4, 0, (const double*)dcoef, (const int*)dsel,
                                   (const DevState*)dst, didx, dval, dnnz, dord, n, (int*)nullptr);
            }
#else
            hipLaunchKernelGGL(KERN, dim3(1), dim3(NTHR), lds, 0, (const double*)dR, (const double*)dz, (const int*)dsel,
                               (const DevState*)dst, kcap, dcoef, didx, dval, dnnz, dord, n, (int*)nullptr);
#endif
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        std::vector<double> got(n);
        CK(hipMemcpy(got.data(), dcoef, n * 8, hipMemcpyDeviceToHost));
        double err = 0;
        for (int i = 0; i < n; ++i) err = fmax(err, fabs(got[i] - x[i]));
        {   // the emission: out_idx ascending, out_val[rank of sel[t]] == x[t]
            std::vector<int64_t> oi(n);
            std::vector<double> ov(n);
            CK(hipMemcpy(oi.data(), didx, n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ov.data(), dval, n * 8, hipMemcpyDeviceToHost));
            for (int t = 0; t + 1 < n; ++t) if (!(oi[t] < oi[t + 1])) err = 1.0;
            for (int t = 0; t < n; ++t) {
                int rank = 0;
                for (int u = 0; u < n; ++u) rank += sel[u] < sel[t];
                if (oi[rank] != sel[t] || fabs(ov[rank] - x[t]) > 1e-9) err = 2.0;
            }
        }
        printf("%s n %d: %.1f us; max |x - x_ref| = %.3e\n", cold ? "cold" : "warm", n, best * 1e3, err);
    }
    return 0;
}
