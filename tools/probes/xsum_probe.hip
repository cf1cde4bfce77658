// Round 6 probe: the wave-wide butterfly sum of a Float64 (v += xor 32, 16, 8, 4, 2, 1) without LDS trips -- v_permlane32_swap /
// v_permlane16_swap (gfx950) for the two cross-row steps, DPP for the four in-row steps -- against the __shfl_xor form
// (ds_bpermute: six dependent LDS round trips).  Same pairs in the same order, additions are commutative: the bits must agree.
// hipcc -O3 --offload-arch=gfx950 -o xsum_probe xsum_probe.hip && ./xsum_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../compressedsensing.jl_amd/csrc/csmp_kernels.hpp"

__global__ void k_check(const double* in, double* fast, double* ref) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    double v = in[t], w = v;
    fast[t] = csmp::wave_xsum(v);
    for (int s = 32; s >= 1; s >>= 1) w += __shfl_xor(w, s, 64);
    ref[t] = w;
}
template <bool FAST>
__global__ void k_time(const double* in, double* out, int reps) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    double v = in[t], acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double w = v + acc * 1e-30;  // a dependent chain, as in the sweep: one reduction per column
        if (FAST)
            w = csmp::wave_xsum(w);
        else
            for (int s = 32; s >= 1; s >>= 1) w += __shfl_xor(w, s, 64);
        acc += w;
    }
    out[t] = acc;
}
int main() {
    const int n = 256 * 64;
    std::vector<double> h(n);
    unsigned long long sd = 88172645463325252ull;
    for (auto& x : h) {
        sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17;
        x = ((double)(sd >> 11) / 9007199254740992.0 - 0.5) * ((sd & 7) ? 1.0 : 1e12);
    }
    double *d, *f, *r;
    hipMalloc(&d, n * 8); hipMalloc(&f, n * 8); hipMalloc(&r, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    k_check<<<n / 256, 256>>>(d, f, r);
    std::vector<double> hf(n), hr(n);
    hipMemcpy(hf.data(), f, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), r, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += std::memcmp(&hf[i], &hr[i], 8) != 0;
    printf("bitwise mismatches: %d of %d\n", bad, n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fast = 0; fast < 2; ++fast) {
        const int reps = 20000;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (fast) k_time<true><<<1, 64>>>(d, f, reps); else k_time<false><<<1, 64>>>(d, f, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f ns per reduction (one wave, dependent chain)\n", fast ? "permlane + DPP" : "__shfl_xor     ", ms * 1e6 / reps);
    }
    return bad != 0;
}
