// cumask_probe.hip -- which compute units does a stream made by hipExtStreamCreateWithCUMask dispatch to, and what does
// a streaming kernel / a latency-bound kernel gain or lose when two streams with DISJOINT masks run side by side?
// Build: hipcc -O3 --offload-arch=gfx950 -o cumask_probe cumask_probe.hip ; run: ./cumask_probe
// (1) mask = the first n bits, n = 32, 80, 128, 176, 256: histogram of HW_REG_XCC_ID and distinct (xcc, se, sh, cu) ids seen.
// (2) a streaming read kernel (16 B per lane, nt) on n CUs: GB/s against n -- the per-CU rate an HBM-bound kernel gets
//     when fewer CUs contend ("CU-partitioned concurrency").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_where(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // hold the CU for a while so that the grid spreads over every CU the mask allows
    long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ p, size_t n, float* sink) {
    f32x4 acc = (f32x4)0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n; i += stride) acc += __builtin_nontemporal_load(p + i);
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x;
}

static hipStream_t masked_stream(int lo, int hi) {  // CUs [lo, hi) of the logical mask
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = lo; b < hi; ++b) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    return s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int nblk = 4096;
    unsigned* d;
    CK(hipMalloc((void**)&d, nblk * 8));
    std::vector<unsigned> h(2 * nblk);
    const int ranges[][2] = {{0, 32}, {0, 80}, {80, 256}, {0, 128}, {128, 256}, {0, 256}};
    for (auto& r : ranges) {
        hipStream_t s = masked_stream(r[0], r[1]);
        CK(hipMemsetAsync(d, 0xff, nblk * 8, s));
        hipLaunchKernelGGL(k_where, dim3(nblk), dim3(64), 0, s, d);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
        int perx[16] = {0};
        std::set<unsigned> cus;
        for (int b = 0; b < nblk; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            perx[xcc]++;
            cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
        }
        int cux[16] = {0};
        for (unsigned c : cus) cux[c >> 12]++;
        printf("mask bits [%3d,%3d): distinct CUs seen %3zu; per XCC:", r[0], r[1], cus.size());
        for (int x = 0; x < 8; ++x) printf(" %d", cux[x]);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    // (2) streaming bandwidth against the number of CUs, alone and with a second masked stream running the same kernel
    const size_t bytes = (size_t)2 << 30;
    f32x4* buf;
    float* sink;
    CK(hipMalloc((void**)&buf, bytes));
    CK(hipMalloc((void**)&sink, 4));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int ns[] = {32, 48, 64, 80, 96, 128, 176, 256};
    for (int n : ns) {
        hipStream_t s = masked_stream(0, n);
        for (int wg = 1; wg <= 4; wg *= 2) {
            hipLaunchKernelGGL(k_stream, dim3(n * wg), dim3(256), 0, s, buf, bytes / 16, sink);
            CK(hipEventRecord(e0, s));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_stream, dim3(n * wg), dim3(256), 0, s, buf, bytes / 16, sink);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("stream on %3d CUs, %d WG/CU: %7.1f GB/s (%5.1f GB/s per CU)\n", n, wg, 3.0 * bytes / ms * 1e-6, 3.0 * bytes / ms * 1e-6 / n);
        }
        CK(hipStreamDestroy(s));
    }
    printf("done\n");
    return 0;
}
