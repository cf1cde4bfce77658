// concurrency_probe.hip -- do two kernels from two HIP streams share the GPU (and a CU) at the same time?
// Spin kernels of a fixed duration (wall clock in-kernel), with a chosen LDS footprint, VGPR footprint and grid.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/concurrency_probe tools/probes/concurrency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int REGS>
__global__ __launch_bounds__(512) void spin(unsigned long long ticks, float* out) {
    extern __shared__ float lds[];
    float acc[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc[i] = threadIdx.x * 0.5f + i;
    lds[threadIdx.x] = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) acc[i] = acc[i] * 1.0001f + lds[threadIdx.x];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

struct Cfg { const char* name; int regs, threads, grid; size_t lds; };

template <int REGS>
static hipError_t launch(const Cfg& c, hipStream_t st, unsigned long long ticks, float* out) {
    if (c.lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)spin<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(spin<REGS>, dim3(c.grid), dim3(c.threads), c.lds, st, ticks, out);
    return hipGetLastError();
}
static hipError_t go(const Cfg& c, hipStream_t st, unsigned long long ticks, float* out) {
    return c.regs >= 180 ? launch<180>(c, st, ticks, out) : c.regs >= 100 ? launch<100>(c, st, ticks, out) : launch<16>(c, st, ticks, out);
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float* out;
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned long long ticks = 30000;  // 300 us
    const Cfg small{"small (16 regs, 1 KiB LDS, 128 WGs x 256)", 16, 256, 128, 1024};
    const Cfg fullA{"A: 512 thr, 128 KiB LDS, ~180 regs, 256 WGs (one per CU, like k_b_screen256)", 180, 512, 256, 128 * 1024};
    const Cfg fullB{"B: 256 thr, 24 KiB LDS, ~100 regs, 256 WGs (like a slim k_b_step)", 100, 256, 256, 24 * 1024};
    const Cfg fullB2{"B2: 256 thr, 36 KiB LDS, ~180 regs, 256 WGs (like today's k_b_step)", 180, 256, 256, 36 * 1024};
    const Cfg a512{"A x2 rounds: 512 WGs", 180, 512, 512, 128 * 1024};
    const Cfg b512{"B x2 rounds: 512 WGs", 100, 256, 512, 24 * 1024};
    struct Pair { Cfg x, y; };
    std::vector<Pair> tests = {{small, small}, {fullA, fullB}, {fullB, fullA}, {fullA, fullB2}, {fullA, fullA}, {a512, b512}};
    for (auto& t : tests) {
        for (int rep = 0; rep < 2; ++rep) {  // first rep warms
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1));
            CK(go(t.x, s1, ticks, out));
            CK(go(t.y, s2, ticks, out));
            CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(e1, s1));
            CK(hipStreamSynchronize(s1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            auto t0 = std::chrono::steady_clock::now();
            (void)t0;
            if (rep == 1) printf("%-90s || %-80s : both done after %.0f us (one alone = 300 us per round)\n", t.x.name, t.y.name, ms * 1e3);
        }
    }
    // wall-clock variant (events on s1 only see s1): time both streams from the host
    for (auto& t : tests) {
        CK(hipDeviceSynchronize());
        auto h0 = std::chrono::steady_clock::now();
        CK(go(t.x, s1, ticks, out));
        CK(go(t.y, s2, ticks, out));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        auto h1 = std::chrono::steady_clock::now();
        printf("host wall: %-60.60s || %-50.50s : %.0f us\n", t.x.name, t.y.name, std::chrono::duration<double, std::micro>(h1 - h0).count());
    }
    return 0;
}
