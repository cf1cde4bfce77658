// host/track.hpp -- every device allocation, page-locked allocation, host registration, event and stream the library makes goes
// through these wrappers (the macros below rename the HIP calls for the rest of the translation unit), so that ONE internal export,
// csmp_live_resources (include/csmp_internal.h), can say what the library holds at any moment: tests/test_gpu_leaks.py runs hundreds of
// create / set_dictionary / solve / destroy cycles, the failing paths among them, and asserts that it all comes back to zero.
// Counting only: no behaviour changes, the HIP calls are made as written.
#pragma once
#include <atomic>
#include <mutex>
#include <unordered_map>

namespace csmp_track {
static std::mutex mu;
static std::unordered_map<void*, size_t> dev, pinned;
static std::atomic<long long> events{0}, streams{0}, registered{0};

static hipError_t malloc_(void** p, size_t n) {
    const hipError_t e = ::hipMalloc(p, n);
    if (e == hipSuccess && *p) {
        std::lock_guard<std::mutex> g(mu);
        dev[*p] = n;
    }
    return e;
}
static hipError_t free_(void* p) {
    if (p) {
        std::lock_guard<std::mutex> g(mu);
        dev.erase(p);
    }
    return ::hipFree(p);
}
static hipError_t host_malloc_(void** p, size_t n, unsigned flags) {
    const hipError_t e = ::hipHostMalloc(p, n, flags);
    if (e == hipSuccess && *p) {
        std::lock_guard<std::mutex> g(mu);
        pinned[*p] = n;
    }
    return e;
}
static hipError_t host_free_(void* p) {
    if (p) {
        std::lock_guard<std::mutex> g(mu);
        pinned.erase(p);
    }
    return ::hipHostFree(p);
}
static hipError_t host_register_(void* p, size_t n, unsigned flags) {
    const hipError_t e = ::hipHostRegister(p, n, flags);
    if (e == hipSuccess) registered += 1;
    return e;
}
static hipError_t host_unregister_(void* p) {
    const hipError_t e = ::hipHostUnregister(p);
    if (e == hipSuccess) registered -= 1;
    return e;
}
static hipError_t event_create_(hipEvent_t* ev) {
    const hipError_t e = ::hipEventCreate(ev);
    if (e == hipSuccess) events += 1;
    return e;
}
static hipError_t event_create_flags_(hipEvent_t* ev, unsigned flags) {
    const hipError_t e = ::hipEventCreateWithFlags(ev, flags);
    if (e == hipSuccess) events += 1;
    return e;
}
static hipError_t event_destroy_(hipEvent_t ev) {
    const hipError_t e = ::hipEventDestroy(ev);
    if (e == hipSuccess) events -= 1;
    return e;
}
static hipError_t stream_create_(hipStream_t* s) {
    const hipError_t e = ::hipStreamCreate(s);
    if (e == hipSuccess) streams += 1;
    return e;
}
static hipError_t stream_create_flags_(hipStream_t* s, unsigned flags) {
    const hipError_t e = ::hipStreamCreateWithFlags(s, flags);
    if (e == hipSuccess) streams += 1;
    return e;
}
static hipError_t stream_destroy_(hipStream_t s) {
    const hipError_t e = ::hipStreamDestroy(s);
    if (e == hipSuccess) streams -= 1;
    return e;
}
}  // namespace csmp_track

#define hipMalloc(p, n) csmp_track::malloc_((void**)(p), (n))
#define hipFree(p) csmp_track::free_((void*)(p))
#define hipHostMalloc(p, n, f) csmp_track::host_malloc_((void**)(p), (n), (f))
#define hipHostFree(p) csmp_track::host_free_((void*)(p))
#define hipHostRegister(p, n, f) csmp_track::host_register_((void*)(p), (n), (f))
#define hipHostUnregister(p) csmp_track::host_unregister_((void*)(p))
#define hipEventCreate(e) csmp_track::event_create_(e)
#define hipEventCreateWithFlags(e, f) csmp_track::event_create_flags_((e), (f))
#define hipEventDestroy(e) csmp_track::event_destroy_(e)
#define hipStreamCreate(s) csmp_track::stream_create_(s)
#define hipStreamCreateWithFlags(s, f) csmp_track::stream_create_flags_((s), (f))
#define hipStreamDestroy(s) csmp_track::stream_destroy_(s)
