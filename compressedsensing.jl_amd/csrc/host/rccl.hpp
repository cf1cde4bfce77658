// host/rccl.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// the signal-sharded solve with the ONE collective inside the library (BASELINE configs[3]): every rank
// (one process per GPU) solves its contiguous block of signals with csmp_omp_batch / csmp_omp_batch_mfma, the results stay on
// the device, are packed there into rows of 2k+1 Float64 (csmp_pack_results' wire layout), moved by ONE ncclAllGather over
// xGMI on the context's stream, and unpacked on the device into global signal order.  A host language needs no collective
// library of its own (the reference's Project.toml has none): the 128-byte communicator id is the only thing ranks exchange,
// by whatever means they were started with.
//
// RCCL is bound LAZILY (dlopen of librccl.so.1 at csmp_comm_id / csmp_comm_init): libcsmp.so does not link it, so a host that
// never shards pays nothing, and a process that already holds an RCCL (PyTorch's) shares that one -- one collective runtime
// per process.  Only the five entry points below are resolved.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only; no symbol of it is referenced at link time

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

static_assert(CSMP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "csmp.h's id size is RCCL's");

bool rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.AllGather) return true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        const char* e = dlerror();
        g_rccl.why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?");
        return false;
    }
    auto sym = [&](const char* n) -> void* {
        void* p = dlsym(g_rccl.handle, n);
        if (!p) g_rccl.why = std::string("librccl has no ") + n;
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    void* ag = sym("ncclAllGather");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GetErrorString || !ag) return false;
    g_rccl.AllGather = (decltype(g_rccl.AllGather))ag;  // (set last: it is the "loaded" flag)
    return true;
}
int rccl_fail(csmp_ctx* ctx, const char* what, ncclResult_t r) {
    return fail(ctx, CSMP_ERCCL, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
}  // namespace

// rows of 2k+1 Float64 [idx | val | nnz] from the batch drivers' device outputs (idx, val: k x nloc, nnz: nloc); rows >= nloc: zeros
// status < 0 (this rank's block could not be solved): the block carries no results and the nnz slot of its row 0 carries the status --
// a count is never negative, so the slot tells every receiver that the rank failed, and with what code (csmp_omp_sharded)
__global__ void k_pack_rows(const int64_t* __restrict__ idx, const double* __restrict__ val, const int64_t* __restrict__ nnz, int64_t k,
                            int64_t nloc, int64_t rows, double* __restrict__ packed, int status) {
    const int64_t w = 2 * k + 1, n = rows * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t s = i / w, t = i % w;
        double v = 0.0;
        if (status < 0) v = (s == 0 && t == 2 * k) ? (double)status : 0.0;
        else if (s < nloc) v = t < k ? (double)idx[s * k + t] : t < 2 * k ? val[s * k + (t - k)] : (double)nnz[s];
        packed[i] = v;
    }
}
// the gathered blocks (world x rows x (2k+1)) -> idx / val (k x nsig) and nnz (nsig) in global signal order; rank r owns the
// contiguous block csmp_shard_range(nsig, r, world): base = nsig / world signals, the first nsig % world ranks one more
__global__ void k_unpack_rows(const double* __restrict__ all, int64_t k, int64_t nsig, int world, int64_t rows, int64_t* __restrict__ idx,
                              double* __restrict__ val, int64_t* __restrict__ nnz) {
    const int64_t w = 2 * k + 1, base = nsig / world, extra = nsig % world;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nsig * w; i += (int64_t)gridDim.x * 256) {
        const int64_t s = i / w, t = i % w;
        const int64_t cut = extra * (base + 1);  // signals below `cut` live on the ranks that hold base + 1
        const int64_t r = s < cut ? s / (base + 1) : extra + (s - cut) / (base > 0 ? base : 1);
        const int64_t lo = r * base + (r < extra ? r : extra);
        const double v = all[(r * rows + (s - lo)) * w + t];
        if (t < k) idx[s * k + t] = (int64_t)v;
        else if (t < 2 * k) val[s * k + (t - k)] = v;
        else nnz[s] = (int64_t)v;
    }
}

// The two device-side halves of the exchange on their own (csmp_omp_sharded runs them around its ncclAllGather): for a host that keeps
// results on the device and brings its own collective, and so that the layout arithmetic for world > 1 can be exercised on one GPU.
extern "C" int csmp_pack_block_device(csmp_ctx* ctx, const int64_t* idx, const double* val, const int64_t* nnz, int64_t k, int64_t nloc,
                                      int64_t rows, double* packed) {
    if (!ctx) return CSMP_EINVAL;
    if (!packed || k < 1 || nloc < 0 || rows < nloc || (nloc > 0 && (!idx || !val || !nnz))) return fail(ctx, CSMP_EINVAL, "pack_block_device: bad arguments");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int64_t w = 2 * k + 1;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (rows * w + 255) / 256));
    hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, ctx->stream, idx, val, nnz, k, nloc, rows, packed, 0);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}
extern "C" int csmp_unpack_gathered_device(csmp_ctx* ctx, const double* gathered, int64_t k, int64_t nsig, int world, int64_t* idx, double* val,
                                           int64_t* nnz) {
    if (!ctx) return CSMP_EINVAL;
    if (!gathered || !idx || !val || !nnz || k < 1 || nsig < 1 || world < 1) return fail(ctx, CSMP_EINVAL, "unpack_gathered_device: bad arguments");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int64_t w = 2 * k + 1, rows = (nsig + world - 1) / world;
    const int grid = (int)std::min<int64_t>(1024, (nsig * w + 255) / 256);
    hipLaunchKernelGGL(k_unpack_rows, dim3(grid), dim3(256), 0, ctx->stream, gathered, k, nsig, world, rows, idx, val, nnz);
    HIPCHECK(hipGetLastError());
    return CSMP_OK;
}

extern "C" int csmp_comm_id(void* id) {
    if (!id) return CSMP_EINVAL;
    if (!rccl_load()) {
        g_create_err = g_rccl.why;
        return CSMP_ERCCL;
    }
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) {
        g_create_err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r);
        return CSMP_ERCCL;
    }
    memcpy(id, u.internal, CSMP_COMM_ID_BYTES);
    return CSMP_OK;
}

extern "C" int csmp_comm_free(csmp_ctx* ctx) {
    if (!ctx) return CSMP_EINVAL;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    return CSMP_OK;
}

extern "C" int csmp_comm_init(csmp_ctx* ctx, const void* id, int rank, int world) {
    if (!ctx) return CSMP_EINVAL;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(ctx, CSMP_EINVAL, "comm_init: bad id / rank / world");
    if (!rccl_load()) return fail(ctx, CSMP_ERCCL, g_rccl.why);
    HIPCHECK(hipSetDevice(ctx->dev));
    CHECK(csmp_comm_free(ctx));
    ncclUniqueId u;
    memcpy(u.internal, id, CSMP_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, world, u, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", r);
    ctx->comm = (void*)c;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return CSMP_OK;
}

// The call is COLLECTIVE: every rank must reach the same collectives with the same counts, or the ranks that did wait for ever.
// So nothing that can differ between the ranks returns before a collective has carried it to all of them:
//   1. AGREEMENT.  The arguments that size the gather -- nsig, k, the element type of B, the method, eps -- and this rank's
//      verdict on its own arguments travel first, in an all-gather of kHdr doubles per rank (a count no argument can change).  Ranks
//      that were called with different arguments, or any rank with invalid ones, make EVERY rank return CSMP_EINVAL here, before any
//      solve and before the gather whose counts would not have matched.  (Only a missing communicator and a NULL context return
//      earlier: such a rank cannot take part in anything.)
//   2. The gather's own two buffers (and the header's) are the one failure that cannot be reported: without them this rank cannot
//      take part.  From then on a failure of THIS rank -- its B missing, its temporaries, its block's solves (CSMP_EDIM,
//      CSMP_ENOMEM, CSMP_ERANGE, a HIP error ...) -- is REMEMBERED, carried through the gather as a status in the block
//      (k_pack_rows) and returned AFTER it, on every rank: the failing rank gets its own code and message, the others the same
//      code and the failing rank's number.  No rank's outputs are valid then.  A HIP error left pending by the failed part is
//      cleared on the spot (hipGetLastError): read later, it would make this rank leave in front of the gather.
constexpr int kHdr = 8;
extern "C" int csmp_omp_sharded(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k, double eps,
                                int method, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!ctx->comm) return fail(ctx, CSMP_ESTATE, "omp_sharded: no communicator (csmp_comm_init)");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    // ---- 1. agreement
    const char* argerr = nullptr;
    if ((b_loc != CSMP_HOST && b_loc != CSMP_DEVICE) || (out_loc != CSMP_HOST && out_loc != CSMP_DEVICE)) argerr = "b_loc / out_loc must be CSMP_HOST or CSMP_DEVICE";
    else if (nsig < 0 || k < 1 || !idx || !val || !nnz || (method != 0 && method != 1)) argerr = "omp_sharded: bad arguments";
    else if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) argerr = "b_dtype must be CSMP_F32 or CSMP_F64";
    else if (!(eps >= 0.0)) argerr = "eps has to be non-negative";  // src/matchingpursuit.jl:74
    {
        DevTmp tHdr, tHdrAll;
        HIPCHECK(tHdr.alloc(kHdr * sizeof(double)));
        HIPCHECK(tHdrAll.alloc((size_t)world * kHdr * sizeof(double)));
        double h[kHdr] = {(double)nsig, (double)k, (double)b_dtype, (double)method, eps == eps ? eps : -1.0, argerr ? 1.0 : 0.0, 0.0, 0.0};
        HIPCHECK(hipMemcpyAsync(tHdr.p, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
        const ncclResult_t rh = g_rccl.AllGather(tHdr.p, tHdrAll.p, (size_t)kHdr, ncclDouble, (ncclComm_t)ctx->comm, ctx->stream);
        if (rh != ncclSuccess) return rccl_fail(ctx, "ncclAllGather (agreement)", rh);
        std::vector<double> all((size_t)world * kHdr);
        HIPCHECK(hipMemcpyAsync(all.data(), tHdrAll.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (argerr) return fail(ctx, CSMP_EINVAL, argerr);
        for (int q = 0; q < world; ++q) {
            const double* hq = all.data() + (size_t)q * kHdr;
            if (hq[5] != 0.0)
                return fail(ctx, CSMP_EINVAL, "omp_sharded: rank " + std::to_string(q) + " was called with invalid arguments; no rank solves anything");
            if (memcmp(hq, h, 5 * sizeof(double)) != 0)
                return fail(ctx, CSMP_EINVAL, "omp_sharded: the ranks disagree on (nsig, k, b_dtype, method, eps): rank " + std::to_string(q) + " has (" +
                                                  std::to_string((int64_t)hq[0]) + ", " + std::to_string((int64_t)hq[1]) + ", " + std::to_string((int)hq[2]) + ", " +
                                                  std::to_string((int)hq[3]) + ", " + std::to_string(hq[4]) + "), rank " + std::to_string(rank) + " (" +
                                                  std::to_string(nsig) + ", " + std::to_string(k) + ", " + std::to_string(b_dtype) + ", " + std::to_string(method) +
                                                  ", " + std::to_string(eps) + "); no rank solves anything");
        }
    }
    if (nsig == 0) return CSMP_OK;  // (on every rank: they agree)
    int64_t lo = 0, hi = 0;
    CHECK(csmp_shard_range(nsig, rank, world, &lo, &hi));
    const int64_t nloc = hi - lo, rows = (nsig + world - 1) / world, w = 2 * k + 1;
    DevTmp tIdx, tVal, tNnz, tPack, tAll, oIdx, oVal, oNnz;
    HIPCHECK(tPack.alloc((size_t)(rows * w) * 8));
    HIPCHECK(tAll.alloc((size_t)(world * rows * w) * 8));
    // ---- this rank's part: a failure is remembered, not returned
    int local = CSMP_OK;
    std::string local_msg;
    auto note = [&](int rc) {
        if (rc != CSMP_OK && local == CSMP_OK) {
            local = rc;
            local_msg = ctx->err;
        }
    };
    if (nloc > 0 && !B) note(fail(ctx, CSMP_EINVAL, "omp_sharded: B == NULL"));
    if (local == CSMP_OK && (tIdx.alloc((size_t)std::max<int64_t>(1, k * nloc) * 8) != hipSuccess || tVal.alloc((size_t)std::max<int64_t>(1, k * nloc) * 8) != hipSuccess ||
                             tNnz.alloc((size_t)std::max<int64_t>(1, nloc) * 8) != hipSuccess)) {
        (void)hipGetLastError();
        note(fail(ctx, CSMP_ENOMEM, "omp_sharded: no device memory for this rank's results"));
    }
    int64_t *d_idx = idx, *d_nnz = nnz;
    double* d_val = val;
    if (local == CSMP_OK && out_loc == CSMP_HOST) {
        if (oIdx.alloc((size_t)(k * nsig) * 8) != hipSuccess || oVal.alloc((size_t)(k * nsig) * 8) != hipSuccess || oNnz.alloc((size_t)nsig * 8) != hipSuccess) {
            (void)hipGetLastError();
            note(fail(ctx, CSMP_ENOMEM, "omp_sharded: no device memory for the gathered results"));
        }
        d_idx = (int64_t*)oIdx.p;
        d_val = (double*)oVal.p;
        d_nnz = (int64_t*)oNnz.p;
    }
    if (local == CSMP_OK && nloc > 0)  // the block's solves: results stay in device memory
        note(method == 1
                 ? csmp_omp_batch_mfma(ctx, B, b_dtype, ldB, nloc, b_loc, k, eps, (int64_t*)tIdx.p, (double*)tVal.p, (int64_t*)tNnz.p, CSMP_DEVICE)
                 : csmp_omp_batch(ctx, B, b_dtype, ldB, nloc, b_loc, k, eps, (int64_t*)tIdx.p, (double*)tVal.p, (int64_t*)tNnz.p, CSMP_DEVICE));
    if (local != CSMP_OK) {
        // whatever the failed part left behind must not reach the gather: a pending HIP error (an allocation that failed inside the
        // block's solves leaves hipErrorOutOfMemory for the next hipGetLastError) and work still queued on the stream
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
    }
    {
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (rows * w + 255) / 256));
        hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t*)tIdx.p, (const double*)tVal.p, (const int64_t*)tNnz.p, k,
                           local == CSMP_OK ? nloc : 0, rows, (double*)tPack.p, local);
        const hipError_t pe = hipGetLastError();  // (remembered like every other failure of this rank: the gather comes first)
        if (pe != hipSuccess) note(fail(ctx, CSMP_EHIP, std::string("omp_sharded: packing this rank's block: ") + hipGetErrorString(pe)));
    }
    // ---- THE collective: every rank's packed block, device memory to device memory, ordered on the context's stream
    const ncclResult_t r = g_rccl.AllGather(tPack.p, tAll.p, (size_t)(rows * w), ncclDouble, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclAllGather", r);
    // every rank's status word (the nnz slot of its row 0) comes down beside the results
    std::vector<double> status((size_t)world, 0.0);
    HIPCHECK(hipMemcpy2DAsync(status.data(), sizeof(double), (const double*)tAll.p + 2 * k, (size_t)(rows * w) * sizeof(double), sizeof(double), (size_t)world,
                              hipMemcpyDeviceToHost, ctx->stream));
    if (local == CSMP_OK) {
        CHECK(csmp_unpack_gathered_device(ctx, (const double*)tAll.p, k, nsig, world, d_idx, d_val, d_nnz));
        if (out_loc == CSMP_HOST) {
            HIPCHECK(hipMemcpyAsync(idx, d_idx, (size_t)(k * nsig) * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(val, d_val, (size_t)(k * nsig) * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(nnz, d_nnz, (size_t)nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    HIPCHECK(hipStreamSynchronize(ctx->stream));  // (the temporaries are released on return)
    if (local != CSMP_OK) return fail(ctx, local, local_msg);
    for (int q = 0; q < world; ++q)
        if (status[(size_t)q] < 0.0)
            return fail(ctx, (int)status[(size_t)q], "omp_sharded: rank " + std::to_string(q) + " could not solve its block (status " +
                                                          std::to_string((int)status[(size_t)q]) + "); no rank's results are valid");
    return CSMP_OK;
}
