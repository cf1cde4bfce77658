// host/batched.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// the batched (MFMA-screened) OMP driver.
// ------------------------------------------------------------------------------------------ batched (MFMA-screened) OMP
// max |A_ij| into out[0] and sum A_ij^2 into the double behind it (out + 2): the second says how FLAT the dictionary is
__global__ void k_absmax_f32(const float* __restrict__ A, int64_t n, float* out) {
    float m = 0.f;
    double q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = A[i];
        m = fmaxf(m, fabsf(v));
        q = fma((double)v, (double)v, q);
    }
    for (int s = 32; s >= 1; s >>= 1) {
        m = fmaxf(m, __shfl_xor(m, s, 64));
        q += __shfl_xor(q, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
        atomicAdd(reinterpret_cast<double*>(out + 2), q);
    }
}
__global__ void k_absmax_f64(const double* __restrict__ A, int64_t n, float* out) {
    float m = 0.f;
    double q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = A[i];
        m = fmaxf(m, (float)fabs(v));
        q = fma(v, v, q);
    }
    for (int s = 32; s >= 1; s >>= 1) {
        m = fmaxf(m, __shfl_xor(m, s, 64));
        q += __shfl_xor(q, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m * 1.0000002f));
        atomicAdd(reinterpret_cast<double*>(out + 2), q);
    }
}

// max_j |a_j|_2 (rounded up), one wave per column
template <typename TA>
__global__ __launch_bounds__(256) void k_colnorm_max(const TA* __restrict__ A, int64_t ld, int M, int64_t N, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    double acc = 0.0;
    for (int m = lane; m < M; m += 64) {
        const double v = (double)A[col * ld + m];
        acc = fma(v, v, acc);
    }
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint((float)sqrt(acc) * 1.0000002f));
}

// tile counts, padded sizes and max|A| of the current dictionary: what both images need
static int batch_meta(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.meta_valid) return CSMP_OK;
    // K is padded (zeros) to an even number of 64-deep tiles, at least four: what the eight-phase screening kernel needs
    b.Mk = (int)std::max<int64_t>(256, ((ctx->M + 127) / 128) * 128);
    b.Npad = ((ctx->N + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT);  // whole 256-atom tiles
    b.n_atiles = (int)(b.Npad / kBT);
    if (!b.amax) HIPCHECK(hipMalloc((void**)&b.amax, 4 * sizeof(float)));  // [max|A|, -, sum A^2 (a double)]
    HIPCHECK(hipMemsetAsync(b.amax, 0, 4 * sizeof(float), ctx->stream));
    const int64_t nel = ctx->ld * ctx->N;
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_absmax_f32, dim3(2048), dim3(256), 0, ctx->stream, (const float*)ctx->dA, nel, b.amax);
    else
        hipLaunchKernelGGL(k_absmax_f64, dim3(2048), dim3(256), 0, ctx->stream, (const double*)ctx->dA, nel, b.amax);
    HIPCHECK(hipGetLastError());
    float h4[4];
    HIPCHECK(hipMemcpyAsync(h4, b.amax, sizeof h4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.amax_host = h4[0];
    double sumsq;
    memcpy(&sumsq, h4 + 2, sizeof sumsq);
    // root mean square of the entries that exist (padding rows of a caller's leading dimension count as what they hold)
    b.arms_host = (float)std::sqrt(sumsq / ((double)ctx->M * (double)ctx->N));
    b.meta_valid = true;
    return CSMP_OK;
}

// bf16 image of the dictionary [Npad][Mk]
static int batch_dict(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.ab_valid) return CSMP_OK;
    CHECK(batch_meta(ctx));
    HIPCHECK(hipMalloc((void**)&b.Ab, (size_t)b.Npad * b.Mk * sizeof(__bf16)));
    const int64_t total = b.Npad * (b.Mk / 8);
    const int grid = (int)((total + 255) / 256);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_b_convert<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.Ab, b.Mk, b.Npad);
    else
        hipLaunchKernelGGL(k_b_convert<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.Ab, b.Mk, b.Npad);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.ab_valid = true;
    return CSMP_OK;
}

// binary16 image of the dictionary [Npad][Mk] under one power-of-two scale
static int batch_dict16(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.ah_valid) return CSMP_OK;
    CHECK(batch_meta(ctx));
    HIPCHECK(hipMalloc((void**)&b.Ah, (size_t)b.Npad * b.Mk * sizeof(_Float16)));
    b.ascale16 = f16_scale(b.amax_host);
    const int64_t total = b.Npad * (b.Mk / 8);
    const int grid = (int)((total + 255) / 256);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_b_convert_f16<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.Ah, b.Mk, b.Npad, b.ascale16);
    else
        hipLaunchKernelGGL(k_b_convert_f16<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.Ah, b.Mk, b.Npad, b.ascale16);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.ah_valid = true;
    return CSMP_OK;
}

// int8 image of the dictionary under one step (max|A| / 127), for the int8 screen
static int batch_dict8(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.a8_valid) return CSMP_OK;
    CHECK(batch_meta(ctx));
    b.Mk8 = (int)std::max<int64_t>(512, ((ctx->M + 255) / 256) * 256);  // bytes per row: an even number (>= 4) of 128-deep K-tiles
    HIPCHECK(hipMalloc((void**)&b.A8, (size_t)b.Npad * b.Mk8));
    b.astep = b.amax_host > 0.f ? b.amax_host / 127.0f : 1.0f;
    const int64_t total = b.Npad * (b.Mk8 / 16);
    const int grid = (int)((total + 255) / 256);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_b_convert_i8<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.A8, b.Mk8, b.Npad, 1.0f / b.astep);
    else
        hipLaunchKernelGGL(k_b_convert_i8<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.A8, b.Mk8, b.Npad, 1.0f / b.astep);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.a8_valid = true;
    return CSMP_OK;
}

// max_j |a_j|_2 (the deterministic screening bound), computed on first use
static int batch_colnorm(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.anorm_host >= 0.f) return CSMP_OK;
    if (!b.amax) HIPCHECK(hipMalloc((void**)&b.amax, 4 * sizeof(float)));  // (a twin that took its parent's meta has none of its own)
    HIPCHECK(hipMemsetAsync(b.amax, 0, sizeof(float), ctx->stream));
    const unsigned grid = (unsigned)((ctx->N + 3) / 4);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_colnorm_max<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.amax);
    else
        hipLaunchKernelGGL(k_colnorm_max<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld, (int)ctx->M, ctx->N, b.amax);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(&b.anorm_host, b.amax, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return CSMP_OK;
}

static int batch_ensure(csmp_ctx* ctx, int nsig, int kcap) {
    Batch& b = ctx->bt;
    const int Bpad = ((nsig + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT);
    if (b.Bcap >= Bpad && b.kcap >= kcap) return CSMP_OK;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    const int nb = std::max(Bpad, b.Bcap), nk = std::max(kcap, b.kcap);
    batch_free(b, true);
    b.Mr = (int)(((ctx->M + 3) / 4) * 4);
    // all of it or none: a capacity is recorded only over a complete set of buffers (a half-built set under nb x nk would let the
    // next call launch on null pointers)
    auto all = [&]() -> int {
        CHECK(dmalloc(ctx, &b.Rb, (size_t)nb * b.Mk));
        CHECK(dmalloc(ctx, &b.r, (size_t)nb * b.Mr));
        CHECK(dmalloc(ctx, &b.b, (size_t)nb * b.Mr));
        CHECK(dmalloc(ctx, &b.T, (size_t)nb * nk * nk));
        CHECK(dmalloc(ctx, &b.Tt, (size_t)nb * nk * nk));
        CHECK(dmalloc(ctx, &b.z, (size_t)nb * nk));
        CHECK(dmalloc(ctx, &b.sel, (size_t)nb * nk));
        CHECK(dmalloc(ctx, &b.bs, (size_t)nb));
        CHECK(dmalloc(ctx, &b.pick, (size_t)nb));
        CHECK(dmalloc(ctx, &b.cand_val, (size_t)nb * b.n_atiles * kTileCand));
        CHECK(dmalloc(ctx, &b.cand_idx, (size_t)nb * b.n_atiles * kTileCand));
        CHECK(dmalloc(ctx, &b.R8, (size_t)nb * (size_t)std::max<int64_t>(512, ((ctx->M + 255) / 256) * 256)));
        CHECK(dmalloc(ctx, &b.sigscale, (size_t)nb));
        return CSMP_OK;
    };
    const int rc = all();
    if (rc != CSMP_OK) {
        batch_free(b, true);
        return rc;
    }
    b.Bcap = nb;
    b.kcap = nk;
    return CSMP_OK;
}

// G = A'A, Float64 products of the exactly promoted dictionary values, upper triangle (row <= column) of an N x N array:
// the option CSMP_OPT_BATCH_GRAM.  8 N^2 bytes (32 GiB at N = 65536) and 2 M N^2 / 2 flops on the Float64 matrix cores
// (k_gram, csmp_gram.hpp: the dictionary is its own "compact copy") -- once per dictionary, like the bf16 image.
static int batch_gram(csmp_ctx* ctx) {
    Batch& b = ctx->bt;
    if (b.gram_valid) return CSMP_OK;
    const int64_t N = ctx->N;
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)N * (size_t)N * sizeof(double);
    if (need + ((size_t)1 << 30) > free_b) return fail(ctx, CSMP_ENOMEM, "CSMP_OPT_BATCH_GRAM: 8 N^2 bytes of HBM are not available");
    HIPCHECK(hipMalloc((void**)&b.Gm, need));
    // k_gram tiles are 128 x 64 over np columns; np = N need not be a multiple of the tile: rows / columns >= np are clamped and
    // never stored.  One slice of the rows (no partials): rows_per_split = the whole column, which the kernel walks in blocks of
    // 16 rows -- a dictionary whose leading dimension is not a multiple of 16 goes through a zero-padded temporary copy.
    const int np = (int)N;
    const size_t es = ctx->dtype == CSMP_F32 ? 4 : 8;
    const int rows = (int)((ctx->M + 15) / 16 * 16);
    const void* src = ctx->dA;
    int64_t ldo = ctx->ld;
    DevTmp padded;
    if (ctx->ld % 16 != 0) {
        ldo = rows;
        HIPCHECK(padded.alloc((size_t)ldo * (size_t)N * es));
        HIPCHECK(hipMemsetAsync(padded.p, 0, (size_t)ldo * (size_t)N * es, ctx->stream));
        HIPCHECK(hipMemcpy2DAsync(padded.p, (size_t)ldo * es, ctx->dA, (size_t)ctx->ld * es, (size_t)ctx->M * es, (size_t)N, hipMemcpyDeviceToDevice,
                                  ctx->stream));
        src = padded.p;
    }
    const dim3 grid((unsigned)((np + kGramWgJ - 1) / kGramWgJ), (unsigned)((np + kGramWgI - 1) / kGramWgI), 1);
    if (ctx->dtype == CSMP_F32)
        hipLaunchKernelGGL(k_gram<float>, grid, dim3(256), 0, ctx->stream, (const float*)src, ldo, np, rows, b.Gm);
    else
        hipLaunchKernelGGL(k_gram<double>, grid, dim3(256), 0, ctx->stream, (const double*)src, ldo, np, rows, b.Gm);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.Ng = N;
    b.gram_valid = true;
    return CSMP_OK;
}

template <typename TA>
static hipError_t b_pick_launch(csmp_ctx* ctx, hipStream_t stream, int sig0, int nsig, double eps, int check_eps, double cert_abs, double cert_rel,
                                int kwin, double cert_abs2) {
    Batch& b = ctx->bt;
    constexpr int U = sizeof(TA) == 4 ? 16 : 8;  // 64-lane chunks of a column in flight per wave (16 bytes per lane each)
    const size_t lds = b_pick_lds_bytes(ctx->Mv, (int)(16 / sizeof(TA)));
    auto kern = k_b_pick<TA, U>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nsig), dim3(256), lds, stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, (const float*)b.cand_val,
                       (const int*)b.cand_idx, b.n_atiles * kTileCand, (const int*)b.sel, b.bs, b.pick, (const double*)b.r, b.Mr, b.kcap, (int)ctx->M, eps,
                       check_eps, cert_abs, cert_rel, kwin, sig0, cert_abs2);
    return hipGetLastError();
}
// DEPTH of the append kernel: columns whose loads are issued together (registers: DEPTH x NI x 16 bytes per lane)
template <typename TA, int NI, bool GRAM>
static hipError_t b_append_launch(csmp_ctx* ctx, hipStream_t stream, int sig0, int nsig, int img) {
    Batch& b = ctx->bt;
    constexpr int DEPTH = (NI >= 8 || (sizeof(TA) == 8 && NI >= 4)) ? 2 : (NI >= 4 || sizeof(TA) == 8) ? 2 : 4;
    const size_t lds = b_append_lds_bytes(ctx->Mv, (int)(16 / sizeof(TA)), b.kcap);
    auto kern = k_b_append<TA, NI, DEPTH, GRAM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nsig), dim3(256), lds, stream, (const TA*)ctx->dA, ctx->ld, ctx->Mv, (const double*)b.Gm, b.Ng, (const BPick*)b.pick, b.T,
                       b.Tt, b.z, b.sel, b.bs, b.r, b.Mr, b.Rb, b.Mk, b.kcap, (int)ctx->M, sig0, img == kOpI8 ? b.R8 : (signed char*)nullptr, b.Mk8, b.sigscale,
                       img == kOpF16 ? 1.0f / b.ascale16 : b.astep, img);
    return hipGetLastError();
}
template <typename TA>
static hipError_t b_step_dispatch(csmp_ctx* ctx, hipStream_t stream, int sig0, int nsig, double eps, int check_eps, double cert_abs, double cert_rel,
                                  int kwin, bool gram, int img, double cert_abs2) {
    const int groups = (ctx->Mv + 1023) / 1024;
    hipError_t e = b_pick_launch<TA>(ctx, stream, sig0, nsig, eps, check_eps, cert_abs, cert_rel, kwin, cert_abs2);
    if (e != hipSuccess) return e;
#define CSMP_BSTEP(NI)                                                                                                  \
    return gram ? b_append_launch<TA, NI, true>(ctx, stream, sig0, nsig, img) : b_append_launch<TA, NI, false>(ctx, stream, sig0, nsig, img);
    if (groups <= 1) { CSMP_BSTEP(1) }
    if (groups <= 2) { CSMP_BSTEP(2) }
    if (groups <= 4) { CSMP_BSTEP(4) }
    CSMP_BSTEP(8)
#undef CSMP_BSTEP
}

// HBM bytes of the batched path's per-signal state (batch_ensure) at support capacity kc
static size_t batch_bytes_per_signal(const csmp_ctx* ctx, int kc) {
    const Batch& b = ctx->bt;
    const size_t Mr = (size_t)((ctx->M + 3) / 4) * 4;
    return (size_t)b.Mk * 2 + 2 * Mr * 8 + 2 * (size_t)kc * kc * 8 + (size_t)kc * 12 + sizeof(BState) + sizeof(BPick) +
           (size_t)b.n_atiles * kTileCand * 8 + (size_t)std::max<int64_t>(512, ((ctx->M + 255) / 256) * 256) + 4;
}

static int omp_batch_mfma_chunk(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                                double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc);

// The entry point: argument checks, then the batch in as many pieces as the free HBM asks for.  The per-signal state is dominated
// by the two k x k Float64 factors (T = R^-1 and its transpose): 1024 signals at k = 128 hold 0.27 GB, at k = 2048 69 GB.  A batch
// whose state does not fit beside what the GPU already holds is solved in chunks of whole 256-signal tiles, one after the other
// (same results; csmp_batch_stats adds the chunks up); if not even one tile fits, csmp_omp_batch's exact sweeps take the batch.
extern "C" int csmp_omp_batch_mfma(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                                   double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if ((b_loc != CSMP_HOST && b_loc != CSMP_DEVICE) || (out_loc != CSMP_HOST && out_loc != CSMP_DEVICE))
        return fail(ctx, CSMP_EINVAL, "b_loc / out_loc must be CSMP_HOST or CSMP_DEVICE");
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");
    if (!B || nsig < 1 || k < 1 || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "omp_batch_mfma: bad arguments");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    HIPCHECK(hipSetDevice(ctx->dev));
    Batch& b = ctx->bt;
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    const int64_t Bpad = ((nsig + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT);
    if (ctx->Mv <= kBatchMaxRows && !(b.Bcap >= Bpad && b.kcap >= kc)) {  // (state of that size is not there yet)
        CHECK(batch_meta(ctx));  // (Mk, n_atiles: the image's geometry)
        size_t free_b = 0, total_b = 0;
        HIPCHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t held = (size_t)b.Bcap * batch_bytes_per_signal(ctx, b.kcap);
        const size_t margin = (size_t)2 << 30;  // (the operand images, temporaries of the re-solves)
        size_t budget = free_b + held > margin ? free_b + held - margin : 0;
        if (ctx->tune_batch_budget_mib > 0) budget = std::min(budget, (size_t)ctx->tune_batch_budget_mib << 20);  // (csmp_tune: a test's stand-in for a full device)
        const size_t per = batch_bytes_per_signal(ctx, std::max(kc, b.kcap));
        if ((size_t)std::max<int64_t>(Bpad, b.Bcap) * per > budget) {
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            batch_free(b, true);  // (whatever an earlier, differently shaped batch left: the budget counted it as free)
            const int64_t fit = (int64_t)(budget / batch_bytes_per_signal(ctx, kc)) / (2 * kBT) * (2 * kBT);
            if (fit < 2 * kBT) {
                b.last_mode = 0;
                b.last_streams = 1;
                b.last_screen_signals = 0;
                b.last_signals = nsig;
                b.last_resolved = b.last_uncertain = b.last_illcond = 0;
                return csmp_omp_batch(ctx, B, b_dtype, ldB, nsig, b_loc, k, eps, idx, val, nnz, out_loc);
            }
            if (fit < Bpad) {
                const size_t es = b_dtype == CSMP_F32 ? 4 : 8;
                int64_t tot_res = 0, tot_unc = 0, tot_ill = 0;
                for (int64_t off = 0; off < nsig; off += fit) {
                    const int64_t n = std::min<int64_t>(fit, nsig - off);
                    CHECK(omp_batch_mfma_chunk(ctx, (const char*)B + (size_t)off * (size_t)ldB * es, b_dtype, ldB, n, b_loc, k, eps, idx + off * k,
                                               val + off * k, nnz + off, out_loc));
                    tot_res += b.last_resolved;
                    tot_unc += b.last_uncertain;
                    tot_ill += b.last_illcond;
                }
                b.last_signals = nsig;
                b.last_resolved = tot_res;
                b.last_uncertain = tot_unc;
                b.last_illcond = tot_ill;
                return CSMP_OK;
            }
        }
    }
    return omp_batch_mfma_chunk(ctx, B, b_dtype, ldB, nsig, b_loc, k, eps, idx, val, nnz, out_loc);
}

static int omp_batch_mfma_chunk(csmp_ctx* ctx, const void* B, int b_dtype, int64_t ldB, int64_t nsig, int b_loc, int64_t k,
                                double eps, int64_t* idx, double* val, int64_t* nnz, int out_loc) {
    if (!ctx) return CSMP_EINVAL;
    if (!(eps >= 0.0)) return fail(ctx, CSMP_EINVAL, "eps has to be non-negative");
    if (!B || nsig < 1 || k < 1 || ldB < ctx->M) return fail(ctx, CSMP_EINVAL, "omp_batch_mfma: bad arguments");
    if (b_dtype != CSMP_F32 && b_dtype != CSMP_F64) return fail(ctx, CSMP_EINVAL, "b_dtype must be CSMP_F32 or CSMP_F64");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (nsig > (1 << 20)) return fail(ctx, CSMP_ERANGE, "omp_batch_mfma: too many signals in one call");
    HIPCHECK(hipSetDevice(ctx->dev));
    const int kc0 = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    if (ctx->Mv > kBatchMaxRows || b_append_lds_bytes(ctx->Mv, ctx->dtype == CSMP_F32 ? 4 : 2, kc0) > (size_t)160 * 1024 - 1024) {
        // The per-signal kernels of this path keep a signal's column slice in registers, its residual (8 M bytes) and three
        // support-length vectors in LDS: beyond 8192 rows, or with a support capacity those vectors do not fit (k ~ 5000 at
        // M = 4096), the contract -- csmp_omp_batch's results -- is met by csmp_omp_batch itself (exact sweeps, three signals
        // in flight).  The statistics say so: no screening kernel, nothing re-solved.
        Batch& b0 = ctx->bt;
        b0.last_mode = 0;
        b0.last_streams = 1;
        b0.last_screen_signals = 0;
        b0.last_signals = nsig;
        b0.last_resolved = b0.last_uncertain = b0.last_illcond = 0;
        return csmp_omp_batch(ctx, B, b_dtype, ldB, nsig, b_loc, k, eps, idx, val, nnz, out_loc);
    }
    const int kc = (int)std::max<int64_t>(1, std::min<int64_t>(k, ctx->M));
    // Operands of the screen (CSMP_OPT_BATCH_SCREEN).  3 (default): binary16 images -- eleven significand bits: the rigorous bound is
    // 2^-10 |a||r| where bf16's is 2^-7.  0: bf16 images (the form of rounds 1-3).  1 / 2: int8 images (2: only where the dictionary is
    // FLAT -- max|A| within 8 root mean squares of its entries: Gaussian unit-norm columns 4.5-5, partial DCT 1.4, few-valued 1.3;
    // spikes beside a dense basis, max / rms = sqrt(M), would coarsen the common step for everything else); the int8 screen has a
    // statistical certificate only, so under the rigorous certificate (the default) an int8 request runs binary16.
    CHECK(batch_meta(ctx));
    const bool flat = ctx->bt.amax_host <= 8.0f * ctx->bt.arms_host;
    const bool i8 = (ctx->opt_batch_screen == 1 || (ctx->opt_batch_screen == 2 && flat)) && ctx->opt_batch_cert == 0;
    const int img = i8 ? kOpI8 : ctx->opt_batch_screen == 0 ? kOpBf16 : kOpF16;
    CHECK(img == kOpI8 ? batch_dict8(ctx) : img == kOpF16 ? batch_dict16(ctx) : batch_dict(ctx));
    CHECK(batch_ensure(ctx, (int)nsig, kc));
    CHECK(solver_ensure(ctx, kc, (int)k));  // the exact path re-solves flagged signals
    ctx->s.begun = false;
    Batch& b = ctx->bt;
    const bool gram = ctx->opt_batch_gram != 0;
    if (gram) CHECK(batch_gram(ctx));
    const size_t es = b_dtype == CSMP_F32 ? 4 : 8;
    void* dB = const_cast<void*>(B);
    DevTmp tB, tIdx, tVal, tNnz;  // freed on every return path
    if (b_loc == CSMP_HOST) {
        HIPCHECK(tB.alloc((size_t)ldB * (size_t)nsig * es));
        dB = tB.p;
        HIPCHECK(hipMemcpy(dB, B, (size_t)ldB * (size_t)nsig * es, hipMemcpyHostToDevice));
    }
    int64_t *d_idx = idx, *d_nnz = nnz;
    double* d_val = val;
    if (out_loc == CSMP_HOST) {
        HIPCHECK(tIdx.alloc((size_t)k * nsig * 8));
        HIPCHECK(tVal.alloc((size_t)k * nsig * 8));
        HIPCHECK(tNnz.alloc((size_t)nsig * 8));
        d_idx = (int64_t*)tIdx.p;
        d_val = (double*)tVal.p;
        d_nnz = (int64_t*)tNnz.p;
    }
    const int Bpad = (int)(((nsig + 2 * kBT - 1) / (2 * kBT)) * (2 * kBT));  // whole 256-signal tiles
    if (b_dtype == CSMP_F32)
        hipLaunchKernelGGL(k_b_init<float>, dim3(Bpad), dim3(256), 0, ctx->stream, (const float*)dB, ldB, (int)ctx->M, (int)nsig, b.r, b.b, b.Mr, b.Rb, b.Mk, b.bs,
                           i8 ? b.R8 : (signed char*)nullptr, b.Mk8, b.sigscale, img == kOpF16 ? 1.0f / b.ascale16 : b.astep, img);
    else
        hipLaunchKernelGGL(k_b_init<double>, dim3(Bpad), dim3(256), 0, ctx->stream, (const double*)dB, ldB, (int)ctx->M, (int)nsig, b.r, b.b, b.Mr, b.Rb, b.Mk, b.bs,
                           i8 ? b.R8 : (signed char*)nullptr, b.Mk8, b.sigscale, img == kOpF16 ? 1.0f / b.ascale16 : b.astep, img);
    HIPCHECK(hipGetLastError());
    // Screening error bound  | |<a_n, r>| - s_n | <= cert_abs |r| + cert_rel s_n  (k_b_pick, csmp_batched.hpp).
    //
    // RIGOROUS (CSMP_OPT_BATCH_CERT = 1, the default; binary16 or bf16 operands).  The images are round-to-nearest: a_i (1 + d_i),
    // r_i (1 + e_i) with |d_i|, |e_i| <= u (u = 2^-11 binary16, 2^-8 bf16; + 2^-23 for the Float64 -> Float32 -> image double rounding);
    // the products of two images are exact in Float32's accumulator width, and the accumulation of the Mk products is charged
    // 2^-23 per addition -- unit roundoff of Float32 under ANY rounding mode (the matrix cores' internal order and rounding are
    // not documented; truncation is covered) -- over at most Mk additions:
    //     |<a,r> - screened| <= (2u + u^2 + Mk 2^-23 (1 + u)^2) sum|a_i r_i| <= (...) |a|_2 |r|_2,
    // with the largest column norm for |a|_2.  binary16 only: the power-of-two scales are exact; image entries below the normal
    // range (2^-14 in scaled units, the largest entry of an operand sitting in [2^14, 2^15)) are charged a FLUSH TO ZERO -- an
    // absolute error of up to 2^-14 against a partner of at most 2^15, Mk of them, both operands: Mk 2^-26 |a||r| -- whether or
    // not the conversion keeps subnormals (it does by default; the bound does not rest on the mode register).  cert_rel: the 2^-15 the packed candidate keys drop and the rounding of the scale multiplication.
    // A proof: nothing about the data is assumed (tests: test_batched_certificate_against_adversarial_residuals).
    //
    // STATISTICAL (CSMP_OPT_BATCH_CERT = 0, opt-in): 8 standard deviations of a model of INDEPENDENT roundings of the M products,
    // sigma = sqrt(2/3) u/2 max|A_ij| |r|, plus a coherent term (an operand whose entries all round the same way is a scaled
    // operand: 2u of the screened value).  Narrower windows; holds for generic data; a residual aligned with the rounding errors
    // of a near-tied atom defeats it (the same test shows that).
    // int8 screen (statistical only): both operands are rounded to a uniform grid -- the dictionary to multiples of astep =
    // max|A| / 127, every residual to multiples of its own rstep = max|r_i| / 127 -- and the integer accumulation is exact: the
    // error of a screened value is sum(da_i r_i) + sum(a_i dr_i) (+ the product of the two), with da_i, dr_i uniform in
    // +-step/2: sigma^2 = astep^2 |r|^2 / 12 + rstep^2 |a|^2 / 12.  8 sigma, the largest column norm; coherent term 2^-6.
    double cert_abs, cert_rel, cert_abs2 = 0.0;
    int kwin;
    const double u_img = (img == kOpF16 ? std::ldexp(1.0, -11) : std::ldexp(1.0, -8)) + std::ldexp(1.0, -23);
    if (i8) {
        CHECK(batch_colnorm(ctx));
        cert_abs = 8.0 * (double)b.astep / std::sqrt(12.0);
        cert_abs2 = 8.0 * (double)b.anorm_host / std::sqrt(12.0);
        cert_rel = std::ldexp(1.0, -6) + std::ldexp(1.0, -14);
        kwin = kWinMax;
    } else if (ctx->opt_batch_cert == 1) {
        CHECK(batch_colnorm(ctx));
        cert_abs = (2.0 * u_img + u_img * u_img + (double)b.Mk * std::ldexp(1.0, -23) * (1.0 + u_img) * (1.0 + u_img) +
                    (img == kOpF16 ? (double)b.Mk * std::ldexp(1.0, -26) : 0.0)) * (double)b.anorm_host;
        cert_rel = std::ldexp(1.0, -14);
        kwin = kWinMax;  // 128
    } else {
        cert_abs = 8.0 * std::sqrt(2.0 / 3.0) * u_img * 0.5 * (double)b.amax_host;
        cert_rel = 2.0 * u_img * 1.01 + std::ldexp(1.0, -14);
        kwin = kWinMax / 2;  // 64
    }
    if (ctx->opt_batch_window > 0) kwin = std::min<int>(kWinMax, (int)ctx->opt_batch_window);
    const int mode = i8 ? kScreen256i8 : img == kOpF16 ? kScreen256f16 : kScreen256p;
    b.last_mode = mode;
    b.last_streams = 1;
    b.last_screen_signals = Bpad;
    // (Two overlap schemes for the MFMA-bound screen and the HBM-bound per-signal kernels were measured in round 3 and lost:
    // profiles/r03_cusplit_experiment.txt, profiles/r03_coresident_experiment.txt.  One stream, phases back to back.)
    const __bf16* dimg = img == kOpI8 ? (const __bf16*)b.A8 : img == kOpF16 ? (const __bf16*)b.Ah : (const __bf16*)b.Ab;
    const __bf16* rimg = img == kOpI8 ? (const __bf16*)b.R8 : (const __bf16*)b.Rb;
    const int mk2 = img == kOpI8 ? b.Mk8 / 2 : b.Mk;  // row length in 2-byte slots
    for (int64_t t = 0; t < k; ++t) {
        if (ctx->prof) {  // HIP events around the screening launch
            if (ctx->ev2_used == ctx->ev2.size()) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); ctx->ev2.push_back(e); }
            HIPCHECK(hipEventRecord(ctx->ev2[ctx->ev2_used++], ctx->stream));
        }
        HIPCHECK(launch_screen(ctx->stream, mode, dimg, rimg, mk2, b.n_atiles, Bpad / kBT, ctx->N, b.cand_val, b.cand_idx, img == kOpBf16 ? nullptr : b.sigscale));
        if (ctx->prof) {
            if (ctx->ev2_used == ctx->ev2.size()) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); ctx->ev2.push_back(e); }
            HIPCHECK(hipEventRecord(ctx->ev2[ctx->ev2_used++], ctx->stream));
        }
        hipError_t e = ctx->dtype == CSMP_F32 ? b_step_dispatch<float>(ctx, ctx->stream, 0, (int)nsig, eps, t > 0, cert_abs, cert_rel, kwin, gram, img, cert_abs2)
                                              : b_step_dispatch<double>(ctx, ctx->stream, 0, (int)nsig, eps, t > 0, cert_abs, cert_rel, kwin, gram, img, cert_abs2);
        HIPCHECK(e);
    }
    hipLaunchKernelGGL(k_b_finish, dim3((int)nsig), dim3(256), (size_t)(b.kcap + 2) * 8, ctx->stream, (const double*)b.T,
                       (const double*)b.z, (const int*)b.sel, (const BState*)b.bs, b.kcap, (int)k, d_idx, d_val, d_nnz);
    HIPCHECK(hipGetLastError());
    // signals whose screen could not be certified (or whose support turned ill-conditioned) are
    // re-solved by the exact single-signal path
    std::vector<BState> hs((size_t)nsig);
    HIPCHECK(hipMemcpyAsync(hs.data(), b.bs, (size_t)nsig * sizeof(BState), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    b.last_signals = nsig;
    b.last_resolved = b.last_uncertain = b.last_illcond = 0;
    int rc = CSMP_OK;
    for (int64_t sgn = 0; sgn < nsig && rc == CSMP_OK; ++sgn) {
        if (!hs[sgn].uncertain && !hs[sgn].illcond) continue;
        b.last_resolved += 1;
        b.last_uncertain += hs[sgn].uncertain ? 1 : 0;
        b.last_illcond += hs[sgn].illcond ? 1 : 0;
        const char* col = (const char*)dB + (size_t)sgn * (size_t)ldB * es;
        rc = b_dtype == CSMP_F32 ? init_from_device_t<float>(ctx, (const float*)col)
                                 : init_from_device_t<double>(ctx, (const double*)col);
        for (int64_t t = 0; t < k && rc == CSMP_OK; ++t) rc = omp_step(ctx, eps, t > 0, false);
        if (rc == CSMP_OK) rc = launch_finish(ctx, d_idx + sgn * k, d_val + sgn * k, d_nnz + sgn, nullptr, (int)k);
    }
    if (out_loc == CSMP_HOST) {
        if (rc == CSMP_OK) {
            HIPCHECK(hipMemcpyAsync(idx, d_idx, (size_t)k * nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(val, d_val, (size_t)k * nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipMemcpyAsync(nnz, d_nnz, (size_t)nsig * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHECK(hipStreamSynchronize(ctx->stream));
    }
    return rc;
}

// name of the screening kernel the last csmp_omp_batch_mfma call used (for the bench's roofline line)
extern "C" const char* csmp_batch_screen_kernel(const csmp_ctx* ctx) { return ctx ? screen_kernel_name(ctx->bt.last_mode) : ""; }

// how the last csmp_omp_batch_mfma call was laid out: signal columns per screening launch, streams used
extern "C" int csmp_batch_layout(const csmp_ctx* ctx, int64_t* screen_signals, int* streams) {
    if (!ctx) return CSMP_EINVAL;
    if (screen_signals) *screen_signals = ctx->bt.last_screen_signals;
    if (streams) *streams = ctx->bt.last_streams;
    return CSMP_OK;
}

extern "C" int csmp_batch_stats(csmp_ctx* ctx, int64_t* signals, int64_t* resolved_exactly, int64_t* uncertain, int64_t* illcond,
                                int64_t* screen_launches, double* screen_ms) {
    if (!ctx) return CSMP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->dev));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 1 < ctx->ev2_used; i += 2) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev2[i], ctx->ev2[i + 1]));
        ctx->prof2_ms += ms;
        ctx->prof2_n += 1;
    }
    ctx->ev2_used = 0;
    if (signals) *signals = ctx->bt.last_signals;
    if (resolved_exactly) *resolved_exactly = ctx->bt.last_resolved;
    if (uncertain) *uncertain = ctx->bt.last_uncertain;
    if (illcond) *illcond = ctx->bt.last_illcond;
    if (screen_launches) *screen_launches = ctx->prof2_n;
    if (screen_ms) *screen_ms = ctx->prof2_ms;
    ctx->prof2_n = 0;
    ctx->prof2_ms = 0.0;
    return CSMP_OK;
}
