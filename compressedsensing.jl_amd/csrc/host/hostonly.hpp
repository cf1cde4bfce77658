// host/hostonly.hpp -- the exports of include/csmp.h that need neither a context nor the GPU: dictionary files (writer, header
// reader) and the wire layout of the signal-sharded gather.  Plain C++ over <cstdio>: included by csmp.hip (the library) and, alone,
// by tools/sanitize/hostonly_driver.cpp, which tools/sanitize_cpu.sh builds with gcc's address and undefined-behaviour sanitizers.
#pragma once
#include "../../../include/csmp.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>

struct DictFileHeader {
    char magic[8];
    uint32_t version, dtype;
    int64_t M, N, ld;
    char pad[24];
};
static_assert(sizeof(DictFileHeader) == 64, "header layout");

extern "C" int csmp_dictionary_file_write(const char* path, const void* A, int64_t M, int64_t N, int64_t ldA, int dtype) {
    if (!path || !A || M < 1 || N < 1 || ldA < M || (dtype != CSMP_F32 && dtype != CSMP_F64)) return CSMP_EINVAL;
    const size_t es = dtype == CSMP_F32 ? 4 : 8;
    const int64_t vec = 16 / (int64_t)es, ld = ((M + vec - 1) / vec) * vec;
    FILE* f = fopen(path, "wb");
    if (!f) return CSMP_EIO;
    DictFileHeader h{};
    memcpy(h.magic, "CSMPDICT", 8);
    h.version = 1;
    h.dtype = (uint32_t)dtype;
    h.M = M;
    h.N = N;
    h.ld = ld;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    const char zeros[16] = {0};
    for (int64_t c = 0; c < N && ok; ++c) {
        ok = fwrite((const char*)A + (size_t)c * (size_t)ldA * es, es, (size_t)M, f) == (size_t)M;
        if (ok && ld > M) ok = fwrite(zeros, es, (size_t)(ld - M), f) == (size_t)(ld - M);
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? CSMP_OK : CSMP_EIO;
}

static int dict_file_open(const char* path, FILE** out, DictFileHeader* h) {
    FILE* f = fopen(path, "rb");
    if (!f) return CSMP_EIO;
    if (fread(h, sizeof *h, 1, f) != 1 || memcmp(h->magic, "CSMPDICT", 8) != 0 || h->version != 1 ||
        (h->dtype != (uint32_t)CSMP_F32 && h->dtype != (uint32_t)CSMP_F64) || h->M < 1 || h->N < 1 || h->ld < h->M ||
        h->ld != ((h->M + (h->dtype == (uint32_t)CSMP_F32 ? 3 : 1)) / (h->dtype == (uint32_t)CSMP_F32 ? 4 : 2)) * (h->dtype == (uint32_t)CSMP_F32 ? 4 : 2)) {
        // (ld is M rounded up to 16 bytes, nothing else: the kernels read the padding rows as part of the columns)
        fclose(f);
        return CSMP_EIO;
    }
    *out = f;
    return CSMP_OK;
}

extern "C" int csmp_dictionary_file_info(const char* path, int64_t* M, int64_t* N, int* dtype) {
    if (!path) return CSMP_EINVAL;
    FILE* f = nullptr;
    DictFileHeader h;
    const int rc = dict_file_open(path, &f, &h);
    if (rc != CSMP_OK) return rc;
    fclose(f);
    if (M) *M = h.M;
    if (N) *N = h.N;
    if (dtype) *dtype = (int)h.dtype;
    return CSMP_OK;
}


extern "C" int csmp_shard_range(int64_t nsig, int rank, int world, int64_t* lo, int64_t* hi) {
    if (nsig < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return CSMP_EINVAL;
    const int64_t base = nsig / world, extra = nsig % world;  // block sizes differ by at most one
    *lo = rank * base + std::min<int64_t>(rank, extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
    return CSMP_OK;
}
extern "C" int csmp_pack_results(const int64_t* idx, const double* val, const int64_t* nnz, int64_t k, int64_t nsig, double* packed) {
    if (!idx || !val || !nnz || !packed || k < 0 || nsig < 0) return CSMP_EINVAL;
    const int64_t w = 2 * k + 1;
    for (int64_t s = 0; s < nsig; ++s) {
        for (int64_t t = 0; t < k; ++t) {
            packed[s * w + t] = (double)idx[s * k + t];
            packed[s * w + k + t] = val[s * k + t];
        }
        packed[s * w + 2 * k] = (double)nnz[s];
    }
    return CSMP_OK;
}
extern "C" int csmp_unpack_results(const double* packed, int64_t k, int64_t nsig, int64_t* idx, double* val, int64_t* nnz) {
    if (!idx || !val || !nnz || !packed || k < 0 || nsig < 0) return CSMP_EINVAL;
    const int64_t w = 2 * k + 1;
    for (int64_t s = 0; s < nsig; ++s) {
        for (int64_t t = 0; t < k; ++t) {
            idx[s * k + t] = (int64_t)packed[s * w + t];
            val[s * k + t] = packed[s * w + k + t];
        }
        nnz[s] = (int64_t)packed[s * w + 2 * k];
    }
    return CSMP_OK;
}
