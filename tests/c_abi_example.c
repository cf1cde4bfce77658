/* The boundary from plain C (C99, nothing but include/csmp.h and libcsmp.so): create a context, hand over a small dictionary, run
 * omp / gomp / the batch form, write and read a dictionary file.  tests/test_abi.py compiles and links it on every run (the header
 * is C, every symbol used resolves); tests/test_gpu_parity.py runs it on the GPU box.  Exit code 0 = every check held. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csmp.h"

#define M 64
#define N 256

static int fail_at(const char *what, csmp_ctx *ctx, int rc) {
    fprintf(stderr, "c_abi_example: %s failed (%d): %s\n", what, rc, csmp_last_error(ctx));
    return 1;
}

int main(int argc, char **argv) {
    static float A[M * N];
    double b[M], B[2 * M];
    int64_t idx[8], nnz = 0, order[8], bidx[2 * 4], bnnz[2];
    double val[8], bval[2 * 4];
    unsigned s = 12345u;
    int i, j, rc;
    /* a deterministic dictionary with unit-norm columns (a plain LCG: no libc rand differences) */
    for (j = 0; j < N; ++j) {
        double n2 = 0.0;
        for (i = 0; i < M; ++i) {
            s = s * 1664525u + 1013904223u;
            A[j * M + i] = (float)((double)(s >> 8) / 8388608.0 - 1.0);
            n2 += (double)A[j * M + i] * A[j * M + i];
        }
        for (i = 0; i < M; ++i) A[j * M + i] = (float)(A[j * M + i] / sqrt(n2));
    }
    for (i = 0; i < M; ++i) {
        b[i] = 2.0 * A[17 * M + i] - 1.5 * A[101 * M + i] + 1.0 * A[200 * M + i];
        B[i] = b[i];
        B[M + i] = 3.0 * A[5 * M + i] + 2.0 * A[250 * M + i];
    }
    csmp_ctx *ctx = NULL;
    rc = csmp_create(&ctx, 0);
    if (rc != CSMP_OK) return fail_at("csmp_create", NULL, rc);
    rc = csmp_set_dictionary(ctx, A, M, N, M, CSMP_F32, CSMP_HOST);
    if (rc != CSMP_OK) return fail_at("csmp_set_dictionary", ctx, rc);
    rc = csmp_omp(ctx, b, CSMP_F64, 3, 1e-6, idx, val, &nnz, order);
    if (rc != CSMP_OK) return fail_at("csmp_omp", ctx, rc);
    if (nnz != 3 || idx[0] != 17 || idx[1] != 101 || idx[2] != 200 || fabs(val[0] - 2.0) > 1e-5 || fabs(val[1] + 1.5) > 1e-5 ||
        fabs(val[2] - 1.0) > 1e-5 || order[0] != 17) {
        fprintf(stderr, "c_abi_example: omp returned nnz=%lld idx=%lld,%lld,%lld val=%g,%g,%g\n", (long long)nnz, (long long)idx[0],
                (long long)idx[1], (long long)idx[2], val[0], val[1], val[2]);
        return 1;
    }
    rc = csmp_gomp(ctx, b, CSMP_F64, 2, 4, 1e-6, idx, val, &nnz, NULL);
    if (rc != CSMP_OK) return fail_at("csmp_gomp", ctx, rc);
    if (nnz < 3) return fail_at("csmp_gomp (support)", ctx, (int)nnz);
    rc = csmp_omp_batch(ctx, B, CSMP_F64, M, 2, CSMP_HOST, 4, 1e-6, bidx, bval, bnnz, CSMP_HOST);
    if (rc != CSMP_OK) return fail_at("csmp_omp_batch", ctx, rc);
    if (bnnz[0] != 3 || bnnz[1] != 2 || bidx[0] != 17 || bidx[4] != 5 || bidx[5] != 250) return fail_at("csmp_omp_batch (supports)", ctx, 0);
    if (csmp_omp(ctx, b, CSMP_F64, 3, -1.0, idx, val, &nnz, NULL) != CSMP_EINVAL) return fail_at("eps < 0 must be refused", ctx, 0);
    if (argc > 1) { /* a dictionary file: write, inspect, load into HBM, solve again */
        int64_t fm = 0, fn = 0;
        int fd = -1;
        rc = csmp_dictionary_file_write(argv[1], A, M, N, M, CSMP_F32);
        if (rc != CSMP_OK) return fail_at("csmp_dictionary_file_write", ctx, rc);
        rc = csmp_dictionary_file_info(argv[1], &fm, &fn, &fd);
        if (rc != CSMP_OK || fm != M || fn != N || fd != CSMP_F32) return fail_at("csmp_dictionary_file_info", ctx, rc);
        rc = csmp_set_dictionary_file(ctx, argv[1], CSMP_DEVICE);
        if (rc != CSMP_OK) return fail_at("csmp_set_dictionary_file", ctx, rc);
        rc = csmp_omp(ctx, b, CSMP_F64, 3, 1e-6, idx, val, &nnz, NULL);
        if (rc != CSMP_OK || nnz != 3 || idx[0] != 17 || idx[2] != 200) return fail_at("csmp_omp on the file's dictionary", ctx, rc);
    }
    csmp_destroy(ctx);
    printf("c_abi_example: ok\n");
    return 0;
}
