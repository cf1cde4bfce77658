"""Randomised GPU-vs-oracle comparison of the core path (omp, gomp, sp, the batch drivers incl. the MFMA-screened one under both
certificates and with the resident Gram matrix, the screened single-signal sweep under both certificates, the in-flight batch forms of gomp / sp, lstsq), on Gaussian and -- every third
round -- structured dictionaries.    python tools/stress_core.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402

cs = load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
runs = bad = 0
rounds = 0
tally = {}


def cmp(name, got, ref, cfg):
    global runs, bad
    runs += 1
    tally[name] = tally.get(name, 0) + 1
    ok = np.array_equal(got[0], ref[0])
    if ok and len(ref[1]) and np.all(np.isfinite(ref[1])):
        ok = np.allclose(got[1], ref[1], rtol=1e-6, atol=1e-6 * float(np.max(np.abs(ref[1]))))
    if not ok:
        bad += 1
        print("MISMATCH", name, cfg, got[0][:10], ref[0][:10], flush=True)


while time.time() - t0 < budget:
    M = int(rng.choice([37, 64, 130, 256, 512, 1024, 2048, 4096, 1500]))
    N = int(rng.choice([300, 1000, 3000, 8000]))
    dtype = rng.choice([np.float32, np.float64])
    k = int(rng.integers(2, max(3, min(M // 6, 48))))
    if M >= 1024 and rng.random() < 0.3:  # large supports: whole-set least squares with the augmented factor and its extension (sp)
        k = int(rng.integers(64, min(M // 4, 260)))
    kind = "gaussian"
    if rounds % 3 == 2:
        kind = str(rng.choice(["few_valued", "partial_dct", "one_magnitude", "signs"]))
        A = cs.structured_dictionary(kind, M, max(N, M), rng=rng, dtype=dtype)
        N = A.shape[1]
    else:
        A = rng.standard_normal((M, N))
        A /= np.linalg.norm(A, axis=0)
        A = np.asfortranarray(A.astype(dtype))
    rounds += 1
    nsig = int(rng.choice([1, 2, 3, 5, 7]))
    B = []
    for _ in range(nsig):
        supp = rng.choice(N, k, replace=False)
        b = A[:, supp].astype(np.float64) @ rng.choice([-1.0, 1.0], k)
        e = rng.standard_normal(M)
        B.append(b + 5e-3 * e / np.linalg.norm(e))
    B = np.asfortranarray(np.stack(B, axis=1))
    cfg = (kind, M, N, k, str(np.dtype(dtype)), nsig)
    eps = float(np.finfo(dtype).eps)
    D = cs.Dictionary(A)
    try:
        refs = [oc.omp(A, B[:, s], k, eps) for s in range(nsig)]
        cmp("omp", D.ctx.omp(B[:, 0], k, eps), refs[0], cfg)
        idx, val, nnz = D.ctx.omp_batch(B, k, eps)
        for s in range(nsig):
            cmp("omp_batch", (idx[:nnz[s], s], val[:nnz[s], s]), refs[s], cfg)
        for cert, gram, scr, name in ((1, 0, 3, "omp_mfma_default_f16_rigorous"), (1, 1, 3, "omp_mfma_f16_rigorous_gram"), (1, 0, 0, "omp_mfma_bf16_rigorous"),
                                      (0, 0, 3, "omp_mfma_f16_statistical"), (0, 0, 0, "omp_mfma_bf16_statistical"), (0, 0, 1, "omp_mfma_int8"),
                                      (0, 1, 1, "omp_mfma_int8_gram")):
            if gram and N > 8000:
                continue
            D.ctx.set_option("batch_cert", cert)
            D.ctx.set_option("batch_gram", gram)
            D.ctx.set_option("batch_screen", scr)
            idx, val, nnz = D.ctx.omp_batch_mfma(B, k, eps)
            for s in range(nsig):
                cmp(name, (idx[:nnz[s], s], val[:nnz[s], s]), refs[s], cfg)
        D.ctx.set_option("batch_cert", 1)
        D.ctx.set_option("batch_gram", 0)
        D.ctx.set_option("batch_screen", 3)
        for cert, img, name in ((1, 3, "omp_screened_f16_rigorous"), (0, 3, "omp_screened_f16_statistical"), (0, 1, "omp_screened_bf16"),
                                (1, 1, "omp_screened_bf16_rigorous"), (0, 2, "omp_screened_int8")):  # CSMP_OPT_SCREENED_SWEEP: lone calls and the batch form
            D.ctx.set_option("batch_cert", cert)
            D.ctx.set_option("screened_sweep", img)
            cmp(name, D.ctx.omp(B[:, 0], k, eps), refs[0], cfg)
            idx, val, nnz = D.ctx.omp_batch(B, k, eps)
            for s in range(nsig):
                cmp(name + "_batch", (idx[:nnz[s], s], val[:nnz[s], s]), refs[s], cfg)
        D.ctx.set_option("screened_sweep", 0)
        D.ctx.set_option("batch_cert", 1)
        l = int(rng.choice([2, 3, 4]))
        gref = [oc.gomp(A, B[:, s], l, k, eps) for s in range(nsig)]
        cmp("gomp", D.ctx.gomp(B[:, 0], l, k, eps), gref[0], cfg + (l,))
        if l <= k:
            idx, val, nnz = D.ctx.gomp_batch(B, l, k, eps)
            for s in range(nsig):
                cmp("gomp_batch", (idx[:nnz[s], s], val[:nnz[s], s]), gref[s], cfg + (l,))
            img = int(rng.integers(1, 4))
            D.ctx.set_option("screened_sweep", img)  # certified top-l picks over the bf16 / int8 / binary16 image
            D.ctx.set_option("batch_cert", int(rng.integers(0, 2)) if img != 2 else 0)
            cmp("gomp_screened", D.ctx.gomp(B[:, 0], l, k, eps), gref[0], cfg + (l,))
            idx, val, nnz = D.ctx.gomp_batch(B, l, k, eps)
            for s in range(nsig):
                cmp("gomp_screened_batch", (idx[:nnz[s], s], val[:nnz[s], s]), gref[s], cfg + (l,))
            D.ctx.set_option("screened_sweep", 0)
            D.ctx.set_option("batch_cert", 1)
        if 2 * k <= M:
            sref = [oc.sp(A, B[:, s], k, 1e-12) for s in range(nsig)]
            cmp("sp", D.ctx.sp(B[:, 0], k, 1e-12), sref[0], cfg)
            D.ctx.set_option("solves_in_flight", int(rng.integers(1, 5)))
            idx, val, nnz, its = D.ctx.sp_batch(B, k, 1e-12)
            for s in range(nsig):
                cmp("sp_batch", (idx[:nnz[s], s], val[:nnz[s], s]), sref[s], cfg)
            D.ctx.set_option("screened_sweep", int(rng.integers(1, 4)))  # certified top-k sets over the bf16 / int8 / binary16 image
            cmp("sp_screened", D.ctx.sp(B[:, 0], k, 1e-12), sref[0], cfg)
            idx, val, nnz, its = D.ctx.sp_batch(B, k, 1e-12)
            for s in range(nsig):
                cmp("sp_screened_batch", (idx[:nnz[s], s], val[:nnz[s], s]), sref[s], cfg)
            D.ctx.set_option("screened_sweep", 0)
            D.ctx.set_option("solves_in_flight", 3)
        if rounds % 4 == 1:  # the same dictionary left in host memory (CSMP_HOST_STREAMED) or read from a dictionary file
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "d.csmp")
                cs.write_dictionary_file(path, A)
                for name, Ds in (("streamed", cs.Dictionary(A, streamed=True)), ("file", cs.Dictionary(path)),
                                 ("file_streamed", cs.Dictionary(path, streamed=True))):
                    try:
                        cmp("omp_" + name, Ds.ctx.omp(B[:, 0], k, eps), refs[0], cfg)
                        cmp("gomp_" + name, Ds.ctx.gomp(B[:, 0], l, k, eps), gref[0], cfg + (l,))
                        idx, val, nnz = Ds.ctx.omp_batch_mfma(B, k, eps)
                        for s in range(nsig):
                            cmp("omp_mfma_" + name, (idx[:nnz[s], s], val[:nnz[s], s]), refs[s], cfg)
                        if 2 * k <= M:
                            cmp("sp_" + name, Ds.ctx.sp(B[:, 0], k, 1e-12), sref[0], cfg)
                    finally:
                        Ds.close()
        cols = np.sort(rng.choice(N, min(3 * k, M // 2, N), replace=False))
        got = D.ctx.lstsq(cols, B[:, 0])
        ref = oc.lstsq_cols(A, cols, B[:, 0])
        cmp("lstsq", (cols, got), (cols, ref), cfg)
    finally:
        D.close()
print(f"runs {runs}  mismatches {bad}  {tally}  {time.time() - t0:.0f} s")
