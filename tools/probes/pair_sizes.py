"""Round 6 probe: from which dictionary size on do two pipelines side by side pay?  csmp_omp_batch on 12 signals, k = 64, one
pipeline (csmp_tune pipelines = 1) against two (= 2), dictionaries of 1 MiB ... 1 GiB.  Usage: python tools/probes/pair_sizes.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load  # noqa: E402
from tools.probes.dyn_probe import dictionary, dev  # noqa: E402

cs = load()
k, nsig = 64, 12
for M, N in ((256, 1024), (512, 4096), (1024, 8192), (2048, 8192), (4096, 4096), (4096, 8192), (4096, 16384), (4096, 32768), (4096, 65536)):
    At = dictionary(M, N, torch.float32, seed=4)
    D = cs.Dictionary(At, device=0)
    g = torch.Generator(device=dev).manual_seed(5)
    B = torch.randn((nsig, M), generator=g, device=dev, dtype=torch.float64)
    idx = torch.full((nsig, k), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((nsig, k), dtype=torch.float64, device=dev)
    nnz = torch.zeros(nsig, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    out = {}
    for p in (1, 2):
        D.ctx.tune("pipelines", p)
        D.ctx.omp_batch_device(B, k, 1e-9, idx, val, nnz)
        D.ctx.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            D.ctx.omp_batch_device(B, k, 1e-9, idx, val, nnz)
            D.ctx.sync()
            best = min(best, time.perf_counter() - t0)
        out[p] = int(nnz.sum()) / best
    print(f"{M:5d} x {N:6d} f32 ({M * N * 4 / 2**20:7.1f} MiB): one pipeline {out[1]:9.1f} atoms/s, two {out[2]:9.1f}  ({out[2] / out[1] - 1:+.1%})", flush=True)
    D.close()
    del At
