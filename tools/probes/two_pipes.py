"""Round 6 probe (verdict item 1b, cross-tick overlap): TWO pipelined batches (csmp_omp_batch, three signals in flight each) on two
contexts / two streams at once, driven from two host threads, each with a share of the sweep workgroups -- the tail and the start
of one pipeline's tick fall under the other pipeline's stream.  Aggregate atoms/s against ONE pipeline on the whole chip.
Usage: python tools/probes/two_pipes.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load  # noqa: E402
from tools.probes.dyn_probe import dictionary, dev  # noqa: E402

cs = load()
M, N, k = 4096, 65536, 256


def run(ctxs, B, idx, val, nnz, grids, offset_us=0):
    n = len(ctxs)
    per = B.shape[0] // n
    for c, g in zip(ctxs, grids):
        c.tune("tick_grid", g)
    for i, c in enumerate(ctxs):  # warm
        c.omp_batch_device(B[i * per:i * per + 3], 8, 1e-7, idx[i * per:i * per + 3, :8].contiguous(), val[i * per:i * per + 3, :8].contiguous(), nnz[i * per:i * per + 3])
        c.sync()
    best = 1e9
    for _ in range(2):
        def work(i):
            if i and offset_us:
                time.sleep(offset_us * 1e-6)
            ctxs[i].omp_batch_device(B[i * per:(i + 1) * per], k, 1e-7, idx[i * per:(i + 1) * per], val[i * per:(i + 1) * per], nnz[i * per:(i + 1) * per])
            ctxs[i].sync()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    return int(nnz[:per * n].sum()) / best


if __name__ == "__main__":
    At = dictionary(M, N, torch.float32, seed=2)
    D = cs.Dictionary(At, device=0)
    c2 = D.ctx.clone()
    c3 = D.ctx.clone()
    K = 18
    g = torch.Generator(device=dev).manual_seed(3)
    B = torch.randn((K, M), generator=g, device=dev, dtype=torch.float64)
    idx = torch.full((K, k), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((K, k), dtype=torch.float64, device=dev)
    nnz = torch.zeros(K, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    one = run([D.ctx], B, idx, val, nnz, [0])
    ref = (idx.cpu().numpy().copy(), val.cpu().numpy().copy())
    print(f"one pipeline, automatic grid: {one:8.1f} atoms/s", flush=True)
    for grids in ([144, 144], [160, 160], [176, 176], [192, 192], [224, 224], [256, 256]):
        for off in (0,):
            v = run([D.ctx, c2], B, idx, val, nnz, grids, off)
            same = np.array_equal(ref[0], idx.cpu().numpy()) and np.array_equal(ref[1], val.cpu().numpy())
            print(f"two pipelines, sweep workgroups {grids}, second started {off} us later: {v:8.1f} atoms/s  identical={same}", flush=True)
    for grids in ([96, 96, 96], [128, 128, 128], [176, 176, 176]):
        v = run([D.ctx, c2, c3], B, idx, val, nnz, grids)
        same = np.array_equal(ref[0], idx.cpu().numpy()) and np.array_equal(ref[1], val.cpu().numpy())
        print(f"three pipelines, sweep workgroups {grids}: {v:8.1f} atoms/s  identical={same}", flush=True)
    c3.close()
    c2.close()
    D.close()
