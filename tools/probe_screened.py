"""Screened single-signal sweep (CSMP_OPT_SCREENED_SWEEP) against the exact path at BASELINE configs[1]: lone csmp_omp calls and
the batch form, time per atom, fallbacks.    python tools/probe_screened.py [nsig]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from csmp_pkg import load

cs = load()
dev = torch.device("cuda", 0)
EPS32 = float(np.finfo(np.float32).eps)
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 8
At = bench.make_dictionary(torch, dev)
B = bench.make_signals(torch, dev, At, 500, nsig + 2)
D = cs.Dictionary(At, device=0)
K = 256
idx = torch.full((nsig, K), -1, dtype=torch.int64, device=dev)
val = torch.zeros((nsig, K), dtype=torch.float64, device=dev)
nnz = torch.zeros(nsig, dtype=torch.int64, device=dev)
res = {}
for scr in (0, 1, 2):
    D.ctx.set_option("screened_sweep", scr)
    D.ctx.screened_stats(reset=True)
    sigs = [B[s].cpu().numpy() for s in range(nsig + 2)]
    for w in range(2):
        D.ctx.omp(sigs[w], K, EPS32)
    t0 = time.perf_counter()
    atoms = 0
    outs = []
    for s in range(2, nsig + 2):
        o = D.ctx.omp(sigs[s], K, EPS32)
        outs.append(o)
        atoms += len(o[0])
    dt = time.perf_counter() - t0
    res[scr] = outs
    line = {"screened": scr, "form": "lone csmp_omp", "us_per_atom": dt / atoms * 1e6, "atoms_per_s": atoms / dt, "stats": D.ctx.screened_stats()}
    print(json.dumps(line), flush=True)
    D.ctx.omp_batch_device(B[2:], K, EPS32, idx, val, nnz)
    D.ctx.sync()
    t0 = time.perf_counter()
    for rep in range(2):
        D.ctx.omp_batch_device(B[2:], K, EPS32, idx, val, nnz)
    D.ctx.sync()
    dt = time.perf_counter() - t0
    atoms = int(nnz.sum()) * 2
    ok = all(np.array_equal(np.sort(idx[s, :int(nnz[s])].cpu().numpy()), np.sort(res[scr][s][0])) for s in range(nsig))
    print(json.dumps({"screened": scr, "form": "csmp_omp_batch", "us_per_atom": dt / atoms * 1e6, "atoms_per_s": atoms / dt,
                      "equals_lone": bool(ok), "stats": D.ctx.screened_stats()}), flush=True)
same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.allclose(a[1], b[1], rtol=1e-9, atol=1e-12)
           for m in (1, 2) for a, b in zip(res[0], res[m]))
print(json.dumps({"screened_equals_exact": bool(same)}))
D.close()
