"""Round 6 probe: the dynamic sweep with a STATIC head -- a workgroup's first columns are the static split's, only the last 1 / n of
its pool is claimed at run time and can be stolen (csmp_tune sweep_dyn = n >= 2; 1 = every column claimed; 0 = the static split).
(1) bit-identity against the static split; (2) the stand-alone 1-GiB sweep; (3) one signal's k = 256 solve (csmp_omp).
Usage: python tools/probes/hybrid_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load  # noqa: E402
from tools.probes.dyn_probe import dictionary, dev  # noqa: E402

cs = load()
MODES = (0, 1, 2, 4, 8, 16)
ok = True
for (M, N, dt) in [(4096, 8192, torch.float32), (1000, 4099, torch.float32), (3000, 777, torch.float64), (4352, 8191, torch.float32), (64, 3, torch.float32),
                   (4096, 130, torch.float32), (12288, 2049, torch.float32)]:
    At = dictionary(M, N, dt)
    D = cs.Dictionary(At, device=0)
    rng = np.random.default_rng(5)
    B = np.asfortranarray(rng.standard_normal((M, 4)))
    k = min(12, M // 2, N)
    res = {}
    for mode in MODES:
        D.ctx.tune("sweep_dyn", mode)
        res[mode] = (D.ctx.sweep(B[:, 0], topk=2), D.ctx.omp(B[:, 1], k, 1e-12), D.ctx.omp_batch(B, k, 1e-12))
    same = all(np.array_equal(np.asarray(u), np.asarray(v)) for m in MODES[1:] for a, b in zip(res[0], res[m]) for u, v in zip(a, b))
    print(f"identity {M}x{N} {str(dt)[6:]}: {'same' if same else 'DIFFERENT'}", flush=True)
    ok = ok and same
    D.close()
    del At
At = dictionary(4096, 65536, torch.float32, seed=2)
D = cs.Dictionary(At, device=0)
for rnd in range(2):
    for mode in MODES:
        D.ctx.tune("sweep_dyn", mode)
        for grid in (0, 176, 208):
            D.ctx.tune("sweep_grid", grid)
            ms = sorted(D.ctx.bench_sweep(reps=40) for _ in range(5))[2]
            print(f"sweep_dyn {mode:2d} grid {grid or 'auto':>4}: {ms * 1e3:7.1f} us  {At.numel() * 4 / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
D.ctx.tune("sweep_grid", 0)
g = np.random.default_rng(3)
y = g.standard_normal(4096)
for mode in MODES:
    D.ctx.tune("sweep_dyn", mode)
    D.ctx.omp(y, 8, 1e-7)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = D.ctx.omp(y, 256, 1e-7)
        best = min(best, time.perf_counter() - t0)
    print(f"sweep_dyn {mode:2d}: one signal, k = 256: {len(out[0]) / best:8.1f} atoms/s", flush=True)
D.close()
sys.exit(0 if ok else 1)
