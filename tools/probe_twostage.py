"""Times fr / srr / ompr (and the kernels behind them, under rocprofv3) at the configs[1] shape.
    python tools/probe_twostage.py [k] [noise]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load  # noqa: E402

cs = load()
M, N = 4096, 65536
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-3
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
At = torch.empty((N, M), dtype=torch.float32, device=dev)
for lo in range(0, N, 8192):
    a = torch.randn((8192, M), generator=g, device=dev, dtype=torch.float64)
    a /= a.norm(dim=1, keepdim=True)
    At[lo:lo + 8192] = a.to(torch.float32)
D = cs.Dictionary(At, device=0)
idx = torch.randperm(N, generator=g, device=dev)[:k + 8]  # a few atoms more than the solvers may keep
sign = torch.randint(0, 2, (k + 8,), generator=g, device=dev).to(torch.float64) * 2 - 1
e = torch.randn(M, generator=g, device=dev, dtype=torch.float64)
b = ((At[idx].to(torch.float64) * sign[:, None]).sum(0) + e * (noise / e.norm())).cpu().numpy()
torch.cuda.synchronize()
for name, fn in (("fr", lambda: D.ctx.fr(b, k)), ("srr init=1", lambda: D.ctx.srr(b, k, 1e-12, -1, 1, 1)),
                 ("srr init=2", lambda: D.ctx.srr(b, k, 1e-12, -1, 2, 1)), ("ompr", lambda: D.ctx.ompr(b, k, 1e-6)),
                 ("omp", lambda: D.ctx.omp(b, k, 0.0))):
    fn()
    t0 = time.perf_counter()
    r = fn()
    dt = time.perf_counter() - t0
    print(f"{name:12s} {dt * 1e3:9.2f} ms  nnz={len(r[0])}  iters/order={r[2] if np.isscalar(r[2]) else len(r[2])}", flush=True)
D.close()
