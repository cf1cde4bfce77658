"""Fixed cost against streaming rate of the sweeps: lone csmp_omp (k atoms) at M = 4096 and several N, exact and screened; run
under rocprofv3 --kernel-trace --stats for the kernel durations.    python tools/probe_screened_n.py N [k]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from csmp_pkg import load

cs = load()
N = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
At = torch.randn((N, 4096), generator=g, device=dev, dtype=torch.float32)
At /= At.norm(dim=1, keepdim=True)
D = cs.Dictionary(At, device=0)
sel = torch.randperm(N, generator=g, device=dev)[:k]
y = At[sel].to(torch.float64).sum(0).cpu().numpy()
for scr in (0, 1):
    D.ctx.set_option("screened_sweep", scr)
    for rep in range(3):
        out = D.ctx.omp(y, k, 1e-7)
    print(scr, len(out[0]), D.ctx.screened_stats())
D.close()
