"""One omp(A, b, 256) call at a time on the configs[1] dictionary (the reference API's own shape of work): ms per solve.
Round 2: 45.3 ms = 176.9 us per atom = sweep 160.0 + k_qr1 7.1 + k_qr2 8.1 (rocprofv3 --kernel-trace --stats), against 158-160 us per
atom when three signals share the tick pipeline (csmp_omp_batch)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
from csmp_pkg import load
cs = load()
dev = torch.device("cuda", 0)
At = bench.make_dictionary(torch, dev)
D = cs.Dictionary(At, device=0)
B = bench.make_signals(torch, dev, At, 0, 4)
b = B[0].cpu().numpy()
D.ctx.omp(b, 256, D.eps)
t0 = time.perf_counter()
for s in range(1, 4):
    D.ctx.omp(B[s].cpu().numpy(), 256, D.eps)
print("lone omp ms per solve", (time.perf_counter() - t0) / 3 * 1e3, flush=True)
