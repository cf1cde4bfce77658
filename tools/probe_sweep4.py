import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load
cs = load()
M, N = 4096, 65536
At = torch.randn((N, M), device="cuda", dtype=torch.float32)
D = cs.Dictionary(At)
ms = min(D.ctx.bench_sweep(0, 30) for _ in range(3))
print(f"CSMP_SWEEP_NBLK={os.environ.get('CSMP_SWEEP_NBLK')}: {ms*1e3:7.1f} us {M*N*4/ms/1e6:7.0f} GB/s")
