# needs the experimental kernel variants: make -C compressedsensing.jl_amd/csrc experiments
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load
cs = load()
M, N = 4096, 65536
At = torch.randn((N, M), device="cuda", dtype=torch.float32)
D = cs.Dictionary(At)
Us = [int(u) for u in os.environ.get("US", "16,8,4").split(",")]
for U in Us:
    v = (3 << 20) | (1 << 8) | U
    ms = min(D.ctx.bench_sweep(v, 30) for _ in range(3))
    print(f"pf U={U} NBLK={os.environ.get('CSMP_SWEEP_NBLK')}: {ms*1e3:7.1f} us {M*N*4/ms/1e6:7.0f} GB/s")
