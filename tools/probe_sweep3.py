"""(needs the experimental kernel variants: `make -C compressedsensing.jl_amd/csrc experiments`)
GPU probe 3: solo sweep (cpw=1, U=16) vs workgroups per CU, incl. 1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load
cs = load()
M, N = 4096, 65536
At = torch.randn((N, M), device="cuda", dtype=torch.float32)
D = cs.Dictionary(At)
gb = M * N * 4 / 1e9
for wg in (1, 2, 3, 4, 6):
    for U in (16, 8):
        v = (1 << 20) | (wg << 8) | U
        ms = min(D.ctx.bench_sweep(v, 20) for _ in range(3))
        print(f"cpw=1 U={U} wg/CU={wg}: {ms*1e3:7.1f} us {gb/ms*1e3:7.0f} GB/s")
