"""(needs the experimental kernel variants: `make -C compressedsensing.jl_amd/csrc experiments`)
GPU probe 2: column-per-wave variants of the sweep (experimental, via csmp_bench_sweep)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load
cs = load()
M, N = 4096, 65536
At = torch.randn((N, M), device="cuda", dtype=torch.float32)
D = cs.Dictionary(At)
gb = M * N * 4 / 1e9
rows = []
base = min(D.ctx.bench_sweep(0, 20) for _ in range(3))
print(f"product: {base*1e3:.1f} us {gb/base*1e3:.0f} GB/s")
for cpw, Us in ((1, (4, 8, 16)), (2, (2, 4, 8))):
    for U in Us:
        for wg in (2, 3, 4, 5, 6, 8):
            v = (cpw << 20) | (wg << 8) | U
            try:
                ms = min(D.ctx.bench_sweep(v, 10) for _ in range(2))
            except Exception as e:
                print("fail", cpw, U, wg, e); continue
            rows.append((gb / ms * 1e3, cpw, U, wg, ms))
for bw, cpw, U, wg, ms in sorted(rows, reverse=True):
    print(f"  {bw:8.1f} GB/s {ms*1e3:7.1f} us  cpw={cpw} U={U} wg/CU={wg}")
