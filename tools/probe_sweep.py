"""(needs the experimental kernel variants: `make -C compressedsensing.jl_amd/csrc experiments`)
GPU probe: sweep-kernel variants at BASELINE config 2 (and 5) shapes -> GB/s per variant.
variant = U + 8*nt + 16*f32acc + 256*workgroups_per_CU (csmp_bench_sweep)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load  # noqa: E402

cs = load()
shapes = [(4096, 65536)] if len(sys.argv) < 2 else [tuple(map(int, a.split("x"))) for a in sys.argv[1:]]
for M, N in shapes:
    At = torch.randn((N, M), device="cuda", dtype=torch.float32)
    D = cs.Dictionary(At)
    gb = M * N * 4 / 1e9
    print(f"== {M}x{N} f32 ({gb:.3f} GB per sweep), grid default")
    rows = []
    for wg in (0, 2, 3, 4, 5, 8):
        for U in (1, 2, 4):
            for nt in (0, 1):
                for f32 in (0, 1):
                    v = U + 8 * nt + 16 * f32 + 256 * wg
                    try:
                        ms = min(D.ctx.bench_sweep(v, 10) for _ in range(2))
                    except Exception as e:  # noqa: BLE001
                        print("variant", v, "failed:", e)
                        continue
                    rows.append((gb / ms * 1e3, wg, U, nt, f32, ms))
    for bw, wg, U, nt, f32, ms in sorted(rows, reverse=True):
        print(f"  {bw:8.1f} GB/s  {ms*1e3:8.1f} us  wg/CU={wg} U={U} nt={nt} f32acc={f32}")
    print("product config:", D.ctx.bench_sweep(0, 20) * 1e3, "us")
    D.close()
    del At
