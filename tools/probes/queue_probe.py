"""Round 6 probe: ONE stand-alone sweep launch with more workgroups than the chip holds at the requested residency (the LDS request
decides how many share a CU): the surplus queues, and a CU that a workgroup leaves goes to the next in line -- balance at a finer
grain than one workgroup per CU for the whole launch.  4096 x 65536 f32 (1 GiB) and 8192 x 131072 f32 (4 GiB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load  # noqa: E402
from tools.probes.dyn_probe import dictionary  # noqa: E402

cs = load()
for (M, N) in ((4096, 65536), (8192, 131072)):
    At = dictionary(M, N, torch.float32, seed=2)
    D = cs.Dictionary(At, device=0)
    for lds in (0, 54, 81):
        D.ctx.tune("sweep_lds_kib", lds)
        for grid in (0, 192, 256, 384, 512, 768, 1024, 2048):
            if lds == 0 and grid not in (0, 192, 256, 512):
                continue
            D.ctx.tune("sweep_grid", grid)
            ms = min(D.ctx.bench_sweep(reps=30) for _ in range(3))
            print(f"{M}x{N} lds request {lds or 'natural':>7} KiB grid {grid or 'auto':>5}: {ms * 1e3:7.1f} us  {At.numel() * 4 / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    D.close()
    del At
    torch.cuda.empty_cache()
