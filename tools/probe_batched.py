"""GPU probe: batched (MFMA-screened) OMP at BASELINE config 3 shape: time, stats, parity sample."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load
cs = load()
M, N = 4096, 65536
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator(device="cuda").manual_seed(1)
At = torch.empty((N, M), device="cuda", dtype=torch.float32)
for lo in range(0, N, 8192):
    a = torch.randn((8192, M), generator=g, device="cuda", dtype=torch.float64)
    a -= 1e-6 * a.mean(dim=1, keepdim=True); a /= a.norm(dim=1, keepdim=True)
    At[lo:lo + 8192] = a.to(torch.float32)
D = cs.Dictionary(At)
B = torch.empty((nsig, M), device="cuda", dtype=torch.float64)
for s in range(nsig):
    gs = torch.Generator(device="cuda").manual_seed(100 + s)
    idx = torch.randperm(N, generator=gs, device="cuda")[:k]
    sign = torch.randint(0, 2, (k,), generator=gs, device="cuda").to(torch.float64) * 2 - 1
    e = torch.randn(M, generator=gs, device="cuda", dtype=torch.float64)
    B[s] = (At[idx].to(torch.float64) * sign[:, None]).sum(0) + e * (5e-3 / e.norm())
eps = D.eps
idx = torch.full((nsig, k), -1, dtype=torch.int64, device="cuda")
val = torch.zeros((nsig, k), dtype=torch.float64, device="cuda")
nnz = torch.zeros(nsig, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
w = min(nsig, 128)
D.ctx.omp_batch_mfma_device(B[:w], 4, eps, idx[:w, :4].contiguous(), val[:w, :4].contiguous(), nnz[:w])  # warm (bf16 copy)
D.ctx.profile_enable(True); D.ctx.batch_stats()
t0 = time.perf_counter()
D.ctx.omp_batch_mfma_device(B, k, eps, idx, val, nnz)
D.ctx.sync()
dt = time.perf_counter() - t0
st = D.ctx.batch_stats()
print(f"batched: {nsig} signals x k={k}: {dt*1e3:.1f} ms  -> {nnz.sum().item()/dt:.0f} atoms/s, {nsig/dt:.1f} signals/s; stats {st}")
if st["screen_launches"]:
    ms = st["screen_ms"] / st["screen_launches"]
    Bpad = -(-nsig // 128) * 128
    print(f"screen GEMM: {ms*1e3:.1f} us avg, {2*M*N*Bpad/ms/1e9:.1f} TFLOP/s (bf16, padded batch {Bpad})")
# parity sample vs the exact single-signal path
i2 = torch.full((8, k), -1, dtype=torch.int64, device="cuda"); v2 = torch.zeros((8, k), dtype=torch.float64, device="cuda"); n2 = torch.zeros(8, dtype=torch.int64, device="cuda")
D.ctx.omp_batch_device(B[:8], k, eps, i2, v2, n2); D.ctx.sync()
print("parity (first 8 signals) idx equal:", bool((i2 == idx[:8]).all()), " max|dval|:", float((v2 - val[:8]).abs().max()))
