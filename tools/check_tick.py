import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load
cs = load()
M, N, k = 4096, 65536, 256
g = torch.Generator(device="cuda").manual_seed(5)
At = torch.randn((N, M), generator=g, device="cuda", dtype=torch.float32); At /= At.norm(dim=1, keepdim=True)
D = cs.Dictionary(At)
nsig = 6
B = torch.empty((nsig, M), dtype=torch.float64, device="cuda")
for s in range(nsig):
    idx = torch.randperm(N, generator=g, device="cuda")[:k]
    sign = torch.randint(0, 2, (k,), generator=g, device="cuda").to(torch.float64) * 2 - 1
    e = torch.randn(M, generator=g, device="cuda", dtype=torch.float64)
    B[s] = (At[idx].to(torch.float64) * sign[:, None]).sum(0) + e * (5e-3 / e.norm())
idx = torch.full((nsig, k), -1, dtype=torch.int64, device="cuda"); val = torch.zeros((nsig, k), dtype=torch.float64, device="cuda"); nnz = torch.zeros(nsig, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
D.ctx.omp_batch_device(B, k, D.eps, idx, val, nnz)
t0 = time.perf_counter(); D.ctx.omp_batch_device(B, k, D.eps, idx, val, nnz); D.ctx.sync(); dt = time.perf_counter() - t0
print("tick path:", nnz.tolist(), f"{dt/nsig/k*1e6:.1f} us/atom")
ok = True
for s in range(nsig):
    i, v, o = D.ctx.omp(B[s].cpu().numpy(), k, D.eps)
    ok &= np.array_equal(i, idx[s, :len(i)].cpu().numpy()) and np.allclose(v, val[s, :len(i)].cpu().numpy(), rtol=1e-12, atol=0) and len(i) == int(nnz[s])
print("tick == solo:", ok, "TICK_NBLK", os.environ.get("CSMP_TICK_NBLK"))
