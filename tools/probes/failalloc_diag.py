import os, sys, faulthandler
import numpy as np
import torch
torch.zeros(1, device='cuda').item()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load
cs = load()
A, x, b = cs.sparse_data(n=256, m=2048, k=8, rng=42, dtype=np.float32)
k, nsig = 8, 11
rng = np.random.default_rng(5)
B = np.asfortranarray(np.stack([cs.perturb(A.astype(np.float64) @ cs.sparse_vector(2048, k, rng=rng).to_dense(), 5e-3, rng=rng) for _ in range(nsig)], axis=1))
eps = float(np.finfo(np.float32).eps)
d0 = cs.Dictionary(A)
d0.ctx.comm_init(cs.comm_id(), 0, 1)
print("first communicator alive:", d0.ctx.omp_sharded(B, nsig, k, eps)[2][:3], flush=True)
for step in ("batch_after_failed_batch", "sharded", "sharded_mfma"):
    d = cs.Dictionary(A)
    print("==", step, flush=True)
    if step.startswith("sharded"):
        d.ctx.comm_init(cs.comm_id(), 0, 1)
    d.ctx.tune("fail_alloc", 3)
    try:
        (d.ctx.omp_sharded(B, nsig, k, eps, "mfma" if step.endswith("mfma") else "exact") if step.startswith("sharded") else d.ctx.omp_batch(B, k, eps))
        print("no failure?!", flush=True)
    except Exception as e:
        print("failed as planned:", e, flush=True)
    d.ctx.tune("fail_alloc", 0)
    print("second call", flush=True)
    r = d.ctx.omp_sharded(B, nsig, k, eps, "mfma" if step.endswith("mfma") else "exact") if step.startswith("sharded") else d.ctx.omp_batch(B, k, eps)
    print("second call ok", r[2][:4], flush=True)
    d.close()
