"""debug: C3 through csmp_omp_batch_mfma with CSMP_BATCH_DEBUG (experiments build): the first failed certificate per signal"""
import os, sys
os.environ["CSMP_BATCH_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from csmp_pkg import load
cs = load()
dev = torch.device("cuda", 0)
At = bench.make_dictionary(torch, dev)
D = cs.Dictionary(At, device=0)
nsig, k = 1024, 128
B = bench.make_signals_fast(torch, dev, At, 0, nsig, k).reshape(nsig, bench.M)
torch.cuda.synchronize()
idx = torch.full((nsig, k), -1, dtype=torch.int64, device=dev)
val = torch.zeros((nsig, k), dtype=torch.float64, device=dev)
nnz = torch.zeros((nsig,), dtype=torch.int64, device=dev)
D.ctx.omp_batch_mfma_device(B, k, D.eps, idx, val, nnz)
D.ctx.sync()
print(D.ctx.batch_stats())
