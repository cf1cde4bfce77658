"""diagnostic: which columns of c = A'r differ between the dynamic and the static sweep"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load
from tools.probes.dyn_probe import dictionary, dev
cs = load()
for (M, N, dt) in [(4096, 8192, torch.float32), (4096, 65536, torch.float32), (2048, 65536, torch.float32)]:
    At = dictionary(M, N, dt)
    D = cs.Dictionary(At, device=0)
    rng = np.random.default_rng(5)
    r = rng.standard_normal(M)
    D.ctx.tune("sweep_dyn", 1)
    ref, _, _ = D.ctx.sweep(r, topk=1)
    D.ctx.tune("sweep_dyn", 0)
    print(M, N, D.ctx.sweep_config())
    for pools in (1, 1, 8):
        D.ctx.tune("claim_pools", pools)
        for rep in range(4):
            c, i, v = D.ctx.sweep(r, topk=1)
            bad = np.nonzero(c != ref)[0]
            print(f"  pools {pools} rep {rep}: {len(bad)} columns differ", bad[:12].tolist(), flush=True)
            if len(bad):
                print("     got", c[bad[:6]], "\n     ref", ref[bad[:6]], "\n     zeros", int((c == 0).sum()), "in ref set", int(np.isin(c[bad[:2000]], ref).sum()), "of", min(len(bad), 2000))
    D.ctx.tune("sweep_dyn", 1)
    c, i, v = D.ctx.sweep(r, topk=1)
    print("  static again:", int((c != ref).sum()), "differ")
    D.close()
