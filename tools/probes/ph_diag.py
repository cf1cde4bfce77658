import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load
from tools.probes.dyn_probe import dictionary, dev
cs = load()
for (M, N, dt) in [(40002, 6710, torch.float32), (40002, 840, torch.float32), (32768, 8192, torch.float32)]:
    At = dictionary(M, N, dt)
    D = cs.Dictionary(At, device=0)
    print(M, N, D.ctx.sweep_config())
    rng = np.random.default_rng(5)
    r = rng.standard_normal(M)
    ref = (At.to(torch.float64) @ torch.from_numpy(r).to(dev)).abs().cpu().numpy()
    for rep in range(3):
        c, i, v = D.ctx.sweep(r, topk=1)
        bad = np.nonzero(~(np.abs(c - ref) < 1e-9))[0]
        print("  rep", rep, len(bad), "bad columns", bad[:16].tolist(), c[bad[:4]], ref[bad[:4]], "nan:", int(np.isnan(c).sum()))
    D.close()
