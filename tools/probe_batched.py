"""C3 (1024 signals sharing A 4096 x 65536 f32, k = 128) through csmp_omp_batch_mfma under a list of option settings
(csmp_set_option, one process): atoms/s, the screening launch average, the batch statistics, and whether every setting
returns the same supports.  usage: python tools/probe_batched.py [only=a,b] [name=option:val,option:val ...]"""
import json, os, sys, time
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
B = bench.make_signals_fast(torch, dev, At, 0, 2 * nsig, k).reshape(2, nsig, bench.M)
torch.cuda.synchronize()
configs = [("default", {}), ("rigorous", {"batch_cert": "1"}), ("gram", {"batch_gram": "1"}), ("gram-rigorous", {"batch_gram": "1", "batch_cert": "1"})]
only = None
for a in sys.argv[1:]:
    if a.startswith("only="):
        only = a[5:].split(",")
        continue
    name, _, kv = a.partition("=")
    configs.append((name, dict(x.split(":") for x in kv.split(",") if x)))
if only:
    configs = [c for c in configs if c[0] in only]
ref = None
for name, env in configs:
    for kk in ("batch_cert", "batch_gram", "batch_window"):
        D.ctx.set_option(kk, int(env.get(kk, 0)))
    idx = torch.full((2, nsig, k), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((2, nsig, k), dtype=torch.float64, device=dev)
    nnz = torch.zeros((2, nsig), dtype=torch.int64, device=dev)
    D.ctx.omp_batch_mfma_device(B[0], k, D.eps, idx[0], val[0], nnz[0])  # warm
    D.ctx.sync()
    D.ctx.profile_enable(True)
    D.ctx.batch_stats()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        D.ctx.omp_batch_mfma_device(B[1], k, D.eps, idx[1], val[1], nnz[1])
    D.ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    st = D.ctx.batch_stats()
    lay = D.ctx.batch_layout()
    D.ctx.profile_enable(False)
    same = None
    if ref is None:
        ref = idx[1].clone()
    else:
        same = bool((ref == idx[1]).all().item())
    scr_us = st["screen_ms"] / max(st["screen_launches"], 1) * 1e3
    print(json.dumps({"config": name, "env": env, "ms_per_batch": dt * 1e3, "atoms_per_s": nsig * k / dt, "ms_per_omp_step": dt / k * 1e3,
                      "screen_us": scr_us, "screen_tflops": 2.0 * bench.M * bench.N * lay["screen_signals"] / (scr_us * 1e-6) / 1e12 if scr_us else None,
                      "layout": lay, "resolved": st["resolved_exactly"], "uncertain": st["uncertain"], "illcond": st["illcond"],
                      "same_supports_as_first": same}), flush=True)
D.close()
