"""The batched (MFMA-screened) path at SMALL batches of the configs[1] problem (A 4096 x 65536, k = 256): where does it overtake the
single-signal paths?  Prints atoms/s per batch size.    python tools/probe_small_batches.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from csmp_pkg import load

cs = load()
dev = torch.device("cuda", 0)
At = bench.make_dictionary(torch, dev)
D = cs.Dictionary(At, device=0)
for nsig in (4, 8, 18, 32, 64, 128, 256):
    out = bench.measure_batched(2, 1, cs, torch, None, dev, 0, 1, At, D, False, nsig=nsig, k=256)
    print(json.dumps({"signals": nsig, "k": 256, "atoms_per_s": round(out["value"], 1), "ms_per_batch": round(out["ms_per_step"], 3),
                      "us_per_omp_step": round(out["roofline"]["whole_step"]["ms_per_omp_step"] * 1e3, 1), "screen_us": round(out["roofline"]["avg_launch_us"], 1),
                      "batch_stats": out["batch_stats"], "matches_exact_path_on_sample": out["matches_exact_path_on_sample"]}), flush=True)
D.close()
