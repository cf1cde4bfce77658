"""Shape table of the product sweep c = A'r (argmaxinner!(P), src/matchingpursuit.jl:181-185): for every (M, element type) a
dictionary of about 1 GiB, the sweep checked against a Float64 product computed by torch, then timed (csmp_bench_sweep: HIP events
around `reps` launches).  bench.py --workload shapes runs table(); the options below exist for tuning on the GPU box:

    python tools/sweep_shapes.py                         # the table, automatic configuration
    python tools/sweep_shapes.py --grids 192,256,384     # ... and the same sweeps under other workgroup counts
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK = 8.0e12  # bytes/s (MI355X_MICROARCH.md)
SHAPES_M = (256, 512, 1000, 3000, 4096, 4352, 8192, 12288, 32768)  # (round 6: 256 and 512 -- configs[0] is 256 x 1024, the reference's tests 32 x 48)


def make_dictionary(torch, dev, M, N, dtype, seed=7):
    """Gaussian unit-norm atoms (src/util.jl:21-27), generated in Float64 and cast once; (N, M) row-major = column-major M x N."""
    g = torch.Generator(device=dev).manual_seed(seed)
    At = torch.empty((N, M), dtype=dtype, device=dev)
    blk = max(1, (64 << 20) // (8 * M))
    for lo in range(0, N, blk):
        n = min(blk, N - lo)
        a = torch.randn((n, M), generator=g, device=dev, dtype=torch.float64)
        a -= 1e-6 * a.mean(dim=1, keepdim=True)
        a /= a.norm(dim=1, keepdim=True)
        At[lo:lo + n] = a.to(dtype)
    return At


def reference_product(torch, At, r):
    """A'r in Float64 on the exactly promoted dictionary values, in blocks (the full Float64 copy would double the footprint)."""
    N, M = At.shape
    out = torch.empty(N, dtype=torch.float64, device=At.device)
    blk = max(1, (256 << 20) // (8 * M))
    for lo in range(0, N, blk):
        out[lo:lo + blk] = At[lo:lo + blk].to(torch.float64) @ r
    return out


def check_sweep(torch, np, D, At, seed=3):
    """the library's |A'r| and arg-max against torch's Float64 product on the same values"""
    N, M = At.shape
    g = torch.Generator(device=At.device).manual_seed(seed)
    r = torch.randn(M, generator=g, device=At.device, dtype=torch.float64)
    ref = reference_product(torch, At, r).abs()
    got, ti, tv = D.ctx.sweep(r.cpu().numpy(), topk=1)
    got = torch.from_numpy(got).to(At.device)
    err = float(((got - ref).abs().max() / ref.max()).item())
    ok_arg = int(ti[0]) == int(ref.argmax().item())
    return err, ok_arg


def measure(torch, np, cs, dev, M, dtype, reps, total_bytes=1 << 30, grids=(), unit=0, check=True, tune=()):
    es = 4 if dtype == torch.float32 else 8
    N = max(8, (total_bytes // (M * es)) // 4 * 4)
    At = make_dictionary(torch, dev, M, N, dtype)
    D = cs.Dictionary(At)
    rows = []
    try:
        for key, value in tune:
            D.ctx.tune(key, value)
        variants = [None] + [g for g in grids]
        for g in variants:
            D.ctx.tune("sweep_unit", unit)
            D.ctx.tune("sweep_grid", 0 if g is None else g)
            cfg = D.ctx.sweep_config()
            err, ok = check_sweep(torch, np, D, At) if check else (None, None)
            D.ctx.bench_sweep(0, reps)  # (clocks and caches settle; the first timed run after a fresh allocation reads ~30 % slow)
            ms = sorted(D.ctx.bench_sweep(0, reps) for _ in range(5))[2]  # median of five runs of `reps` launches
            nbytes = M * N * es
            rows.append({"M": M, "N": N, "dtype": "f32" if es == 4 else "f64", "bytes": nbytes, "us": round(ms * 1e3, 2),
                         "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4),
                         "max_rel_err": err, "argmax_ok": ok, **cfg})
    finally:
        D.close()
        del At
        torch.cuda.empty_cache()
    return rows


def table(torch, np, cs, dev, reps=20, Ms=SHAPES_M, dtypes=None, **kw):
    out = []
    for M in Ms:
        for dt in (dtypes or (torch.float32, torch.float64)):
            out += measure(torch, np, cs, dev, M, dt, reps, **kw)
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--reps", type=int, default=20)
    p.add_argument("--M", type=str, default="")
    p.add_argument("--dtypes", type=str, default="f32,f64")
    p.add_argument("--grids", type=str, default="")
    p.add_argument("--unit", type=int, default=0)
    p.add_argument("--no-check", action="store_true")
    p.add_argument("--out", type=str, default="")
    p.add_argument("--tune", type=str, default="", help="csmp_tune overrides, e.g. sweep_short=1")
    a = p.parse_args()
    import numpy as np
    import torch
    from csmp_pkg import load
    cs = load()
    dev = torch.device("cuda:0")
    Ms = tuple(int(x) for x in a.M.split(",")) if a.M else SHAPES_M
    dts = tuple({"f32": torch.float32, "f64": torch.float64}[x] for x in a.dtypes.split(","))
    grids = tuple(int(x) for x in a.grids.split(",")) if a.grids else ()
    tune = tuple((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.tune.split(",") if kv)
    rows = table(torch, np, cs, dev, a.reps, Ms, dts, grids=grids, unit=a.unit, check=not a.no_check, tune=tune)
    for r in rows:
        print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
