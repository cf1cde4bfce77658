"""Round 6 probe: the product sweep with its columns handed out at run time (k_sweep_dyn / the DYN tick) against the static split.
(1) bit-identity of c = A'r, arg-max and complete omp_batch results on several shapes (ragged M, N mod 4 != 0, few columns);
(2) the stand-alone sweep's time per launch over grids; (3) the pipelined batch (k_tick) over tick grids and dispatch orders.
Usage: python tools/probes/dyn_probe.py [quick]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csmp_pkg import load  # noqa: E402

cs = load()
dev = torch.device("cuda", 0)


def dictionary(M, N, dtype, seed=1):
    g = torch.Generator(device=dev).manual_seed(seed)
    At = torch.empty((N, M), dtype=dtype, device=dev)
    blk = max(1, min(N, (1 << 27) // max(M, 1)))
    for lo in range(0, N, blk):
        n = min(blk, N - lo)
        a = torch.randn((n, M), generator=g, device=dev, dtype=torch.float64)
        a /= a.norm(dim=1, keepdim=True)
        At[lo:lo + n] = a.to(dtype)
    torch.cuda.synchronize()  # (the library sweeps on a stream of its own)
    return At


def identity_checks():
    ok = True
    for (M, N, dt) in [(4096, 65536, torch.float32), (1000, 4099, torch.float32), (3000, 777, torch.float64), (4352, 8191, torch.float32),
                       (256, 4098, torch.float64), (64, 3, torch.float32), (4096, 130, torch.float32), (12288, 2049, torch.float32)]:
        At = dictionary(M, N, dt)
        D = cs.Dictionary(At, device=0)
        rng = np.random.default_rng(5)
        r = rng.standard_normal(M)
        res = {}
        for mode in (0, 1):
            D.ctx.tune("sweep_dyn", mode)
            cfg = D.ctx.sweep_config()
            cabs, idx, absv = D.ctx.sweep(r, topk=1)
            k = min(16, M // 2, N)
            Bs = torch.from_numpy(np.stack([rng.standard_normal(M) for _ in range(4)])).to(dev)
            rng = np.random.default_rng(5)  # (same signals in both modes)
            rng.standard_normal(M)
            i2 = torch.full((4, k), -1, dtype=torch.int64, device=dev)
            v2 = torch.zeros((4, k), dtype=torch.float64, device=dev)
            n2 = torch.zeros(4, dtype=torch.int64, device=dev)
            D.ctx.omp_batch_device(Bs, k, 1e-12, i2, v2, n2)
            D.ctx.sync()
            res[mode] = (cfg["dynamic"], int(idx[0]), float(absv[0]), i2.cpu().numpy().copy(), v2.cpu().numpy().copy(), n2.cpu().numpy().copy(), cabs.copy())
        ref = (At.to(torch.float64) @ torch.from_numpy(r).to(dev)).abs()
        same = res[0][1] == res[1][1] and res[0][2] == res[1][2] and np.array_equal(res[0][3], res[1][3]) and \
            np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5]) and res[0][1] == int(ref.argmax().item()) and \
            np.array_equal(res[0][6], res[1][6]) and float(np.abs(res[0][6] - ref.cpu().numpy()).max()) < 1e-12 * float(np.linalg.norm(r))
        ok = ok and same and res[0][0] == 1 and res[1][0] == 0
        print(f"identity {M}x{N} {str(dt)[6:]}: dynamic flags {res[0][0]}/{res[1][0]} argmax {res[0][1]}/{res[1][1]}/{int(ref.argmax().item())} "
              f"omp nnz {res[0][5].tolist()} same={same}", flush=True)
        D.close()
        del At
    return ok


def sweep_times(At, D):
    for mode, pools in ((1, 1), (0, 1), (0, 4), (0, 8), (0, 16)):
        D.ctx.tune("sweep_dyn", mode)
        D.ctx.tune("claim_pools", pools)
        for grid in (0, 176, 208, 256, 512):
            D.ctx.tune("sweep_grid", grid)
            ms = min(D.ctx.bench_sweep(reps=40) for _ in range(3))
            print(f"sweep {'static' if mode else 'dynamic pools %d' % pools} grid {grid or 'auto':>4}: {ms * 1e3:7.1f} us  {At.numel() * 4 / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    D.ctx.tune("sweep_grid", 0)
    D.ctx.tune("sweep_dyn", 0)
    D.ctx.tune("claim_pools", 8)


def tick_times(At, D, K=6):
    k = 256
    g = torch.Generator(device=dev).manual_seed(3)
    B = torch.randn((K, At.shape[1]), generator=g, device=dev, dtype=torch.float64)
    idx = torch.full((K, k), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((K, k), dtype=torch.float64, device=dev)
    nnz = torch.zeros(K, dtype=torch.int64, device=dev)
    ref = None
    for mode, order, grid in [(1, 0, 0), (0, 0, 0), (0, 1, 0), (0, 0, 192), (0, 1, 192), (0, 1, 224), (0, 1, 256)]:
        D.ctx.tune("sweep_dyn", mode)
        D.ctx.tune("tick_order", order)
        D.ctx.tune("tick_grid", grid)
        D.ctx.omp_batch_device(B[:3], 16, 1e-7, idx[:3, :16].contiguous(), val[:3, :16].contiguous(), nnz[:3])
        D.ctx.sync()
        best = 1e9
        for _ in range(2):
            D.ctx.profile_enable(4)
            D.ctx.profile_read(reset=True)
            t0 = time.perf_counter()
            D.ctx.omp_batch_device(B, k, 1e-7, idx, val, nnz)
            D.ctx.sync()
            dt = time.perf_counter() - t0
            n, ms = D.ctx.profile_read(reset=True)
            D.ctx.profile_enable(False)
            best = min(best, dt)
        ov = D.ctx.profile_overhead(32)
        out = (idx.cpu().numpy().copy(), val.cpu().numpy().copy())
        if ref is None:
            ref = out
        same = np.array_equal(ref[0], out[0]) and np.array_equal(ref[1], out[1])
        print(f"tick {'static ' if mode else 'dynamic'} order {order} grid {grid or 'auto':>4}: {int(nnz.sum()) / best:8.1f} atoms/s, "
              f"tick {(ms / max(n, 1) - ov) * 1e3:6.1f} us ({n} timed) identical={same}", flush=True)
    D.ctx.tune("sweep_dyn", 0)
    D.ctx.tune("tick_order", 0)
    D.ctx.tune("tick_grid", 0)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = identity_checks()
    print("identity:", "OK" if ok else "FAILED", flush=True)
    At = dictionary(4096, 65536, torch.float32, seed=2)
    D = cs.Dictionary(At, device=0)
    sweep_times(At, D)
    if not quick:
        tick_times(At, D)
    D.close()
    sys.exit(0 if ok else 1)
