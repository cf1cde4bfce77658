#!/usr/bin/env python3
"""bench.py -- single-signal OMP on MI355X at BASELINE.json configs[1]:
A 4096 x 65536 Float32 Gaussian dictionary (unit-norm atoms), k = 256.

A "step" is one complete omp(A, b, k) solve of one synthetic signal = 256 atoms selected, each by
one full sweep of the 1 GiB dictionary (K1 of SURVEY.md section 2.3) plus the on-device QR append.
Every atom of every signal is selected by its own single-signal sweep (1 GiB streamed per atom);
independent signals are pipelined three at a time so that the short append chain of two signals
runs underneath the sweep of the third (csmp_omp_batch, DESIGN.md "tick kernel").
Inputs (dictionary and signals) are resident in HBM before the timed region starts; results stay
on the device and, with N > 1 ranks, are exchanged by ONE all_gather (RCCL) inside the timed
region.  Signals are independent: weak scaling, K signals per rank.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).  At N = 1 the line also
carries, under "secondary", the results of the other single-GPU configurations of BASELINE.json measured
by the same process right after the headline: configs[2] (batched, bf16 MFMA screen) and configs[4] (GOMP
S = 4 and Subspace Pursuit on 8192 x 131072, k = 512) -- each with its own roofline block.  The headline
stays configs[1]; --no-secondary skips them.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

M, N, K_ATOMS = 4096, 65536, 256
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
NOISE = 5e-3  # ||e||_2, as test/matchingpursuit.jl:12-13 (perturb(b, delta/2), delta = 1e-2)
SEED_A = 0xC0FFEE


def free_port():
    """A free TCP port on 127.0.0.1 for the launcher's rendezvous (no fixed port: two benches may share a box)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def exchange_results(idx, val, nnz, group=None):
    """The ONE collective of the signal-sharded path: every rank's (idx, val, nnz) block,
    packed on the device the tensors live on into rows of 2k+1 Float64 (csmp_pack_results' layout) and moved by a
    single all_gather (RCCL over xGMI under the "nccl" backend; the gloo CPU test drives this very function).
    idx, val: (n, k); nnz: (n,), the same n on every rank.  Returns the (world * n, 2k+1) tensor in rank order."""
    import torch.distributed as dist
    from csmp_pkg import load
    cs = load()
    packed = cs.pack_t(idx, val, nnz)
    return cs.gather_packed(packed, packed.shape[0] * dist.get_world_size(group), group)


def library_collective(cs, dist, D, use_dist, share_gpu):
    """True when the ONE collective of the signal-sharded path runs inside the library (csmp_omp_sharded: ncclAllGather on the
    context's stream): every N > 1 run under the "nccl" backend.  The gloo rehearsal (--share-gpu: RCCL refuses two ranks on one
    device) and a box whose RCCL cannot be bound keep the host-side all_gather of the same packed rows."""
    if not use_dist or share_gpu or dist.get_backend() != "nccl":
        return False
    import torch
    ok = 1
    try:
        cs.library_comm(D.ctx)
    except Exception as e:  # noqa: BLE001
        ok = 0
        print(f"bench.py: csmp_comm_init failed ({e!r}); the gather falls back to torch.distributed", file=sys.stderr, flush=True)
    # every rank must take the same path: one that could not bind RCCL or join the communicator sends everybody to the host-side gather
    flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if ok:
            D.ctx.comm_free()
            D.ctx._comm_key = None
        return False
    return True


def rocprof_row(kernel_substr, pattern="r*_bench_kernel_stats.csv"):
    """AverageNs of the newest COMMITTED rocprofv3 --kernel-trace --stats row whose kernel name contains `kernel_substr`
    (profiles/, named per round).  It was measured on the build that was profiled, not in this process: the line carries it as
    `committed_profile` with its file name and no derived fraction; every fraction in the line comes from this run's own timers."""
    import csv
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            for row in csv.DictReader(open(f)):
                if kernel_substr in row.get("Name", ""):
                    best = {"file": os.path.relpath(f, ROOT), "kernel": row["Name"].split("(")[0], "calls": int(row["Calls"]),
                            "avg_launch_us": float(row["AverageNs"]) / 1e3,
                            "note": "historical: rocprofv3 --kernel-trace --stats of the same command on the build that file was taken from"}
        except Exception:  # noqa: BLE001
            pass
    return best


LINE_BUDGET = 3000  # bytes: the driver keeps ~8 KB of stdout tail; round 3's 30 KB line was cut and the record did not parse
DETAIL_FILE = "bench_secondary.json"
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_timed", "avg_launch_us",
              "algorithmic_bytes_per_launch", "flops_per_launch", "launch_duration_us", "launches_in_flight", "whole_job_GBps", "whole_job_frac")
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "signals_per_sec", "config", "roofline", "cpu_baseline", "secondary", "secondary_file", "gather_check",
             "ranks_seen", "devices", "matches_exact_path_on_sample", "batch_stats", "equals_unsharded_omp", "ranks_agree_on_first_support", "error")


def _rnd(x):
    """Floats to 6 significant digits (the line is a report, not a checkpoint)."""
    if isinstance(x, float):
        return float("%.6g" % x)
    if isinstance(x, dict):
        return {k: _rnd(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rnd(v) for v in x]
    return x


def headline(out):
    """The ONE compact JSON object bench.py prints as its LAST stdout line (<= LINE_BUDGET bytes): the contract's fields, the
    headline roofline and cpu_baseline, and ONE scalar per secondary workload.  Everything else (per-workload rooflines, notes,
    option dumps) is the detail written to DETAIL_FILE by emit()."""
    o = {k: out[k] for k in _TOP_KEYS if k in out}
    if "roofline" in o:
        r = out["roofline"]
        o["roofline"] = {k: r[k] for k in _ROOF_KEYS if k in r}
        if isinstance(o["roofline"].get("kernel"), str):
            o["roofline"]["kernel"] = o["roofline"]["kernel"].split(" = ")[0].split(" (")[0][:64]
    if "config" in o:
        o["config"] = {k: (v if not isinstance(v, str) else v[:200]) for k, v in out["config"].items() if isinstance(v, (str, int, float, bool)) or v is None}
    if isinstance(o.get("cpu_baseline"), dict):
        c = out["cpu_baseline"]
        o["cpu_baseline"] = {k: (c[k][:160] if isinstance(c[k], str) else c[k]) for k in ("value", "unit", "cores", "kind", "sample", "selection_order_matches_gpu", "error") if k in c}
    if isinstance(o.get("secondary"), dict):
        o["secondary"] = {name: (blk.get("value") if isinstance(blk, dict) and "error" not in blk else None) for name, blk in out["secondary"].items()}
        for nm in ("ompr_8f2", "srr_8f2"):  # the same solves three in flight (the headline itself is three signals pipelined)
            blk = out["secondary"].get(nm)
            if isinstance(blk, dict) and isinstance(blk.get("three_in_flight"), dict):
                o["secondary"][nm + "_three_in_flight"] = blk["three_in_flight"]["solves_per_s"]
        b3 = out["secondary"].get("batched_c3")
        if isinstance(b3, dict) and "roofline_composite" in b3:  # the batched step against BOTH of its ceilings (verdict round 4, item 7)
            o["secondary"]["batched_c3_frac_of_mfma_only_ceiling"] = b3["roofline"]["whole_step"]["frac"]
            o["secondary"]["batched_c3_frac_of_composite_ceiling"] = b3["roofline_composite"]["frac"]
    if isinstance(o.get("metric"), str):
        o["metric"] = o["metric"][:200]
    o = _rnd(o)
    if isinstance(o.get("devices"), list):
        o["devices"] = [str(d)[:48] for d in o["devices"]]
    line = json.dumps(o, separators=(",", ":"))
    for drop in ("batch_stats", "devices", "gather_check", "secondary"):  # never needed at today's sizes: a guard, not a plan
        if len(line) <= LINE_BUDGET:
            break
        o.pop(drop, None)
        line = json.dumps(o, separators=(",", ":"))
    return line


def emit(out):
    """Detail to DETAIL_FILE in the cwd (best effort), then the compact headline as the last stdout line."""
    try:
        with open(DETAIL_FILE, "w") as f:
            json.dump(out, f, indent=1)
        out = dict(out, secondary_file=DETAIL_FILE)
    except OSError:
        pass
    print(headline(out), flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=18)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch-cert", choices=["statistical", "rigorous"], default="rigorous", help="--workload batched: CSMP_OPT_BATCH_CERT (rigorous is the library's default)")
    p.add_argument("--batch-gram", action="store_true", help="--workload batched: CSMP_OPT_BATCH_GRAM (resident G = A'A, 32 GiB)")
    p.add_argument("--no-in-flight", action="store_true", help="--workload ompr / srr: skip the three-solves-in-flight part (a kernel trace of the one-solve-at-a-time loop alone)")
    p.add_argument("--tune", type=str, default="", help="measurement overrides of the sweep configuration (csmp_internal.h), e.g. tick_grid=224,sweep_grid=192")
    p.add_argument("--workload", choices=["omp", "shapes", "screened", "streamed", "batched", "gomp", "gomp_single", "sp", "sp_single", "fr", "ompr", "srr", "colsharded"], default="omp",
                   help="omp = configs[1] (default, the headline metric); batched = configs[2]/[3]: 1024 signals per GPU, "
                        "k=128, bf16 MFMA screening GEMM + Float64 rescoring (a step = one batch); gomp / sp = configs[4]: "
                        "A 8192x131072, k=512, GOMP with S=4 atoms per sweep / Subspace Pursuit (a step = one solve)")
    p.add_argument("--share-gpu", action="store_true",
                   help="rehearsal of the N > 1 path on a node with fewer GPUs than ranks: rank r uses GPU r mod (visible GPUs) and the "
                        "exchange runs over gloo through host memory (RCCL refuses two ranks on one device) -- exercises every line of the "
                        "multi-rank code; the numbers are NOT a scaling measurement")
    p.add_argument("--in-flight", type=int, default=0, help="--workload sp: CSMP_OPT_SOLVES_IN_FLIGHT (1..4; 0 = the library's default)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--batch-screen", choices=["f16", "bf16", "int8"], default="f16", help="--workload batched: CSMP_OPT_BATCH_SCREEN (operands of the screening GEMM; int8 needs --batch-cert statistical)")
    p.add_argument("--screened", action="store_true", help="--workload gomp / gomp_single / sp / sp_single: CSMP_OPT_SCREENED_SWEEP (image sweeps, certified selections)")
    p.add_argument("--screen-image", choices=["f16", "bf16", "int8"], default="f16", help="--workload screened, --screened: the image the sweeps read (CSMP_OPT_SCREENED_SWEEP = 3 / 1 / 2)")
    p.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / configs[4] blocks of the default line")
    p.add_argument("--profile-every", type=int, default=8, help="time every n-th sweep launch with HIP events (1 = all)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    return p.parse_args()


def make_dictionary(torch, dev):
    """src/util.jl:21-27 on the device: randn in Float64, subtract 1e-6 * column mean, unit 2-norm,
    cast ONCE to Float32.  Stored as (N, M) row-major = column-major M x N; same on every rank."""
    g = torch.Generator(device=dev).manual_seed(SEED_A)
    At = torch.empty((N, M), dtype=torch.float32, device=dev)
    blk = 8192
    for lo in range(0, N, blk):
        a = torch.randn((blk, M), generator=g, device=dev, dtype=torch.float64)
        a -= 1e-6 * a.mean(dim=1, keepdim=True)
        a /= a.norm(dim=1, keepdim=True)
        At[lo:lo + blk] = a.to(torch.float32)
    return At


def make_signals(torch, dev, At, first_id, count):
    """Planted k-sparse +-1 x0 (src/util.jl:13-19), b = A x0 + e with ||e||_2 = 5e-3
    (src/util.jl:50-55), formed in Float64 from the CAST dictionary.  Seeded by global signal id."""
    B = torch.empty((count, M), dtype=torch.float64, device=dev)
    for s in range(count):
        g = torch.Generator(device=dev).manual_seed(1_000_003 * (first_id + s) + 17)
        idx = torch.randperm(N, generator=g, device=dev)[:K_ATOMS]
        sign = torch.randint(0, 2, (K_ATOMS,), generator=g, device=dev).to(torch.float64) * 2 - 1
        b = (At[idx].to(torch.float64) * sign[:, None]).sum(dim=0)
        e = torch.randn(M, generator=g, device=dev, dtype=torch.float64)
        B[s] = b + e * (NOISE / e.norm())
    return B


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota).  (The GPU
    pool's boxes show 256 logical CPUs but cap the container at a 16-CPU quota; oversubscribing
    OpenMP beyond the quota makes the sweep 10-30x slower.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return n


def cpu_baseline(At, Bsig, gpu_order0, seconds):
    """The oracle (oracle/csmp_oracle.c: the reference algorithm restated, OpenMP over the host
    cores the container may use) timed on a bounded sample of THE SAME workload: complete k=256
    solves of the first timed signals on the same dictionary, until `seconds` of CPU work are done.
    Doubles as an in-bench parity check of the GPU's selection order on signal 0."""
    import numpy as np
    from oracle import oracle_c
    oracle_c.build()
    A = At.cpu().numpy().T  # (M, N) Fortran-ordered view, no copy
    cores = usable_cores()
    eps = float(np.finfo(np.float32).eps)
    oracle_c.omp(A, Bsig[0].cpu().numpy(), 2, eps, nthreads=cores)  # page in / spin up the team
    atoms, solved, same = 0, 0, None
    t0 = time.perf_counter()
    while solved < Bsig.shape[0] and (time.perf_counter() - t0) < seconds:
        idx, val, order = oracle_c.omp(A, Bsig[solved].cpu().numpy(), K_ATOMS, eps, nthreads=cores)
        if solved == 0:
            same = bool(np.array_equal(order, gpu_order0))
        atoms += len(order)
        solved += 1
    dt = time.perf_counter() - t0
    return {"value": atoms / dt, "unit": "atoms/s", "cores": cores, "kind": "port",
            "sample": f"{solved} complete k={K_ATOMS} solves ({atoms} atoms) of the timed signals, same 4096x65536 f32 "
                      f"dictionary, oracle/csmp_oracle.c with {cores} OpenMP threads, {dt:.1f} s",
            "selection_order_matches_gpu": same}


def make_signals_fast(torch, dev, At, first_id, count, k):
    """Same recipe as make_signals, vectorised over signals (the batched workload needs thousands)."""
    g = torch.Generator(device=dev).manual_seed(7_000_003 * (first_id + 1) + 29)
    B = torch.empty((count, M), dtype=torch.float64, device=dev)
    for lo in range(0, count, 64):
        n = min(64, count - lo)
        idx = torch.stack([torch.randperm(N, generator=g, device=dev)[:k] for _ in range(n)])
        sign = torch.randint(0, 2, (n, k), generator=g, device=dev).to(torch.float64) * 2 - 1
        b = torch.einsum("skm,sk->sm", At[idx].to(torch.float64), sign)
        e = torch.randn((n, M), generator=g, device=dev, dtype=torch.float64)
        B[lo:lo + n] = b + e * (NOISE / e.norm(dim=1, keepdim=True))
    return B


MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (the 2:1-sparse figure is not used)


def per_signal_roofline(nsig, k, gram, us):
    """HBM roofline of k_b_pick + k_b_append over one OMP step of a batch (averaged over the k steps of a solve)."""
    j_avg = (k - 1) / 2.0
    j2_avg = sum(j * j for j in range(k)) / k
    per_signal = (1 if gram else 2) * j_avg * M * 4 + M * 4 + 2 * M * 8 + M * 2 + j2_avg * 8 + (N // 128) * 4 * 8 + M * 8
    b = nsig * per_signal
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_step": b, "us_per_step": us,
            "achieved": b / (us * 1e-6) / 1e9 if us > 0 else 0.0, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else 0.0,
            "traffic": None, "note": "everything of the step that is not the screening launch; rescored window columns not counted"}


def composite_roofline(flops, peak_tf, hbm_bytes, ms_per_omp_step, nsig):
    t_mfma = flops / (peak_tf * 1e12)
    t_hbm = hbm_bytes / (HBM_PEAK_GBS * 1e9)
    return {"bound": "mfma then hbm, serial", "mfma_floor_us": t_mfma * 1e6, "hbm_floor_us": t_hbm * 1e6, "floor_us": (t_mfma + t_hbm) * 1e6,
            "measured_us": ms_per_omp_step * 1e3, "frac": (t_mfma + t_hbm) / (ms_per_omp_step * 1e-3),
            "ceiling_atoms_per_s": nsig / (t_mfma + t_hbm)}


def measure_batched(K, W, cs, torch, dist, dev, rank, world, At, D, use_dist, cert=1, gram=0, nsig=1024, k=128, screen=3):
    """configs[2] (1 GPU) / configs[3] (8192 signals over 8 GPUs): 1024 signals per GPU sharing A, k = 128.
    A step = one batch of 1024 complete solves.  cert / gram: the options CSMP_OPT_BATCH_CERT / CSMP_OPT_BATCH_GRAM of the
    contexts (include/csmp.h).  Returns the result dict on rank 0, None elsewhere."""
    eps = D.eps
    dsync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)  # (the gloo CPU test drives this function too)
    D.ctx.set_option("batch_cert", cert)
    D.ctx.set_option("batch_screen", screen)  # 3: binary16 operands (the default); 0: bf16; 1: int8 (v_mfma_i32_16x16x64_i8; statistical certificate only)
    t_setup = time.perf_counter()
    D.ctx.set_option("batch_gram", gram)
    B = make_signals_fast(torch, dev, At, rank * (K + W), (K + W) * nsig, k).reshape(K + W, nsig, M)
    idx = torch.full((K + W, nsig, k), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((K + W, nsig, k), dtype=torch.float64, device=dev)
    nnz = torch.zeros((K + W, nsig), dtype=torch.int64, device=dev)
    dsync()
    gram_seconds = None
    if gram:  # the resident Gram matrix is built by the first call that needs it: once per dictionary, like the bf16 image
        D.ctx.omp_batch_mfma_device(B[0][:256].contiguous(), 1, eps, idx[0][:256, :1].contiguous(), val[0][:256, :1].contiguous(), nnz[0][:256].contiguous())
        D.ctx.sync()
        gram_seconds = time.perf_counter() - t_setup
    for w in range(W):
        D.ctx.omp_batch_mfma_device(B[w], k, eps, idx[w], val[w], nnz[w])
    D.ctx.sync()
    D.ctx.profile_enable(True)
    D.ctx.batch_stats()
    dsync()
    if use_dist:  # warm the collective at the size and through the packing kernels of the timed one
        exchange_results(idx[W:].reshape(-1, k), val[W:].reshape(-1, k), nnz[W:].reshape(-1))
        dist.barrier()
        dsync()
    t0 = time.perf_counter()
    resolved = uncertain = illcond = 0
    screen_n, screen_ms = 0, 0.0
    for s in range(W, W + K):
        D.ctx.omp_batch_mfma_device(B[s], k, eps, idx[s], val[s], nnz[s])
        st = D.ctx.batch_stats()
        resolved += st["resolved_exactly"]
        uncertain += st["uncertain"]
        illcond += st["illcond"]
        screen_n += st["screen_launches"]
        screen_ms += st["screen_ms"]
    D.ctx.sync()
    if use_dist:  # one gather of every rank's packed results
        exchange_results(idx[W:].reshape(-1, k), val[W:].reshape(-1, k), nnz[W:].reshape(-1))
    dsync()
    if use_dist:
        dist.barrier()
        dsync()
    dt = time.perf_counter() - t0
    D.ctx.profile_enable(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    atoms = torch.tensor([float(nnz[W:].sum().item())], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(atoms, op=dist.ReduceOp.SUM)
    tmax, atoms = tmax.item(), atoms.item()
    if rank != 0:
        return None
    flops = 2.0 * M * N * nsig  # per OMP step of the batch
    lay = D.ctx.batch_layout()  # signal columns of one screening launch (the whole batch, padded to 256-signal tiles)
    flops_launch = 2.0 * M * N * lay["screen_signals"]
    tf = flops_launch / (screen_ms / max(screen_n, 1) / 1e3) / 1e12 if screen_n else 0.0
    ms_per_omp_step = tmax / K / k * 1e3
    # parity spot check against the exact single-signal path (first 4 signals of the first timed batch)
    i2 = torch.full((4, k), -1, dtype=torch.int64, device=dev)
    v2 = torch.zeros((4, k), dtype=torch.float64, device=dev)
    n2 = torch.zeros(4, dtype=torch.int64, device=dev)
    D.ctx.omp_batch_device(B[W][:4].contiguous(), k, eps, i2, v2, n2)
    D.ctx.sync()
    same = bool((i2 == idx[W][:4]).all().item()) and float((v2 - val[W][:4]).abs().max().item()) < 1e-9
    D.ctx.set_option("batch_cert", 1)  # (back to the library's defaults)
    D.ctx.set_option("batch_screen", 3)
    D.ctx.set_option("batch_gram", 0)  # (releases the 8 N^2 bytes)
    i8 = screen == 1 and not cert
    opname = "int8" if i8 else "bf16" if screen == 0 else "f16"
    return {
        "metric": "batched OMP atoms selected/sec, 1024 signals per GPU sharing A 4096x65536, k=128 (%s MFMA screen + f64 rescoring, %s certificate)" % (
            opname, "rigorous" if cert else "statistical"),
        "value": atoms / tmax, "unit": "atoms/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": tmax / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("int8 MFMA screen (i32 accumulate)" if i8 else opname + " MFMA screen (f32 accumulate)") + " + f64 rescoring/append", "data": "synthetic",
        "signals_per_sec": K * nsig * world / tmax,
        "config": {"workload": "configs[2]/[3]: batched OMP, 1024 signals per GPU sharing A 4096x65536 Float32, k=128",
                   "signals_per_gpu_per_step": nsig, "sharding": f"signals over {world} GPU(s), A replicated, one all_gather"},
        # (int8 operands: the dense int8 peak is twice the bf16 one -- MI355X_MICROARCH.md; whole_step stays priced against the bf16
        # ceiling, the figure the earlier rounds and their verdicts use, with the int8-peak fraction beside it)
        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TF * (2 if i8 else 1), "unit": "TOP/s (int8)" if i8 else "TFLOP/s",
                     "frac": tf / (MFMA_PEAK_TF * (2 if i8 else 1)), "traffic": None,
                     "kernel": D.ctx.batch_screen_kernel(),
                     "launches_timed": int(screen_n), "flops_per_launch": flops_launch, "avg_launch_us": screen_ms / max(screen_n, 1) * 1e3,
                     "signals_per_launch": lay["screen_signals"], "streams": lay["streams"],
                     # the whole OMP step of the batch (screen + rescoring/append of every signal) against the same ceiling
                     "whole_step": {"ms_per_omp_step": ms_per_omp_step, "achieved": flops / (ms_per_omp_step / 1e3) / 1e12,
                                    "frac": flops / (ms_per_omp_step / 1e3) / 1e12 / MFMA_PEAK_TF, "frac_is_of": "the dense bf16 peak (2.5 PFLOP/s)",
                                    "frac_of_int8_peak": (flops / (ms_per_omp_step / 1e3) / 1e12 / (2 * MFMA_PEAK_TF)) if i8 else None}},
        # the per-signal kernels of a step (k_b_pick + k_b_append) against the HBM roofline: their ALGORITHMIC bytes -- per signal and step
        # (1 or 2) j columns of A_S, the new column, the residual in and out, its bf16 image, T and T' (j^2/2 x 8 B each), the tile
        # candidates and the residual again for the selection; the window's rescored columns (~5 per signal and step at this
        # workload, profiles/r03_batched_traffic.json) are NOT counted: a lower bound -- over everything of the step that is not the screen
        "per_signal_kernels": per_signal_roofline(nsig, k, gram, ms_per_omp_step * 1e3 - (screen_ms / max(screen_n, 1) * 1e3 if screen_n else 0.0)),
        # The step is an MFMA-bound launch and an HBM-bound pass BACK TO BACK (two overlap schemes were measured and lost:
        # profiles/r03_cusplit_experiment.txt, r03_coresident_experiment.txt), so its honest ceiling is the serial composite: the
        # screen's flop at the dense 16-bit peak plus the per-signal kernels' algorithmic bytes at the HBM peak
        "roofline_composite": composite_roofline(flops, MFMA_PEAK_TF * (2 if i8 else 1), per_signal_roofline(nsig, k, gram, 1.0)["algorithmic_bytes_per_step"],
                                                 ms_per_omp_step, nsig),
        "batch_stats": {"resolved_by_exact_path": int(resolved), "uncertain": int(uncertain), "illcond": int(illcond)},
        "options": {"certificate": "rigorous" if cert else "statistical", "resident_gram": bool(gram), "screen_operands": opname,
                    "gram_setup_seconds": gram_seconds,
                    "gram_bytes": (8 * N * N) if gram else 0},
        "matches_exact_path_on_sample": same,
    }


def measure_lone_omp(K, W, B, D, eps):
    """configs[1] as the reference's API shapes it: one csmp_omp call at a time (sweep -> k_qr1 -> k_qr2 per atom, nothing of
    another signal underneath), b handed over as a host vector, results returned to the host."""
    sigs = [B[s].cpu().numpy() for s in range(W + K)]
    for w in range(W):
        D.ctx.omp(sigs[w], K_ATOMS, eps)
    t0 = time.perf_counter()
    atoms = 0
    for s in range(W, W + K):
        i, v, o = D.ctx.omp(sigs[s], K_ATOMS, eps)
        atoms += len(i)
    dt = time.perf_counter() - t0
    us_atom = dt / max(atoms, 1) * 1e6
    return {"metric": "OMP atoms selected/sec at m=4096,n=65536,k=256, one omp(A,b,k) call at a time (no pipelining across signals)",
            "value": atoms / dt, "unit": "atoms/s", "steps": K, "warmup": W, "ms_per_solve": dt / K * 1e3, "us_per_atom": us_atom,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": M * N * 4 / (us_atom * 1e-6) / 1e9,
                         "frac": M * N * 4 / (us_atom * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "ALL-IN: algorithmic bytes per atom / wall time per atom of the whole call (upload of b, sweep, both append "
                                 "stages, kernel boundaries, back substitution, download) -- not a kernel duration"}}


def measure_lone_omp_device(K, W, torch, dev, B, D, eps):
    """configs[1] read literally -- ONE signal, nothing of another signal in flight -- with the inputs resident in HBM when the
    clock starts and the results left there: csmp_omp_batch with one signal per call (device pointers in and out).  The call's
    one synchronisation (the per-signal factorisation flag) stays inside the timed region."""
    idx = torch.full((1, K_ATOMS), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((1, K_ATOMS), dtype=torch.float64, device=dev)
    nnz = torch.zeros(1, dtype=torch.int64, device=dev)
    sig = [B[s:s + 1].contiguous() for s in range(W + K)]
    torch.cuda.synchronize()
    for w in range(W):
        D.ctx.omp_batch_device(sig[w], K_ATOMS, eps, idx, val, nnz)
    D.ctx.sync()
    atoms = 0
    t0 = time.perf_counter()
    for s in range(W, W + K):
        D.ctx.omp_batch_device(sig[s], K_ATOMS, eps, idx, val, nnz)
        D.ctx.sync()
        atoms += int(nnz.item())  # (8 bytes: the count of atoms this solve selected)
    dt = time.perf_counter() - t0
    us_atom = dt / max(atoms, 1) * 1e6
    return {"metric": "OMP atoms selected/sec at m=4096,n=65536,k=256, ONE signal at a time, b and the results resident in HBM",
            "value": atoms / dt, "unit": "atoms/s", "steps": K, "warmup": W, "ms_per_solve": dt / K * 1e3, "us_per_atom": us_atom,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": M * N * 4 / (us_atom * 1e-6) / 1e9,
                         "frac": M * N * 4 / (us_atom * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "ALL-IN: algorithmic bytes per atom / wall time per atom of the whole solve (sweep, both append stages, kernel "
                                 "boundaries, back substitution); no host transfer of b or of the results inside the clock"}}


def measure_f64_dictionary(K, W, cs, torch, dev):
    """The headline workload on a Float64 dictionary of the same bytes -- A 4096 x 32768 Matrix{Float64} (1 GiB), k = 256, three
    signals pipelined: the reference is generic over eltype(A) and its own tests are Float64 (test/matchingpursuit.jl:10-13)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sweep_shapes
    n64 = N // 2
    At = sweep_shapes.make_dictionary(torch, dev, M, n64, torch.float64, seed=SEED_A + 1)
    D = cs.Dictionary(At)
    try:
        eps = D.eps  # eps(Float64)
        B = torch.empty((K + W, M), dtype=torch.float64, device=dev)
        for s in range(K + W):
            g = torch.Generator(device=dev).manual_seed(7_000_003 * s + 29)
            idx0 = torch.randperm(n64, generator=g, device=dev)[:K_ATOMS]
            sign = torch.randint(0, 2, (K_ATOMS,), generator=g, device=dev).to(torch.float64) * 2 - 1
            e = torch.randn(M, generator=g, device=dev, dtype=torch.float64)
            B[s] = (At[idx0] * sign[:, None]).sum(dim=0) + e * (NOISE / e.norm())
        idx = torch.full((K + W, K_ATOMS), -1, dtype=torch.int64, device=dev)
        val = torch.zeros((K + W, K_ATOMS), dtype=torch.float64, device=dev)
        nnz = torch.zeros(K + W, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        if W:
            D.ctx.omp_batch_device(B[:W], K_ATOMS, eps, idx[:W], val[:W], nnz[:W])
        D.ctx.sync()
        D.ctx.profile_enable(8)
        D.ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        D.ctx.omp_batch_device(B[W:], K_ATOMS, eps, idx[W:], val[W:], nnz[W:])
        D.ctx.sync()
        dt = time.perf_counter() - t0
        sweeps, sweep_ms = D.ctx.profile_read(reset=True)
        D.ctx.profile_enable(False)
        bracket = D.ctx.profile_overhead(64)
        atoms = int(nnz[W:].sum().item())
        i1, v1, o1 = D.ctx.omp(B[W].cpu().numpy(), K_ATOMS, eps)  # one call at a time: the same support
        same = bool(np.array_equal(np.sort(idx[W].cpu().numpy()[:len(i1)]), i1))
        us = max(sweep_ms / max(sweeps, 1) - bracket, 0.0) * 1e3
        alg = M * n64 * 8
        return {"metric": "OMP atoms selected/sec on a Float64 dictionary, m=4096,n=32768,k=256 (3 signals pipelined)", "value": atoms / dt,
                "unit": "atoms/s", "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "pipelined_equals_single_call_support": same,
                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": alg / (us * 1e-6) / 1e9 if us else 0.0,
                             "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us else 0.0, "traffic": None, "avg_launch_us": us,
                             "algorithmic_bytes_per_launch": alg, "kernel": "csmp::" + tick_kernel_name(D).replace("float", "double"),
                             "sweep_config": D.ctx.sweep_config()}}
    finally:
        D.close()
        del At
        torch.cuda.empty_cache()


def measure_streamed_omp(cs, torch, dev, At, D, B, eps, k=16, solves=2):
    """SURVEY 8(f-4), second half: the configs[1] dictionary left in HOST memory (CSMP_HOST_STREAMED) -- what a dictionary larger
    than HBM has to do: every sweep reads its M N 4 bytes over the host link.  k atoms per solve (a k = 256 solve would be 256 GiB
    over the link); the support must be the resident dictionary's, atom for atom."""
    import numpy as np
    Ah = np.asfortranarray(At.cpu().numpy().T)  # (M, N) column-major host copy of the same dictionary
    t0 = time.perf_counter()
    Ds = cs.Dictionary(Ah, streamed=True)
    t_map = time.perf_counter() - t0
    sigs = [B[s].cpu().numpy() for s in range(solves + 1)]
    Ds.ctx.omp(sigs[0], 2, eps)  # warm-up: two sweeps
    same = True
    t0 = time.perf_counter()
    atoms = 0
    res = []
    for s in range(1, solves + 1):
        res.append(Ds.ctx.omp(sigs[s], k, eps))
        atoms += len(res[-1][0])
    dt = time.perf_counter() - t0
    for s in range(1, solves + 1):
        i2, v2, o2 = D.ctx.omp(sigs[s], k, eps)
        same = same and np.array_equal(res[s - 1][2], o2) and np.array_equal(res[s - 1][1], v2)
    Ds.close()
    gbs = atoms * M * N * 4 / dt / 1e9
    return {"metric": "OMP atoms selected/sec at m=4096,n=65536 with the dictionary STREAMED from host memory (CSMP_HOST_STREAMED), k=%d" % k,
            "value": atoms / dt, "unit": "atoms/s", "steps": solves, "warmup": 1, "ms_per_solve": dt / solves * 1e3,
            "identical_to_resident_dictionary": bool(same), "map_seconds": t_map,
            "roofline": {"bound": "host link (PCIe 5 x16)", "unit": "GB/s", "peak": 64.0, "achieved": gbs, "frac": gbs / 64.0, "traffic": None,
                         "note": "ALL-IN: algorithmic bytes per atom (the dictionary once) / wall time per atom; peak = the link's raw 64 GB/s per direction"}}


def measure_screened_omp(K, W, torch, dev, At, D, eps, cert=1, image=3):
    """configs[1] with the screened sweep (CSMP_OPT_SCREENED_SWEEP): every sweep reads the bf16 image (M N 2 bytes) and the
    pick is certified against the f32 dictionary in Float64, an uncertified solve repeated exactly -- the results are the exact
    path's, and this function checks that on every timed signal.  Two forms: one csmp_omp call at a time, and csmp_omp_batch
    (up to three solves in flight, one sweep apart).  The headline stays the exact path: this one depends on the certificate holding
    (it does on these dictionaries: `fallbacks`), which is a property of the data."""
    import numpy as np
    B = make_signals(torch, dev, At, 500, K + W)
    sigs = [B[s].cpu().numpy() for s in range(W + K)]
    D.ctx.set_option("screened_sweep", 0)
    exact = [D.ctx.omp(sigs[s], K_ATOMS, eps) for s in range(W, W + K)]
    if image == 2:
        cert = 0  # (the int8 image has the statistical bound only)
    D.ctx.set_option("batch_cert", cert)
    D.ctx.set_option("screened_sweep", image)  # 3: binary16 image, 1: bf16 image, 2: int8 image
    iname, ibytes = {1: ("bf16", 2), 2: ("int8", 1), 3: ("f16", 2)}[image]
    out = {"metric": "OMP atoms selected/sec at m=4096,n=65536,k=256, screened sweep (%s image, certified picks, exact results)" % iname,
           "unit": "atoms/s", "certificate": "rigorous" if cert else "statistical", "image": iname, "steps": K, "warmup": W}
    try:
        for w in range(W):
            D.ctx.omp(sigs[w], K_ATOMS, eps)
        D.ctx.screened_stats(reset=True)
        t0 = time.perf_counter()
        got = [D.ctx.omp(sigs[s], K_ATOMS, eps) for s in range(W, W + K)]
        dt = time.perf_counter() - t0
        atoms = sum(len(g[0]) for g in got)
        same = all(np.array_equal(g[0], e[0]) and np.array_equal(g[2], e[2]) and np.allclose(g[1], e[1], rtol=1e-9, atol=1e-12)
                   for g, e in zip(got, exact))
        us_atom = dt / max(atoms, 1) * 1e6
        out["lone"] = {"value": atoms / dt, "us_per_atom": us_atom, "ms_per_solve": dt / K * 1e3, "equals_exact_path": bool(same),
                       "stats": D.ctx.screened_stats(reset=True)}
        idx = torch.full((K + W, K_ATOMS), -1, dtype=torch.int64, device=dev)
        val = torch.zeros((K + W, K_ATOMS), dtype=torch.float64, device=dev)
        nnz = torch.zeros(K + W, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        if W > 0:
            D.ctx.omp_batch_device(B[:W], K_ATOMS, eps, idx[:W], val[:W], nnz[:W])
        D.ctx.sync()
        D.ctx.screened_stats(reset=True)
        D.ctx.profile_enable(16)
        D.ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        D.ctx.omp_batch_device(B[W:], K_ATOMS, eps, idx[W:], val[W:], nnz[W:])
        D.ctx.sync()
        dt = time.perf_counter() - t0
        sweeps, sweep_ms = D.ctx.profile_read(reset=True)
        D.ctx.profile_enable(False)
        atoms = int(nnz[W:].sum().item())
        same = all(int(nnz[W + s]) == len(exact[s][0]) and np.array_equal(idx[W + s, :len(exact[s][0])].cpu().numpy(), exact[s][0])
                   for s in range(K))
        avg = sweep_ms / max(sweeps, 1) / 1e3
        out["value"] = atoms / dt
        out["ms_per_step"] = dt / K * 1e3
        out["batch"] = {"value": atoms / dt, "us_per_atom": dt / max(atoms, 1) * 1e6, "equals_exact_path": bool(same),
                        "signals_in_flight": min(3, D.ctx.get_option("solves_in_flight"), K), "stats": D.ctx.screened_stats(reset=True)}
        alg = M * N * ibytes  # the image, streamed once per atom
        out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": alg / avg / 1e9 if sweeps else 0.0,
                           "frac": alg / avg / 1e9 / HBM_PEAK_GBS if sweeps else 0.0, "traffic": None,
                           "kernel": "csmp::k_sweep_%s<2,3,true> (while the other solve's pick / append stages run beside it)" % {1: "bf16", 2: "i8", 3: "f16"}[image],
                           "launches_timed": int(sweeps), "avg_launch_us": avg * 1e6, "algorithmic_bytes_per_launch": alg,
                           "f32_equivalent_frac_all_in": M * N * 4 / (dt / max(atoms, 1)) / 1e9 / HBM_PEAK_GBS,
                           "note": "algorithmic bytes of THIS path are M N x the image's element size; f32_equivalent_frac_all_in prices the whole "
                                   "batch's time per atom against the exact path's M N 4 bytes -- above 1 means faster than any exact sweep can be"}
    finally:
        D.ctx.set_option("screened_sweep", 0)
        D.ctx.set_option("batch_cert", 1)
    return out


def measure_config5(workload, K, W, cs, torch, dev, D5=None, At5=None, delta=1e-2, screened=False):
    """configs[4]: GOMP (S = 4 atoms per step) or Subspace Pursuit on A 8192 x 131072 Float32, k = 512.
    A step = one complete solve.  delta: sp's residual tolerance (the reference's default is 1e-12, src/twostage.jl:87: with
    noisy data it keeps iterating until the residual stops decreasing; 1e-2 stops after the first update!).
    screened (gomp workloads): CSMP_OPT_SCREENED_SWEEP -- sweeps over the bf16 image, the top-S pick certified in Float64
    (identical results; the line carries the fallback count and a check of the first timed solve against the exact path)."""
    M5, N5, k, S = 8192, 131072, 512, 4
    own = D5 is None
    if own:
        At5, D5 = make_dictionary5(cs, torch, dev)
    sigs = []
    for s_ in range(K + W):
        gs = torch.Generator(device=dev).manual_seed(99 + s_)
        idx = torch.randperm(N5, generator=gs, device=dev)[:k]
        sign = torch.randint(0, 2, (k,), generator=gs, device=dev).to(torch.float64) * 2 - 1
        e = torch.randn(M5, generator=gs, device=dev, dtype=torch.float64)
        sigs.append(((At5[idx].to(torch.float64) * sign[:, None]).sum(0) + e * (NOISE / e.norm())).cpu().numpy())
    torch.cuda.synchronize()
    eps = D5.eps
    screened = int(screened)
    isgw = workload in ("gomp", "gomp_single")
    if screened:
        exact0 = D5.ctx.gomp(sigs[W], S, k, eps) if isgw else D5.ctx.sp(sigs[W], k, delta)
        D5.ctx.set_option("batch_cert", 1 if int(screened) == 3 else 0)  # binary16 image: the rigorous certificate (the default); bf16 / int8: statistical
        D5.ctx.set_option("screened_sweep", int(screened))  # 3: binary16 image, 1: bf16 image, 2: int8 image
        D5.ctx.screened_stats(reset=True)

    def solve(b):
        if workload in ("gomp", "gomp_single"):
            i, v, o = D5.ctx.gomp(b, S, k, eps)
            return len(i), 0
        i, v, it = D5.ctx.sp(b, k, delta)
        return len(i), it
    if workload == "gomp":
        # the caller's loop over signals as ONE call: csmp_gomp_batch keeps two solves in flight (a signal's short stages under
        # the other signal's sweep); signals resident in HBM, results left on the device
        Bd = torch.stack([torch.from_numpy(x) for x in sigs]).to(dev)
        oi = torch.full((K + W, k), -1, dtype=torch.int64, device=dev)
        ov = torch.zeros((K + W, k), dtype=torch.float64, device=dev)
        on = torch.zeros(K + W, dtype=torch.int64, device=dev)
        if W:
            D5.ctx.gomp_batch_device(Bd[:W].contiguous(), S, k, eps, oi[:W], ov[:W], on[:W])
        D5.ctx.sync()
        torch.cuda.synchronize()
        D5.ctx.profile_enable(True)
        D5.ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        D5.ctx.gomp_batch_device(Bd[W:].contiguous(), S, k, eps, oi[W:], ov[W:], on[W:])
        D5.ctx.sync()
        dt = time.perf_counter() - t0
        atoms, iters = int(on[W:].sum().item()), 0
        sweeps, sweep_ms = D5.ctx.profile_read(reset=True)
        D5.ctx.profile_enable(False)
    elif workload == "sp":
        # the caller's loop over signals as ONE call: csmp_sp_batch keeps several solves in flight (contexts on their own streams
        # and host threads); the signals are handed over as a host matrix, as csmp_sp takes its b
        import numpy as np
        Bh = np.asfortranarray(np.stack(sigs, axis=1))
        if W:
            D5.ctx.sp_batch(Bh[:, :W], k, delta)
        D5.ctx.profile_enable(True)
        D5.ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        bi, bv, bn, bits = D5.ctx.sp_batch(Bh[:, W:], k, delta)
        dt = time.perf_counter() - t0
        atoms, iters = int(bn.sum()), int(bits.sum())
        sweeps, sweep_ms = D5.ctx.profile_read(reset=True)
        D5.ctx.profile_enable(False)
    else:
        for w in range(W):
            solve(sigs[w])
        D5.ctx.profile_enable(True)
        D5.ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        atoms, iters = 0, 0
        for s_ in range(W, W + K):
            n, it = solve(sigs[s_])
            atoms += n
            iters += it
        dt = time.perf_counter() - t0
        sweeps, sweep_ms = D5.ctx.profile_read(reset=True)
        D5.ctx.profile_enable(False)
    alg = M5 * N5 * ({1: 2, 2: 1, 3: 2}[int(screened)] if screened else 4)
    avg = sweep_ms / max(sweeps, 1) / 1e3
    isg = workload in ("gomp", "gomp_single")
    out = {"metric": ("GOMP (S=4) atoms selected/sec" + ((", three solves in flight (csmp_gomp_batch)" if screened else ", two solves in flight (csmp_gomp_batch)") if workload == "gomp" else ", one gomp call at a time")
                      if isg else "Subspace Pursuit solves/sec" + (", several solves in flight (csmp_sp_batch)" if workload == "sp" else ", one sp call at a time"))
           + " at m=8192,n=131072,k=512",
           "value": (atoms / dt) if isg else K / dt, "unit": "atoms/s" if isg else "solves/s",
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64 (f32 dictionary, Float64 accumulate/QR)", "data": "synthetic",
           "config": {"workload": f"configs[4]: {workload} on A 8192x131072 Float32 Gaussian unit-norm, k=512" + (", S=4" if isg else f", delta={delta:g}"),
                      "sweeps": int(sweeps), "sp_update_calls": int(iters), "sp_update_calls_per_solve": iters / K if workload in ("sp", "sp_single") else None,
                      "signals_in_flight": (min(3, D5.ctx.get_option("solves_in_flight")) if screened else 2) if workload == "gomp"
                      else (D5.ctx.get_option("solves_in_flight") if workload == "sp" else 1)},
           "roofline": {"bound": "hbm", "achieved": alg / avg / 1e9 if sweeps else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (alg / avg / 1e9 / HBM_PEAK_GBS) if sweeps else 0.0, "traffic": None, "kernel": "csmp::k_sweep_gen<float,16,2,false>",
                        "launches_timed": int(sweeps), "avg_launch_us": avg * 1e6, "algorithmic_bytes_per_launch": alg}}
    if isg:  # the whole solve against the same roofline: one dictionary pass per S atoms is all the algorithm needs
        out["roofline"]["whole_solve"] = {"achieved": alg / S * atoms / dt / 1e9, "frac": alg / S * atoms / dt / 1e9 / HBM_PEAK_GBS,
                                          "note": "ALL-IN: M*N*%d bytes per S atoms / wall time per atom" % ({1: 2, 2: 1, 3: 2}[int(screened)] if screened else 4)}
    if screened:
        import numpy as np
        st_ = D5.ctx.screened_stats(reset=True)
        got0 = D5.ctx.gomp(sigs[W], S, k, eps) if isgw else D5.ctx.sp(sigs[W], k, delta)
        D5.ctx.screened_stats(reset=True)
        D5.ctx.set_option("screened_sweep", 0)
        D5.ctx.set_option("batch_cert", 1)
        if not isgw:  # sp returns (idx, val, update! calls): the count has to agree too
            exact0 = (exact0[0], exact0[1], np.asarray([exact0[2]]))
            got0 = (got0[0], got0[1], np.asarray([got0[2]]))
        iname5 = {1: "bf16", 2: "int8", 3: "f16"}[int(screened)]
        out["metric"] += ", screened sweep (%s image, %s certificate, certified top-%s, exact results)" % (iname5, "rigorous" if int(screened) == 3 else "statistical", "S picks" if isgw else "k sets")
        out["roofline"]["kernel"] = "csmp::k_sweep_i8<2,3,true> (M*N bytes per sweep)" if int(screened) == 2 else "csmp::k_sweep_%s<2,3,true> (M*N*2 bytes per sweep)" % iname5
        out["screened"] = {"stats": st_, "first_timed_solve_equals_exact_path": bool(
            np.array_equal(got0[0], exact0[0]) and np.array_equal(got0[2], exact0[2]) and np.allclose(got0[1], exact0[1], rtol=1e-9, atol=1e-12)),
            "f32_equivalent_frac_all_in": (M5 * N5 * 4 / S * atoms / dt / 1e9 / HBM_PEAK_GBS) if isgw else None,
            "stats_count": "solves" if isgw else "acquisitions (one selection each: the first and one per update!)"}
    if workload == "gomp":
        # two sweeps share the HBM most of the time: a launch bracketed by HIP events on ONE stream takes about twice as long as
        # the kernel alone, so the per-launch figure says nothing here -- the block's achieved / frac are the all-in ones
        r = out["roofline"]
        r["sweep_launch_while_sharing_the_gpu"] = {"avg_launch_us": r["avg_launch_us"], "achieved": r["achieved"], "frac": r["frac"]}
        r["achieved"], r["frac"] = r["whole_solve"]["achieved"], r["whole_solve"]["frac"]
        r["note"] = ("achieved / frac = ALL-IN (M*N*%d bytes per S atoms / wall time per atom): with two solves in flight the sweeps of the two "
                     "streams overlap and a per-launch duration measures the sharing, not the kernel (gomp_c5_single has the kernel alone)"
                     % ({1: 2, 2: 1, 3: 2}[int(screened)] if screened else 4))
    if workload == "sp":
        r = out["roofline"]  # (as for gomp: overlapping solves share the HBM, a per-launch duration measures the sharing)
        r["sweep_launch_while_sharing_the_gpu"] = {"avg_launch_us": r["avg_launch_us"], "achieved": r["achieved"], "frac": r["frac"]}
        nsweep = K + iters  # one sweep per acquisition: the first one and one per update!
        r["achieved"] = alg * nsweep / dt / 1e9
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
        r["note"] = ("achieved / frac = ALL-IN: M*N*4 bytes per acquisition sweep x sweeps / wall time of the batch -- the share of the HBM roofline "
                     "the whole solves reach; sp_c5_single has the sweep kernel alone")
    if workload == "sp_single":
        # the factorisations beside the sweeps: the first acquisition factorises k columns, every update! 2k and then k
        # (src/twostage.jl:74-83,104-107); thin QR of n columns = 2 M n^2 flop.  Time = the solves minus their sweeps.
        flop = 2.0 * M5 * (K * k * k + iters * ((2 * k) ** 2 + k * k))
        rest = max(dt - sweep_ms / 1e3, 1e-9)
        out["factorisation"] = {"flop": flop, "seconds_outside_sweeps": rest, "achieved_tflops_f64": flop / rest / 1e12,
                                "note": "everything of a solve that is not a dictionary sweep: panel appends, top-k selection, back substitutions, host loop"}
    if own:
        D5.close()
    return out


def make_dictionary5(cs, torch, dev):
    M5, N5 = 8192, 131072
    g = torch.Generator(device=dev).manual_seed(SEED_A + 5)
    At5 = torch.empty((N5, M5), dtype=torch.float32, device=dev)
    for lo in range(0, N5, 8192):
        a = torch.randn((8192, M5), generator=g, device=dev, dtype=torch.float64)
        a -= 1e-6 * a.mean(dim=1, keepdim=True)
        a /= a.norm(dim=1, keepdim=True)
        At5[lo:lo + 8192] = a.to(torch.float32)
    return At5, cs.Dictionary(At5, device=dev.index)


def run_twostage(args, cs, torch, dev, At, D, show=True):
    """ompr / srr (src/twostage.jl) at the configs[1] shape, k = 256.  The signals carry 8 planted atoms more
    than the solvers may keep and noise 0.3, so that the replacement loops have work to do (tens of
    iterations); one step = one complete solve."""
    import numpy as np
    K, W = args.steps, args.warmup
    sigs = []
    for s_ in range(K + W):
        g = torch.Generator(device=dev).manual_seed(31_337 + s_)
        idx = torch.randperm(N, generator=g, device=dev)[:K_ATOMS + 8]
        sign = torch.randint(0, 2, (K_ATOMS + 8,), generator=g, device=dev).to(torch.float64) * 2 - 1
        e = torch.randn(M, generator=g, device=dev, dtype=torch.float64)
        sigs.append(((At[idx].to(torch.float64) * sign[:, None]).sum(0) + e * (0.3 / e.norm())).cpu().numpy())
    torch.cuda.synchronize()

    def solve(b):
        if args.workload == "ompr":
            return D.ctx.ompr(b, K_ATOMS, 1e-6)[2]
        return D.ctx.srr(b, K_ATOMS, 1e-12, -1, 1, 1)[2]
    # ompr with the screened sweep (--screened): the result of the first timed signal is compared with the exact path's
    image = 0
    if args.workload == "ompr" and getattr(args, "screened", False):
        image = {"f16": 3, "bf16": 1, "int8": 2}[getattr(args, "screen_image", "f16")]
        exact0 = D.ctx.ompr(sigs[W], K_ATOMS, 1e-6)
        D.ctx.set_option("batch_cert", 1 if image == 3 else 0)
        D.ctx.set_option("screened_sweep", image)
        D.ctx.screened_stats(reset=True)
    for w in range(W):
        solve(sigs[w])
    D.ctx.profile_enable(1)
    D.ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    iters = 0
    for s_ in range(W, W + K):
        iters += solve(sigs[s_])
    dt = time.perf_counter() - t0
    sweeps, sweep_ms = D.ctx.profile_read(reset=True)
    alg = M * N * ({0: 4, 1: 2, 2: 1, 3: 2}[image])
    avg = sweep_ms / max(sweeps, 1) / 1e3
    scr_info = None
    if image:
        st_ = D.ctx.screened_stats(reset=True)
        got0 = D.ctx.ompr(sigs[W], K_ATOMS, 1e-6)
        D.ctx.set_option("screened_sweep", 0)
        D.ctx.set_option("batch_cert", 1)
        scr_info = {"image": {1: "bf16", 2: "int8", 3: "f16"}[image], "stats": st_, "stats_count": "sweeps (one certified selection each)",
                    "first_timed_solve_equals_exact_path": bool(np.array_equal(got0[0], exact0[0]) and got0[2] == exact0[2]
                                                                and np.allclose(got0[1], exact0[1], rtol=1e-9, atol=1e-12))}
    name = {"ompr": "OMP with replacement", "srr": "stepwise regression with replacement (oblivious start, l=1)"}[args.workload]
    out = {"metric": f"{name} solves/sec at m=4096,n=65536,k=256", "value": K / dt, "unit": "solves/s",
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64 (f32 dictionary, Float64 accumulate/QR)", "data": "synthetic",
           "config": {"workload": f"SURVEY 8(f)2: {args.workload} on A 4096x65536 Float32 Gaussian unit-norm, k=256, 264 planted atoms, noise 0.3",
                      "iterations": int(iters), "iterations_per_s": iters / dt, "sweeps_timed": int(sweeps)},
           "roofline": {"bound": "hbm", "achieved": alg / avg / 1e9 if sweeps else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (alg / avg / 1e9 / HBM_PEAK_GBS) if sweeps else 0.0, "traffic": None,
                        "kernel": ("csmp::k_sweep_i8<2,3,true>" if image == 2 else "csmp::k_sweep_bf16<2,3,true>" if image == 1 else "csmp::k_sweep_f16<2,3,true>" if image == 3 else "csmp::k_sweep_gen<float,16,2,false>")
                        if args.workload == "ompr" else "csmp::k_fr_sweep<float,16,true,NQ> (NQ = 2 in the loop, 1 / -1 at the start)",
                        "launches_timed": int(sweeps), "avg_launch_us": avg * 1e6, "algorithmic_bytes_per_launch": alg}}
    D.ctx.profile_enable(False)
    if scr_info:
        out["metric"] += ", screened sweep (%s image, certified selections, exact results)" % scr_info["image"]
        out["screened"] = scr_info
    # the same solves, three in flight (api.solve_in_flight: the Dictionary's context + two clones, a host thread each): one
    # signal's long latency-bound chain (rank-one exchanges, selections, host decisions) under the others' sweeps
    def one(c, b):
        return c.ompr(b, K_ATOMS, 1e-6)[2] if args.workload == "ompr" else c.srr(b, K_ATOMS, 1e-12, -1, 1, 1)[2]
    if not getattr(args, "no_in_flight", False):
        many = [sigs[W + (i % K)] for i in range(3 * K)]
        cs.solve_in_flight(D, many[:3], one, in_flight=3)  # (the clones' first call allocates their solver slots)
        t0 = time.perf_counter()
        its = cs.solve_in_flight(D, many, one, in_flight=3)
        dt3 = time.perf_counter() - t0
        out["three_in_flight"] = {"solves": len(many), "ms_per_solve": dt3 / len(many) * 1e3, "solves_per_s": len(many) / dt3,
                                  "iterations_equal_single": bool(sum(its[:K]) == iters)}
    if show:
        emit(out)
    return out


def run_fr(args, cs, torch, dev, At, D, show=True):
    """Forward regression / OLS (src/forward.jl) at the configs[1] shape, k = 256 atoms per signal.  One step = one
    complete fr(A, b, sparsity=256) solve; the signals are resident in HBM and go through csmp_fr_batch (three in
    flight per tick kernel, like the default workload)."""
    K, W = args.steps, args.warmup
    Bsig = make_signals(torch, dev, At, 5000, K + W)
    idx = torch.empty((K + W, K_ATOMS), dtype=torch.int64, device=dev)
    val = torch.empty((K + W, K_ATOMS), dtype=torch.float64, device=dev)
    nnz = torch.empty((K + W,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if W:
        D.ctx.fr_batch_device(Bsig[:W], K_ATOMS, 0.0, 0.0, idx[:W], val[:W], nnz[:W])
        D.ctx.sync()
    D.ctx.profile_enable(args.profile_every)
    D.ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    D.ctx.fr_batch_device(Bsig[W:], K_ATOMS, 0.0, 0.0, idx[W:], val[W:], nnz[W:])
    D.ctx.sync()
    dt = time.perf_counter() - t0
    atoms = int(nnz[W:].sum().item())
    window = D.ctx.profile_window()
    sweeps, sweep_ms = D.ctx.profile_read(reset=True)
    alg = M * N * 4
    duration = sweep_ms / max(sweeps, 1) / 1e3  # one launch, start to end
    two = window["streams"] == 2 and window["launches"] > 0
    # two pipelines side by side: their launches overlap and are priced as ONE window, as in the default workload
    avg = window["window_ms"] / 1e3 / window["launches"] if two else duration
    out = {"metric": "forward regression (OLS) atoms selected/sec at m=4096,n=65536,k=256", "value": atoms / dt, "unit": "atoms/s",
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64 (f32 dictionary, Float64 accumulate/QR)", "data": "synthetic",
           "config": {"workload": "SURVEY 8(f)3: fr/ols on A 4096x65536 Float32 Gaussian unit-norm, k=256, signals resident in HBM, "
                                  "2 x 3 in flight (csmp_fr_batch)" if two else "three in flight (csmp_fr_batch)", "launches_timed": int(sweeps)},
           "roofline": {"bound": "hbm", "achieved": alg / avg / 1e9 if sweeps else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (alg / avg / 1e9 / HBM_PEAK_GBS) if sweeps else 0.0, "traffic": None,
                        "kernel": "csmp::k_tick_fr<float,8,1> = forward-regression sweep of one signal (c = A'r and the OLS rescaling in one "
                                  "dictionary pass) fused with the two short append stages of two other signals",
                        "launches_timed": int(window["launches"] if two else sweeps), "avg_launch_us": avg * 1e6, "algorithmic_bytes_per_launch": alg,
                        "launch_duration_us": duration * 1e6, "launches_in_flight": 2 if two else 1}}
    D.ctx.profile_enable(False)
    if show:
        emit(out)
    return out


def colsharded_solves(K, W, sigs, k, eps, rank, world, group, make_shard, barrier, all_supports):
    """The measured part of the column-sharded workload, free of GPU specifics (the gloo CPU test drives it with numpy
    shards): W + K solves of omp_colsharded, the K timed ones between two barriers; then every rank's support of the first
    timed signal is compared (all ranks run the same replicated append chain: they must agree bit for bit).
    make_shard() -> this rank's shard object; all_supports(idx) -> list of every rank's idx array."""
    from csmp_pkg import load
    cs = load()
    shard = make_shard()
    first = None
    for w in range(W):
        cs.omp_colsharded(shard, sigs[w], k, eps, group=group)
    barrier()
    t0 = time.perf_counter()
    atoms = 0
    for s_ in range(W, W + K):
        idx, val, order = cs.omp_colsharded(shard, sigs[s_], k, eps, group=group)
        atoms += len(idx)
        if first is None:
            first = (idx, val, order)
    barrier()
    dt = time.perf_counter() - t0
    import numpy as np
    pad = -np.ones(k, np.int64)
    pad[:len(first[0])] = first[0]
    sup = all_supports(pad)
    agree = all(np.array_equal(sup[0], x) for x in sup)
    return {"seconds": dt, "atoms": atoms, "first": first, "ranks_agree_on_first_support": bool(agree), "supports_gathered": len(sup)}


def run_colsharded(args, cs, torch, dist, dev, rank, world, use_dist, ranks_seen, devices):
    """SURVEY 8(f)4 / 8(e) "natural": ONE signal at a time, the dictionary's COLUMNS sharded over the ranks -- the only way
    several GPUs speed up a single omp(A, b, k) (src/matchingpursuit.jl:73-82).  Every rank holds N / world columns and a
    replica of the solver state; one all_gather of one 16-KiB record per rank and step (sharded.omp_colsharded)."""
    import numpy as np
    K, W = (args.steps, args.warmup) if (args.steps, args.warmup) != (18, 3) else (3, 1)
    At = make_dictionary(torch, dev)
    B = make_signals(torch, dev, At, 0, K + W)  # the SAME signals on every rank
    sigs = [B[s].cpu().numpy() for s in range(K + W)]
    lo, hi = cs.column_range(N, rank, world)
    Aloc = At[lo:hi].contiguous() if world > 1 else At
    D = cs.Dictionary(Aloc, device=dev.index)
    eps = D.eps
    ref = None
    if world == 1:  # the whole dictionary is here: the unsharded call is the comparison
        ref = D.ctx.omp(sigs[W], K_ATOMS, eps)
    del At
    group = None

    def barrier():
        D.ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def all_supports(idx):
        if not use_dist:
            return [idx]
        t = torch.from_numpy(idx) if dist.get_backend() == "gloo" else torch.from_numpy(idx).to(dev)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.cpu().numpy() for o in out]

    res = colsharded_solves(K, W, sigs, K_ATOMS, eps, rank, world, group, lambda: cs.HipColumnShard(D.ctx, lo, dev), barrier, all_supports)
    tmax = torch.tensor([res["seconds"]], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = tmax.item()
    out = None
    if rank == 0:
        us_atom = tmax / max(res["atoms"], 1) * 1e6
        alg = M * N * 4
        out = {"metric": "column-sharded OMP atoms selected/sec at m=4096,n=65536,k=256 (ONE signal, columns over the GPUs)",
               "value": res["atoms"] / tmax, "unit": "atoms/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": tmax / K * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 (f32 dictionary, Float64 accumulate/QR)",
               "data": "synthetic", "us_per_atom": us_atom,
               "config": {"workload": "SURVEY 8(f)4: single-signal OMP, A 4096x65536 Float32 Gaussian unit-norm, k=256, columns sharded",
                          "columns_per_gpu": int(hi - lo), "collective": "one all_gather_into_tensor of one record (32 B + M*4 B) per rank and step",
                          "sharding": f"columns over {world} GPU(s), solver state replicated"},
               "ranks_seen": ranks_seen, "devices": devices,
               "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS * world, "achieved": alg / (us_atom * 1e-6) / 1e9,
                            "frac": alg / (us_atom * 1e-6) / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                            "note": "ALL-IN per atom (sweep of N/world columns + record exchange + replicated append) against the sum of the ranks' HBM peaks"},
               "ranks_agree_on_first_support": res["ranks_agree_on_first_support"], "supports_gathered": res["supports_gathered"]}
        if ref is not None:
            out["equals_unsharded_omp"] = bool(np.array_equal(ref[0], res["first"][0]) and np.array_equal(ref[2], res["first"][2]))
        if not res["ranks_agree_on_first_support"] or res["supports_gathered"] != world:
            out["error"] = "ranks disagree on the support of the first timed signal"
        emit(out)
    D.close()
    return out


def run_shapes(args, cs, torch, np, dev):
    """The product sweep c = A'r over dictionaries of ~1 GiB with M = 256 .. 32768 rows, Float32 and Float64 (the reference is
    generic over shape and element type, src/matchingpursuit.jl:54-60; its own tests are Float64): every row checked against torch's
    Float64 product, timed with HIP events (csmp_bench_sweep, median of five runs), priced against 8 TB/s.  Writes the table to
    profiles/r06_sweep_shapes.json when run from the repository (tools/sweep_shapes.py holds the loop)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sweep_shapes
    rows = sweep_shapes.table(torch, np, cs, dev, reps=max(5, min(args.steps, 20)))
    worst = min(rows, key=lambda r: r["frac"])
    out = {"metric": "product sweep A'r: worst fraction of the HBM roofline over %d shapes (M = 256..32768, f32 and f64, ~1 GiB each)" % len(rows),
           "value": worst["frac"], "unit": "fraction of 8 TB/s", "n_gpus": 1, "steps": len(rows), "warmup": 0,
           "ms_per_step": sum(r["us"] for r in rows) / len(rows) / 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "config": {"workload": "shape table of the A'r sweep (verdict round 4 item 1, round 5 item 2)"},
           "roofline": {"bound": "hbm", "achieved": worst["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": worst["frac"], "traffic": None,
                        "kernel": "csmp::k_sweep_gen (worst row: M = %d %s)" % (worst["M"], worst["dtype"])},
           "all_correct": all(r["argmax_ok"] and r["max_rel_err"] < 1e-12 for r in rows), "rows": rows}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_sweep_shapes.json"), "w") as f:
            json.dump(rows, f, indent=1)
    except OSError:
        pass
    return out


def tick_kernel_name(D):
    """the symbol of the steady-state tick for this dictionary, as rocprofv3 prints it"""
    c = D.ctx.sweep_config()
    return "k_tick<float, %d, %s, true, %s>" % (c["unit_loads"], "true" if c["phases"] > 1 else "false", "true" if c.get("dynamic") else "false")


def rank_census(torch, dist, dev, use_dist, world):
    """ranks_seen = the process group's own world size; devices = every rank's GPU as IT names it, gathered."""
    name = torch.cuda.get_device_name(dev) + f" (cuda:{dev.index})"
    if not use_dist:
        return 1, [name]
    names = [None] * world
    dist.all_gather_object(names, name)
    return dist.get_world_size(), names


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not under a launcher: start one (child processes; this process never touches the GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    import numpy as np
    import torch
    import torch.distributed as dist
    from csmp_pkg import load
    cs = load()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()
    if local >= torch.cuda.device_count():  # (device_count() does not initialise the GPU)
        print(f"bench.py: rank {rank} wants GPU {local} but this node shows {torch.cuda.device_count()} GPU(s): one process per GPU, "
              f"--gpus must not exceed the visible devices", file=sys.stderr, flush=True)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    if use_dist:
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ranks_seen, devices = rank_census(torch, dist, dev, use_dist, world)

    def finish():
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()

    if args.workload == "colsharded":
        out = run_colsharded(args, cs, torch, dist, dev, rank, world, use_dist, ranks_seen, devices)
        finish()
        if rank == 0 and out and "error" in out:
            sys.exit(3)
        return
    if args.workload in ("gomp", "sp", "gomp_single", "sp_single"):
        if args.steps == 18 and args.warmup == 3:
            args.steps, args.warmup = (9, 3) if args.workload in ("gomp", "sp") else (3, 1)
        if rank == 0:
            At5, D5 = make_dictionary5(cs, torch, dev)
            for kv in filter(None, args.tune.split(",")):
                D5.ctx.tune(kv.split("=")[0], int(kv.split("=")[1]))
            if args.in_flight:
                D5.ctx.set_option("solves_in_flight", args.in_flight)
            emit(measure_config5(args.workload, args.steps, args.warmup, cs, torch, dev, D5, At5,
                                 screened={"f16": 3, "bf16": 1, "int8": 2}[args.screen_image] if args.screened else 0))
            D5.close()
        return finish()
    if args.workload == "shapes":
        if rank == 0:
            emit(run_shapes(args, cs, torch, np, dev))
        return finish()
    At = make_dictionary(torch, dev)
    D = cs.Dictionary(At, device=local)  # borrowed, zero-copy
    for kv in filter(None, args.tune.split(",")):
        D.ctx.tune(kv.split("=")[0], int(kv.split("=")[1]))
    if args.workload in ("fr", "ompr", "srr"):
        if args.steps == 18 and args.warmup == 3:
            args.steps, args.warmup = (9, 3) if args.workload == "fr" else (6, 1)
        if rank == 0:
            (run_fr if args.workload == "fr" else run_twostage)(args, cs, torch, dev, At, D)
        D.close()
        return finish()
    if args.workload == "screened":
        if args.steps == 18 and args.warmup == 3:
            args.steps, args.warmup = 9, 3
        if rank == 0:
            emit(measure_screened_omp(args.steps, args.warmup, torch, dev, At, D, D.eps,
                                      cert=1 if args.batch_cert == "rigorous" else 0, image={"f16": 3, "bf16": 1, "int8": 2}[args.screen_image]))
        D.close()
        return finish()
    if args.workload == "streamed":
        if rank == 0:
            Bs = make_signals(torch, dev, At, 0, 4)
            out = measure_streamed_omp(cs, torch, dev, At, D, Bs, D.eps, solves=3 if args.steps == 18 else max(1, min(args.steps, 3)))
            out.update({"n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "ms_per_step": out["ms_per_solve"], "config": {"workload": "configs[1] dictionary (4096x65536 Float32, 1 GiB) in host memory, k=16"}})
            emit(out)
        D.close()
        return finish()
    if args.workload == "batched":
        if args.steps == 18 and args.warmup == 3:
            args.steps, args.warmup = 3, 1
        out = measure_batched(args.steps, args.warmup, cs, torch, dist, dev, rank, world, At, D, use_dist,
                              cert=1 if args.batch_cert == "rigorous" else 0, gram=1 if args.batch_gram else 0,
                              screen={"f16": 3, "bf16": 0, "int8": 1}[args.batch_screen])
        if rank == 0:
            out["ranks_seen"], out["devices"] = ranks_seen, devices
            emit(out)
        D.close()
        return finish()
    eps = D.eps  # eps(Float32): omp(A, b, k) default (src/matchingpursuit.jl:85)
    K, W = args.steps, args.warmup
    B = make_signals(torch, dev, At, rank * (K + W), K + W)
    idx = torch.full((K + W, K_ATOMS), -1, dtype=torch.int64, device=dev)
    val = torch.zeros((K + W, K_ATOMS), dtype=torch.float64, device=dev)
    nnz = torch.zeros(K + W, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # inputs are resident before the library's stream touches them

    def barrier():
        D.ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    lib_gather = library_collective(cs, dist, D, use_dist, args.share_gpu)
    if lib_gather:  # every rank's results of all world * K timed signals, on the device, in global order
        g_idx = torch.full((world * K, K_ATOMS), -1, dtype=torch.int64, device=dev)
        g_val = torch.zeros((world * K, K_ATOMS), dtype=torch.float64, device=dev)
        g_nnz = torch.zeros(world * K, dtype=torch.int64, device=dev)
    if W > 0:
        D.ctx.omp_batch_device(B[:W], K_ATOMS, eps, idx[:W], val[:W], nnz[:W])
    if lib_gather:  # warm the collective too (one signal per rank through the same call)
        D.ctx.omp_sharded_device(B[W:W + 1].contiguous(), world, K_ATOMS, eps, g_idx[:world], g_val[:world], g_nnz[:world])
    elif use_dist:
        exchange_results(idx[W:], val[W:], nnz[W:])
    D.ctx.profile_enable(args.profile_every)  # HIP events around every n-th sweep launch of the timed region
    D.ctx.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    if lib_gather:  # the block's solves + the single collective of the path, inside the library (csmp_omp_sharded)
        D.ctx.omp_sharded_device(B[W:], world * K, K_ATOMS, eps, g_idx, g_val, g_nnz)
        idx[W:], val[W:], nnz[W:] = g_idx[rank * K:(rank + 1) * K], g_val[rank * K:(rank + 1) * K], g_nnz[rank * K:(rank + 1) * K]
        gathered = cs.pack_t(g_idx, g_val, g_nnz)
    else:
        D.ctx.omp_batch_device(B[W:], K_ATOMS, eps, idx[W:], val[W:], nnz[W:])
        D.ctx.sync()
        if use_dist:  # the single collective of the path: every rank's (idx, val, nnz) shard over the host-side group
            gathered = exchange_results(idx[W:], val[W:], nnz[W:])
    barrier()
    dt = time.perf_counter() - t0
    window = D.ctx.profile_window()  # (before profile_read, which consumes the events)
    sweeps, sweep_ms = D.ctx.profile_read(reset=True)
    D.ctx.profile_enable(False)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    atoms = torch.tensor([float(nnz[W:].sum().item())], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(atoms, op=dist.ReduceOp.SUM)
    tmax, atoms = tmax.item(), atoms.item()

    if rank == 0:
        alg_bytes = M * N * 4  # bytes/atom = M*N*sizeof(Float32), A streamed once
        # a timed launch's bracket reads the launch plus the event pair itself: the empty-pair reading is measured and subtracted
        bracket_ms = D.ctx.profile_overhead(64)
        duration_s = max(sweep_ms / max(sweeps, 1) - bracket_ms, 0.0) / 1e3  # one launch, start to end
        two = window["streams"] == 2 and window["launches"] > 0
        if two:
            # Two pipelines side by side (csmp_omp_batch from two signals on): the sweep launches of the two streams OVERLAP -- while one
            # launch runs, the other pipeline's launch moves its bytes too, so bytes / (one launch's duration) is not a bandwidth.
            # The launches are priced as one window instead: all sweep launches from the first to the last timed one on each stream,
            # over the time from the earliest start event to the latest end event (HIP events on both streams, one clock).
            avg_sweep_s = window["window_ms"] / 1e3 / window["launches"]
        else:
            avg_sweep_s = duration_s
        achieved = alg_bytes / avg_sweep_s / 1e9 if sweeps else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "sweep_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 PMC passes (see profiles/README.md)
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "csmp::" + tick_kernel_name(D) + (" = the sweep launch of a tick: the product sweep of one signal (one column per wave, a ring of "
                          "32 nt loads = up to 32 KiB in flight per wave), one workgroup per CU; two pipelines of three signals run side by side on "
                          "two streams, so two such launches overlap and a CU that a workgroup of one leaves goes to the next workgroup of the other"
                          if two else " = the steady-state tick of the pipelined batch: the product sweep of one "
                          "signal (one column per wave, a ring of 32 nt loads = up to 32 KiB in flight per wave) fused with the two "
                          "short append stages of two other signals") + "; csmp::k_sweep_gen<float,16,2,false> when a signal runs alone",
                "sweep_config": D.ctx.sweep_config(),
                "launches_timed": int(window["launches"] if two else sweeps), "avg_launch_us": avg_sweep_s * 1e6, "algorithmic_bytes_per_launch": alg_bytes,
                "launch_duration_us": duration_s * 1e6, "launches_in_flight": 2 if two else 1,
                # the same bound from the wall clock alone: every atom moves the dictionary once, so the job's own throughput is a
                # bandwidth -- appends, launch gaps and the pipelines' fill and drain included
                "whole_job_GBps": atoms / tmax * alg_bytes / 1e9 / world, "whole_job_frac": atoms / tmax * alg_bytes / 1e9 / world / HBM_PEAK_GBS,
                "timer": ("HIP events on BOTH pipelines' streams around every %d-th sweep launch of the timed region: avg_launch_us = (latest end - "
                          "earliest start) / (the %d launches from the first to the last timed one on each stream) -- the launches of the two streams "
                          "overlap, each lasting launch_duration_us (mean of the timed ones minus the reading of an empty event pair, %.2f us; the "
                          "rocprofv3 row's AverageNs is THIS number)" % (args.profile_every, window["launches"], bracket_ms * 1e3)) if two else
                         ("HIP events on the library's stream around every %d-th steady-state tick of the timed region, minus the reading of "
                          "an empty event pair (%.2f us)" % (args.profile_every, bracket_ms * 1e3)), "event_pair_us": bracket_ms * 1e3}
        rp = rocprof_row(tick_kernel_name(D), pattern="r06_bench_kernel_stats.csv")
        if rp:
            roof["committed_profile"] = rp
        out = {
            "metric": "OMP atoms selected/sec at m=4096,n=65536,k=256 (single-signal sweeps, 2 x 3 signals pipelined)",
            "value": atoms / tmax, "unit": "atoms/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": tmax / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "signals_per_sec": K * world / tmax,
            "config": {"workload": "configs[1]: single-signal OMP, A 4096x65536 Float32 Gaussian unit-norm, "
                                   "k=256, planted +-1 256-sparse x0 + noise 5e-3, eps=eps(Float32)",
                       "signals_per_gpu": K, "signals_in_flight": 6 if two else 3,
                       "sharding": f"signals over {world} GPU(s), A replicated, one all_gather",
                       "collective": "ncclAllGather inside csmp_omp_sharded" if lib_gather else ("torch.distributed all_gather" if use_dist else "none (one rank)")},
            "ranks_seen": ranks_seen, "devices": devices,
            "roofline": roof,
            "atoms_selected": int(atoms),
        }
        if use_dist:
            # the one exchange, checked: a row for every signal of every rank, and the first timed signal of every rank
            # recomputed HERE (rank 0 regenerates it from its global id) must be the gathered row
            gi, gv, gn = cs.unpack_t(gathered, K_ATOMS)
            ok = []
            for r in range(world):
                b_r = make_signals(torch, dev, At, r * (K + W) + W, 1)[0].cpu().numpy()
                i_r, v_r, o_r = D.ctx.omp(b_r, K_ATOMS, eps)
                col = r * K
                ok.append(bool(int(gn[col]) == len(i_r) and np.array_equal(gi[:len(i_r), col], i_r)))
            out["gathered_rows"] = int(gathered.shape[0])
            out["gather_check"] = {"rows_expected": world * K, "rows_ok": int(gathered.shape[0]) == world * K,
                                   "first_signal_of_every_rank_equals_rank0_recomputation": ok}
            if int(gathered.shape[0]) != world * K or not all(ok):
                out["error"] = "gathered results do not match: see gather_check"
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs on rank 0 at N = 1 only
            try:
                # selection order of signal W (first timed signal) for the parity cross-check
                i0, v0, o0 = D.ctx.omp(B[W].cpu().numpy(), K_ATOMS, eps)
                out["cpu_baseline"] = cpu_baseline(At, B[W:], o0, args.cpu_seconds)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        if not args.no_secondary and world == 1:
            # the other single-GPU configurations of BASELINE.json, measured by this same process (builder-run lines
            # of them also sit under profiles/; these are the driver-run ones)
            sec = {}
            try:  # the reference API's own shape of work: ONE omp(A, b, k) at a time (src/matchingpursuit.jl:73-86), nothing pipelined
                sec["lone_omp_c2"] = measure_lone_omp(3, 1, B, D, eps)
            except Exception as e:  # noqa: BLE001
                sec["lone_omp_c2"] = {"error": repr(e)}
            try:  # the same dictionary left in host memory, every sweep over the host link
                sec["omp_c2_streamed"] = measure_streamed_omp(cs, torch, dev, At, D, B, eps)
            except Exception as e:  # noqa: BLE001
                sec["omp_c2_streamed"] = {"error": repr(e)}
            try:  # the same, with b and the results resident in HBM: the device-resident one-signal figure
                sec["lone_omp_c2_device"] = measure_lone_omp_device(3, 1, torch, dev, B, D, eps)
            except Exception as e:  # noqa: BLE001
                sec["lone_omp_c2_device"] = {"error": repr(e)}
            try:  # opt-in: sweeps over the binary16 image with certified picks (same results, half the bytes)
                sec["omp_c2_screened_f16"] = measure_screened_omp(6, 2, torch, dev, At, D, eps)  # (binary16 image, rigorous certificate)
            except Exception as e:  # noqa: BLE001
                sec["omp_c2_screened_f16"] = {"error": repr(e)}
            # Only modes whose results are PROVABLY the exact path's are measured here (the rigorous certificate): library defaults /
            # + the resident Gram matrix / bf16 operands.  The statistical certificate and the int8 images stay opt-in and out of this
            # line (tests/test_gpu_parity.py::test_batched_certificate_against_adversarial_residuals shows what they can miss);
            # `--workload batched --batch-cert statistical --batch-screen int8` still measures them.
            for name, cert, gram, scr in (("batched_c3", 1, 0, 3), ("batched_c3_gram", 1, 1, 3), ("batched_c3_rigorous_bf16", 1, 0, 0)):
                try:
                    sec[name] = measure_batched(2, 1, cs, torch, dist, dev, 0, 1, At, D, False, cert=cert, gram=gram, screen=scr)
                except Exception as e:  # noqa: BLE001
                    sec[name] = {"error": repr(e)}
            # SURVEY 8(f) rows 2 and 3 at the configs[1] shape: forward regression (batched ticks), ompr, srr
            import copy
            for name, wl, st_, wu in (("fr_8f3", "fr", 6, 3), ("ompr_8f2", "ompr", 3, 1), ("srr_8f2", "srr", 3, 1)):
                a2 = copy.copy(args)
                a2.workload, a2.steps, a2.warmup = wl, st_, wu
                a2.screened = False
                try:
                    sec[name] = (run_fr if wl == "fr" else run_twostage)(a2, cs, torch, dev, At, D, show=False)
                except Exception as e:  # noqa: BLE001
                    sec[name] = {"error": repr(e)}
            D.close()
            del B, idx, val, nnz, At
            torch.cuda.empty_cache()
            try:  # the reference's own element type (verdict round 4, item 1)
                sec["omp_f64_dictionary"] = measure_f64_dictionary(6, 3, cs, torch, dev)
            except Exception as e:  # noqa: BLE001
                sec["omp_f64_dictionary"] = {"error": repr(e)}
            try:
                At5, D5 = make_dictionary5(cs, torch, dev)
                sec["gomp_c5"] = measure_config5("gomp", 6, 2, cs, torch, dev, D5, At5)
                sec["gomp_c5_single"] = measure_config5("gomp_single", 2, 1, cs, torch, dev, D5, At5)
                sec["gomp_c5_screened_f16"] = measure_config5("gomp", 6, 2, cs, torch, dev, D5, At5, screened=3)
                sec["sp_c5"] = measure_config5("sp", 9, 3, cs, torch, dev, D5, At5)
                sec["sp_c5_single"] = measure_config5("sp_single", 3, 1, cs, torch, dev, D5, At5)
                sec["sp_c5_default_delta"] = measure_config5("sp", 9, 3, cs, torch, dev, D5, At5, delta=1e-12)
                sec["sp_c5_screened_f16"] = measure_config5("sp", 9, 3, cs, torch, dev, D5, At5, screened=3)
                D5.close()
            except Exception as e:  # noqa: BLE001
                sec["config5"] = {"error": repr(e)}
            out["secondary"] = sec
            try:  # the fastest PROVABLY identical path of this line (opt-in: binary16 image, rigorous certificate)
                sc = sec["omp_c2_screened_f16"]
                out["fastest_provably_identical_results"] = {"value": sc["value"], "unit": "atoms/s", "where": "secondary.omp_c2_screened_f16 (csmp_omp_batch, CSMP_OPT_SCREENED_SWEEP = 3, rigorous certificate)",
                                                             "equals_exact_path": sc["batch"]["equals_exact_path"], "fallbacks": sc["batch"]["stats"]["fallbacks"]}
            except Exception:  # noqa: BLE001
                pass
        emit(out)
    D.close()
    finish()


if __name__ == "__main__":
    main()
