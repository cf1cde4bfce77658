"""Full-size, FULL-k comparison of the HIP path with the oracle at BASELINE configs[1], [2] and [4] -- the GPU tests
compare an oracle PREFIX at these sizes (12-16 atoms) and rely on properties for the rest, because complete CPU solves take
minutes; this tool spends them once per round.  Prints one JSON line per check (supports / selection order exact,
coefficients to 1e-6 relative, iteration counts), then a summary.
    python tools/full_size_oracle_check.py [c2 c3 c5]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from csmp_pkg import load
from oracle import oracle_c as oc

cs = load()
dev = torch.device("cuda", 0)
EPS32 = float(np.finfo(np.float32).eps)
which = set(sys.argv[1:]) or {"c2", "c3", "c5"}
bad = 0


def report(name, ok, **kw):
    global bad
    bad += not ok
    print(json.dumps({"check": name, "ok": bool(ok), **kw}), flush=True)


def same(got_idx, got_val, ref_idx, ref_val):
    ok = len(got_idx) == len(ref_idx) and np.array_equal(got_idx, ref_idx)
    rel = None
    if ok and len(ref_val):
        rel = float(np.abs(np.asarray(got_val) - ref_val).max() / max(np.abs(ref_val).max(), 1e-300))
        ok = rel <= 1e-6
    return ok, rel


if which & {"c2", "c3"}:
    At = bench.make_dictionary(torch, dev)
    A = np.asfortranarray(At.cpu().numpy().T)
    D = cs.Dictionary(At, device=0)
    if "c2" in which:  # configs[1]: k = 256, every atom of three signals
        B = bench.make_signals(torch, dev, At, 500, 3)
        for s in range(3):
            y = B[s].cpu().numpy()
            t0 = time.time()
            ref = oc.omp(A, y, 256, EPS32)
            t1 = time.time()
            got = D.ctx.omp(y, 256, EPS32)
            ok, rel = same(got[0], got[1], ref[0], ref[1])
            ok = ok and np.array_equal(got[2], ref[2])
            report("configs[1] omp k=256, signal %d: support, selection order, coefficients" % s, ok, atoms=len(ref[0]), max_rel_coef_err=rel,
                   oracle_seconds=round(t1 - t0, 1))
        idx = torch.full((3, 256), -1, dtype=torch.int64, device=dev)
        val = torch.zeros((3, 256), dtype=torch.float64, device=dev)
        nnz = torch.zeros(3, dtype=torch.int64, device=dev)
        D.ctx.omp_batch_device(B, 256, EPS32, idx, val, nnz)
        D.ctx.sync()
        for s in range(3):
            ref = oc.omp(A, B[s].cpu().numpy(), 256, EPS32)
            ok, rel = same(idx[s, :int(nnz[s])].cpu().numpy(), val[s, :int(nnz[s])].cpu().numpy(), ref[0], ref[1])
            report("configs[1] omp_batch (k_tick pipeline) k=256, signal %d" % s, ok, max_rel_coef_err=rel)
        D.ctx.set_option("screened_sweep", 1)  # the screened single-signal sweep: every atom of signal 0 against the oracle
        got = D.ctx.omp(B[0].cpu().numpy(), 256, EPS32)
        ref = oc.omp(A, B[0].cpu().numpy(), 256, EPS32)
        ok, rel = same(got[0], got[1], ref[0], ref[1])
        report("configs[1] omp k=256 with the screened sweep (bf16 image, certified picks)", ok and np.array_equal(got[2], ref[2]), max_rel_coef_err=rel,
               stats=D.ctx.screened_stats())
        D.ctx.set_option("screened_sweep", 0)
        # forward regression at the same size, 64 atoms
        y = B[0].cpu().numpy()
        ref = oc.fr(A, y, 64)
        got = D.ctx.fr(y, 64)
        ok, rel = same(got[0], got[1], ref[0], ref[1])
        report("configs[1] shape, fr / ols k=64", ok, max_rel_coef_err=rel)
    if "c3" in which:  # configs[2]: 1024 signals, k = 128: six of them against the oracle for all 128 atoms, under every option
        nsig, k = 1024, 128
        B = bench.make_signals_fast(torch, dev, At, 7000, nsig, k).reshape(nsig, bench.M)
        sample = [0, 255, 256, 511, 777, 1023]
        refs = {s: oc.omp(A, B[s].cpu().numpy(), k, EPS32) for s in sample}
        for name, cert, gram, scr in (("statistical certificate", 0, 0, 0), ("rigorous certificate", 1, 0, 0), ("resident Gram matrix", 0, 1, 0),
                                      ("int8 screen", 0, 0, 1), ("int8 screen + resident Gram matrix", 0, 1, 1)):
            D.ctx.set_option("batch_cert", cert)
            D.ctx.set_option("batch_gram", gram)
            D.ctx.set_option("batch_screen", scr)
            idx = torch.full((nsig, k), -1, dtype=torch.int64, device=dev)
            val = torch.zeros((nsig, k), dtype=torch.float64, device=dev)
            nnz = torch.zeros(nsig, dtype=torch.int64, device=dev)
            D.ctx.omp_batch_mfma_device(B, k, EPS32, idx, val, nnz)
            D.ctx.sync()
            st = D.ctx.batch_stats()
            allok, worst = True, 0.0
            for s in sample:
                ok, rel = same(idx[s, :int(nnz[s])].cpu().numpy(), val[s, :int(nnz[s])].cpu().numpy(), refs[s][0], refs[s][1])
                allok &= ok
                worst = max(worst, rel or 0.0)
            report("configs[2] omp_batch_mfma k=128, %s: 6 signals x 128 atoms vs oracle" % name, allok, max_rel_coef_err=worst,
                   uncertain=st["uncertain"], illcond=st["illcond"])
        D.ctx.set_option("batch_cert", 0)
        D.ctx.set_option("batch_gram", 0)
        D.ctx.set_option("batch_screen", 2)
    D.close()
    del At, A

if "c5" in which:  # configs[4]: 8192 x 131072, k = 512: GOMP S = 4 and Subspace Pursuit, complete solves
    At5, D5 = bench.make_dictionary5(cs, torch, dev)
    A5 = np.asfortranarray(At5.cpu().numpy().T)
    M5, N5, k = 8192, 131072, 512
    g = torch.Generator(device=dev).manual_seed(2026)
    sel = torch.randperm(N5, generator=g, device=dev)[:k]
    sign = torch.randint(0, 2, (k,), generator=g, device=dev).to(torch.float64) * 2 - 1
    e = torch.randn(M5, generator=g, device=dev, dtype=torch.float64)
    y = ((At5[sel].to(torch.float64) * sign[:, None]).sum(0) + e * (5e-3 / e.norm())).cpu().numpy()
    t0 = time.time()
    ref = oc.gomp(A5, y, 4, k, EPS32)
    t1 = time.time()
    got = D5.ctx.gomp(y, 4, k, EPS32)
    ok, rel = same(got[0], got[1], ref[0], ref[1])
    report("configs[4] gomp S=4 k=512: support, selection order, coefficients", ok and np.array_equal(got[2], ref[2]), atoms=len(ref[0]),
           max_rel_coef_err=rel, oracle_seconds=round(t1 - t0, 1))
    D5.ctx.set_option("screened_sweep", 1)
    got = D5.ctx.gomp(y, 4, k, EPS32)
    ok, rel = same(got[0], got[1], ref[0], ref[1])
    report("configs[4] gomp S=4 k=512 with the screened sweep (certified top-S picks)", ok and np.array_equal(got[2], ref[2]), max_rel_coef_err=rel,
           stats=D5.ctx.screened_stats())
    D5.ctx.set_option("screened_sweep", 0)
    bi, bv, bn = D5.ctx.gomp_batch(np.asfortranarray(np.stack([y, -y], axis=1)), 4, k, EPS32)
    ok, rel = same(bi[:bn[0], 0], bv[:bn[0], 0], ref[0], ref[1])
    ok2, rel2 = same(bi[:bn[1], 1], -bv[:bn[1], 1], ref[0], ref[1])
    report("configs[4] gomp_batch (two in flight): both signals", ok and ok2, max_rel_coef_err=max(rel or 0, rel2 or 0))
    for delta in (1e-2, 1e-12):
        t0 = time.time()
        ref = oc.sp(A5, y, k, delta)
        t1 = time.time()
        got = D5.ctx.sp(y, k, delta)
        ok, rel = same(got[0], got[1], ref[0], ref[1])
        report("configs[4] sp k=512 delta=%g: support, coefficients, update! calls" % delta, ok and got[2] == ref[2], update_calls=int(ref[2]),
               max_rel_coef_err=rel, oracle_seconds=round(t1 - t0, 1))
        D5.ctx.set_option("screened_sweep", 2)
        D5.ctx.screened_stats(reset=True)
        got = D5.ctx.sp(y, k, delta)
        ok, rel = same(got[0], got[1], ref[0], ref[1])
        report("configs[4] sp k=512 delta=%g with the screened sweep (int8 image, certified top-k sets)" % delta, ok and got[2] == ref[2],
               max_rel_coef_err=rel, stats=D5.ctx.screened_stats())
        D5.ctx.set_option("screened_sweep", 0)
    D5.close()

print(json.dumps({"summary": "all checks passed" if bad == 0 else "%d check(s) FAILED" % bad}))
sys.exit(1 if bad else 0)
