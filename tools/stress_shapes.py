"""Randomised GPU-vs-oracle comparison of what round 5 added: every driver on ragged and tall dictionaries (any M: the shape-general
sweep, the residual staged in phases beyond M ~ 20 400, both element types), the forward-regression family where its LDS images do
not fit, ompr on the inverse Gram matrix (and its fallback), the step-level SP / OMPR functors call by call, and column removal
at capacities on either side of 1023.    python tools/stress_shapes.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402

cs = load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
runs = bad = 0
tally = {}


def cmp(name, got, ref, cfg, iters=None):
    global runs, bad
    runs += 1
    tally[name] = tally.get(name, 0) + 1
    ok = np.array_equal(got[0], ref[0])
    if ok and len(ref[1]) and np.all(np.isfinite(ref[1])):
        ok = np.allclose(got[1], ref[1], rtol=1e-6, atol=1e-6 * float(np.max(np.abs(ref[1]))))
    if ok and iters is not None:
        ok = iters[0] == iters[1]
    if not ok:
        bad += 1
        print("MISMATCH", name, cfg, got[0][:10], ref[0][:10], iters, flush=True)


while time.time() - t0 < budget:
    M = int(rng.choice([1000, 1001, 3000, 4352, 4097, 6000, 8000, 12000, 20500, 32768, 40002, 777]))
    N = int(rng.choice([300, 700, 1500])) if M >= 8000 else int(rng.choice([600, 1500, 4000]))
    dtype = rng.choice([np.float32, np.float64])
    k = int(rng.integers(3, 28))
    A = rng.standard_normal((M, N))
    A -= 1e-6 * A.mean(axis=0, keepdims=True)
    A /= np.linalg.norm(A, axis=0, keepdims=True)
    A = np.asfortranarray(A.astype(dtype))
    eps = float(np.finfo(dtype).eps)
    cfg = (M, N, k, str(np.dtype(dtype)))

    def signal(extra=0, noise=5e-3):
        supp = rng.choice(N, k + extra, replace=False)
        e = rng.standard_normal(M)
        return A[:, supp].astype(np.float64) @ rng.choice([-1.0, 1.0], k + extra) + noise * e / np.linalg.norm(e)

    D = cs.Dictionary(A)
    try:
        y = signal()
        cmp("omp", D.ctx.omp(y, k, eps), oc.omp(A, y, k, eps), cfg)
        l = int(rng.choice([2, 3, 4]))
        cmp("gomp", D.ctx.gomp(y, l, k, eps), oc.gomp(A, y, l, k, eps), cfg + (l,))
        cmp("mp", D.ctx.mp(y, k + 5), oc.mp(A, y, k + 5), cfg)
        if 2 * k <= M:
            rs = oc.sp(A, y, k, 1e-10)
            gs = D.ctx.sp(y, k, 1e-10)
            cmp("sp", gs, rs, cfg, (gs[2], rs[2]))
        Y = np.asfortranarray(np.stack([signal() for _ in range(4)], axis=1))
        idx, val, nnz = D.ctx.omp_batch(Y, k, eps)
        for s in range(4):
            cmp("omp_batch", (idx[:nnz[s], s], val[:nnz[s], s]), oc.omp(A, Y[:, s], k, eps), cfg)
        # forward regression family (the tall path beyond M ~ 10 000 / ~ 6 800)
        yf = signal(noise=0.05)
        cmp("fr", D.ctx.fr(yf, k, 0.0, 0.0), oc.fr(A, yf, k), cfg)
        ys = signal(extra=2, noise=0.2)
        init = int(rng.choice([1, 2]))
        rr = oc.srr(A, ys, k, 1e-12, -1, init, 1)
        gr = D.ctx.srr(ys, k, 1e-12, -1, init, 1)
        cmp("srr", gr, rr, cfg + (init,), (gr[2], rr[2]))
        cmp("rmp", D.ctx.rmp(yf, 0.02, 2), oc.rmp(A, yf, 0.02, 2), cfg)
        cmp("foba", D.ctx.foba(yf, 0.02), oc.foba(A, yf, 0.02), cfg)
        # ompr: exchanges on the inverse Gram matrix, and the same solve with the guard failing (QR path)
        ro = oc.ompr(A, ys, k, 1e-9)
        go = D.ctx.ompr(ys, k, 1e-9)
        cmp("ompr", go, ro, cfg, (go[2], ro[2]))
        D.ctx.tune("swap_refuse", 1)
        go = D.ctx.ompr(ys, k, 1e-9)
        D.ctx.tune("swap_refuse", 0)
        cmp("ompr_qr_fallback", go, ro, cfg, (go[2], ro[2]))
        # step-level functors, call by call
        if 2 * k <= M:
            P = cs.SP(D, ys, k)
            x = cs.sp_acquisition(P)
            ri, rv = onp.sp_acquisition(A, ys, [], [], k)
            cmp("SP.acquisition", (x.nzind, x.nzval), (ri, rv), cfg)
            for _ in range(3):
                x = P(x)
                ri, rv = onp.sp_update(A, ys, ri, rv, k)
                cmp("SP.update", (x.nzind, x.nzval), (ri, rv), cfg)
            P.close()
        P = cs.OMPR(D, ys, k)
        x = cs.oblivious_acquisition(P, None, k)
        ri, rv = onp.oblivious_acquisition(A, ys, k)
        cmp("OMPR.acquisition", (x.nzind, x.nzval), (ri, rv), cfg)
        for _ in range(3):
            x = P(x)
            ri, rv = onp.ompr_update(A, ys, ri, rv)
            cmp("OMPR.update", (x.nzind, x.nzval), (ri, rv), cfg)
        P.close()
        # dropindex! at a capacity on either side of 1023 (and, rarely, beyond 4095)
        kcap = int(rng.choice([k + 4, 1500, 4200])) if M >= 4200 else int(rng.choice([k + 4, min(M, 1500)]))
        D.ctx.solver_begin(cs._lib.ALGO_OMP, y, kcap)
        for _ in range(k):
            D.ctx.solver_step(1)
        i0, v0, res, order, stop = D.ctx.solver_state(kcap)
        supp = list(order)
        for _ in range(2):
            atom = supp[int(rng.integers(0, len(supp)))]
            D.ctx.solver_remove(atom)
            supp.remove(atom)
        i1, v1, res, order, stop = D.ctx.solver_state(kcap)
        S = np.array(sorted(supp))
        coef = np.linalg.lstsq(A[:, S].astype(np.float64), y, rcond=None)[0]
        cmp("solver_remove", (i1, v1), (S, coef), cfg + (kcap,))
    finally:
        D.close()
print(f"runs {runs}  mismatches {bad}  {tally}  {time.time() - t0:.0f} s")
