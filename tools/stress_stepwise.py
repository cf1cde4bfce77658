"""Randomised GPU-vs-oracle comparison of the SURVEY 8(f) drivers (fr, srr, ompr, rmp, foba, br, lace).
    python tools/stress_stepwise.py [seconds] [seed]
Prints every disagreement (support, iteration count, coefficients beyond 1e-6 relative) and a tally."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csmp_pkg import load  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402

cs = load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
runs = bad = 0
tally = {}


def cmp(name, got, ref, cfg, iters=False):
    global runs, bad
    runs += 1
    tally[name] = tally.get(name, 0) + 1
    ok = np.array_equal(got[0], ref[0])
    if ok and len(ref[1]):
        ok = np.allclose(got[1], ref[1], rtol=1e-6, atol=1e-6 * float(np.max(np.abs(ref[1]))))
    if ok and iters:
        ok = got[2] == ref[2]
    if not ok:
        bad += 1
        print("MISMATCH", name, cfg, "got", got[0][:12], got[2] if iters else "", "ref", ref[0][:12], ref[2] if iters else "", flush=True)


while time.time() - t0 < budget:
    M = int(rng.choice([24, 40, 64, 100, 128, 200, 256, 384]))
    N = int(rng.choice([M // 2, M, 2 * M, 4 * M, 8 * M]))
    N = max(N, 8)
    dtype = rng.choice([np.float32, np.float64])
    k = int(rng.integers(1, max(2, min(M // 4, N // 2, 40))))
    noise = float(rng.choice([0.0, 1e-3, 5e-2, 0.3]))
    coherent = rng.random() < 0.25
    A = rng.standard_normal((M, N))
    if coherent:
        A += rng.uniform(0.5, 2.0) * rng.standard_normal((M, 1))
    A /= np.linalg.norm(A, axis=0)
    A = np.asfortranarray(A.astype(dtype))
    kk = min(N, k + int(rng.integers(0, 3)))
    supp = rng.choice(N, kk, replace=False)
    # distinct magnitudes: with equal +-1 coefficients and k < #planted, two planted atoms tie EXACTLY in exact
    # arithmetic (|<a_i, b>| = |1 + s g| for both) and rounding picks the winner -- the undefined-parity regime
    b = A[:, supp].astype(np.float64) @ (rng.choice([-1.0, 1.0], kk) * rng.uniform(0.5, 2.0, kk))
    if noise:
        e = rng.standard_normal(M)
        b = b + noise * e / np.linalg.norm(e)
    cfg = (M, N, k, str(np.dtype(dtype)), noise, coherent)
    D = cs.Dictionary(A)
    try:
        cmp("fr", D.ctx.fr(b, k), oc.fr(A, b, k), cfg)
        if k + 1 <= M and k <= N:
            l = int(rng.choice([1, 1, 2, 3]))
            if k + l <= M:
                for init in (1, 2):
                    cmp(f"srr{init}", D.ctx.srr(b, k, 1e-12, -1, init, l), oc.srr(A, b, k, 1e-12, -1, init, l), cfg + (l,), iters=True)
            ref_ompr = oc.ompr(A, b, k, 1e-9)
            cmp("ompr", D.ctx.ompr(b, k, 1e-9), ref_ompr, cfg, iters=True)
            D.ctx.set_option("screened_sweep", int(rng.integers(1, 3)))  # image sweeps, certified selections (bf16 / int8 image)
            cmp("ompr_screened", D.ctx.ompr(b, k, 1e-9), ref_ompr, cfg, iters=True)
            D.ctx.set_option("screened_sweep", 0)
        if noise > 0:
            cmp("rmp_d", D.ctx.rmp(b, noise), oc.rmp(A, b, noise), cfg)
            cmp("foba", D.ctx.foba(b, noise), oc.foba(A, b, noise), cfg)
        if N <= M and noise > 0:
            cmp("br", D.ctx.br(b, k=k), oc.br(A, b, k=k), cfg)
            cmp("lace", D.ctx.br(b, max_eps=2 * noise, lace=True), oc.br(A, b, max_eps=2 * noise, lace=True), cfg)
            if M <= 128:
                cmp("rmp_k", D.ctx.rmp(b, k), oc.rmp(A, b, k), cfg)
    finally:
        D.close()
print(f"runs {runs}  mismatches {bad}  {tally}  {time.time() - t0:.0f} s")
