"""Generates tests/golden/golden_small.npz.

The reference (Julia) cannot run in the build container and its tests hold no golden vectors
(test/matchingpursuit.jl:7 -- unseeded random data), so these vectors are produced by the C
oracle (oracle/csmp_oracle.c) and accepted only where the independent numpy twin
(oracle/oracle_np.py) returns the identical support/order and coefficients to 1e-10: they pin
the oracle against regressions and give the GPU path committed known answers, including the
edge cases the reference's code path has but its tests never exercise (ties, stagnation,
eps-stop, full support, the gomp remainder step).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from csmp_pkg import load  # noqa: E402
from oracle import oracle_c as oc, oracle_np as on  # noqa: E402

cs = load()
EPS64 = float(np.finfo(np.float64).eps)
EPS32 = float(np.finfo(np.float32).eps)
out = {}
names = []


def agree(a, b, what):
    assert np.array_equal(a[0], b[0]), (what, a[0], b[0])
    assert np.allclose(a[1], b[1], rtol=1e-10, atol=1e-13), (what, np.abs(a[1] - b[1]).max())
    if len(a) > 2 and isinstance(a[2], np.ndarray):
        assert np.array_equal(a[2], b[2]), (what, "order")


def add(name, algo, A, b, params, res):
    names.append(name)
    out[f"{name}.A"] = A
    out[f"{name}.b"] = b
    out[f"{name}.algo"] = np.array(algo)
    out[f"{name}.params"] = np.array(params, dtype=np.float64)
    out[f"{name}.idx"] = res[0]
    out[f"{name}.val"] = res[1]
    out[f"{name}.order"] = res[2] if len(res) > 2 and isinstance(res[2], np.ndarray) else np.zeros(0, np.int64)


def omp_case(name, A, b, k, eps, twin=True):
    r = oc.omp(A, b, k, eps)
    if twin:
        agree(r, on.omp(A, b, k, eps), name)
    add(name, "omp", A, b, [k, eps], r)


def gomp_case(name, A, b, l, k, eps, twin=True):
    r = oc.gomp(A, b, l, k, eps)
    if twin:
        agree(r, on.gomp(A, b, l, k, eps), name)
    add(name, "gomp", A, b, [l, k, eps], r)


def mp_case(name, A, b, k):
    r = oc.mp(A, b, k)
    agree(r, on.mp(A, b, k), name)
    add(name, "mp", A, b, [k], r)


def sp_case(name, A, b, k, delta):
    r = oc.sp(A, b, k, delta)
    r2 = on.sp(A, b, k, delta)
    agree(r[:2], r2[:2], name)
    assert r[2] == r2[2]
    add(name, "sp", A, b, [k, delta, r[2]], r[:2])


# 1-2: the reference's own test shape (test/matchingpursuit.jl:10-29), seeded, noiseless + noisy
A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=11)
y = cs.perturb(b, 5e-3, rng=12)
omp_case("omp_ref_32x48_noisy", A, y, 3, EPS64)
mp_case("mp_ref_32x48", A, b, 30)
gomp_case("gomp_ref_32x48_l2", A, y, 2, 3, EPS64)
out["omp_ref_32x48_noisy.x0_idx"] = x.nzind
out["omp_ref_32x48_noisy.x0_val"] = x.nzval

# 3-4: medium f64 / f32 dictionaries
A, x, b = cs.sparse_data(n=64, m=256, k=8, rng=21)
y = cs.perturb(b, 5e-3, rng=22)
omp_case("omp_f64_64x256_k8", A, y, 8, EPS64)
A32 = np.asfortranarray(A.astype(np.float32))
y32 = cs.perturb(A32[:, x.nzind].astype(np.float64) @ x.nzval, 5e-3, rng=23)
omp_case("omp_f32_64x256_k8", A32, y32, 8, EPS32)
gomp_case("gomp_f32_64x256_k7_l3", A32, y32, 3, 7, EPS32)  # 2 full steps + remainder 1 (:134-137)
sp_case("sp_f64_64x256_k8", A, y, 8, 1e-12)
mp_case("mp_f32_64x256_k40", A32, y32, 40)

# 5: ragged shape (M not a multiple of any vector width), f32
A, x, b = cs.sparse_data(n=37, m=101, k=4, rng=31, dtype=np.float32)
y = cs.perturb(b, 5e-3, rng=32)
omp_case("omp_f32_ragged_37x101", A, y, 4, EPS32)
gomp_case("gomp_f32_ragged_37x101", A, y, 3, 5, EPS32)

# 6: duplicated columns -> exact tie, the lower index must win (argmax = first max, :184)
A, x, b = cs.sparse_data(n=32, m=40, k=3, rng=41)
A = np.asfortranarray(np.concatenate([A, A[:, x.nzind]], axis=1))  # copies at 40, 41, 42
y = cs.perturb(b, 5e-3, rng=42)
# (the numpy twin is not consulted here: BLAS gemv may round the two copies of a column
#  differently depending on their position, which breaks the exact tie -- the C oracle and the
#  GPU kernel both compute every column with one fixed summation order, so the tie is exact)
omp_case("omp_dupcols", A, y, 3, EPS64, twin=False)
assert np.all(out["omp_dupcols.idx"] < 40)
t = on.omp(A, y, 3, EPS64)
assert np.array_equal(np.sort(np.where(t[0] >= 40, x.nzind[np.clip(t[0] - 40, 0, 2)], t[0])), out["omp_dupcols.idx"])
gomp_case("gomp_dupcols", A, y, 2, 4, EPS64, twin=False)

# 7: eps-stop: noiseless 3-sparse signal, generous k, eps well above round-off
A, x, b = cs.sparse_data(n=48, m=128, k=3, rng=51)
omp_case("omp_eps_stop", A, b, 10, 1e-8)
assert len(out["omp_eps_stop.idx"]) == 3
gomp_case("gomp_eps_stop_remainder", A, b, 2, 9, 1e-8)  # eps-break, then the remainder step still runs

# 8: stagnation: every atom already selected, eps = 0 never stops (:66)
A, x, b = cs.sparse_data(n=16, m=3, k=2, rng=61)
y = cs.perturb(b, 1e-2, rng=62)
omp_case("omp_stagnation_N3", A, y, 6, 0.0)
assert len(out["omp_stagnation_N3.idx"]) == 3

# 9: support fills the whole row space: nnz(x) == size(A,1) guard (:63)
A, x, b = cs.sparse_data(n=4, m=12, k=2, rng=71)
y = cs.perturb(b, 1e-1, rng=72)
omp_case("omp_full_M4", A, y, 8, 0.0)
assert len(out["omp_full_M4.idx"]) == 4

# 10: b = 0: the first update! still adds atom 0 with coefficient 0 before the eps check (:77-79)
A, x, b = cs.sparse_data(n=16, m=24, k=2, rng=81)
omp_case("omp_zero_b", A, np.zeros(16), 3, EPS64)
assert out["omp_zero_b.idx"].tolist() == [0]

# 11: the twostage test shape (test/twostage.jl:42-52)
A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=91)
y = cs.perturb(b, 5e-3, rng=92)
sp_case("sp_ref_32x64", A, y, 3, 1e-2)
sp_case("sp_ref_32x64_noiseless", A, b, 3, 1e-12)



# 12-: forward regression / OLS (src/forward.jl).  params = [k, max_eps, min_delta]
def fr_case(name, A, b, k, max_eps=0.0, min_delta=0.0, twin=True):
    r = oc.fr(A, b, k, max_eps, min_delta)
    if twin:
        agree(r, on.fr(A, b, k, max_eps, min_delta), name)
    add(name, "fr", A, b, [k, max_eps, min_delta], r)


A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=11)  # test/forward.jl:8-22
y = cs.perturb(b, 1e-2, rng=12)
fr_case("fr_ref_32x48", A, b, 3)
fr_case("fr_ref_32x48_noisy", A, y, 3)
assert np.array_equal(out["fr_ref_32x48.idx"], x.nzind) and np.array_equal(out["fr_ref_32x48_noisy.idx"], x.nzind)
A, x, b = cs.sparse_data(n=64, m=256, k=8, rng=21)
A32 = np.asfortranarray(A.astype(np.float32))
y32 = cs.perturb(A32[:, x.nzind].astype(np.float64) @ x.nzval, 5e-3, rng=23)
fr_case("fr_f32_64x256_k12", A32, y32, 12)
fr_case("fr_f64_64x256_eps_stop", A, cs.perturb(b, 5e-3, rng=22), 40, max_eps=0.02)  # norm(r) > max_eps (:60)
assert 0 < len(out["fr_f64_64x256_eps_stop.idx"]) < 40
fr_case("fr_f64_64x256_delta_stop", A, cs.perturb(b, 5e-3, rng=22), 40, min_delta=0.05)  # min_delta^2 < max (:64)
assert 0 < len(out["fr_f64_64x256_delta_stop.idx"]) < 40
A, x, b = cs.sparse_data(n=37, m=101, k=4, rng=31, dtype=np.float32)
fr_case("fr_f32_ragged_37x101", A, cs.perturb(b, 5e-3, rng=32), 6)
A, x, b = cs.sparse_data(n=32, m=40, k=3, rng=41)
A = np.asfortranarray(np.concatenate([A, A[:, x.nzind]], axis=1))
fr_case("fr_dupcols", A, cs.perturb(b, 5e-3, rng=42), 3, twin=False)  # exact tie -> lower index (findmax)
assert np.all(out["fr_dupcols.idx"] < 40)
# a coherent dictionary, where the OLS rule and the OMP rule select different atoms
rng = np.random.default_rng(11)
A = rng.standard_normal((48, 600)) + 1.5 * rng.standard_normal((48, 1))
A = np.asfortranarray(A / np.linalg.norm(A, axis=0))
b = A[:, rng.choice(600, 16, replace=False)] @ rng.standard_normal(16) + 1e-3 * rng.standard_normal(48)
fr_case("fr_coherent_48x600", A, b, 16)
assert not np.array_equal(out["fr_coherent_48x600.order"], oc.omp(A, b, 16, 0.0)[2])



# stepwise regression with replacement (src/twostage.jl:3-33).  params = [k, delta, initialization, l, iterations]
def srr_case(name, A, b, k, init, l):
    r = oc.srr(A, b, k, 1e-12, -1, init, l)
    t = on.srr(A, b, k, 1e-12, None, init, l)
    agree(r[:2], t[:2], name)
    assert r[2] == t[2]
    add(name, "srr", A, b, [k, 1e-12, init, l, r[2]], r[:2])


A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=91)  # test/twostage.jl:6-9
srr_case("srr_ref_32x64", A, cs.perturb(b, 5e-3, rng=92), 3, 1, 1)
srr_case("srr_ref_32x64_l3", A, cs.perturb(b, 5e-3, rng=92), 3, 1, 3)
A, x, b = cs.sparse_data(n=128, m=512, k=14, rng=93, dtype=np.float32)
y = cs.perturb(b, 2e-1, rng=94)
srr_case("srr_f32_128x512_k12", A, y, 12, 1, 1)  # two atoms fewer than planted, noisy: several replacements
assert out["srr_f32_128x512_k12.params"][4] >= 3
srr_case("srr_f32_128x512_k12_init2", A, y, 12, 2, 1)



# relevance matching pursuit and FoBa (src/stepwise.jl).  params: rmp_k [k]; rmp_delta [delta, maxiter]; foba [delta]
def stepwise_case(name, algo, A, b, params, f):
    r, t = f(oc), f(on)
    agree(r, t, name)
    add(name, algo, A, b, params, r)


A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=95)  # test/stepwise.jl:6-9
y = cs.perturb(b, 1e-2, rng=96)
stepwise_case("rmp_k_ref_32x64", "rmp_k", A, y, [3], lambda o: o.rmp(A, y, 3))
stepwise_case("rmp_delta_ref_32x64", "rmp_delta", A, y, [1e-2, 3], lambda o: o.rmp(A, y, 1e-2, 3))
stepwise_case("foba_ref_32x64", "foba", A, y, [1e-2], lambda o: o.foba(A, y, 1e-2))
A, x, b = cs.sparse_data(n=96, m=300, k=8, rng=97, dtype=np.float32)
y = cs.perturb(b, 5e-2, rng=98)
stepwise_case("foba_f32_96x300", "foba", A, y, [0.05], lambda o: o.foba(A, y, 0.05))
for nm in ("rmp_k_ref_32x64", "rmp_delta_ref_32x64", "foba_ref_32x64"):
    assert len(out[nm + ".idx"]) == 3



# backward regression and LACE (src/backward.jl).  params = [max_eps, max_delta, k]
A, x, b = cs.sparse_data(n=32, m=32, k=3, rng=101)  # test/backward.jl:10-14
y = cs.perturb(b, 5e-3, rng=102)
BIG = 1e300  # stands for Inf in the fixture
stepwise_case("br_ref_32x32_k3", "br", A, y, [BIG, BIG, 3], lambda o: o.br(A, y, k=3))
stepwise_case("br_ref_32x32_eps", "br", A, y, [1e-2, BIG, 0], lambda o: o.br(A, y, max_eps=1e-2))
stepwise_case("lace_ref_32x32_delta", "lace", A, y, [BIG, 1e-2, 0], lambda o: o.br(A, y, max_delta=1e-2, lace=True))
for nm in ("br_ref_32x32_k3", "br_ref_32x32_eps", "lace_ref_32x32_delta"):
    assert np.array_equal(out[nm + ".idx"], x.nzind)


# round 5: OMP with replacement (the driver) and the step-level functors SP / OMPR: x after the acquisition and after each of
# `steps` update! calls is what a host that steps sees; the fixture holds the LAST iterate (the numpy twin's step functions),
# cross-checked against the C oracle's driver where the driver has not stopped earlier.  params = [k, steps]
def ompr_case(name, A, b, k, delta):
    r = oc.ompr(A, b, k, delta)
    r2 = on.ompr(A, b, k, delta)
    agree(r[:2], r2[:2], name)
    assert r[2] == r2[2]
    add(name, "ompr", A, b, [k, delta, r[2]], r[:2])


def sp_steps_case(name, A, b, k, steps):
    idx, val = on.sp_acquisition(A, b, [], [], k)
    for t in range(steps):
        idx, val = on.sp_update(A, b, idx, val, k)
    c = oc.sp(A, b, k, 0.0, maxiter=steps)
    if c[2] == steps:
        agree((idx, val), c[:2], name)
    add(name, "sp_steps", A, b, [k, steps], (idx, val))


def ompr_steps_case(name, A, b, k, steps):
    idx, val = on.oblivious_acquisition(A, b, k)
    for t in range(steps):
        idx, val = on.ompr_update(A, b, idx, val)
    c = oc.ompr(A, b, k, 0.0, maxiter=steps)
    if c[2] == steps:
        agree((idx, val), c[:2], name)
    add(name, "ompr_steps", A, b, [k, steps], (idx, val))


A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=111)  # test/twostage.jl's shape
y = cs.perturb(b, 5e-3, rng=112)
ompr_case("ompr_ref_32x64", A, y, 3, 1e-2)
sp_steps_case("sp_steps_ref_32x64", A, y, 3, 2)
ompr_steps_case("ompr_steps_ref_32x64", A, y, 3, 2)
A, x, b = cs.sparse_data(n=64, m=256, k=10, rng=113, dtype=np.float32)
y = cs.perturb(b, 0.3, rng=114)  # heavy noise, two atoms fewer than planted: the exchanges have work to do
ompr_case("ompr_f32_64x256_k8", A, y, 8, 1e-6)
sp_steps_case("sp_steps_f32_64x256_k8", A, y, 8, 3)
ompr_steps_case("ompr_steps_f32_64x256_k8", A, y, 8, 4)

out["names"] = np.array(names)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
np.savez_compressed(path, **out)
print("wrote", path, len(names), "cases", os.path.getsize(path), "bytes")
