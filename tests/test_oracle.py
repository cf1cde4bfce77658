"""CPU tests of the oracle (oracle/): the reference's own known-answer tests re-run on seeded
data, the C restatement against its independent numpy twin and scikit-learn, the golden vectors,
and the edge semantics the C-ABI path is later held to."""
import numpy as np
import pytest

from oracle import oracle_np as on

EPS64 = float(np.finfo(np.float64).eps)
EPS32 = float(np.finfo(np.float32).eps)


def run_case(oracle, c):
    A, b, p = c["A"], c["b"], c["params"]
    if c["algo"] == "omp":
        return oracle.omp(A, b, int(p[0]), float(p[1]))
    if c["algo"] == "gomp":
        return oracle.gomp(A, b, int(p[0]), int(p[1]), float(p[2]))
    if c["algo"] == "mp":
        return oracle.mp(A, b, int(p[0]))
    if c["algo"] == "sp":
        return oracle.sp(A, b, int(p[0]), float(p[1]))
    if c["algo"] in ("br", "lace"):
        return oracle.br(A, b, float(p[0]), float(p[1]), int(p[2]), lace=c["algo"] == "lace")
    if c["algo"] == "rmp_k":
        return oracle.rmp(A, b, int(p[0]))
    if c["algo"] == "rmp_delta":
        return oracle.rmp(A, b, float(p[0]), int(p[1]))
    if c["algo"] == "foba":
        return oracle.foba(A, b, float(p[0]))
    if c["algo"] == "srr":
        return oracle.srr(A, b, int(p[0]), float(p[1]), -1, int(p[2]), int(p[3]))
    if c["algo"] == "fr":
        return oracle.fr(A, b, int(p[0]), float(p[1]), float(p[2]))
    if c["algo"] == "ompr":
        return oracle.ompr(A, b, int(p[0]), float(p[1]))
    if c["algo"] in ("sp_steps", "ompr_steps"):  # the functor's iterate after `steps` update! calls: the numpy twin's step functions
        from oracle import oracle_np
        k, steps = int(p[0]), int(p[1])
        if c["algo"] == "sp_steps":
            idx, val = oracle_np.sp_acquisition(A, b, [], [], k)
            for _ in range(steps):
                idx, val = oracle_np.sp_update(A, b, idx, val, k)
        else:
            idx, val = oracle_np.oblivious_acquisition(A, b, k)
            for _ in range(steps):
                idx, val = oracle_np.ompr_update(A, b, idx, val)
        return idx, val
    raise AssertionError(c["algo"])


def test_golden_vectors(oracle, golden):
    assert len(golden) >= 44
    for name, c in golden.items():
        r = run_case(oracle, c)
        assert np.array_equal(r[0], c["idx"]), name
        if np.all(np.isfinite(c["val"])):  # (gomp_dupcols is a singular LS problem: support only)
            np.testing.assert_allclose(r[1], c["val"], rtol=1e-10 if c["algo"].endswith("_steps") else 1e-12, atol=1e-15, err_msg=name)
        if c["algo"] in ("omp", "gomp", "fr"):
            assert np.array_equal(r[2], c["order"]), name
        if c["algo"] in ("sp", "ompr"):
            assert r[2] == int(c["params"][2]), name
        if c["algo"] == "srr":
            assert r[2] == int(c["params"][4]), name


# ---- the reference's known-answer tests (planted recovery), on seeded instances
def test_reference_mp_property(oracle, cs):
    # test/matchingpursuit.jl:15-19: mp(A,b,10k): A*xmp ~ b and xmp ~ x with atol = 3e-2
    ok = 0
    for seed in range(30):
        A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=1000 + seed)
        idx, val = oracle.mp(A, b, 30)
        xm = np.zeros(48)
        xm[idx] = val
        ok += np.allclose(A @ xm, b, atol=3e-2) and np.allclose(xm, x.to_dense(), atol=3e-2)
    assert ok >= 24  # "these tests may rarely fail" (test/matchingpursuit.jl:7)


def test_reference_omp_property(oracle, cs):
    # test/matchingpursuit.jl:21-30
    ok = okn = 0
    for seed in range(40):
        A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=2000 + seed)
        y = cs.perturb(b, 5e-3, rng=seed)
        idx, val, _ = oracle.omp(A, b, 3, EPS64)
        ok += np.array_equal(idx, x.nzind) and np.allclose(val, x.nzval, rtol=1.5e-8)
        idx, val, _ = oracle.omp(A, y, 3, EPS64)
        okn += np.array_equal(idx, x.nzind) and np.allclose(val, x.nzval, atol=2e-2)
    assert ok >= 36 and okn >= 36


def test_reference_gomp_property(oracle, cs):
    # test/matchingpursuit.jl:32-45 (l = 2)
    ok = 0
    for seed in range(40):
        A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=3000 + seed)
        idx, val, _ = oracle.gomp(A, b, 2, 3, EPS64)
        ok += np.array_equal(idx, x.nzind) and np.allclose(val, x.nzval, rtol=1.5e-8)
    assert ok >= 34


def test_reference_sp_property(oracle, cs):
    # test/twostage.jl:42-52
    ok = 0
    for seed in range(40):
        A, x, b = cs.gaussian_data(32, 64, 3, rng=4000 + seed)
        y = cs.perturb(b, 5e-3, rng=seed)
        idx, val, _ = oracle.sp(A, b, 3)
        good = np.array_equal(idx, x.nzind) and np.allclose(val, x.nzval, rtol=1.5e-8)
        idx, val, _ = oracle.sp(A, y, 3, 1e-2)
        good &= np.array_equal(idx, x.nzind) and np.allclose(val, x.nzval, atol=3e-2)
        ok += good
    assert ok >= 34


def test_reference_qr_solve_pin(oracle, cs):
    # test/forward.jl:23-28: AiQR \ y ~ A[:, nzind] \ y -- the only direct pin on the third-party QR
    for seed in range(5):
        A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=5000 + seed)
        y = cs.perturb(b, 1e-2, rng=seed)
        cols = np.array([0, 1, 2])
        np.testing.assert_allclose(oracle.lstsq_cols(A, cols, y), np.linalg.lstsq(A[:, cols], y, rcond=None)[0], rtol=1e-11)
    A, x, b = cs.sparse_data(n=128, m=64, k=3, rng=7)
    cols = np.array([5, 60, 3, 17, 40, 41, 2])
    np.testing.assert_allclose(oracle.lstsq_cols(A, cols, b), np.linalg.lstsq(A[:, cols], b, rcond=None)[0], rtol=1e-10, atol=1e-13)


# ---- C restatement against the independent numpy twin
@pytest.mark.parametrize("shape", [(32, 48, 3), (64, 256, 8), (37, 101, 5), (256, 1024, 16)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_c_vs_numpy_twin(oracle, cs, shape, dtype):
    n, m, k = shape
    eps = float(np.finfo(dtype).eps)
    for seed in range(3):
        A, x, b = cs.sparse_data(n=n, m=m, k=k, rng=seed * 7 + n, dtype=dtype)
        y = cs.perturb(b, 5e-3, rng=seed)
        a, t = oracle.omp(A, y, k, eps), on.omp(A, y, k, eps)
        assert np.array_equal(a[0], t[0]) and np.array_equal(a[2], t[2])
        np.testing.assert_allclose(a[1], t[1], rtol=1e-10)
        a, t = oracle.gomp(A, y, 3, k + 1, eps), on.gomp(A, y, 3, k + 1, eps)
        assert np.array_equal(a[0], t[0]) and np.array_equal(a[2], t[2])
        np.testing.assert_allclose(a[1], t[1], rtol=1e-9, atol=1e-12)
        a, t = oracle.mp(A, y, 2 * k), on.mp(A, y, 2 * k)
        assert np.array_equal(a[0], t[0])
        np.testing.assert_allclose(a[1], t[1], rtol=1e-10, atol=1e-13)
        if 2 * k <= n:
            a, t = oracle.sp(A, y, k), on.sp(A, y, k)
            assert np.array_equal(a[0], t[0]) and a[2] == t[2]
            np.testing.assert_allclose(a[1], t[1], rtol=1e-9, atol=1e-12)


def test_c256x1024_k32_config1(oracle, cs):
    # BASELINE config 1: omp(A,b,k) on Gaussian A 256x1024, k=32, Float64
    A, x, b = cs.sparse_data(n=256, m=1024, k=32, rng=123)
    y = cs.perturb(b, 5e-3, rng=124)
    a, t = oracle.omp(A, y, 32, EPS64), on.omp(A, y, 32, EPS64)
    assert np.array_equal(a[0], t[0]) and np.array_equal(a[2], t[2])
    np.testing.assert_allclose(a[1], t[1], rtol=1e-10)
    # LS optimality of the result on its own support: A_S' (b - A_S c) = 0
    r = oracle.residual(A, a[0], a[1], y)
    assert np.abs(A[:, a[0]].T @ r).max() < 1e-12


def test_against_sklearn_omp(oracle, cs):
    from sklearn.linear_model import orthogonal_mp
    for seed in range(4):
        A, x, b = cs.sparse_data(n=64, m=256, k=6, rng=600 + seed)
        y = cs.perturb(b, 5e-3, rng=seed)
        idx, val, _ = oracle.omp(A, y, 6, 0.0)
        w = orthogonal_mp(A, y, n_nonzero_coefs=6, precompute=False)
        assert np.array_equal(np.flatnonzero(w), idx)
        np.testing.assert_allclose(w[idx], val, rtol=1e-8)


# ---- step primitives and edge semantics
def test_sweep_and_topk_semantics(oracle):
    rng = np.random.default_rng(0)
    A = np.asfortranarray(rng.standard_normal((20, 30)))
    r = rng.standard_normal(20)
    out, best = oracle.sweep_abs(A, r)
    np.testing.assert_allclose(out, np.abs(A.T @ r), rtol=1e-13)
    assert best == int(np.argmax(out))
    v = np.array([1.0, 3.0, 3.0, 0.5, 3.0, 2.0])
    assert oracle.topk_desc(v, 4).tolist() == [1, 2, 4, 5]  # ties by ascending index
    assert oracle.topk_desc(v, 1).tolist() == [1]


def test_eps_negative_is_an_error(oracle, cs):
    A, x, b = cs.sparse_data(n=16, m=24, k=2, rng=0)
    with pytest.raises(ValueError):
        oracle.omp(A, b, 2, -1.0)
    with pytest.raises(ValueError):
        oracle.gomp(A, b, 2, 2, -1e-3)
    with pytest.raises(ValueError):
        oracle.sp(A, b, 9)  # 2k > M


def test_f32_dictionary_is_promoted_exactly(oracle, cs):
    A, x, b = cs.sparse_data(n=64, m=256, k=8, rng=5, dtype=np.float32)
    y = cs.perturb(b, 5e-3, rng=6)
    a = oracle.omp(A, y, 8, EPS32)
    t = oracle.omp(A.astype(np.float64), y, 8, EPS32)
    assert np.array_equal(a[0], t[0]) and np.array_equal(a[1], t[1])


def test_mp_warm_start(oracle, cs):
    A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=9)
    i1, v1 = oracle.mp(A, b, 5)
    i2, v2 = oracle.mp(A, b, 7, x0=(i1, v1))
    i3, v3 = oracle.mp(A, b, 12)
    assert np.array_equal(i2, i3)
    np.testing.assert_allclose(v2, v3, rtol=1e-12, atol=1e-15)


def test_reference_ompr_property_and_twin(oracle, cs):
    # test/twostage.jl "OMP with replacement": ompr(A,b,k,1e-6) recovers the planted x; C vs numpy twin
    ok = 0
    for seed in range(30):
        A, x, b = cs.gaussian_data(32, 64, 3, rng=7000 + seed)
        y = cs.perturb(b, 5e-3, rng=seed)
        a, t = oracle.ompr(A, b, 3, 1e-6), on.ompr(A, b, 3, 1e-6)
        assert np.array_equal(a[0], t[0]) and a[2] == t[2]
        np.testing.assert_allclose(a[1], t[1], rtol=1e-9, atol=1e-12)
        a2, t2 = oracle.ompr(A, y, 3, 1e-2), on.ompr(A, y, 3, 1e-2)
        assert np.array_equal(a2[0], t2[0]) and a2[2] == t2[2]
        ok += np.array_equal(a[0], x.nzind) and np.allclose(a[1], x.nzval, rtol=1.5e-8)
    assert ok >= 26
    A, x, b = cs.sparse_data(n=128, m=512, k=12, rng=5, dtype=np.float32)
    y = cs.perturb(b, 5e-3, rng=6)
    a, t = oracle.ompr(A, y, 12, 1e-6), on.ompr(A, y, 12, 1e-6)
    assert np.array_equal(a[0], t[0]) and a[2] == t[2] and a[2] >= 2


def test_reference_fr_property_and_twin(oracle, cs):
    """test/forward.jl:14-22 (planted 3-sparse recovery at 32 x 48, noiseless and perturbed by 1e-2) on
    seeded data; and the C restatement (rescaling downdated one Q column per step) against the numpy
    twin (rescaling recomputed from a fresh QR at every step, as ols_rescaling! does)."""
    from oracle import oracle_np
    ok = 0
    for seed in range(40):
        A, x, b = cs.sparse_data(n=32, m=48, k=3, rng=3000 + seed)
        y = cs.perturb(b, 1e-2, rng=seed)
        i1, v1, o1 = oracle.fr(A, b, 3)
        i2, v2, o2 = oracle.fr(A, y, 3)
        t1, t2 = oracle_np.fr(A, b, 3), oracle_np.fr(A, y, 3)
        assert np.array_equal(o1, t1[2]) and np.array_equal(o2, t2[2])
        np.testing.assert_allclose(v2, t2[1], rtol=1e-10)
        ok += (np.array_equal(i1, x.nzind) and np.allclose(v1, x.nzval) and np.array_equal(i2, x.nzind)
               and np.allclose(v2, x.nzval, atol=2e-2))
    assert ok >= 36  # "may rarely fail" (test/matchingpursuit.jl:7)
    for (n, m, k, dt) in [(64, 256, 12, np.float64), (100, 333, 20, np.float32), (256, 1024, 40, np.float32)]:
        A, x, b = cs.sparse_data(n=n, m=m, k=k, rng=n + m, dtype=dt)
        y = cs.perturb(b, 5e-3, rng=1)
        for kw in ({}, {"max_eps": 0.05}, {"min_delta": 0.05}):
            a, t = oracle.fr(A, y, k + 5, **kw), oracle_np.fr(A, y, k + 5, **kw)
            assert np.array_equal(a[2], t[2]), (n, m, kw)
            np.testing.assert_allclose(a[1], t[1], rtol=1e-9, atol=1e-12)
    # k >= M: the nnz(x) < size(A,1) guard ends the loop (src/forward.jl:58)
    A, x, b = cs.sparse_data(n=6, m=20, k=2, rng=5)
    assert len(oracle.fr(A, cs.perturb(b, 0.1, rng=2), 15)[0]) <= 6


def test_reference_srr_property_and_twin(oracle, cs):
    """test/twostage.jl:11-39 (planted recovery with srr, noiseless / noisy / k = 1 / l = k) on seeded data,
    and the C restatement against the numpy twin (dense inverse for the backward scores)."""
    from oracle import oracle_np
    ok = 0
    for seed in range(20):
        A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=4000 + seed)
        y = cs.perturb(b, 5e-3, rng=seed)
        good = True
        for bb, l in ((b, 1), (y, 1), (b, 3), (y, 3)):
            r = oracle.srr(A, bb, 3, l=l)
            t = oracle_np.srr(A, bb, 3, l=l)
            assert np.array_equal(r[0], t[0]) and r[2] == t[2]
            np.testing.assert_allclose(r[1], t[1], rtol=1e-9, atol=1e-12)
            good &= np.array_equal(r[0], x.nzind) and np.allclose(r[1], x.nzval, atol=3e-2)
        x1 = cs.sparse_vector(64, 1, rng=seed)
        r = oracle.srr(A, A[:, x1.nzind] @ x1.nzval, 1)
        good &= np.array_equal(r[0], x1.nzind) and np.allclose(r[1], x1.nzval)
        ok += good
    assert ok >= 18
    for (n, m, k, l, init) in [(128, 512, 12, 1, 1), (128, 512, 8, 4, 2), (100, 333, 9, 2, 1), (256, 1024, 20, 1, 2)]:
        A, x, b = cs.sparse_data(n=n, m=m, k=k + 2, rng=n + k, dtype=np.float32)
        y = cs.perturb(b, 1e-1, rng=3)
        r, t = oracle.srr(A, y, k, 1e-12, -1, init, l), oracle_np.srr(A, y, k, 1e-12, None, init, l)
        assert np.array_equal(r[0], t[0]) and r[2] == t[2], (n, m, k, l, init)
        np.testing.assert_allclose(r[1], t[1], rtol=1e-8, atol=1e-12)
        # initialization = 3 (random_acquisition!, src/matchingpursuit.jl:195-204) on a fixed draw: C restatement == twin,
        # and the draw's order does not matter (sort!(ind), :197)
        draw = np.random.default_rng(n + l).choice(m, size=k, replace=False)
        r = oracle.srr(A, y, k, 1e-12, -1, 3, l, init=draw)
        t = oracle_np.srr(A, y, k, 1e-12, None, 3, l, init=draw)
        assert np.array_equal(r[0], t[0]) and r[2] == t[2], (n, m, k, l, "random")
        np.testing.assert_allclose(r[1], t[1], rtol=1e-8, atol=1e-12)
        r2 = oracle.srr(A, y, k, 1e-12, -1, 3, l, init=draw[::-1].copy())
        assert np.array_equal(r[0], r2[0]) and np.array_equal(r[1], r2[1])


def test_reference_rmp_foba_property_and_twin(oracle, cs):
    """test/stepwise.jl:11-39 (planted recovery with rmp(A,b,k), rmp(A,y,δ), rmp(A,y,δ,3), foba(A,·,δ)) on seeded
    data, and the C restatement against the numpy twin."""
    from oracle import oracle_np
    ok = 0
    for seed in range(16):
        A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=5000 + seed)
        y = cs.perturb(b, 1e-2, rng=seed)
        good = True
        for f in (lambda o: o.rmp(A, y, 3), lambda o: o.rmp(A, y, 1e-2), lambda o: o.rmp(A, y, 1e-2, 3),
                  lambda o: o.foba(A, b, 1e-2), lambda o: o.foba(A, y, 1e-2)):
            r, t = f(oracle), f(oracle_np)
            assert np.array_equal(r[0], t[0])
            np.testing.assert_allclose(r[1], t[1], rtol=1e-8, atol=1e-12)
            good &= np.array_equal(r[0], x.nzind) and np.allclose(r[1], x.nzval, atol=2e-2)
        ok += good
    assert ok >= 14
    A, x, b = cs.sparse_data(n=128, m=512, k=10, rng=3, dtype=np.float32)
    y = cs.perturb(b, 5e-2, rng=4)
    for f in (lambda o: o.rmp(A, y, 0.05), lambda o: o.rmp(A, y, 0.05, 3), lambda o: o.foba(A, y, 0.05), lambda o: o.rmp(A, y, 12)):
        r, t = f(oracle), f(oracle_np)
        assert np.array_equal(r[0], t[0])
        np.testing.assert_allclose(r[1], t[1], rtol=1e-8, atol=1e-12)


def test_reference_br_lace_property_and_twin(oracle, cs):
    """test/backward.jl:17-41 (br / lace / fbr with sparsity = k, max_residual = δ, max_increase = δ on a square
    32 x 32 system) on seeded data, and the C restatement against the numpy twin (whose LACE step measures
    the residual increase by re-solving, as the reference does)."""
    from oracle import oracle_np
    ok = tot = 0
    for seed in range(12):
        A, x, b = cs.sparse_data(n=32, m=32, k=3, rng=6000 + seed)
        y = cs.perturb(b, 5e-3, rng=seed)
        for lace in (False, True):
            for kw in (dict(k=3), dict(max_eps=1e-2), dict(max_delta=1e-2)):
                r, t = oracle.br(A, y, lace=lace, **kw), oracle_np.br(A, y, lace=lace, **kw)
                assert np.array_equal(r[0], t[0])
                np.testing.assert_allclose(r[1], t[1], rtol=1e-7, atol=1e-10)
                tot += 1
                ok += np.array_equal(r[0], x.nzind) and np.allclose(r[1], x.nzval, atol=2e-2)
    assert ok >= tot - 4
    A, x, b = cs.sparse_data(n=200, m=150, k=12, rng=5, dtype=np.float32)
    y = cs.perturb(b, 5e-2, rng=4)
    for kw in (dict(k=12), dict(max_eps=0.06), dict(max_delta=0.02), dict(k=20, lace=True), dict(max_eps=0.06, lace=True)):
        r, t = oracle.br(A, y, **kw), oracle_np.br(A, y, **kw)
        assert np.array_equal(r[0], t[0]), kw
