"""ctypes binding of oracle/libcsmp_oracle.so (the C restatement, csmp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  Parity-pin status: see csmp_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64 = C.c_int64
p_i64 = C.POINTER(C.c_int64)
p_f64 = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libcsmp_oracle.so")
    src = os.path.join(_HERE, "csmp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("CSMP_ORACLE_SO")  # tools/sanitize_cpu.sh: the same source built with gcc's sanitizers
        if not so:
            so = os.path.join(_HERE, "libcsmp_oracle.so")
            if not os.path.exists(so):
                build()
        _LIB = C.CDLL(so)
        _LIB.cso_sweep_abs.restype = i64
    return _LIB


def _prep(A, b):
    A = np.asarray(A)
    if A.dtype not in (np.float32, np.float64):
        A = A.astype(np.float64)
    A = np.asfortranarray(A)
    M, N = A.shape
    dtype = 0 if A.dtype == np.float32 else 1
    b = np.ascontiguousarray(np.asarray(b), dtype=np.float64)  # exact promotion
    assert b.shape == (M,)
    return A, b, M, N, dtype


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def omp(A, b, k, eps, nthreads=0):
    """-> (idx sorted 0-based, val, order).  Raises ValueError for eps < 0 (reference throws)."""
    A, b, M, N, dtype = _prep(A, b)
    cap = max(int(k), 1)
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    order = np.zeros(cap, np.int64)
    nnz = i64(0)
    rc = lib().cso_omp(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), C.c_double(eps),
                       _vp(idx), _vp(val), C.byref(nnz), _vp(order), int(nthreads))
    if rc == -1:
        raise ValueError("eps has to be non-negative")
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy(), order[:n].copy()


def gomp(A, b, l, k, eps, nthreads=0):
    A, b, M, N, dtype = _prep(A, b)
    cap = max(int(k) + int(l), 1)
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    order = np.zeros(cap, np.int64)
    nnz = i64(0)
    rc = lib().cso_gomp(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(l)), i64(int(k)),
                        C.c_double(eps), _vp(idx), _vp(val), C.byref(nnz), _vp(order), int(nthreads))
    if rc == -1:
        raise ValueError("eps has to be non-negative")
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy(), order[:n].copy()


def mp(A, b, k, x0=None, nthreads=0):
    A, b, M, N, dtype = _prep(A, b)
    if x0 is None:
        i0 = np.zeros(0, np.int64)
        v0 = np.zeros(0, np.float64)
    else:
        i0 = np.ascontiguousarray(x0[0], dtype=np.int64)
        v0 = np.ascontiguousarray(x0[1], dtype=np.float64)
    cap = max(int(k) + len(i0), 1)
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    nnz = i64(0)
    rc = lib().cso_mp(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), _vp(i0), _vp(v0),
                      i64(len(i0)), _vp(idx), _vp(val), C.byref(nnz), int(nthreads))
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy()


def sp(A, b, k, delta=1e-12, maxiter=-1, nthreads=0):
    A, b, M, N, dtype = _prep(A, b)
    cap = max(2 * int(k), 1)
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    nnz = i64(0)
    iters = i64(0)
    rc = lib().cso_sp(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), C.c_double(delta),
                      i64(int(maxiter)), _vp(idx), _vp(val), C.byref(nnz), C.byref(iters), int(nthreads))
    if rc == -3:
        raise ValueError("2k > length(b) is invalid for Subspace Pursuit")
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy(), iters.value


def ompr(A, b, k, delta, maxiter=-1, nthreads=0):
    A, b, M, N, dtype = _prep(A, b)
    idx = np.zeros(max(int(k), 1) + 1, np.int64)
    val = np.zeros(max(int(k), 1) + 1, np.float64)
    nnz = i64(0)
    iters = i64(0)
    if nthreads <= 0:
        nthreads = min(16, os.cpu_count() or 1)  # (GPU-pool boxes: 256 logical CPUs, 16-CPU cgroup quota)
    rc = lib().cso_ompr(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), C.c_double(delta),
                        i64(int(maxiter)), _vp(idx), _vp(val), C.byref(nnz), C.byref(iters), int(nthreads))
    if rc == -3:
        raise ValueError("k out of range")
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy(), iters.value


def fr(A, b, k, max_eps=0.0, min_delta=0.0, nthreads=0):
    """fr/ols/oomp/ormp (src/forward.jl:44-54) -> (idx sorted 0-based, val, order)."""
    A, b, M, N, dtype = _prep(A, b)
    cap = max(int(k), 1)
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    order = np.zeros(cap, np.int64)
    nnz = i64(0)
    rc = lib().cso_fr(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), C.c_double(max_eps),
                      C.c_double(min_delta), _vp(idx), _vp(val), C.byref(nnz), _vp(order), int(nthreads))
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy(), order[:n].copy()


def srr(A, b, k, delta=1e-12, maxiter=-1, initialization=1, l=1, nthreads=0, init=None):
    """srr (src/twostage.jl:3-33) -> (idx sorted 0-based, val, iters).  initialization = 3 takes the k atoms of
    random_acquisition! (src/matchingpursuit.jl:195-204) from `init`: the draw is the caller's."""
    A, b, M, N, dtype = _prep(A, b)
    cap = int(k) + int(l) + 1
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    nnz = i64(0)
    iters = i64(0)
    if initialization == 3:
        init = np.ascontiguousarray(init, np.int64)
        assert init.size == int(k)
        rc = lib().cso_srr_from(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), C.c_double(delta),
                                i64(int(maxiter)), _vp(init), i64(int(l)), _vp(idx), _vp(val), C.byref(nnz),
                                C.byref(iters), int(nthreads))
        if rc == -3:
            raise ValueError("k / l out of range")
        assert rc == 0, rc
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), iters.value
    rc = lib().cso_srr(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), i64(int(k)), C.c_double(delta),
                       i64(int(maxiter)), int(initialization), i64(int(l)), _vp(idx), _vp(val), C.byref(nnz),
                       C.byref(iters), int(nthreads))
    if rc == -3:
        raise ValueError("k / l out of range")
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy(), iters.value


def _stepwise(fn, A, b, *args, nthreads=0):
    A, b, M, N, dtype = _prep(A, b)
    cap = min(M, N) + 1
    idx = np.zeros(cap, np.int64)
    val = np.zeros(cap, np.float64)
    nnz = i64(0)
    rc = fn(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(b), *args, _vp(idx), _vp(val), C.byref(nnz), int(nthreads))
    assert rc == 0, rc
    n = nnz.value
    return idx[:n].copy(), val[:n].copy()


def rmp(A, b, delta_or_k, maxiter=1, nthreads=0):
    """rmp(A,b,δ,maxiter) for a float second argument, rmp(A,b,k) for an int (src/stepwise.jl:5-43)."""
    if isinstance(delta_or_k, (int, np.integer)):
        return _stepwise(lib().cso_rmp_k, A, b, i64(int(delta_or_k)), nthreads=nthreads)
    return _stepwise(lib().cso_rmp_delta, A, b, C.c_double(float(delta_or_k)), i64(int(maxiter)), nthreads=nthreads)


def foba(A, b, delta, nthreads=0):
    return _stepwise(lib().cso_foba, A, b, C.c_double(float(delta)), nthreads=nthreads)


def br(A, b, max_eps=np.inf, max_delta=np.inf, k=0, lace=False, nthreads=0):
    """br / fbr (lace=False) and lace (lace=True): src/backward.jl:27-35,154-162,233-270."""
    return _stepwise(lib().cso_br, A, b, C.c_double(float(max_eps)), C.c_double(float(max_delta)), i64(int(k)),
                     int(bool(lace)), nthreads=nthreads)


def sweep_abs(A, r, nthreads=0):
    A, r, M, N, dtype = _prep(A, r)
    out = np.zeros(N, np.float64)
    best = lib().cso_sweep_abs(_vp(A), dtype, i64(M), i64(N), i64(M), _vp(r), _vp(out), int(nthreads))
    return out, int(best)


def topk_desc(v, k):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(min(int(k), len(v)), np.int64)
    lib().cso_topk_desc(_vp(v), i64(len(v)), i64(int(k)), _vp(out))
    return out


def lstsq_cols(A, cols, b):
    A, b, M, N, dtype = _prep(A, b)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    coef = np.zeros(len(cols), np.float64)
    rc = lib().cso_lstsq_cols(_vp(A), dtype, i64(M), i64(M), _vp(cols), i64(len(cols)), _vp(b), _vp(coef))
    assert rc == 0
    return coef


def residual(A, idx, val, b):
    A, b, M, N, dtype = _prep(A, b)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float64)
    r = np.zeros(M, np.float64)
    lib().cso_residual(_vp(A), dtype, i64(M), i64(M), _vp(idx), _vp(val), i64(len(idx)), _vp(b), _vp(r))
    return r
