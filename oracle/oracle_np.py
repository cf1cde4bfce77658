"""Independent numpy twin of the C oracle (csmp_oracle.c): same reference semantics, different code.

TEST INFRASTRUCTURE ONLY (see csmp_oracle.h).  Written separately from the C restatement so
that the two can be diffed against each other: this one re-solves the least-squares problem
from scratch with LAPACK (numpy.linalg.lstsq) at every step instead of updating a QR, and
uses numpy's BLAS products for the sweep.  Small/medium sizes only.

Reference lines restated (paths relative to /root/reference):
  mp    src/matchingpursuit.jl:26-40      omp   src/matchingpursuit.jl:62-82
  gomp  src/matchingpursuit.jl:116-139    sp    src/twostage.jl:54-107
  helpers src/matchingpursuit.jl:152-193, src/util.jl:118-134
  fr    src/forward.jl:44-114
All arithmetic is Float64 on exactly promoted inputs; indices are 0-based.
"""
import numpy as np


def _f64(A, b):
    return np.asarray(A, dtype=np.float64), np.asarray(b, dtype=np.float64)


def _residual(A, b, idx, val):  # residual!: src/matchingpursuit.jl:158-161
    if len(idx) == 0:
        return b.copy()
    return b - A[:, idx] @ val


def _abs_corr(A, r):  # argmaxinner!: src/matchingpursuit.jl:182-183
    return np.abs(A.T @ r)


def _topk(v, k):  # partialsortperm(v, 1:k, rev=true): desc by value, ties by ascending index
    order = np.lexsort((np.arange(len(v)), -v))
    return order[:k]


def _ls(A, idx, b):  # AiQR \ b on the sorted support
    return np.linalg.lstsq(A[:, idx], b, rcond=None)[0]


def omp(A, b, k, eps):
    A, b = _f64(A, b)
    if not eps >= 0:
        raise ValueError("eps has to be non-negative")  # :74
    M, N = A.shape
    idx = np.zeros(0, np.int64)
    val = np.zeros(0)
    order = []
    for _ in range(k):  # :77
        progressed = False
        if len(idx) < M:  # :63
            r = _residual(A, b, idx, val)
            i = int(np.argmax(_abs_corr(A, r)))  # first max, :184
            if i not in idx:  # :66
                idx = np.sort(np.append(idx, i))  # util.jl:120-122
                order.append(i)
                val = _ls(A, idx, b)  # :68
                progressed = True
        if not (np.linalg.norm(_residual(A, b, idx, val)) >= eps):  # :79
            break
        if not progressed:  # a no-op step repeats forever: same result as running it out
            break
    return idx, val, np.array(order, np.int64)


def gomp(A, b, l, k, eps):
    A, b = _f64(A, b)
    if not eps >= 0:
        raise ValueError("eps has to be non-negative")  # :127
    M, N = A.shape
    idx = np.zeros(0, np.int64)
    val = np.zeros(0)
    order = []

    def update(idx, val, l):  # :116-123
        if not len(idx) < M:
            return idx, val
        r = _residual(A, b, idx, val)
        for i in _topk(_abs_corr(A, r), l):  # :119, util.jl:129-134 (skip duplicates)
            if i not in idx and len(idx) < M:
                idx = np.sort(np.append(idx, i))
                order.append(int(i))
        val = _ls(A, idx, b)
        return idx, val

    for _ in range(k // l):  # :130
        idx, val = update(idx, val, l)
        if not (np.linalg.norm(_residual(A, b, idx, val)) >= eps):  # :132
            break
    if k % l > 0:  # :134-137 (runs even after an eps-break)
        idx, val = update(idx, val, k % l)
    return idx, val, np.array(order, np.int64)


def mp(A, b, k, x0=None):
    A, b = _f64(A, b)
    x = {}
    if x0 is not None:
        for i, v in zip(*x0):
            x[int(i)] = float(v)
    for _ in range(k):  # :36
        idx = np.array(sorted(x), np.int64)
        val = np.array([x[i] for i in idx])
        r = _residual(A, b, idx, val)  # :27
        i = int(np.argmax(_abs_corr(A, r)))  # :28
        d = float(A[:, i] @ r)  # :29
        if i in x:
            x[i] += d
        elif d != 0.0:
            x[i] = d
    idx = np.array(sorted(x), np.int64)
    return idx, np.array([x[i] for i in idx])


def sp_acquisition(A, b, idx, val, k):
    """sp_acquisition!(P, x, k): src/twostage.jl:67-72 -- (idx, val) -> the union with the k atoms best correlated with the
    residual of x, and the least-squares solution on it"""
    A, b = _f64(A, b)
    idx, val = np.asarray(idx, np.int64), np.asarray(val, np.float64)
    r = _residual(A, b, idx, val)
    idx = np.union1d(idx, _topk(_abs_corr(A, r), k)).astype(np.int64)
    return idx, _ls(A, idx, b)


def sp_update(A, b, idx, val, k):
    """update!(P::SP, x): src/twostage.jl:75-83 (nnz(x) == k is the caller's to hold, :76)"""
    A, b = _f64(A, b)
    idx, val = sp_acquisition(A, b, idx, val, k)  # :77
    drop = len(idx) - k
    if drop > 0:  # :78-81
        kill = np.lexsort((np.arange(len(val)), np.abs(val)))[:drop]
        keep = np.setdiff1d(np.arange(len(idx)), kill)
        idx = idx[keep]
    return idx, _ls(A, idx, b)  # :82


def sp(A, b, k, delta=1e-12, maxiter=None):
    A, b = _f64(A, b)
    M, N = A.shape
    if 2 * k > M:
        raise ValueError("2k > length(b) is invalid for Subspace Pursuit")  # twostage.jl:55
    if maxiter is None:
        maxiter = 16 * k
    idx, val = sp_acquisition(A, b, np.zeros(0, np.int64), np.zeros(0), k)  # :90
    resnorm = np.linalg.norm(_residual(A, b, idx, val))
    iters = 0
    for _ in range(maxiter):  # :92
        oldnorm = resnorm
        idx, val = sp_update(A, b, idx, val, k)
        iters += 1
        resnorm = np.linalg.norm(_residual(A, b, idx, val))
        if resnorm <= delta or oldnorm <= resnorm:  # :96
            break
    return idx, val, iters


def oblivious_acquisition(A, b, k):
    """oblivious_acquisition!(P, x, k) on an empty x: src/matchingpursuit.jl:207-216"""
    A, b = _f64(A, b)
    idx = np.sort(_topk(_abs_corr(A, b), k)).astype(np.int64)
    return idx, _ls(A, idx, b)


def ompr_update(A, b, idx, val):
    """update!(P::OMPR, x) with eta = 1: src/twostage.jl:134-180"""
    A, b = _f64(A, b)
    idx, val = np.asarray(idx, np.int64), np.asarray(val, np.float64)
    N = A.shape[1]
    r = _residual(A, b, idx, val)
    Ar = A.T @ r  # eta = 1
    Ar[idx] += val  # copy!(P.Ar, x); mul!(P.Ar, A', r, eta, 1)
    mask = np.ones(N, bool)
    mask[idx] = False
    cand = np.abs(Ar) * mask
    if cand.max() > 0:
        i = int(np.argmax(cand))  # first maximum among atoms outside the support
        idx2 = np.sort(np.append(idx, i))
        v2 = Ar[idx2]
        j = int(np.argmin(np.abs(v2)))  # first minimum
        idx = np.delete(idx2, j)
        val = _ls(A, idx, b)
    return idx, val


def ompr(A, b, k, delta, maxiter=None):
    """OMP with replacement: src/twostage.jl:110-202 (x starts empty)."""
    A, b = _f64(A, b)
    M, N = A.shape
    if maxiter is None:
        maxiter = M  # :185
    idx, val = oblivious_acquisition(A, b, k)
    resnorm = np.linalg.norm(_residual(A, b, idx, val))
    iters = 0
    for _ in range(maxiter):
        oldnorm = resnorm
        idx, val = ompr_update(A, b, idx, val)
        iters += 1
        resnorm = np.linalg.norm(_residual(A, b, idx, val))
        if resnorm <= delta or oldnorm <= resnorm:
            break
    return idx, val, iters


def fr(A, b, k, max_eps=0.0, min_delta=0.0):
    """Forward regression / OLS: src/forward.jl:44-114, with the rescaling recomputed from scratch
    at every step exactly as ols_rescaling! does (:99-114): |a_j|^2 - |Q' a_j|^2 with a fresh QR
    of the active columns."""
    A, b = _f64(A, b)
    M, N = A.shape
    idx = np.zeros(0, np.int64)
    val = np.zeros(0)
    order = []
    for _ in range(k):  # :51
        if not len(idx) < M:  # :58
            break
        r = _residual(A, b, idx, val)
        if not np.linalg.norm(r) > max_eps:  # :60-61
            break
        resc = np.sum(A * A, axis=0)  # :108
        if len(idx):
            Q = np.linalg.qr(A[:, idx])[0]
            resc = resc - np.sum((Q.T @ A) ** 2, axis=0)  # :104,:109-113
        with np.errstate(divide="ignore", invalid="ignore"):
            d2 = (A.T @ r) ** 2 / resc  # :77-79
        d2[idx] = 0.0  # :80
        d2 = np.where(np.isnan(d2), -1.0, d2)  # a NaN never wins (see csmp_oracle.c)
        i = int(np.argmax(d2))  # findmax: first maximum
        if not min_delta ** 2 < d2[i]:  # :64
            break
        idx = np.sort(np.append(idx, i))
        order.append(i)
        val = _ls(A, idx, b)  # :67
    return idx, val, np.array(order, np.int64)


def srr(A, b, k, delta=1e-12, maxiter=None, initialization=1, l=1, init=None):
    """Stepwise regression with replacement: src/twostage.jl:3-33 (x starts empty).  Forward scores as
    in fr() above; backward scores x_i^2 / diag(inv(As'As))_i (src/backward.jl:70-83) from a dense
    inverse -- independent of the C restatement's triangular solves."""
    A, b = _f64(A, b)
    M, N = A.shape
    if maxiter is None:
        maxiter = 4 * k
    norm2 = np.sum(A * A, axis=0)

    def fit(idx):
        val = _ls(A, idx, b) if len(idx) else np.zeros(0)
        return val, _residual(A, b, idx, val)

    def forward(idx, val, r, guarded):
        if not len(idx) < M:
            return idx, val, r, False
        if guarded and not np.linalg.norm(r) > 0:
            return idx, val, r, False
        resc = norm2.copy()
        if len(idx):
            Q = np.linalg.qr(A[:, idx])[0]
            resc = resc - np.sum((Q.T @ A) ** 2, axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            d2 = (A.T @ r) ** 2 / resc
        d2[idx] = 0.0
        d2 = np.where(np.isnan(d2), -1.0, d2)
        i = int(np.argmax(d2))
        if (guarded and not 0 < d2[i]) or i in idx:
            return idx, val, r, False
        idx = np.sort(np.append(idx, i))
        val, r = fit(idx)
        return idx, val, r, True

    def backward(idx, val, r):
        As = A[:, idx]
        gamma = np.diag(np.linalg.inv(As.T @ As))
        j = int(np.argmin(val ** 2 / gamma))  # first minimum in nzind order
        idx = np.delete(idx, j)
        val, r = fit(idx)
        return idx, val, r

    idx = np.zeros(0, np.int64)
    val, r = fit(idx)
    if initialization == 1:
        idx = np.sort(_topk(_abs_corr(A, b), k)).astype(np.int64)
        val, r = fit(idx)
    elif initialization == 3:  # random_acquisition! (src/matchingpursuit.jl:195-204) with the caller's draw
        idx = np.sort(np.asarray(init, np.int64))
        val, r = fit(idx)
    else:
        for _ in range(k):
            idx, val, r, _ok = forward(idx, val, r, False)
    resnorm = np.linalg.norm(r)
    iters = 0
    for _ in range(maxiter):
        oldnorm = resnorm
        for _s in range(l):
            idx, val, r, ok = forward(idx, val, r, True)
            if not ok:
                break
        while len(idx) > k:
            idx, val, r = backward(idx, val, r)
        resnorm = np.linalg.norm(r)
        iters += 1
        if resnorm <= delta or oldnorm <= resnorm:
            break
    return idx, val, iters


class _Stepwise:
    """StepwiseRegression object (src/forward.jl:14-32) with everything recomputed from scratch per step."""

    def __init__(self, A, b):
        self.A, self.b = _f64(A, b)
        self.M, self.N = self.A.shape
        self.norm2 = np.sum(self.A * self.A, axis=0)
        self.idx = np.zeros(0, np.int64)
        self.val = np.zeros(0)
        self.r = self.b.copy()
        self.last_max_d2 = 0.0

    def _fit(self):
        self.val = _ls(self.A, self.idx, self.b) if len(self.idx) else np.zeros(0)
        self.r = _residual(self.A, self.b, self.idx, self.val)

    def forward(self, max_eps, min_delta):  # forward_step!: src/forward.jl:56-73
        A = self.A
        if not len(self.idx) < self.M or not np.linalg.norm(self.r) > max_eps:
            return False
        resc = self.norm2.copy()
        if len(self.idx):
            Q = np.linalg.qr(A[:, self.idx])[0]
            resc = resc - np.sum((Q.T @ A) ** 2, axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            d2 = (A.T @ self.r) ** 2 / resc
        d2[self.idx] = 0.0
        d2 = np.where(np.isnan(d2), -1.0, d2)
        i = int(np.argmax(d2))
        self.last_max_d2 = float(d2[i])
        if not min_delta ** 2 < d2[i] or i in self.idx:
            return False
        self.idx = np.sort(np.append(self.idx, i))
        self._fit()
        return True

    def backward(self, max_eps, max_delta):  # backward_step!: src/backward.jl:51-67
        if not len(self.idx) > 0:
            return False
        As = self.A[:, self.idx]
        gamma = np.diag(np.linalg.inv(As.T @ As))
        d2 = self.val ** 2 / gamma
        j = int(np.argmin(d2))
        if not (np.sqrt(d2[j] + np.linalg.norm(self.r) ** 2) < max_eps and d2[j] < max_delta ** 2):
            return False
        self.idx = np.delete(self.idx, j)
        self._fit()
        return True

    def backward_lace(self, max_eps, max_delta):  # backward_step!(P::LACE, ...): src/backward.jl:247-270
        if not len(self.idx) > 0:
            return False
        normr = np.linalg.norm(self.r)
        j = int(np.argmin(np.abs(self.val)))
        keep_idx, keep_val, keep_r = self.idx, self.val, self.r
        self.idx = np.delete(self.idx, j)
        self._fit()
        d2 = np.linalg.norm(self.r) ** 2 - normr ** 2  # measured, as the reference does
        if np.sqrt(normr ** 2 + d2) < max_eps and d2 < max_delta ** 2:
            return True
        self.idx, self.val, self.r = keep_idx, keep_val, keep_r
        return False

    def dense(self):
        x = np.zeros(self.N)
        x[self.idx] = self.val
        return x


def rmp(A, b, delta_or_k, maxiter=1):
    """src/stepwise.jl:5-43: rmp(A,b,δ,maxiter) for a float, rmp(A,b,k) for an int."""
    P = _Stepwise(A, b)
    if isinstance(delta_or_k, (int, np.integer)):
        k = int(delta_or_k)
        for _ in range(P.M):
            if not P.forward(0.0, 0.0):
                break
        for _ in range(len(P.idx), k, -1):
            if not P.backward(np.inf, np.inf):
                break
        return P.idx, P.val
    delta = float(delta_or_k)
    xt = P.dense()
    for _ in range(maxiter):
        for _ in range(P.M):
            if not P.forward(0.0, delta):
                break
        if np.allclose(xt, P.dense(), rtol=np.sqrt(np.finfo(float).eps), atol=0) and np.linalg.norm(xt - P.dense()) <= np.sqrt(np.finfo(float).eps) * max(np.linalg.norm(xt), np.linalg.norm(P.dense())):
            break
        xt = P.dense()
        for _ in range(len(P.idx), 0, -1):
            if not P.backward(np.inf, delta):
                break
        if np.linalg.norm(xt - P.dense()) <= np.sqrt(np.finfo(float).eps) * max(np.linalg.norm(xt), np.linalg.norm(P.dense())):
            break
        xt = P.dense()
    return P.idx, P.val


def foba(A, b, delta):
    """src/stepwise.jl:47-56."""
    P = _Stepwise(A, b)
    for _ in range(P.M):
        if not P.forward(0.0, delta):
            break
        half = np.sqrt(P.last_max_d2) / 2
        while P.backward(np.inf, half):
            pass
    return P.idx, P.val


def br(A, b, max_eps=np.inf, max_delta=np.inf, k=0, lace=False):
    """br / fbr / lace: src/backward.jl:27-35,154-162,233-242 (all N <= M columns, then backward steps)."""
    P = _Stepwise(A, b)
    assert P.N <= P.M
    P.idx = np.arange(P.N, dtype=np.int64)
    P._fit()
    for _ in range(P.N, k, -1):
        if not (P.backward_lace(max_eps, max_delta) if lace else P.backward(max_eps, max_delta)):
            break
    return P.idx, P.val
