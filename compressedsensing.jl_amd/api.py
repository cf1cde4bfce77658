"""Host-side mirror of the reference's matching-pursuit interface, over libcsmp.so.

Same names, argument meaning and error behaviour as the reference (paths relative to it):

    mp(A, b, k[, x])                         src/matchingpursuit.jl:34-40
    omp(A, b, k) / omp(A, b, eps, k) / omp(A, b; max_residual, sparsity)   :73-91
    gomp(A, b, l, k) / gomp(A, b, l, eps, k) / gomp(A, b, l; ...)          :126-148
    sp(A, b, k, delta=1e-12; maxiter=16k)    src/twostage.jl:87-101
    fr(A, b, max_eps, min_delta, k) / fr(A, b; max_residual, min_decrease, sparsity) = ols = oomp = ormp
                                             src/forward.jl:34-54
    FR functor with update!                  src/forward.jl:88-95
    srr(A, b, k, delta=1e-12; maxiter=4k, initialization=1, l=1)          src/twostage.jl:3-33
    rmp(A, b, delta, maxiter=1) / rmp(A, b, k) / foba(A, b, delta)        src/stepwise.jl:5-56
    br / fbr / lace (A, b, max_eps, max_delta, k) and keyword forms       src/backward.jl:27-41,148-162,226-242
    ompr(A, b, k, delta; maxiter)            src/twostage.jl:184-202
    MP / OMP / GOMP functors with update!    src/matchingpursuit.jl:10-31,44-70,95-123
    argmaxinner!(P[, k])                     src/matchingpursuit.jl:181-193

`A` is either a numpy matrix (uploaded to HBM for the duration of the call) or a `Dictionary`
(uploaded once, resident -- what a caller looping over many signals wants).  Results are
`SparseVector`s with 0-based sorted `nzind`.  The reference throws bare strings
(`throw("eps = ... has to be non-negative")`); here those become ValueError with the same text.
There is no CPU path: without libcsmp.so and an MI355X every call raises `CsmpError`.
"""
import numpy as np

from . import _lib
from ._lib import CsmpError
from .sparsevec import SparseVector, spzeros


class Dictionary:
    """The measurement matrix resident in HBM (the `A` field of MP/OMP/GOMP/SP)."""

    def __init__(self, A, device=0, streamed=False):
        """A: numpy matrix / torch CUDA tensor of atoms (see Context.set_dictionary), or the path of a dictionary file.
        streamed=True: A stays in host memory and is read over the host link by every sweep -- a dictionary larger than HBM."""
        self.ctx = _lib.Context(device)
        if isinstance(A, (str, bytes)) or hasattr(A, "__fspath__"):
            self.ctx.set_dictionary_file(A, streamed=streamed)
        else:
            self.ctx.set_dictionary(A, streamed=streamed)
        self.shape = (self.ctx.M, self.ctx.N)
        self.dtype = np.dtype(self.ctx.dtype)

    @property
    def eps(self):  # eps(eltype(A)): the drivers' default tolerance (:85,89,142,146)
        return float(np.finfo(self.dtype).eps)

    def close(self):
        self.ctx.close()


def _dict(A):
    return (A, False) if isinstance(A, Dictionary) else (Dictionary(A), True)


def _meta(A):
    """(M, N, eps(eltype(A))) without touching the GPU, so argument errors surface first."""
    if isinstance(A, Dictionary):
        return A.shape[0], A.shape[1], A.eps
    A = np.asarray(A) if not hasattr(A, "dtype") else A
    if A.ndim != 2:
        raise ValueError("A must be a matrix")
    dt = np.dtype(A.dtype) if A.dtype in (np.float32, np.float64) else np.dtype(np.float64)
    return A.shape[0], A.shape[1], float(np.finfo(dt).eps)


def _is_int(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, bool)


def _check_eps(eps):
    if not eps >= 0:
        raise ValueError(f"ε = {eps} has to be non-negative")  # src/matchingpursuit.jl:74,127


# ------------------------------------------------------------------------------------ drivers
def omp(A, b, *args, max_residual=None, sparsity=None):
    """omp(A,b,k) | omp(A,b,eps[,k]) | omp(A,b; max_residual=eps(T), sparsity=min(size(A)...))"""
    M, N, eps0 = _meta(A)
    if len(args) == 0:  # keyword form :88-91
        eps = eps0 if max_residual is None else max_residual
        k = min(M, N) if sparsity is None else sparsity
    elif len(args) == 1 and _is_int(args[0]):  # omp(A,b,k::Int) :84-86
        eps, k = eps0, args[0]
    elif len(args) == 1:  # omp(A,b,eps) with k = size(A,1) :73
        eps, k = args[0], M
    elif len(args) == 2:
        eps, k = args
    else:
        raise TypeError("omp(A, b, [eps,] k)")
    _check_eps(eps)
    D, tmp = _dict(A)
    try:
        idx, val, _ = D.ctx.omp(b, int(k), float(eps))
        return SparseVector(N, idx, val)
    finally:
        if tmp:
            D.close()


def gomp(A, b, l, *args, max_residual=None, sparsity=None):
    """gomp(A,b,l,k) | gomp(A,b,l,eps[,k]) | gomp(A,b,l; max_residual=eps(T), sparsity=size(A,2))"""
    M, N, eps0 = _meta(A)
    if len(args) == 0:  # :145-148
        eps = eps0 if max_residual is None else max_residual
        k = N if sparsity is None else sparsity
    elif len(args) == 1 and _is_int(args[0]):  # :141-143
        eps, k = eps0, args[0]
    elif len(args) == 1:  # gomp(A,b,l,eps) with k = size(A,1) :126
        eps, k = args[0], M
    elif len(args) == 2:
        eps, k = args
    else:
        raise TypeError("gomp(A, b, l, [eps,] k)")
    _check_eps(eps)
    if int(l) < 1:
        raise ValueError("l has to be positive")
    D, tmp = _dict(A)
    try:
        idx, val, _ = D.ctx.gomp(b, int(l), int(k), float(eps))
        return SparseVector(N, idx, val)
    finally:
        if tmp:
            D.close()


def mp(A, b, k, x=None):
    """mp(A,b,k,x=spzeros(N)): k steps; x is a warm start and is updated in place like the reference."""
    D, tmp = _dict(A)
    try:
        M, N = D.shape
        i0 = v0 = None
        if x is not None and x.nnz:
            i0, v0 = x.nzind, x.nzval
        idx, val = D.ctx.mp(b, int(k), i0, v0)
        if x is None:
            return SparseVector(N, idx, val)
        x.nzind, x.nzval = idx, val
        return x
    finally:
        if tmp:
            D.close()


def sp(A, b, k, delta=1e-12, maxiter=None):
    """sp(A,b,k,delta=1e-12; maxiter=16k).  2k > length(b) is an error (src/twostage.jl:55)."""
    M, N, _ = _meta(A)
    if 2 * k > M:
        raise ValueError(f"2k = {2 * k} > {M} = length(b) is invalid for Subspace Pursuit")
    D, tmp = _dict(A)
    try:
        idx, val, _ = D.ctx.sp(b, int(k), float(delta), -1 if maxiter is None else int(maxiter))
        return SparseVector(N, idx, val)
    finally:
        if tmp:
            D.close()


def ompr(A, b, k, delta, maxiter=None):
    """ompr(A,b,k,delta; maxiter=size(A,1)): OMP with replacement, src/twostage.jl:184-202 (x empty)."""
    D, tmp = _dict(A)
    try:
        idx, val, _ = D.ctx.ompr(b, int(k), float(delta), -1 if maxiter is None else int(maxiter))
        return SparseVector(D.shape[1], idx, val)
    finally:
        if tmp:
            D.close()


def fr(A, b, *args, max_residual=0.0, min_decrease=0.0, sparsity=None):
    """fr(A,b,max_eps,min_delta,k=size(A,1)) and fr(A,b; max_residual=0, min_decrease=0, sparsity=size(A,2)):
    forward regression / orthogonal least squares (src/forward.jl:34-54), x starting empty."""
    M, N, _ = _meta(A)
    if len(args) == 0:  # keyword form :34-37
        max_eps, min_delta, k = max_residual, min_decrease, (N if sparsity is None else sparsity)
    elif len(args) in (2, 3):
        max_eps, min_delta = args[0], args[1]
        k = args[2] if len(args) == 3 else M  # :45
    else:
        raise TypeError("fr(A, b, max_eps, min_delta[, k]) or fr(A, b; max_residual, min_decrease, sparsity)")
    if np.shape(b)[0] != M:  # FR(A, b): DimensionMismatch (:22)
        raise ValueError(f"DimensionMismatch: size(A, 1) = {M} != {np.shape(b)[0]} = length(b)")
    D, tmp = _dict(A)
    try:
        idx, val, _ = D.ctx.fr(b, min(int(k), M), float(max_eps), float(min_delta))
        return SparseVector(N, idx, val)
    finally:
        if tmp:
            D.close()


ols = oomp = ormp = fr  # src/forward.jl:52-54


def srr(A, b, k, delta=1e-12, maxiter=None, initialization=1, l=1, rng=None, init=None):
    """srr(A,b,k,delta=1e-12; maxiter=4k, initialization=1, l=1): stepwise regression with replacement,
    src/twostage.jl:3-33 (x empty).  initialization 3 = random_acquisition! (src/matchingpursuit.jl:195-204): the k initial
    atoms are drawn HERE, without replacement, from `rng` (a numpy Generator or a seed) -- the reference draws them from Julia's
    global RNG, so only the distribution matches, not the draw; pass `init` (k atom indices) to fix the draw itself."""
    D, tmp = _dict(A)
    try:
        if int(initialization) == 3 and init is None:
            init = np.random.default_rng(rng).choice(D.shape[1], size=int(k), replace=False)
        idx, val, _ = D.ctx.srr(b, int(k), float(delta), -1 if maxiter is None else int(maxiter), int(initialization), int(l), init)
        return SparseVector(D.shape[1], idx, val)
    finally:
        if tmp:
            D.close()


def rmp(A, b, delta_or_k, maxiter=1, kmax=None):
    """rmp(A, b, δ, maxiter=1) -- δ a float -- and rmp(A, b, k) -- k an int: relevance matching pursuit,
    src/stepwise.jl:5-43 (x empty).  kmax bounds the support of the forward stage (default min(M, N, 4095))."""
    D, tmp = _dict(A)
    try:
        idx, val = D.ctx.rmp(b, delta_or_k, maxiter, kmax)
        return SparseVector(D.shape[1], idx, val)
    finally:
        if tmp:
            D.close()


def foba(A, b, delta, kmax=None, isfast=True):
    """foba(A, b, δ; isfast): adaptive forward-backward greedy algorithm, src/stepwise.jl:47-56 (x empty).
    `isfast` is accepted for signature parity and has no effect: the reference's isfast = Val(false) computes the SAME
    backward scores |r_{-i}|^2 - |r|^2 by re-factorising without each column in turn (naive_backward_δ!, src/backward.jl:87-105)
    instead of x_i^2 / γ_i (backward_δ!, :79-83) -- one quantity, two costs; the device evaluates the closed form."""
    del isfast
    D, tmp = _dict(A)
    try:
        idx, val = D.ctx.foba(b, float(delta), kmax)
        return SparseVector(D.shape[1], idx, val)
    finally:
        if tmp:
            D.close()


def _backward(A, b, args, max_residual, max_increase, sparsity, lace, name):
    if len(args) == 0:  # keyword form (src/backward.jl:38-41,150-153,228-231)
        max_eps, max_delta, k = max_residual, max_increase, sparsity
    elif len(args) == 3:
        max_eps, max_delta, k = args
    else:
        raise TypeError(f"{name}(A, b, max_eps, max_delta, k) or {name}(A, b; max_residual, max_increase, sparsity)")
    M, N, _ = _meta(A)
    if N > M:
        raise ValueError(f"A needs to be overdetermined but is of size ({M}, {N})")  # :218
    D, tmp = _dict(A)
    try:
        idx, val = D.ctx.br(b, float(max_eps), float(max_delta), int(k), lace)
        return SparseVector(N, idx, val)
    finally:
        if tmp:
            D.close()


def br(A, b, *args, max_residual=float("inf"), max_increase=float("inf"), sparsity=0):
    """br(A,b,max_eps,max_delta,k) / br(A,b; max_residual=Inf, max_increase=Inf, sparsity=0): backward regression,
    src/backward.jl:27-41 (the default isfast = true scores)."""
    return _backward(A, b, args, max_residual, max_increase, sparsity, False, "br")


def fbr(A, b, *args, max_residual=float("inf"), max_increase=float("inf"), sparsity=0):
    """fbr: the same algorithm as br on the normal equations (src/backward.jl:148-162); here they share one path."""
    return _backward(A, b, args, max_residual, max_increase, sparsity, False, "fbr")


def lace(A, b, *args, max_residual=float("inf"), max_increase=float("inf"), sparsity=0):
    """lace(A,b,eps,delta,k): least absolute coefficient elimination, src/backward.jl:226-270."""
    return _backward(A, b, args, max_residual, max_increase, sparsity, True, "lace")


def omp_batch(A, B, k, eps=None):
    """[omp(A, B[:, s], eps, k) for s in axes(B, 2)] on one GPU; returns a list of SparseVectors."""
    eps = _meta(A)[2] if eps is None else eps
    _check_eps(eps)
    D, tmp = _dict(A)
    try:
        idx, val, nnz = D.ctx.omp_batch(B, int(k), float(eps))
        return [SparseVector(D.shape[1], idx[:n, s], val[:n, s]) for s, n in enumerate(nnz)]
    finally:
        if tmp:
            D.close()


def gomp_batch(A, B, l, k, eps=None):
    """[gomp(A, B[:, s], l, eps, k) for s in axes(B, 2)] on one GPU (two solves in flight): list of SparseVectors."""
    eps = _meta(A)[2] if eps is None else eps
    _check_eps(eps)
    D, tmp = _dict(A)
    try:
        idx, val, nnz = D.ctx.gomp_batch(B, int(l), int(k), float(eps))
        return [SparseVector(D.shape[1], idx[:n, s], val[:n, s]) for s, n in enumerate(nnz)]
    finally:
        if tmp:
            D.close()


def sp_batch(A, B, k, delta=1e-12, maxiter=None):
    """[sp(A, B[:, s], k, delta; maxiter) for s in axes(B, 2)] on one GPU (several solves in flight): list of SparseVectors."""
    D, tmp = _dict(A)
    try:
        idx, val, nnz, its = D.ctx.sp_batch(B, int(k), float(delta), -1 if maxiter is None else int(maxiter))
        return [SparseVector(D.shape[1], idx[:n, s], val[:n, s]) for s, n in enumerate(nnz)]
    finally:
        if tmp:
            D.close()


def solve_in_flight(A, signals, solve, in_flight=3):
    """Any single-signal driver for many signals with `in_flight` solves at once on ONE GPU: `solve(ctx, b)` is called with a
    context of its own (the Dictionary's and `in_flight - 1` clones: own stream, own solver state, the same resident dictionary)
    from a host thread of its own -- the library calls release the GIL -- so that one signal's short, latency-bound stages run
    under another signal's dictionary sweeps (what csmp_gomp_batch / csmp_sp_batch do inside the library).  Returns the list
    of `solve`'s results in signal order.  Example: solve_in_flight(D, cols, lambda c, b: c.ompr(b, 64, 1e-6))."""
    import threading
    D, tmp = _dict(A)
    signals = list(signals)
    T = max(1, min(int(in_flight), len(signals)))
    ctxs = [D.ctx] + [D.ctx.clone() for _ in range(T - 1)]
    out = [None] * len(signals)
    err = [None] * T

    def work(t):
        try:
            for s in range(t, len(signals), T):
                out[s] = solve(ctxs[t], signals[s])
        except Exception as e:  # noqa: BLE001
            err[t] = e
    th = [threading.Thread(target=work, args=(t,)) for t in range(1, T)]
    try:
        for x in th:
            x.start()
        work(0)
    finally:
        for x in th:  # (also when work(0) left through a BaseException: no clone is closed under a thread that still uses it)
            if x.is_alive():
                x.join()
        for c in ctxs[1:]:
            c.close()
        if tmp:
            D.close()
    for e in err:
        if e is not None:
            raise e
    return out


def fr_batch(A, B, k, max_eps=0.0, min_delta=0.0):
    """[fr(A, B[:, s], max_eps, min_delta, k) for s in axes(B, 2)]: list of SparseVectors (pipelined on the device)."""
    D, tmp = _dict(A)
    try:
        idx, val, nnz = D.ctx.fr_batch(B, int(k), float(max_eps), float(min_delta))
        return [SparseVector(D.shape[1], idx[:nnz[s], s].copy(), val[:nnz[s], s].copy()) for s in range(idx.shape[1])]
    finally:
        if tmp:
            D.close()


def omp_batch_mfma(A, B, k, eps=None, cert=None, gram=None):
    """omp_batch through the batched variant (BASELINE configs 3/4): one bf16 MFMA screening GEMM per step for all
    signals, Float64 rescoring of the screened atoms that could still be the exact maximum, and a per-step certificate;
    signals that fail it are re-solved by the exact path, so certified results equal omp_batch's.
    cert: "rigorous" (the library's default: a deterministic error bound, a passed certificate proves the pick) or "statistical"
    (opt-in: 8 sigma of independent roundings + a coherent term; narrower windows, faster, NOT a proof -- see
    tests/test_gpu_parity.py::test_batched_certificate_against_adversarial_residuals); gram: True keeps G = A'A resident on the
    GPU (8 N^2 bytes) and halves the append traffic (csmp_set_option CSMP_OPT_BATCH_CERT / CSMP_OPT_BATCH_GRAM, include/csmp.h).
    Options given here hold for this call only: a caller-owned Dictionary gets its own values back."""
    eps = _meta(A)[2] if eps is None else eps
    _check_eps(eps)
    D, tmp = _dict(A)
    saved = {}
    try:
        if cert is not None:
            saved["batch_cert"] = D.ctx.get_option("batch_cert")
            D.ctx.set_option("batch_cert", {"statistical": 0, "rigorous": 1}[cert])
        if gram is not None:
            saved["batch_gram"] = D.ctx.get_option("batch_gram")
            D.ctx.set_option("batch_gram", int(bool(gram)))
        idx, val, nnz = D.ctx.omp_batch_mfma(B, int(k), float(eps))
        return [SparseVector(D.shape[1], idx[:n, s], val[:n, s]) for s, n in enumerate(nnz)]
    finally:
        if tmp:
            D.close()
        else:
            for key, v in saved.items():
                D.ctx.set_option(key, v)


# ------------------------------------------------------------------------------------ functors
class _Update:
    """abstract type Update; (U::Update)(x) = update!(U, x)   (src/CompressedSensing.jl:22-23)"""

    def __call__(self, x=None):
        return self.update_(x)


class _DevicePursuit(_Update):
    ALGO = None

    def __init__(self, A, b, kcap, l=1):
        self.D, self._tmp = _dict(A)
        self.A, self.b = A, np.asarray(b)
        self.l = int(l)
        M, N = self.D.shape
        self.kcap = int(kcap)
        # a context of its own that borrows the resident dictionary: like the reference's P objects, two functors on
        # one Dictionary (or a functor and a driver call) share A and nothing else
        self.ctx = self.D.ctx.clone()
        self.ctx.solver_begin(self.ALGO, b, self.kcap)
        self._x = spzeros(N)  # the x this object has been evolving

    def _sync_x(self, x):
        idx, val, res, order, stop = self.ctx.solver_state(max(self.kcap, 1))
        self._x = SparseVector(self.D.shape[1], idx, val)
        self.resnorm, self.order, self.stop = res, order, stop
        if x is None:
            return self._x.copy()
        x.nzind, x.nzval = idx.copy(), val.copy()
        return x

    def _same(self, x):
        return x is None or (x.nnz == self._x.nnz and np.array_equal(x.nzind, self._x.nzind))

    def residual_norm(self):
        return self.ctx.solver_state(max(self.kcap, 1))[2]

    def close(self):
        self.ctx.close()
        if self._tmp:
            self.D.close()


class OMP(_DevicePursuit):
    """OMP(A, b, k=size(A,1)); update!(P, x): src/matchingpursuit.jl:54-70.  The updatable QR and
    the residual live on the device, tied to the x this object returned last."""
    ALGO = _lib.ALGO_OMP

    def __init__(self, A, b, k=None):
        M = A.shape[0]
        super().__init__(A, b, M if k is None else min(int(k), M))

    def update_(self, x=None):
        if not self._same(x):
            raise ValueError("update!(P::OMP, x): x is not the support this OMP object's QR was built for")
        self.ctx.solver_step(1)
        return self._sync_x(x)


class GOMP(_DevicePursuit):
    """GOMP(A, b, l, k=size(A,1)); update!(P, x, l=P.l): src/matchingpursuit.jl:108-123."""
    ALGO = _lib.ALGO_GOMP

    def __init__(self, A, b, l, k=None):
        M = A.shape[0]
        super().__init__(A, b, M if k is None else min(int(k), M), l)

    def update_(self, x=None, l=None):
        if not self._same(x):
            raise ValueError("update!(P::GOMP, x): x is not the support this GOMP object's QR was built for")
        self.ctx.solver_step(self.l if l is None else int(l))
        return self._sync_x(x)


class FR(_DevicePursuit):
    """FR(A, b) (= OLS = OOMP = ORMP = StepwiseRegression, src/forward.jl:14-19); update!(P, x): :88-95.
    `delta2` is P.δ² of the last step (:11)."""
    ALGO = _lib.ALGO_FR

    def __init__(self, A, b, k=None):
        M = A.shape[0]
        if np.shape(b)[0] != M:
            raise ValueError(f"DimensionMismatch: size(A, 1) = {M} != {np.shape(b)[0]} = length(b)")
        super().__init__(A, b, M if k is None else min(int(k), M))

    def update_(self, x=None):
        if not self._same(x):
            raise ValueError("update!(P::FR, x): x is not the support this FR object's QR was built for")
        self.ctx.solver_step(1)
        return self._sync_x(x)

    @property
    def delta2(self):
        return self.ctx.fr_scores()


OLS = OOMP = ORMP = StepwiseRegression = FR


class MP(_DevicePursuit):
    """MP(A, b); update!(P, x): src/matchingpursuit.jl:19-31.  Any x may be passed (the reference
    recomputes the residual from x every step): a foreign x restarts the device residual from it."""
    ALGO = _lib.ALGO_MP

    def __init__(self, A, b, steps=4096):
        """steps: how many update! calls this object can record (MP keeps no factorisation, so this is only the
        length of its (atom, coefficient) log on the device)."""
        super().__init__(A, b, steps)

    def update_(self, x=None):
        if x is not None and not (self._same(x) and np.array_equal(x.nzval, self._x.nzval)):
            self.ctx.solver_begin(self.ALGO, self.b, self.kcap, x.nzind, x.nzval)
            self._base = x.copy()
        self.ctx.solver_step(1)
        idx, val, res, order, stop = self.ctx.solver_state(max(self.kcap, 1))
        base = getattr(self, "_base", None)
        out = base.copy() if base is not None else spzeros(self.D.shape[1])
        for i, v in zip(idx, val):  # steps since the (re)start, merged per atom by the library
            out[i] = out[i] + v
        self._x = out
        if x is None:
            return out.copy()
        x.nzind, x.nzval = out.nzind.copy(), out.nzval.copy()
        return x


class _TwoStage(_Update):
    """SP and OMPR: functors whose x the reference's caller owns and passes back in (update!(P, x)).  The device solver holds the x
    it produced last; an x that is not that one -- other atoms or other values -- restarts the solver from it where the reference
    allows that (SP: the acquisition starts from residual!(P, x) whatever x is, src/twostage.jl:68)."""
    ALGO = None

    def __init__(self, A, b, k):
        self.D, self._tmp = _dict(A)
        self.A, self.b, self.k = A, np.asarray(b), int(k)
        self.ctx = self.D.ctx.clone()
        self.ctx.solver_begin(self.ALGO, self.b, self.k)
        self._x = spzeros(self.D.shape[1])

    def _sync_x(self, x):
        idx, val, res, order, stop = self.ctx.solver_state(max(2 * self.k, 1))
        self._x = SparseVector(self.D.shape[1], idx, val)
        self.resnorm = res
        if x is None:
            return self._x.copy()
        x.nzind, x.nzval = idx.copy(), val.copy()
        return x

    def _is_mine(self, x):
        return x is None or (x.nnz == self._x.nnz and np.array_equal(x.nzind, self._x.nzind) and np.array_equal(x.nzval, self._x.nzval))

    def _check_nnz(self, x):
        n = self._x.nnz if x is None else x.nnz
        if n != self.k:  # the reference throws this String (src/twostage.jl:76,135)
            raise ValueError(f"nnz(x) = {n} \u2260 {self.k} = k")

    def residual_norm(self):
        return self.ctx.solver_state(max(2 * self.k, 1))[2]

    def close(self):
        self.ctx.close()
        if self._tmp:
            self.D.close()


class SP(_TwoStage):
    """SP(A, b, k) = SubspacePursuit (src/twostage.jl:42-61): sp_acquisition!(P, x, k=P.k) (:67-72) and update!(P, x) (:75-83) one
    call at a time on the device -- the phases csmp_sp strings together (sweep + top-k, least squares on the union, prune, least
    squares on the k kept)."""
    ALGO = _lib.ALGO_SP

    def __init__(self, A, b, k):
        M = A.shape[0]
        if 2 * k > M:  # error(...) at :55
            raise ValueError(f"2k = {2 * k} > {M} = length(b) is invalid for Subspace Pursuit")
        super().__init__(A, b, k)

    def _load(self, x):
        if not self._is_mine(x):  # a foreign x: the solver restarts from it (its values too: the residual is b - A x)
            self.ctx.solver_begin(self.ALGO, self.b, self.k, x.nzind, x.nzval)
            self._x = x.copy()

    def acquisition_(self, x=None, k=None):
        self._load(x)
        self.ctx.solver_acquire(self.k if k is None else int(k))
        return self._sync_x(x)

    def update_(self, x=None):
        self._check_nnz(x)
        self._load(x)
        self.ctx.solver_step(1)
        return self._sync_x(x)


SubspacePursuit = SP


class OMPR(_TwoStage):
    """OMPR(A, b, k) (src/twostage.jl:110-132): oblivious_acquisition!(P, x, k) fills the empty x (src/matchingpursuit.jl:207-216;
    how ompr starts, src/twostage.jl:190), update!(P, x) (:134-180, eta = 1) swaps one atom.  The updatable QR (with its Givens
    down-date) lives on the device, tied to the x this object returned last."""
    ALGO = _lib.ALGO_OMPR

    def acquisition_(self, x=None, k=None):
        if x is not None and x.nnz:
            raise ValueError("oblivious_acquisition!(P::OMPR, x, k): x must be empty (OMPR(A, b, k) starts from an empty factorisation)")
        if self._x.nnz:  # a fresh start of the same object
            self.ctx.solver_begin(self.ALGO, self.b, self.k)
        self.ctx.solver_acquire(self.k if k is None else int(k))
        return self._sync_x(x)

    def update_(self, x=None, eta=1.0):
        if eta != 1.0:
            raise NotImplementedError("update!(P::OMPR, x, eta): only eta = 1 (the value every driver of the reference uses) runs on the device")
        self._check_nnz(x)
        if not self._is_mine(x):
            raise ValueError("update!(P::OMPR, x): x is not the vector this OMPR object's QR was built for")
        self.ctx.solver_step(1)
        return self._sync_x(x)


def sp_acquisition(P, x=None, k=None):
    """sp_acquisition!(P::SP, x, k = P.k): src/twostage.jl:67-72"""
    return P.acquisition_(x, k)


# ------------------------------------------------------------------------------------ oblivious
def oblivious(A, b, k):
    """oblivious(A, b, k): src/oblivious.jl:3-8 -- the k atoms most correlated with b
    (partialsortperm(abs.(A'b), 1:k, rev=true)), then least squares on them.  Two device
    primitives: one sweep + top-k, one on-device QR solve.  (The reference allocates the result as
    spzeros(size(b)), i.e. of length M -- an upstream slip; here x has the dictionary's N entries.)"""
    D, tmp = _dict(A)
    try:
        _, ti, _ = D.ctx.sweep(np.asarray(b, dtype=np.float64), int(k), want_abs=False)
        coef = D.ctx.lstsq(ti, b)
        return SparseVector(D.shape[1], ti, coef)
    finally:
        if tmp:
            D.close()


def oblivious_acquisition(A, b, x=None, k=None):
    """oblivious_acquisition!(P, x, k): src/matchingpursuit.jl:207-216 -- residual of the current x,
    the k atoms best correlated with it are added (x[ind] = NaN placeholders in the reference), then
    least squares on the enlarged support.  x is updated in place and returned.
    Two call forms: (A, b, x, k) on a matrix / Dictionary, and (P, x, k) on a functor that keeps an updatable QR
    (OMPR, OMP, GOMP: csmp_solver_acquire)."""
    if isinstance(A, (_TwoStage, _DevicePursuit)):  # (P, x, k)
        P, x, k = A, b, (x if k is None else k)
        if isinstance(P, OMPR):
            return P.acquisition_(x, k)
        if not isinstance(P, (OMP, GOMP)):
            raise TypeError("oblivious_acquisition!(P, x, k): P must keep an updatable QR (OMP, GOMP, OMPR)")
        if not P._same(x):
            raise ValueError("oblivious_acquisition!(P, x, k): x is not the support this object's QR was built for")
        P.ctx.solver_acquire(int(k))
        return P._sync_x(x)
    D, tmp = _dict(A)
    try:
        b = np.asarray(b, dtype=np.float64)
        if x.nnz:
            raise NotImplementedError("oblivious_acquisition! is built for an empty x (how srr/ompr call it, "
                                      "src/twostage.jl:10,190); a non-empty x needs QR column insertion order handling")
        r = b
        _, ti, _ = D.ctx.sweep(r, int(k), want_abs=False)
        cols = np.sort(ti)
        coef = D.ctx.lstsq(cols, b)
        x.nzind, x.nzval = cols.astype(np.int64), coef
        return x
    finally:
        if tmp:
            D.close()


def random_acquisition(A, b, x, k, rng=None):
    """random_acquisition!(P, x, k): src/matchingpursuit.jl:195-204 -- k random atoms, least squares."""
    D, tmp = _dict(A)
    try:
        rng = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)
        cols = np.sort(rng.choice(D.shape[1], size=int(k), replace=False))
        coef = D.ctx.lstsq(cols, b)
        x.nzind, x.nzval = cols.astype(np.int64), coef
        return x
    finally:
        if tmp:
            D.close()


def update_(P, x=None, *a):
    """update!(P, x)"""
    return P.update_(x, *a)


def argmaxinner(A, r, k=None):
    """argmaxinner!(P) -> index of the largest |<a_i, r>| (first on ties);
    argmaxinner!(P, k) -> the k largest, descending, ties by ascending index (:181-193)."""
    D, tmp = _dict(A)
    try:
        _, ti, _ = D.ctx.sweep(r, 1 if k is None else int(k), want_abs=False)
        return int(ti[0]) if k is None else ti
    finally:
        if tmp:
            D.close()
