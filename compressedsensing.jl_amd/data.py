"""Synthetic inputs and the parity predicate of the reference (src/util.jl:4-55).

Argument names follow the REFERENCE's convention: `n` = rows (measurements), `m` = columns
(atoms) -- the opposite of BASELINE.json's "m=4096, n=65536" (SURVEY.md naming trap).
numpy's generators replace Julia's `randn` / `StatsBase.sample`; the distributions are the same,
the streams are not (the reference seeds nothing, test/matchingpursuit.jl:7).
"""
import numpy as np

from .sparsevec import SparseVector


def _rng(rng):
    return rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)


def sparse_vector(m, k, gaussian=False, rng=None):
    """src/util.jl:13-19: k-sparse vector of length m, entries +-1 (or N(0,1))."""
    if m < k:
        raise ValueError(f"m = {m} < {k} = k")
    rng = _rng(rng)
    ind = np.sort(rng.choice(m, size=k, replace=False))
    val = rng.standard_normal(k) if gaussian else rng.choice(np.array([-1.0, 1.0]), size=k)
    return SparseVector(m, ind, val)


def sparse_data(n=32, m=64, k=3, rescaled=True, rng=None, dtype=np.float64):
    """src/util.jl:21-31: Gaussian dictionary (n rows x m atoms), planted k-sparse x, b = A x.

    Generated in Float64; if `dtype` is float32 the dictionary is cast ONCE and b is formed
    from the cast values in Float64, so an f32 run and the f64 oracle see
    identical inputs.  Returns (A column-major, x, b float64)."""
    rng = _rng(rng)
    A = rng.standard_normal((n, m))
    if rescaled:
        A -= 1e-6 * A.mean(axis=0, keepdims=True)  # util.jl:24-25
        A /= np.sqrt((A * A).sum(axis=0, keepdims=True))  # util.jl:26
    A = np.asfortranarray(A.astype(dtype))
    x = sparse_vector(m, k, rng=rng)
    b = A[:, x.nzind].astype(np.float64) @ x.nzval
    return A, x, b


gaussian_data = sparse_data  # util.jl:32


def perturb(b, delta, rng=None):
    """src/util.jl:50-55: b + e with ||e||_2 = delta exactly."""
    rng = _rng(rng)
    e = rng.standard_normal(np.shape(b))
    e *= delta / np.linalg.norm(e)
    return np.asarray(b, dtype=np.float64) + e


def samesupport(x, y):
    """src/util.jl:4-9: equality of the sorted supports."""
    def supp(v):
        if isinstance(v, SparseVector):
            return np.sort(v.nzind)
        return np.flatnonzero(np.asarray(v))
    sx, sy = supp(x), supp(y)
    return len(sx) == len(sy) and bool(np.all(sx == sy))


def structured_dictionary(kind, n, m, rng=None, dtype=np.float32):
    """Unit-norm dictionaries that are NOT Gaussian (n rows x m atoms, reference naming) -- not in the reference; they
    exist to probe the bf16 screen of csmp_omp_batch_mfma where rounding errors are not independent:
      few_valued        entries in {+-0.3, +-0.7}: every entry of a class rounds alike
      partial_dct       n random rows of the m-point DCT-II
      common_component  Gaussian + a strong common vector (coherent: cond(A_S) large)
      one_magnitude     entries in {-c_j, 0, +c_j}, a different c_j per column: in bf16 a whole column is SCALED by one factor
      signs             +-1/sqrt(n): exactly representable when n is a power of 4"""
    rng = _rng(rng)
    if kind == "few_valued":
        A = rng.choice(np.array([-0.7, -0.3, 0.3, 0.7]), size=(n, m))
    elif kind == "partial_dct":
        rows = np.sort(rng.choice(m, size=n, replace=False))
        A = np.cos(np.pi * (np.arange(m)[None, :] + 0.5) * rows[:, None] / m)
    elif kind == "common_component":
        A = rng.standard_normal((n, m)) + 1.5 * rng.standard_normal((n, 1))
    elif kind == "one_magnitude":
        dens = rng.uniform(0.2, 0.6, size=m)
        A = (rng.random((n, m)) < dens[None, :]) * rng.choice(np.array([-1.0, 1.0]), size=(n, m))
        A[0, np.abs(A).sum(axis=0) == 0] = 1.0
    elif kind == "signs":
        A = rng.choice(np.array([-1.0, 1.0]), size=(n, m))
    else:
        raise ValueError(kind)
    A = A / np.sqrt((A * A).sum(axis=0, keepdims=True))
    return np.asfortranarray(A.astype(dtype))
