"""SparseVector -- the result container of the reference's drivers.

The reference returns `SparseVector{Float64,Int64}` regardless of the dictionary's element type
(`spzeros(size(A, 2))`, src/matchingpursuit.jl:34,76,129; src/twostage.jl:89): sorted `nzind`
plus aligned `nzval`.  This class mirrors the fields the reference's code and tests touch
(`nzind`, `nzval`, `nnz`, `x[i] = v`, `x[i] += v`).  Indices are 0-BASED here (Python); the Julia
wrapper in `julia/CompressedSensingAMD.jl` adds 1.
"""
import numpy as np


class SparseVector:
    __slots__ = ("n", "nzind", "nzval")

    def __init__(self, n, nzind=None, nzval=None):
        self.n = int(n)
        self.nzind = np.zeros(0, np.int64) if nzind is None else np.asarray(nzind, np.int64).copy()
        self.nzval = np.zeros(0, np.float64) if nzval is None else np.asarray(nzval, np.float64).copy()
        if self.nzind.shape != self.nzval.shape:
            raise ValueError("nzind and nzval must have the same length")
        if len(self.nzind) > 1 and not np.all(np.diff(self.nzind) > 0):
            order = np.argsort(self.nzind, kind="stable")
            self.nzind, self.nzval = self.nzind[order], self.nzval[order]
            if not np.all(np.diff(self.nzind) > 0):
                raise ValueError("duplicate indices in nzind")
        if len(self.nzind) and (self.nzind[0] < 0 or self.nzind[-1] >= self.n):
            raise IndexError("index out of range")

    @property
    def nnz(self):
        return len(self.nzind)

    def __len__(self):
        return self.n

    def copy(self):
        return SparseVector(self.n, self.nzind, self.nzval)

    def _find(self, i):
        p = int(np.searchsorted(self.nzind, i))
        return p, (p < len(self.nzind) and self.nzind[p] == i)

    def __getitem__(self, i):
        p, hit = self._find(int(i))
        return float(self.nzval[p]) if hit else 0.0

    def __setitem__(self, i, v):
        i = int(i)
        if not 0 <= i < self.n:
            raise IndexError(i)
        p, hit = self._find(i)
        if hit:
            self.nzval[p] = v
        elif v != 0 or v != v:  # SparseVector setindex! stores NaN but not a structural zero
            self.nzind = np.insert(self.nzind, p, i)
            self.nzval = np.insert(self.nzval, p, v)

    def to_dense(self):
        x = np.zeros(self.n)
        x[self.nzind] = self.nzval
        return x

    def __repr__(self):
        return f"SparseVector(n={self.n}, nzind={self.nzind.tolist()}, nzval={self.nzval.tolist()})"


def spzeros(n):
    return SparseVector(n)
