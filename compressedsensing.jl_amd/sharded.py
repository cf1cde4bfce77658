"""Signals sharded across the GPUs of one node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), the dictionary replicated, signal s owned by the rank whose contiguous
block holds it, NO communication during the solves and ONE gather of the packed results at the
end.  The reference has no distributed code; this is the loop
`[omp(A, B[:, s], eps, k) for s in 1:nsig]` a caller of the reference writes, spread over ranks.
"""
import numpy as np


def shard_range(nsig, rank, world):
    """Contiguous block of signals owned by `rank`: sizes differ by at most one."""
    base, extra = divmod(int(nsig), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack(idx, val, nnz):
    """(k x n) int64, (k x n) float64, (n,) int64 -> one float64 buffer [n, 2k+1] (indices are
    < 2^53, exactly representable), so the whole shard travels in a single collective."""
    k, n = idx.shape
    buf = np.empty((n, 2 * k + 1), np.float64)
    buf[:, :k] = idx.T
    buf[:, k:2 * k] = val.T
    buf[:, 2 * k] = nnz
    return buf


def unpack(buf, k):
    idx = buf[:, :k].T.astype(np.int64)
    val = buf[:, k:2 * k].T.copy()
    nnz = buf[:, 2 * k].astype(np.int64)
    return idx, val, nnz


def omp_sharded(D, B, k, eps=None, group=None, solver=None, device=None):
    """Solve omp for every column of B (M x nsig, identical on all ranks) with the ranks of the
    default (or given) process group; every rank returns the full (idx, val, nnz) arrays.

    `solver(B_local, k, eps) -> (idx, val, nnz)` defaults to the HIP path `D.ctx.omp_batch`;
    the gloo CPU tests inject a stand-in there to exercise the sharding/gather logic only."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nsig = B.shape[1]
    lo, hi = shard_range(nsig, rank, world)
    if solver is None:
        eps = D.eps if eps is None else eps
        solver = D.ctx.omp_batch
    idx, val, nnz = solver(np.asfortranarray(B[:, lo:hi]), k, eps)
    maxn = -(-nsig // world)
    mine = np.zeros((maxn, 2 * k + 1), np.float64)
    mine[:hi - lo] = pack(idx, val, nnz)
    backend = dist.get_backend(group)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    t = torch.from_numpy(mine).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)  # the single collective of the path
    bufs = []
    for r in range(world):
        rlo, rhi = shard_range(nsig, r, world)
        bufs.append(out[r][:rhi - rlo].cpu().numpy())
    return unpack(np.concatenate(bufs, axis=0), k)


def sharded_solve(B, cap, one, group=None, device=None):
    """The same sharding for ANY single-signal solver of the package: `one(b) -> (idx, val[, ...])` is called
    for the signals of this rank's block (e.g. `lambda b: D.ctx.fr(b, k)`, `lambda b: D.ctx.srr(b, k)`); `cap`
    bounds nnz.  Every rank returns (idx cap x nsig padded with -1, val, nnz) after the single all_gather."""
    def solver(Bl, kk, eps):
        n = Bl.shape[1]
        idx = -np.ones((kk, n), np.int64)
        val = np.zeros((kk, n))
        nnz = np.zeros(n, np.int64)
        for s in range(n):
            r = one(Bl[:, s])
            i, v = r[0], r[1]
            idx[:len(i), s], val[:len(i), s], nnz[s] = i, v, len(i)
        return idx, val, nnz
    return omp_sharded(None, B, cap, eps=0.0, group=group, solver=solver, device=device)


def fr_sharded(D, B, k, max_eps=0.0, min_delta=0.0, group=None):
    """fr (forward regression / OLS) for every column of B, signals sharded over the ranks."""
    return sharded_solve(B, k, lambda b: D.ctx.fr(b, k, max_eps, min_delta), group=group)
