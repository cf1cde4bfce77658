"""Multi-GPU drivers: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Two shardings, neither of which exists in the reference (it has no distributed code):

* SIGNALS sharded, dictionary replicated (BASELINE configs[3]): the loop
  `[omp(A, B[:, s], eps, k) for s in 1:nsig]` a caller of the reference writes, spread over ranks.  Signal s
  is owned by the rank whose contiguous block holds it, NO communication during the solves and ONE
  all_gather of the packed results at the end (`omp_sharded`, `sharded_solve`, `fr_sharded`).  The wire layout
  -- per signal a row of 2k+1 Float64 [idx | val | nnz] -- is the C ABI's (`csmp_pack_results`).

* COLUMNS sharded, one signal: `omp_colsharded` -- every rank sweeps its slice
  of the dictionary, ONE all_gather of a 16-KiB record per rank and step carries (|c|, global index, the
  column itself), and every rank appends the winner to its replica of the factorisation
  (csrc/csmp_shard.hpp).  Results are identical to `omp(A, b, eps, k)` on the whole dictionary
  (src/matchingpursuit.jl:73-82), ties across shards going to the lower global index.
"""
import numpy as np


def shard_range(nsig, rank, world):
    """Contiguous block of signals owned by `rank`: sizes differ by at most one (= csmp_shard_range)."""
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


# ---------------------------------------------------------------------------------- the one exchange, on tensors
def pack_t(idx, val, nnz):
    """torch tensors idx (n, k) int64, val (n, k) float64, nnz (n,) int64 (device or CPU) -> (n, 2k+1) float64,
    the same rows as `pack` / csmp_pack_results.  Stays on the tensors' device: no host round trip."""
    import torch
    return torch.cat([idx.to(torch.float64), val, nnz[:, None].to(torch.float64)], dim=1).contiguous()


class ShardedSolveError(RuntimeError):
    """A rank could not solve its block of a signal-sharded call; raised on EVERY rank after the one collective (csmp_omp_sharded
    returns the failing rank's code on every rank the same way)."""

    def __init__(self, failed, status):
        super().__init__(f"signal-sharded solve: rank(s) {failed} could not solve their block (status {status}); no rank's results are valid")
        self.failed, self.status = failed, status


class ShardedArgumentError(ValueError):
    """The ranks of a signal-sharded call were given different (nsig, k): raised on EVERY rank by the agreement exchange, before any
    collective whose counts would not have matched (csmp_omp_sharded returns CSMP_EINVAL on every rank the same way)."""

    def __init__(self, rows):
        super().__init__(f"signal-sharded solve: the ranks disagree on (nsig, row width): {rows}; no rank's results are valid")
        self.rows = rows


def agree(values, group=None, device=None):
    """All ranks exchange a fixed-size header (a count no argument can change) and compare: different values on any two ranks raise
    ShardedArgumentError on every rank.  The collective in front of the gather, whose counts depend on exactly these values."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    on_host = device is None or device.type != "cuda" or dist.get_backend(group) == "gloo"
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=torch.device("cpu") if on_host else device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    rows = [tuple(o.cpu().tolist()) for o in out]
    if any(r != rows[0] for r in rows):
        raise ShardedArgumentError(rows)


def gather_packed(packed, nsig, group=None, status=0):
    """ONE all_gather of every rank's packed block (padded to the largest block, ceil(nsig / world) rows), behind the fixed-size
    agreement exchange on (nsig, 2k+1); returns the (nsig, 2k+1) tensor of all signals in global order, on `packed`'s device.
    status < 0: THIS rank could not solve its block.  It still takes part -- a rank that left before the collective would leave
    the others waiting for ever -- with an empty block whose row 0 carries the status in its nnz slot (a count is never negative:
    libcsmp's k_pack_rows writes the same word); every rank then raises ShardedSolveError."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    # the gather's counts follow from (nsig, row width): ranks that disagree on them would post mismatched all_gathers and hang
    agree((int(nsig), int(packed.shape[1])), group, packed.device)
    maxn = -(-int(nsig) // world)
    if maxn == 0:  # an empty batch (on every rank: they agree): nothing to gather, nothing that could have failed
        return torch.zeros((0, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    mine = packed
    if status < 0:
        mine = torch.zeros((maxn, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        mine[0, -1] = float(status)
    elif packed.shape[0] != maxn:
        mine = torch.zeros((maxn, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        mine[:packed.shape[0]] = packed
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only (rehearsals of the multi-rank path with several ranks on one GPU: bench.py --share-gpu)
        host = mine.cpu()
        outh = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(outh, host, group=group)
        out = [o.to(mine.device) for o in outh]
    else:
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine, group=group)  # the single collective of the path
    failed = [r for r in range(world) if float(out[r][0, -1]) < 0.0]
    if failed:
        raise ShardedSolveError(failed, int(float(out[failed[0]][0, -1])))
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(nsig, r, world)
        parts.append(out[r][:rhi - rlo])
    return torch.cat(parts, dim=0)


def unpack_t(full, k):
    """(nsig, 2k+1) float64 tensor -> numpy (idx k x nsig, val k x nsig, nnz) like the batch drivers return."""
    return unpack(full.cpu().numpy(), k)


def library_comm(ctx, group=None):
    """Give `ctx` its RCCL communicator over the ranks of `group` (csmp_comm_init), once per (context, group): rank 0 draws the id
    (csmp_comm_id), the process group broadcasts those 128 bytes -- the only thing the host-side group is used for."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    key = (id(group), rank, world)
    if getattr(ctx, "_comm_key", None) == key:
        return
    import torch
    from . import _lib
    # every rank runs the same sequence of collectives whatever fails where: rank 0 broadcasts the id OR a failure marker, and the
    # ranks agree on the outcome of csmp_comm_init before any of them uses (or gives up on) the communicator
    err = None
    ident = None
    if rank == 0:
        try:
            ident = _lib.comm_id()
        except Exception as e:  # noqa: BLE001  (RCCL cannot be bound, ncclGetUniqueId failed)
            err = e
    ids = [ident]
    dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if ids[0] is None:
        raise _lib.CsmpError(_lib.ERCCL, "rank 0 could not draw a communicator id" + (f": {err}" if err else ""))
    ok = 1
    try:
        ctx.comm_init(ids[0], rank, world)
    except Exception as e:  # noqa: BLE001
        ok, err = 0, e
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        if ok:
            ctx.comm_free()
        raise _lib.CsmpError(_lib.ERCCL, "csmp_comm_init failed on " + ("this rank: " + str(err) if not ok else "another rank"))
    ctx._comm_key = key


def omp_sharded(D, B, k, eps=None, group=None, solver=None, device=None, method="exact"):
    """Solve omp for every column of B (M x nsig, identical on all ranks) with the ranks of the
    default (or given) process group; every rank returns the full (idx, val, nnz) arrays.

    method: "exact" = csmp_omp_batch (single-signal sweeps, three signals pipelined), "mfma" = csmp_omp_batch_mfma
    (the batched variant BASELINE configs[3] names: bf16 MFMA screening GEMM + Float64 rescoring; same results).
    With the "nccl" backend the whole exchange runs inside the library (csmp_omp_sharded: the block's solves, the packing, ONE
    ncclAllGather over xGMI and the unpacking, all in device memory).  `solver(B_local, k, eps) -> (idx, val, nnz)` replaces the HIP path; the gloo CPU tests
    inject a stand-in there to exercise the sharding/gather logic only."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nsig = B.shape[1]
    lo, hi = shard_range(nsig, rank, world)
    backend = dist.get_backend(group)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    if method not in ("exact", "mfma"):
        raise ValueError('omp_sharded: method must be "exact" or "mfma"')
    if solver is None and dev.type == "cuda" and backend == "nccl":
        # the whole exchange inside the library (csmp_omp_sharded): solves, packing, ONE ncclAllGather on the context's stream,
        # unpacking -- device memory throughout; the process group only carries the 128-byte communicator id, once
        eps = D.eps if eps is None else eps
        library_comm(D.ctx, group)
        Bl = torch.from_numpy(np.ascontiguousarray(np.asarray(B)[:, lo:hi].T)).to(dev)  # (n, M): rows = signals
        idx = torch.full((nsig, k), -1, dtype=torch.int64, device=dev)
        val = torch.zeros((nsig, k), dtype=torch.float64, device=dev)
        nnz = torch.zeros(nsig, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        D.ctx.omp_sharded_device(Bl, nsig, k, eps, idx, val, nnz, method)
        return idx.T.cpu().numpy(), val.T.cpu().numpy(), nnz.cpu().numpy()
    if solver is None and dev.type == "cuda":
        # device-resident path under a host-side collective (gloo rehearsals): results packed on the GPU
        eps = D.eps if eps is None else eps
        Bl = torch.from_numpy(np.ascontiguousarray(np.asarray(B)[:, lo:hi].T)).to(dev)  # (n, M): rows = signals
        n = hi - lo
        idx = torch.full((n, k), -1, dtype=torch.int64, device=dev)
        val = torch.zeros((n, k), dtype=torch.float64, device=dev)
        nnz = torch.zeros(n, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        status, err = 0, None
        try:
            if n:
                (D.ctx.omp_batch_mfma_device if method == "mfma" else D.ctx.omp_batch_device)(Bl, k, eps, idx, val, nnz)
            D.ctx.sync()
        except Exception as e:  # noqa: BLE001  (this rank still joins the collective: see gather_packed)
            status, err = min(-1, int(getattr(e, "code", -1))), e
        packed = pack_t(idx, val, nnz)
    else:
        if solver is None:
            eps = D.eps if eps is None else eps
            solver = D.ctx.omp_batch_mfma if method == "mfma" else D.ctx.omp_batch
        status, err = 0, None
        try:
            idx, val, nnz = solver(np.asfortranarray(B[:, lo:hi]), k, eps)
            packed = torch.from_numpy(pack(idx, val, nnz)).to(dev)
        except Exception as e:  # noqa: BLE001
            status, err = min(-1, int(getattr(e, "code", -1))), e
            packed = torch.zeros((0, 2 * int(k) + 1), dtype=torch.float64, device=dev)
    try:
        full = gather_packed(packed, nsig, group, status=status)
    except ShardedSolveError as e:
        if err is not None:
            raise e from err  # (the failing rank keeps its own exception as the cause)
        raise
    return unpack_t(full, k)


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


# ---------------------------------------------------------------------------------- one signal, columns sharded
def column_range(N, rank, world):
    """Contiguous block of dictionary columns held by `rank` (sizes differ by at most one)."""
    return shard_range(N, rank, world)


class HipColumnShard:
    """One rank's side of the column-sharded solve on the GPU: a Context holding columns
    [col_offset, col_offset + N_local) of the dictionary (csmp_shard_* of include/csmp.h)."""

    def __init__(self, ctx, col_offset, device=None):
        import torch
        self.ctx = ctx
        self.dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        ctx.shard_config(col_offset)
        self.record_bytes = ctx.shard_record_bytes()
        self._rec = torch.zeros(self.record_bytes, dtype=torch.uint8, device=self.dev)

    def use_stream(self, handle):
        """Run this shard's kernels on the given hipStream_t (None: the library's own stream again)."""
        self.ctx.set_stream(handle)

    def begin(self, b, k):
        from . import _lib
        self.k = int(k)
        self.ctx.solver_begin(_lib.ALGO_OMP, b, max(int(k), 1))

    def sweep(self, eps, check_eps):
        self.ctx.shard_sweep(eps, check_eps, self._rec)
        return self._rec

    def append(self, recs, nrec):
        self.ctx.shard_append(recs, nrec)

    def state(self):
        idx, val, res, order, stop = self.ctx.solver_state(max(self.k, 1))
        return idx, val, order


def omp_colsharded(shards, b, k, eps, group=None):
    """omp(A, b, eps, k) (src/matchingpursuit.jl:73-82) with the dictionary's columns sharded.

    `shards`: ONE shard object (this process is a rank of `group`: one all_gather of one record per rank and
    step) or a LIST of shard objects living in this process ("virtual ranks", e.g. two contexts on one GPU: the
    records are concatenated instead of gathered).  A shard offers begin / sweep / append / state
    (`HipColumnShard`; the gloo CPU test supplies a numpy one).  Returns (idx sorted, val, selection order)."""
    import torch
    virtual = isinstance(shards, (list, tuple))
    mine = list(shards) if virtual else [shards]
    if not virtual:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            virtual = True  # one process, no group: the single shard IS the whole dictionary (world size 1)
        else:
            world = dist.get_world_size(group)
    # On the GPU the sweeps, the record exchange and the appends are ordered by ONE stream: the library's kernels are
    # enqueued on the torch stream the collective (or the concatenation) runs on; the host never waits inside the loop.
    gpu = all(hasattr(sh, "use_stream") for sh in mine)
    stream = torch.cuda.Stream(mine[0].dev) if gpu else None
    if gpu:
        for sh in mine:
            sh.use_stream(stream.cuda_stream)
    try:
        ctxmgr = torch.cuda.stream(stream) if gpu else _Null()
        with ctxmgr:
            for sh in mine:
                sh.begin(b, k)
            for t in range(int(k)):
                recs = [sh.sweep(float(eps), t > 0) for sh in mine]
                if virtual:
                    allrec, nrec = torch.cat([r.reshape(-1) for r in recs]), len(mine)
                else:
                    if recs[0].is_cuda and dist.get_backend(group) == "gloo":  # (rehearsal with ranks sharing a GPU: through host memory)
                        host = recs[0].reshape(-1).cpu()
                        allh = torch.empty(world * host.numel(), dtype=host.dtype)
                        dist.all_gather_into_tensor(allh, host, group=group)
                        allrec = allh.to(recs[0].device)
                    else:
                        allrec = torch.empty(world * recs[0].numel(), dtype=recs[0].dtype, device=recs[0].device)
                        dist.all_gather_into_tensor(allrec, recs[0].reshape(-1), group=group)  # the ONE collective of a step
                    nrec = world
                for sh in mine:
                    sh.append(allrec, nrec)
            out = mine[0].state()
    finally:
        if gpu:
            for sh in mine:
                sh.use_stream(None)
    return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
