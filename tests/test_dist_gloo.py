"""world_size-2 gloo test of the sharded driver (the N>1 path of bench.py / omp_sharded): the
per-rank solver is injected (the C oracle stands in for the HIP path -- tests may use it as the
checker) so that shard assignment, packing and the single all_gather are exercised on CPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nsig, k, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from csmp_pkg import load
    from oracle import oracle_c
    cs = load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    A, x, b = cs.sparse_data(n=48, m=160, k=k, rng=5)
    rng = np.random.default_rng(77)
    B = np.asfortranarray(np.stack([cs.perturb(A @ cs.sparse_vector(160, k, rng=rng).to_dense(), 5e-3, rng=rng) for _ in range(nsig)], axis=1))
    calls = []

    def solver(Bl, kk, eps):
        calls.append(Bl.shape[1])
        idx = -np.ones((kk, Bl.shape[1]), np.int64)
        val = np.zeros((kk, Bl.shape[1]))
        nnz = np.zeros(Bl.shape[1], np.int64)
        for s in range(Bl.shape[1]):
            i, v, _ = oracle_c.omp(A, Bl[:, s], kk, eps, nthreads=1)
            idx[:len(i), s], val[:len(i), s], nnz[s] = i, v, len(i)
        return idx, val, nnz

    idx, val, nnz = cs.omp_sharded(None, B, k, eps=1e-12, solver=solver)
    lo, hi = cs.shard_range(nsig, rank, world)
    ok = calls == [hi - lo] and idx.shape == (k, nsig)
    for s in range(nsig):  # every rank holds every signal's answer after the gather
        i, v, _ = oracle_c.omp(A, B[:, s], k, 1e-12, nthreads=1)
        ok &= nnz[s] == len(i) and np.array_equal(idx[:len(i), s], i) and np.array_equal(val[:len(i), s], v)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("nsig", [5, 8])
def test_omp_sharded_gloo_world2(nsig, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nsig, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _worker_generic(rank, world, port, nsig, k, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from csmp_pkg import load
    from oracle import oracle_c
    cs = load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    A, x, b = cs.sparse_data(n=48, m=160, k=k, rng=6)
    rng = np.random.default_rng(78)
    B = np.asfortranarray(np.stack([cs.perturb(A @ cs.sparse_vector(160, k, rng=rng).to_dense(), 5e-3, rng=rng) for _ in range(nsig)], axis=1))
    # forward regression with a residual stop: signals end with different nnz, the gather must keep them apart
    idx, val, nnz = cs.sharded_solve(B, k + 2, lambda bb: oracle_c.fr(A, bb, k + 2, 0.02, 0.0, nthreads=1))
    ok = idx.shape == (k + 2, nsig)
    for s in range(nsig):
        i, v, _ = oracle_c.fr(A, B[:, s], k + 2, 0.02, 0.0, nthreads=1)
        ok &= nnz[s] == len(i) and np.array_equal(idx[:len(i), s], i) and np.array_equal(val[:len(i), s], v)
        ok &= bool(np.all(idx[len(i):, s] == -1))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_solve_generic_gloo_world2(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_generic, args=(r, 2, port, 7, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
