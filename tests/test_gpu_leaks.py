"""GPU tests (pytest -m gpu) of what the library HOLDS and of its failing paths:
* create -> set_dictionary (resident, host-streamed, from a file) -> one solve of every family -> destroy, 200 times over, with the
  failing paths in between (an unreadable dictionary file, a signal of the wrong length, the batched path under its HBM-budget hook,
  a rank that cannot solve its block on the world-1 communicator): the device's free memory comes back to where it started, and the
  library's own count of live device blocks, page-locked bytes, registered host ranges, events and streams (csmp_live_resources,
  host/track.hpp) reads what it read before;
* every device allocation of every family made to FAIL in turn (csmp_tune fail_alloc: a real hipMalloc failure that leaves
  hipErrorOutOfMemory pending): the call reports it, nothing is left half-built, and the same context then gives the clean
  context's result bit for bit.
The reference has no such state (Julia's GC owns its arrays: src/matchingpursuit.jl:44-60); this is the price of the C ABI."""
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _families(cs, L, A, eps):
    """name -> f(ctx) -> tuple of numpy arrays, one small solve per entry point family"""
    M, N = A.shape
    rng = np.random.default_rng(3)
    k = 6
    xs = cs.sparse_vector(N, k, rng=rng)
    y = cs.perturb(A[:, xs.nzind].astype(np.float64) @ xs.nzval, 5e-3, rng=rng)
    B = np.asfortranarray(np.stack([cs.perturb(A[:, (v := cs.sparse_vector(N, k, rng=rng)).nzind].astype(np.float64) @ v.nzval, 5e-3, rng=rng)
                                    for _ in range(7)], axis=1))

    def functor(c):
        c.solver_begin(L.ALGO_OMP, y, k)
        for _ in range(3):
            c.solver_step()
        return c.solver_state(k)[:2]

    fam = {
        "omp": lambda c: c.omp(y, k, eps),
        "gomp": lambda c: c.gomp(y, 2, k, eps),
        "mp": lambda c: c.mp(y, 2 * k),
        "sp": lambda c: c.sp(y, k, 1e-12)[:2],
        "fr": lambda c: c.fr(y, k)[:2],
        "ompr": lambda c: c.ompr(y, k, 1e-9)[:2],
        "srr": lambda c: c.srr(y, k, 1e-12)[:2],
        "rmp": lambda c: c.rmp(y, k)[:2],
        "foba": lambda c: c.foba(y, 1e-2)[:2],
        "omp_batch": lambda c: c.omp_batch(B, k, eps),          # seven signals: two pipelines, a twin context and its stream
        "omp_batch_mfma": lambda c: c.omp_batch_mfma(B, k, eps),
        "gomp_batch": lambda c: c.gomp_batch(B, 2, k, eps),
        "sp_batch": lambda c: c.sp_batch(B, k, 1e-12)[:3],
        "fr_batch": lambda c: c.fr_batch(B, k)[:3],
        "sweep": lambda c: c.sweep(y, topk=5),
        "lstsq": lambda c: (c.lstsq(np.sort(xs.nzind), y),),
        "functor": functor,
    }
    return fam, y, B


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(a, b))


def _device_free():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_no_resource_leaks(cs, tmp_path):
    from csmp_pkg import load
    L = load()._lib
    gc.collect()
    A32, _, _ = cs.sparse_data(n=96, m=384, k=6, rng=11, dtype=np.float32)
    A64, _, _ = cs.sparse_data(n=64, m=300, k=6, rng=12, dtype=np.float64)
    paths = {}
    for name, A in (("a32", A32), ("a64", A64)):
        paths[name] = str(tmp_path / (name + ".csmp"))
        L.write_dictionary_file(paths[name], A)
    bad_file = str(tmp_path / "truncated.csmp")  # a good header over too few bytes: the read fails AFTER the context let its dictionary go
    with open(bad_file, "wb") as f:
        f.write(open(paths["a32"], "rb").read()[:64 + 1000])
    # one warm cycle: what the process keeps for good (the runtime's pools, RCCL's first communicator) is not a leak of a cycle
    d = cs.Dictionary(A32)
    d.ctx.comm_init(cs.comm_id(), 0, 1)
    d.ctx.omp_sharded(np.asfortranarray(np.zeros((96, 2))), 2, 2, 1e-7)
    d.close()
    gc.collect()
    base_live = L.live_resources()
    base_free = _device_free()
    cycles = 200
    for c in range(cycles):
        A = A32 if c % 2 == 0 else A64
        eps = float(np.finfo(A.dtype).eps)
        fam, y, B = _families(cs, L, A, eps)
        how = c % 3
        if how == 0:
            d = cs.Dictionary(A)                                    # resident (a library-owned copy in HBM)
        elif how == 1:
            d = cs.Dictionary(A, streamed=True)                     # the dictionary stays in host memory, mapped into the device
        else:
            ctx = L.Context(0)
            ctx.set_dictionary_file(paths["a32" if c % 2 == 0 else "a64"])
            d = None
        ctx = d.ctx if d is not None else ctx
        ctx.tune("pipelines", 2)                                     # (two pipelines for these small dictionaries too: the twin context and its stream)
        names = sorted(fam)
        for name in names[c % 4::4] if c >= 8 else names:           # (every family in the first cycles, a rotating quarter afterwards)
            fam[name](ctx)
        # ---- the failing paths, in turn
        which = c % 5
        if which == 0:
            with pytest.raises(cs.CsmpError):
                ctx.call("csmp_set_dictionary_file", os.fsencode(bad_file), L.DEVICE if c % 2 else L.HOST_STREAMED)
            with pytest.raises(cs.CsmpError):                        # the failed call leaves NO dictionary behind
                ctx.call("csmp_omp", L.ptr(y), L.F64, L.i64(3), L.C.c_double(eps), L.ptr(np.zeros(3, np.int64)), L.ptr(np.zeros(3)),
                         L.C.byref(L.i64(0)), None)
        elif which == 1:
            with pytest.raises(cs.CsmpError):
                ctx.omp(y[:-1], 3, eps)                              # length(b) != size(A, 1)
            with pytest.raises(cs.CsmpError):
                ctx.omp_batch(B[:-1], 3, eps)
        elif which == 2:
            ctx.tune("batch_budget_mib", 1)                          # csmp_omp_batch_mfma: solves in chunks / hands over to the exact batch
            ctx.omp_batch_mfma(B, 6, eps)
            ctx.tune("batch_budget_mib", 0)
        elif which == 3:
            ctx.comm_init(cs.comm_id(), 0, 1)
            with pytest.raises(cs.CsmpError):                        # this rank cannot solve its block: it still goes through the gather
                ctx.call("csmp_omp_sharded", None, L.F64, L.i64(A.shape[0]), L.i64(5), L.HOST, L.i64(4), L.C.c_double(eps), 0,
                         L.ptr(np.zeros((4, 5), np.int64, order="F")), L.ptr(np.zeros((4, 5), order="F")), L.ptr(np.zeros(5, np.int64)), L.HOST)
            ctx.omp_sharded(B, B.shape[1], 4, eps)
        else:
            ctx.tune("fail_alloc", 1 + c % 7)                        # a real allocation failure somewhere inside a batch
            try:
                ctx.omp_batch(B, 6, eps)
            except cs.CsmpError:
                pass
            ctx.tune("fail_alloc", 0)
            ctx.omp_batch(B, 6, eps)
        if d is not None:
            d.close()
        else:
            ctx.close()
    gc.collect()
    live = L.live_resources()
    assert live == base_live, (live, base_live)
    freed = _device_free()
    assert abs(freed - base_free) <= 64 << 20, (freed, base_free)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_allocation_may_fail(cs, dtype):
    """fail_alloc = n makes the n-th device allocation of solver / batch state from now fail for real.  For every family and every n
    up to the number of allocations the family makes: the failing call raises (or, past the last allocation, succeeds), and the
    SAME context then returns the clean context's result.  A state recorded over a half-built set of buffers (round 6: the batched
    path's capacity was set before its buffers existed) shows up as a GPU fault or as a different result here."""
    from csmp_pkg import load
    L = load()._lib
    A, _, _ = cs.sparse_data(n=96, m=384, k=6, rng=21, dtype=dtype)
    eps = float(np.finfo(dtype).eps)
    fam, y, B = _families(cs, L, A, eps)
    gc.collect()
    base = L.live_resources()
    for name, f in sorted(fam.items()):
        clean = cs.Dictionary(A)
        clean.ctx.tune("pipelines", 2)  # (the twin context's allocations are among those that fail in turn)
        want = f(clean.ctx)
        clean.close()
        n, seen_ok = 0, 0
        while seen_ok < 2 and n < 200:  # (two successes in a row: n is past every allocation of the call)
            n += 1
            d = cs.Dictionary(A)
            d.ctx.tune("pipelines", 2)
            d.ctx.tune("fail_alloc", n)
            try:
                got = f(d.ctx)
                assert _same(got, want), (name, n, "the call went through with a result of its own")
                seen_ok += 1
            except cs.CsmpError as e:
                seen_ok = 0
                assert e.code in (L.EHIP, L.ENOMEM), (name, n, e.code, str(e))
            d.ctx.tune("fail_alloc", 0)
            got = f(d.ctx)
            assert _same(got, want), (name, n, "after the failed call")
            d.close()
        assert n < 200, name
    gc.collect()
    assert L.live_resources() == base
