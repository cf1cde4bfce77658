"""GPU parity tests (pytest -m gpu) of round 6's launch forms of the SAME arithmetic: csmp_omp_batch with two pipelines of three
signals side by side (a twin context and stream, one workgroup per CU, append stages and sweep as two launches), the product
sweep with its columns handed out at run time (csmp_tune sweep_dyn), and the window clock of csmp_profile_window.  Each must give
the bits of the plain form -- every column's sum is one wave's, in one lane order, whoever sweeps it -- and the oracle's supports
(src/matchingpursuit.jl:62-91, 181-185)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def close(v, ref, tol=1e-9):
    return np.allclose(v, ref, rtol=tol, atol=tol * (float(np.max(np.abs(ref))) if len(ref) else 0.0))


def signals(cs, A, k, nsig, seed):
    rng = np.random.default_rng(seed)
    m = A.shape[1]
    cols = []
    for _ in range(nsig):
        xs = cs.sparse_vector(m, k, rng=rng)
        cols.append(cs.perturb(A[:, xs.nzind].astype(np.float64) @ xs.nzval, 5e-3, rng=rng))
    return np.asfortranarray(np.stack(cols, axis=1))


@pytest.mark.parametrize("cfg", [(256, 1024, 12, 6, np.float64), (256, 1024, 12, 7, np.float64), (1000, 3000, 9, 13, np.float32),
                                 (4096, 2500, 10, 12, np.float32), (2048, 1500, 8, 8, np.float64), (300, 700, 5, 17, np.float32)])
def test_two_pipelines_give_the_bits_of_one(cs, oracle, cfg):
    """csmp_omp_batch runs two pipelines side by side, the second on a twin context (host/omp.hpp: omp_ticks_pair), in rounds of 3 + 3
    signals, then 1 + 1, then a lone one (host/forward.hpp).  Whole rounds only (6, 12), a lone last signal (7, 13), rounds of 1 + 1
    (8), both (17)."""
    n, m, k, nsig, dtype = cfg
    eps = float(np.finfo(dtype).eps)
    A, _, _ = cs.sparse_data(n=n, m=m, k=k, rng=n + m + nsig, dtype=dtype)
    d = cs.Dictionary(A)
    B = signals(cs, A, k, nsig, nsig)
    out = {}
    for mode in (1, 2, 0):  # one pipeline; two; the automatic choice (two)
        d.ctx.tune("pipelines", mode)
        out[mode] = d.ctx.omp_batch(B, k, eps)
    d.ctx.tune("pair_split", 1)  # two pipelines with the fused tick (one launch per tick under the large LDS request)
    out["fused"] = d.ctx.omp_batch(B, k, eps)
    d.ctx.tune("pair_split", 0)
    for key in (2, 0, "fused"):
        for a, b in zip(out[1], out[key]):
            assert np.array_equal(a, b), key
    idx, val, nnz = out[0]
    for s in range(nsig):
        ref = oracle.omp(A, B[:, s], k, eps)
        assert nnz[s] == len(ref[0]) and np.array_equal(idx[:nnz[s], s], ref[0]), s
        assert close(val[:nnz[s], s], ref[1]), s
    d.close()


@pytest.mark.parametrize("cfg", [(256, 1024, 10, 6, np.float64), (1024, 3000, 9, 7, np.float32), (512, 2048, 8, 11, np.float32), (256, 700, 6, 2, np.float64)])
def test_two_pipelines_of_forward_regression_give_the_bits_of_one(cs, oracle, cfg):
    """csmp_fr_batch side by side on a twin context (host/forward.hpp: fr_ticks_pair, the ticks under the one-workgroup-per-CU LDS
    request): rounds of 3 + 3, 1 + 1 and a lone signal; supports, coefficients and counts of one pipeline bit for bit, and the oracle's
    forward regression (src/forward.jl:44-72)."""
    n, m, k, nsig, dtype = cfg
    A, _, _ = cs.sparse_data(n=n, m=m, k=k, rng=n + m + nsig, dtype=dtype)
    d = cs.Dictionary(A)
    B = signals(cs, A, k, nsig, 3 * nsig)
    out = {}
    for mode in (1, 2):
        d.ctx.tune("pipelines", mode)
        out[mode] = d.ctx.fr_batch(B, k, 0.0, 0.0)
    for a, b in zip(out[1], out[2]):
        assert np.array_equal(a, b)
    idx, val, nnz = out[2]
    for s in range(nsig):
        ref = oracle.fr(A, B[:, s], k, 0.0, 0.0)
        assert nnz[s] == len(ref[0]) and np.array_equal(idx[:nnz[s], s], ref[0]), s
        assert close(val[:nnz[s], s], ref[1]), s
    d.close()


def test_two_pipelines_device_buffers_and_repeated_calls(cs, oracle):
    """device-resident signals and results (the bench's form), the same context called again with another batch size: the twin's
    stream joins the context's before the call returns its work to the caller's stream order"""
    import torch
    n, m, k = 512, 2048, 10
    A, _, _ = cs.sparse_data(n=n, m=m, k=k, rng=77, dtype=np.float32)
    eps = float(np.finfo(np.float32).eps)
    d = cs.Dictionary(A)
    for nsig in (9, 6, 14):
        B = signals(cs, A, k, nsig, 100 + nsig)
        Bt = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()
        idx = torch.full((nsig, k), -1, dtype=torch.int64, device="cuda")
        val = torch.zeros((nsig, k), dtype=torch.float64, device="cuda")
        nnz = torch.zeros(nsig, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        d.ctx.omp_batch_device(Bt, k, eps, idx, val, nnz)
        d.ctx.sync()
        for s in range(nsig):
            ref = oracle.omp(A, B[:, s], k, eps)
            c = int(nnz[s])
            assert c == len(ref[0]) and np.array_equal(idx[s, :c].cpu().numpy(), ref[0]) and close(val[s, :c].cpu().numpy(), ref[1]), (nsig, s)
    d.close()


@pytest.mark.parametrize("shape", [(4096, 1300, np.float32), (1000, 4099, np.float32), (3000, 777, np.float64), (4352, 519, np.float32),
                                   (64, 3, np.float32), (4096, 130, np.float64), (12288, 205, np.float32)])
def test_dynamic_sweep_gives_the_bits_of_the_static_one(cs, oracle, shape):
    """csmp_tune(sweep_dyn): a fifth wave per workgroup claims groups of four columns from per-workgroup counters and steals from its
    neighbours' (csmp_kernels.hpp: sweep_body_dyn) -- all of them (1) or only the last 1 / n of every pool after a static head (n >= 2).
    Opt-in (measured 0-3 % slower than the static split); its c = A'r, arg-max and whole solves are the static split's bit for bit.
    N mod 4 != 0, fewer columns than waves, ragged M."""
    M, N, dtype = shape
    g = np.random.default_rng(M + N)
    A = g.standard_normal((M, N))
    A /= np.linalg.norm(A, axis=0, keepdims=True)
    A = np.asfortranarray(A.astype(dtype))
    d = cs.Dictionary(A)
    k = min(8, M // 2, N)
    B = np.asfortranarray(g.standard_normal((M, 4)))
    eps = 1e-12
    res = {}
    for mode in (0, 1, 8):  # the static split; every column claimed; a static head and the last eighth of every pool claimed
        d.ctx.tune("sweep_dyn", mode)
        cfg = d.ctx.sweep_config()
        assert cfg["dynamic"] == ((1 if mode else 0) if cfg["workgroups"] <= 512 and cfg["tick_workgroups"] <= 512 else 0), cfg
        sw = [d.ctx.sweep(B[:, 0], topk=1) for _ in range(3)]  # (three launches: the two counter sets take turns)
        for t in sw[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(sw[0], t))
        res[mode] = (sw[0], d.ctx.omp_batch(B, k, eps), d.ctx.omp(B[:, 1], k, eps))
    for mode in (1, 8):
        for part in range(3):
            for a, b in zip(res[0][part], res[mode][part]):
                assert np.array_equal(a, b), (mode, part)
    ref = np.abs(A.astype(np.float64).T @ B[:, 0])
    assert np.allclose(res[1][0][0], ref, rtol=0, atol=4e-14 * np.linalg.norm(B[:, 0]))
    assert int(res[1][0][1][0]) == int(np.argmax(ref))
    d.close()


def test_profile_window_counts_both_pipelines(cs):
    """csmp_profile_window (include/csmp_internal.h): the sampled sweep launches of the context AND of its twin as one window"""
    import torch
    n, m, k, nsig = 1024, 4096, 16, 12
    A, _, _ = cs.sparse_data(n=n, m=m, k=k, rng=5, dtype=np.float32)
    d = cs.Dictionary(A)
    B = signals(cs, A, k, nsig, 9)
    Bt = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()
    idx = torch.full((nsig, k), -1, dtype=torch.int64, device="cuda")
    val = torch.zeros((nsig, k), dtype=torch.float64, device="cuda")
    nnz = torch.zeros(nsig, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for pipes, streams in ((2, 2), (1, 1)):
        d.ctx.tune("pipelines", pipes)
        d.ctx.profile_enable(2)
        d.ctx.profile_read(reset=True)
        d.ctx.omp_batch_device(Bt, k, 1e-7, idx, val, nnz)
        d.ctx.sync()
        w = d.ctx.profile_window()
        launches, ms = d.ctx.profile_read(reset=True)
        d.ctx.profile_enable(False)
        # one pipeline: the ticks with all three stages live (all but the two that fill and the two that drain a triple); two pipelines:
        # a tick's sweep is a launch of its own and every one counts -- k per signal
        steady = (nsig // 3) * (3 * k + 2 - 4) if streams == 1 else nsig * k
        assert w["streams"] == streams, w
        assert steady - 2 * streams * 2 <= w["launches"] <= steady, (w, steady)  # (first .. last SAMPLED launch on each stream)
        assert launches >= w["launches"] // 2 - 2 and ms > 0.0
        assert 0.0 < w["mean_launch_ms"] <= w["window_ms"]
        # every launch lies inside the window: the window is at least as long as launches / streams back to back would be short of
        assert w["window_ms"] * streams >= 0.5 * w["launches"] * w["mean_launch_ms"] / max(streams, 1)
    d.close()


@pytest.mark.parametrize("cfg", [(512, 4096, 160, 0.7, 2.0, 5), (384, 3000, 150, 0.8, 3.0, 6)])
def test_ompr_many_exchanges_on_a_coherent_dictionary(cs, oracle, cfg):
    """OMPR's fast path exchanges atoms on H = (A_S'A_S)^-1 by rank-one corrections (csmp_swap.hpp) and never re-derived H, c, x: the
    advisor's round-5 finding.  Every 32 accepted exchanges they are now rebuilt from the factorisation of the support
    (OmprJob::gram_refresh_if_due).  A coherent dictionary (a shared component in every atom, cond(A_S) ~ 20) under heavy noise runs
    54-69 exchanges: the supports and iteration counts are the oracle's (which solves against b afresh after every exchange, as the
    reference does: src/twostage.jl:176), and the coefficients are THE least-squares solution on the final support."""
    M, N, k, mix, noise, seed = cfg
    g = np.random.default_rng(seed)
    sh = g.standard_normal((M, 1))
    A = g.standard_normal((M, N)) + mix * np.sqrt(M) * sh / np.linalg.norm(sh)
    A /= np.linalg.norm(A, axis=0, keepdims=True)
    A = np.asfortranarray(A.astype(np.float32))
    xs = cs.sparse_vector(N, k, rng=g)
    y = cs.perturb(A[:, xs.nzind].astype(np.float64) @ xs.nzval, noise, rng=g)
    d = cs.Dictionary(A)
    got = d.ctx.ompr(y, k, 0.0, 400)
    ref = oracle.ompr(A, y, k, 0.0, 400)
    assert got[2] >= 50 and got[2] == ref[2], (got[2], ref[2])
    assert np.array_equal(got[0], ref[0])
    S = A[:, got[0]].astype(np.float64)
    xls = np.linalg.lstsq(S, y, rcond=None)[0]
    assert np.allclose(got[1], xls, rtol=1e-9, atol=1e-10 * np.abs(xls).max()), np.abs(got[1] - xls).max()
    assert close(got[1], ref[1])
    d.close()
