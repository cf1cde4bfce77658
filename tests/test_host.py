"""CPU tests of the host-side mirror: SparseVector, the synthetic-data generators of
src/util.jl, and the drivers' argument handling / error behaviour (no GPU compute)."""
import numpy as np
import pytest


def test_sparsevector_semantics(cs):
    x = cs.spzeros(16)
    for i, v in enumerate((1.0, 2.0, 3.0)):  # test/util.jl:48-58
        x[i] = v
    assert x.nnz == 3 and x.nzind.tolist() == [0, 1, 2]
    x[9] = float("nan")  # util.jl:120 placeholder is stored
    x[5] = 0.0  # structural zero is not
    assert x.nzind.tolist() == [0, 1, 2, 9]
    x[1] = 0.0  # an existing entry keeps its slot
    assert x.nnz == 4 and x[1] == 0.0
    y = cs.SparseVector(8, [5, 1], [2.0, 3.0])
    assert y.nzind.tolist() == [1, 5] and y.nzval.tolist() == [3.0, 2.0]
    with pytest.raises(ValueError):
        cs.SparseVector(8, [1, 1], [1.0, 2.0])
    assert np.array_equal(y.to_dense(), [0, 3, 0, 0, 0, 2, 0, 0])


def test_sparse_data_matches_reference_recipe(cs):
    A, x, b = cs.sparse_data(n=32, m=64, k=3, rng=0)
    assert A.shape == (32, 64) and A.flags.f_contiguous and A.dtype == np.float64
    np.testing.assert_allclose(np.linalg.norm(A, axis=0), 1.0, rtol=1e-12)  # util.jl:26
    assert x.nnz == 3 and set(np.abs(x.nzval)) == {1.0}  # util.jl:17
    np.testing.assert_allclose(A @ x.to_dense(), b, rtol=1e-12, atol=1e-15)
    A32, x32, b32 = cs.sparse_data(n=32, m=64, k=3, rng=0, dtype=np.float32)
    assert A32.dtype == np.float32 and b32.dtype == np.float64
    np.testing.assert_array_equal(A32[:, x32.nzind].astype(np.float64) @ x32.nzval, b32)
    with pytest.raises(ValueError):
        cs.sparse_vector(3, 4)


def test_perturb_norm_is_exact(cs):
    b = np.ones(50)
    y = cs.perturb(b, 5e-3, rng=1)
    assert abs(np.linalg.norm(y - b) - 5e-3) < 1e-15  # util.jl:50-55


def test_samesupport(cs):
    x = cs.SparseVector(10, [1, 4], [1.0, -1.0])
    assert cs.samesupport(x, cs.SparseVector(10, [4, 1], [3.0, 2.0]))
    assert not cs.samesupport(x, cs.SparseVector(10, [1, 5], [1.0, 1.0]))
    assert cs.samesupport(x, x.to_dense())


def test_driver_argument_errors_come_before_any_gpu_work(cs):
    A, x, b = cs.sparse_data(16, 24, 2, rng=0)
    with pytest.raises(ValueError, match="non-negative"):
        cs.omp(A, b, -1e-3, 2)  # src/matchingpursuit.jl:74
    with pytest.raises(ValueError, match="non-negative"):
        cs.gomp(A, b, 2, -1.0, 2)  # :127
    with pytest.raises(ValueError, match="invalid for Subspace Pursuit"):
        cs.sp(A, b, 9)  # src/twostage.jl:55
    with pytest.raises(ValueError):
        cs.omp_batch(A, np.zeros((16, 2)), 2, eps=-1.0)


def test_shard_range_partitions(cs):
    for nsig in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [cs.shard_range(nsig, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nsig
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_signal_sharding_wire_layout_helpers(cs):
    """csmp_shard_range / csmp_pack_results / csmp_unpack_results (host helpers of the C ABI, no GPU involved) agree with
    the Python mirror's shard_range / pack / unpack: any host language shards and gathers the same way."""
    L = cs._lib
    for nsig, world in [(8192, 8), (5, 2), (7, 3), (3, 8), (0, 4)]:
        blocks = [L.shard_range(nsig, r, world) for r in range(world)]
        assert blocks == [cs.shard_range(nsig, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == nsig and all(b[1] == c[0] for b, c in zip(blocks, blocks[1:]))
        assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
    rng = np.random.default_rng(0)
    k, n = 6, 11
    idx = rng.integers(0, 1 << 40, size=(n, k)).astype(np.int64)
    val = rng.standard_normal((n, k))
    nnz = rng.integers(0, k + 1, size=n).astype(np.int64)
    packed = L.pack_results(idx, val, nnz)
    assert packed.shape == (n, 2 * k + 1)
    from csmp_pkg import load
    sh = load().sharded
    assert np.array_equal(packed, sh.pack(idx.T, val.T, nnz))
    i2, v2, n2 = L.unpack_results(packed, k)
    assert np.array_equal(i2, idx) and np.array_equal(v2, val) and np.array_equal(n2, nnz)


def test_dictionary_file_format_round_trip(cs, tmp_path):
    """include/csmp.h: 64-byte header ("CSMPDICT", version, dtype, M, N, ld) + the columns padded to 16 bytes.  Host-only entry
    points of the library (no GPU is touched)."""
    import struct
    rng = np.random.default_rng(0)
    for dtype, M, N in [(np.float32, 5, 7), (np.float64, 6, 3), (np.float32, 8, 2)]:
        A = np.asfortranarray(rng.standard_normal((M, N)).astype(dtype))
        path = str(tmp_path / f"d_{M}_{N}.csmp")
        cs.write_dictionary_file(path, A)
        es = np.dtype(dtype).itemsize
        vec = 16 // es
        ld = (M + vec - 1) // vec * vec
        raw = open(path, "rb").read()
        assert len(raw) == 64 + ld * N * es
        magic, version, code, m_, n_, ld_ = struct.unpack("<8sIIqqq", raw[:40])
        assert magic == b"CSMPDICT" and version == 1 and (m_, n_, ld_) == (M, N, ld) and raw[40:64] == bytes(24)
        body = np.frombuffer(raw[64:], dtype=dtype).reshape(N, ld)
        assert np.array_equal(body[:, :M].T, A) and not body[:, M:].any()
        assert cs.dictionary_file_info(path) == (M, N, dtype)
    bad = tmp_path / "bad.csmp"
    bad.write_bytes(b"not a dictionary" * 8)
    with pytest.raises(cs.CsmpError):
        cs.dictionary_file_info(str(bad))


def test_dictionary_file_golden_bytes(cs, tmp_path):
    """The on-disk format is pinned by two committed files (tests/golden/dict_*.csmp, written by csmp_dictionary_file_write and
    checked by hand against include/csmp.h's layout): today's writer must reproduce them byte for byte and the reader must
    return their shapes."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    A32 = np.asfortranarray((np.arange(15, dtype=np.float32).reshape(3, 5).T - 7) / 4)   # 5 x 3
    A64 = np.asfortranarray((np.arange(12, dtype=np.float64).reshape(4, 3).T - 5) / 8)   # 3 x 4
    for name, A in (("dict_5x3_f32.csmp", A32), ("dict_3x4_f64.csmp", A64)):
        path = str(tmp_path / name)
        cs.write_dictionary_file(path, A)
        assert open(path, "rb").read() == open(os.path.join(gold, name), "rb").read(), name
        assert cs.dictionary_file_info(os.path.join(gold, name)) == (A.shape[0], A.shape[1], A.dtype.type)
