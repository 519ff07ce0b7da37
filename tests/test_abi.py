"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/hyperdb_hip.h declares, argument errors map to the reference's exception types, and the
Python shim keeps the reference's function table.  No compute is launched (no GPU here)."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "hyperdb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hdb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hyperdb import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hyperdb_hip.h but not exported"
    assert set(_native.EXPORTS) <= set(names)
    assert set(names) <= set(_native.EXPORTS), set(names) - set(_native.EXPORTS)


def test_version_and_error_string():
    from hyperdb import _native
    assert _native.lib().hdb_version() >= 100
    assert isinstance(_native.lib().hdb_last_error(), bytes)


def test_argument_errors_without_gpu():
    from hyperdb import _native
    lib = _native.lib()
    h = ctypes.c_void_p()
    # d <= 0 -> HDB_ERR_ARG before any HIP call
    rc = lib.hdb_index_create(ctypes.byref(h), None, 0, 0, 1, 0, 0, None)
    assert rc == -1 and b"d > 0" in lib.hdb_last_error()
    rc = lib.hdb_index_create(ctypes.byref(h), None, 10, 4, 1, 0, 0, None)
    assert rc == -1 and b"null" in lib.hdb_last_error()
    rc = lib.hdb_index_create(ctypes.byref(h), ctypes.c_void_p(16), 10, 4, 9, 0, 0, None)
    assert rc == -1 and b"dtype" in lib.hdb_last_error()
    with pytest.raises(ValueError):
        _native._check(-1, "x")
    with pytest.raises(NotImplementedError):
        _native._check(-3, "x")
    with pytest.raises(RuntimeError):
        _native._check(-2, "x")
    assert _native.packed_bytes(1, 100) % 16 == 0 and _native.packed_bytes(1, 100) >= 1204
    assert _native.packed_bytes(256, 100) == (256 * 100 * 12 + 256 * 4 + 15) // 16 * 16


def test_shim_keeps_reference_function_table():
    """Names and signatures of reference hyperdb/ranking_algorithm.py:8,24,32,44,54,63,77,116,128,149."""
    import hyperdb.ranking_algorithm as ranking
    expect = {
        "get_norm_vector": ["vector"],
        "dot_product": ["vectors", "query_vector"],
        "cosine_similarity": ["vectors", "query_vector"],
        "euclidean_metric": ["vectors", "query_vector", "get_similarity_score"],
        "manhattan_distance": ["vectors", "query_vector"],
        "jaccard_similarity": ["vectors", "query_vector"],
        "pearson_correlation": ["vectors", "query_vector"],
        "check_and_binarize_vectors": ["vectors"],
        "hamming_distance": ["vectors", "query_vector"],
        "hyperDB_ranking_algorithm_sort": ["vectors", "query_vector", "top_k", "metric", "timestamps", "recency_bias"],
    }
    for name, params in expect.items():
        sig = inspect.signature(getattr(ranking, name))
        assert list(sig.parameters) == params, name
    sig = inspect.signature(ranking.hyperDB_ranking_algorithm_sort)
    assert sig.parameters["top_k"].default == 5
    assert sig.parameters["metric"].default == "cosine_similarity"
    assert sig.parameters["timestamps"].default is None
    assert sig.parameters["recency_bias"].default == 0
    assert inspect.signature(ranking.euclidean_metric).parameters["get_similarity_score"].default is True


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import hyperdb.ranking_algorithm as ranking
    from hyperdb._native import HyperDBNativeError
    with pytest.raises(HyperDBNativeError):
        ranking.hyperDB_ranking_algorithm_sort(np.eye(3), np.ones(3))
    with pytest.raises(HyperDBNativeError):
        ranking.dot_product(np.eye(3), np.ones(3))


def test_host_helpers_match_reference_semantics():
    import numpy as np
    import hyperdb.ranking_algorithm as ranking
    from oracle import ranking_oracle as orc
    x = np.array([[3.0, 4.0], [0.0, 0.0]])
    assert np.array_equal(ranking.get_norm_vector(x), orc.unit_rows(x.copy()))
    q = np.array([0.5, -2.0, 0.0, 3.0])
    a, b = q.copy(), q.copy()
    assert np.array_equal(ranking.check_and_binarize_vectors(a), orc.binarize_inplace(b)) and np.array_equal(a, b)
    with pytest.raises(ValueError):
        ranking._validate_metric("unknown_metric")
    for m in ("dot_product", "cosine_similarity", "euclidean_metric", "manhattan_distance", "jaccard_similarity",
              "pearson_correlation", "hamming_distance"):
        ranking._validate_metric(m)          # every metric of the reference's dispatch table is accepted


@pytest.mark.parametrize("parts,k,levels", [(8, 100, 7), (2, 5, 2), (5, 333, 50), (1, 64, 4)])
def test_host_merge_ties_padding_and_interleaved_rows(parts, k, levels):
    """hdb_merge_topk_host (host code of the library, no GPU) against numpy: heavy score ties across parts, row ids
    interleaved between parts (tie order follows the row id, not the part), ragged lists padded with -1, status OR."""
    import numpy as np
    from hyperdb import _native
    rng = np.random.default_rng(parts * 1000 + k)
    nq = 3
    nb = _native.packed_bytes(nq, k)
    recs = np.zeros(parts * nb, dtype=np.uint8)
    want_i = np.full((nq, k), -1, np.int64)
    want_s = np.full((nq, k), -np.inf, np.float32)
    want_st = np.zeros(nq, np.int32)
    views = [_native.record_views(recs[p * nb:(p + 1) * nb], nq, k) for p in range(parts)]
    for p in range(parts):
        views[p][0][:], views[p][1][:] = -1, -np.inf
    for q in range(nq):
        rows = rng.permutation(parts * k * 2)[:parts * k].reshape(parts, k)
        allp = []
        for p in range(parts):
            m = int(rng.integers(0, k + 1)) if q == 1 else k          # query 1: ragged lists
            sc = rng.integers(0, levels, size=m).astype(np.float32)
            order = np.lexsort((rows[p, :m], -sc))
            views[p][0][q, :m], views[p][1][q, :m] = rows[p, :m][order], sc[order]
            allp += list(zip(-sc, rows[p, :m]))
            st = int(rng.integers(0, 2)) << int(rng.integers(0, 3))
            views[p][2][q] = st
            want_st[q] |= st
        allp.sort()
        for i, (ns, r) in enumerate(allp[:k]):
            want_i[q, i], want_s[q, i] = r, -ns
    gi, gs, gst = _native.merge_topk_host(recs, parts, nq, k, np.empty(nb, dtype=np.uint8))
    assert np.array_equal(gi, want_i) and np.array_equal(gs, want_s) and np.array_equal(gst, want_st)
