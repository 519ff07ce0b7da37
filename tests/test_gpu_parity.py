"""GPU parity tests (-m gpu): the HIP path, called through the C ABI via the drop-in shim, against

  * tests/golden/*.npz  -- outputs of the real reference (made by tests/golden/make_golden.py);
  * oracle/ranking_oracle.py on the same seeded inputs, at sizes the oracle finishes in seconds;
  * size-independent properties at BASELINE.json's full sizes (fused path == on-device exact path,
    validity of every returned row against float64-exact scores, shard-merge == global).

Tolerances (BASELINE.json north_star): scores within 1e-3 (fp16 data) / 1e-5 (fp32/fp64 data),
applied as tol*max(1,|s|); hamming bit-exact.  Index lists must match the reference except for swaps
inside that band (the fp16 reference rounds its own scores to fp16 -- SURVEY.md section 8a rule 3).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GPU_METRICS = ("dot_product", "cosine_similarity", "euclidean_metric", "hamming_distance",
               "manhattan_distance", "jaccard_similarity", "pearson_correlation")
INTEGER_METRICS = ("hamming_distance",)


@pytest.fixture(scope="module")
def ranking():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import hyperdb.ranking_algorithm as r
    return r


@pytest.fixture(scope="module")
def orc():
    from oracle import ranking_oracle
    return ranking_oracle


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    return z, json.loads(str(z["manifest"]))


def _tol(dtype, metric):
    if metric == "hamming_distance":
        return 0.0
    if metric == "jaccard_similarity":
        return 1e-6                      # ratio of two integers <= d, computed in float32 on the device
    return 1e-3 if np.dtype(dtype) == np.float16 else 1e-5


# ------------------------------------------------------------------------------------------------
# 1. the reference's own unit tests, run against the shim (tests/test_ranking_algorithm.py)
# ------------------------------------------------------------------------------------------------
class TestReferenceKnownAnswersOnGpu:
    def test_euclidean_shape_and_values(self, ranking):
        r = ranking.euclidean_metric(np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]), np.array([1, 1, 1]))
        assert r.shape == (3,) and np.all(r > 0)

    def test_euclidean_empty(self, ranking):
        with pytest.raises(ValueError):
            ranking.euclidean_metric(np.array([]), np.array([]))

    def test_cosine_values(self, ranking):
        r = ranking.cosine_similarity(np.array([[1, 0], [0, 1]]), np.array([1, 0]))
        assert np.array_equal(r, [1.0, 0.0])

    def test_hamming(self, ranking):
        r = ranking.hamming_distance(np.array([[1, 1], [0, 1], [1, 0]]), np.array([1, 1]))
        assert np.array_equal(r, [2, 1, 1])

    def test_manhattan(self, ranking):
        r = ranking.manhattan_distance(np.array([[1, 0], [0, 1]]), np.array([1, 0]))
        assert np.allclose(r, [1.0, 1 / 3])

    def test_jaccard(self, ranking):
        assert np.array_equal(ranking.jaccard_similarity(np.array([[1, 1], [1, 0], [0, 0]]), np.array([1, 1])), [1.0, 0.5, 0.0])
        assert np.array_equal(ranking.jaccard_similarity(np.array([[2, 2], [2, 0], [0, 0]]), np.array([1, 1])), [1.0, 0.5, 0.0])

    def test_pearson(self, ranking):
        r = ranking.pearson_correlation(np.array([[1, 1], [0, 1], [1, 0]]), np.array([1, 1]))
        assert np.isnan(r[0]) and r[1] != 0.0 and r[2] != 0.0
        assert np.all(np.isnan(ranking.pearson_correlation(np.array([[1, 1], [0, 0], [1, 1]]), np.array([1, 1]))))

    @pytest.mark.parametrize("metric, rb, expected", [
        ("cosine_similarity", 0, [0, 2, 1]), ("cosine_similarity", 1, [2, 0, 1]),
        ("euclidean_metric", 0, [0, 2, 1]), ("hamming_distance", 0, [0, 2, 1]), ("dot_product", 0, [0, 2, 1]),
        ("manhattan_distance", 0, [0, 2, 1]), ("jaccard_similarity", 0, [0, 2, 1]), ("pearson_correlation", 0, [0, 1, 2])])
    def test_sort(self, ranking, metric, rb, expected):
        V = np.array([[1, 0], [0, 1], [0.5, 0.5]])
        idx, _ = ranking.hyperDB_ranking_algorithm_sort(
            V, np.array([1, 0]), metric=metric, timestamps=[1627825200.0, 1627911600.0, 1627998000.0], recency_bias=rb)
        assert list(idx) == expected

    def test_unknown_metric(self, ranking):
        with pytest.raises(ValueError):
            ranking.hyperDB_ranking_algorithm_sort(np.array([[1, 0], [0, 1]]), np.array([1, 0]), metric="unknown_metric")

    def test_1d_vectors(self, ranking):
        with pytest.raises(ValueError):
            ranking.hyperDB_ranking_algorithm_sort(np.array([1, 0]), np.array([1, 0]), metric="euclidean_metric")

    def test_nan(self, ranking):
        with pytest.raises(ValueError):
            ranking.hyperDB_ranking_algorithm_sort(np.array([[1, 0], [0, 1], [np.nan, np.nan]]), np.array([1, 0]))
        with pytest.raises(ValueError):
            ranking.hyperDB_ranking_algorithm_sort(np.array([[1.0, 0], [0, 1]]), np.array([np.nan, 0]))


# ------------------------------------------------------------------------------------------------
# 2. golden vectors produced by the real reference
# ------------------------------------------------------------------------------------------------
def test_kat_golden_scores(ranking, golden_dir):
    z, cases = _load(golden_dir, "kat.npz")
    for c in cases:
        if c["kind"] != "sort" or c["metric"] not in GPU_METRICS:
            continue
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(z["sort.V"].copy(), z["sort.q"].copy(), metric=c["metric"],
                                                         timestamps=list(z["sort.ts"]), recency_bias=c["recency_bias"])
        assert idx.dtype == np.int64 and sc.dtype == np.float64
        assert list(idx) == list(z[c["name"] + ".idx"]), c["name"]
        assert np.allclose(sc, z[c["name"] + ".scores"], atol=1e-6, rtol=0), c["name"]


def test_sweep_golden_topk(ranking, orc, golden_dir):
    """All (matrix, query, metric, k, recency) cases of sweep.npz for the four GPU metrics."""
    z, cases = _load(golden_dir, "sweep.npz")
    handles = {}
    n_checked = 0
    for c in cases:
        if c["metric"] not in GPU_METRICS:
            continue
        V, q = z[c["mat"] + ".V"], z[f"{c['mat']}.{c['query']}"]
        if c["mat"] not in handles:
            handles[c["mat"]] = ranking.register_vectors(V)
        ts = {"none": None, "unix": z[c["mat"] + ".ts"], "small": z[c["mat"] + ".ts_small"]}[c["recency"]]
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(handles[c["mat"]], q.copy(), top_k=c["top_k"], metric=c["metric"],
                                                         timestamps=ts, recency_bias=c["recency_bias"])
        ref_idx, ref_sc = z[c["name"] + ".idx"], z[c["name"] + ".scores"]
        tol = _tol(V.dtype, c["metric"])
        bias = None if ts is None else c["recency_bias"] * np.exp(ts - np.max(ts))
        if bias is not None:
            tol = max(tol, 1e-5)        # the recency term is a float32 on the device (integer scores stop being integers)
        orc.check_topk(idx, sc, V, q, c["metric"], c["top_k"], bias=bias, tol=tol)
        if c["metric"] in ("hamming_distance", "jaccard_similarity") and bias is None:
            # massive ties: bit-exact (hamming) / float32-exact (jaccard = ratio of two small integers) score
            # multiset; tie order canonical on our side, arbitrary in the reference
            assert np.allclose(np.sort(sc), np.sort(ref_sc), rtol=1e-6, atol=0, equal_nan=True), c["name"]
        else:
            assert orc.same_result_modulo_ties(idx, sc, ref_idx, ref_sc, tol), c["name"]
        n_checked += 1
    for h in handles.values():
        h.close()
    assert n_checked == 7 * 8 * 2 * 4 + 7 * 8 * 2       # metrics x mats x queries x k  + recency cases


def test_sweep_golden_full_vectors(ranking, golden_dir):
    z, cases = _load(golden_dir, "sweep.npz")
    seen = set()
    for c in cases:
        key = (c["mat"], c["query"], c["metric"])
        if key in seen or c["metric"] not in GPU_METRICS:
            continue
        seen.add(key)
        V, q = z[c["mat"] + ".V"], z[f"{c['mat']}.{c['query']}"]
        fn = getattr(ranking, c["metric"])
        got = fn(V, q.copy())
        want = z[f"{c['mat']}.{c['query']}.{c['metric']}.full"]
        assert got.shape == want.shape and got.dtype == want.dtype, key
        if c["metric"] == "hamming_distance":
            assert np.array_equal(got, want), key
        else:
            tol = _tol(V.dtype, c["metric"])
            g, w = got.astype(np.float64), want.astype(np.float64)
            assert np.array_equal(np.isnan(g), np.isnan(w)), key           # pearson / jaccard NaN rules
            ok = ~np.isnan(w)
            assert np.all(np.abs(g[ok] - w[ok]) <= 2 * tol * np.maximum(1, np.abs(w[ok]))), (key, np.abs(g[ok] - w[ok]).max())


def test_edge_golden(ranking, orc, golden_dir, capsys):
    z, cases = _load(golden_dir, "edge.npz")
    for c in cases:
        name = c["name"]
        if c["metric"] not in GPU_METRICS:
            continue
        kw = {k: c[k] for k in ("top_k", "metric", "recency_bias") if k in c}
        if name + ".ts" in z.files:
            kw["timestamps"] = z[name + ".ts"]
        q = z[name + ".q"].copy()
        V = z[name + ".V"]
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(V.copy(), q, **kw)
        printed = capsys.readouterr().out
        ref_idx, ref_sc = z[name + ".idx"], z[name + ".scores"]
        assert np.asarray(idx).shape == ref_idx.shape, name
        assert np.asarray(sc).shape == ref_sc.shape, name          # includes the (1,1) single-row quirk
        assert printed == c["printed"], name
        if c["metric"] in ("hamming_distance", "jaccard_similarity"):
            assert np.array_equal(q, z[name + ".q_after"]), name    # in-place binarisation of the query
            assert np.allclose(np.sort(np.asarray(sc).ravel()), np.sort(ref_sc.ravel()), rtol=1e-6, atol=0), name
            continue
        tol = _tol(V.dtype, c["metric"])
        if len(ref_idx) == 0:
            continue
        assert np.allclose(np.asarray(sc, dtype=np.float64).ravel(), ref_sc.ravel(), atol=10 * tol, rtol=0), name
        if name in ("duplicate_rows", "zero_rows_cosine", "zero_query_cosine"):
            # exact float ties: reference order arbitrary, ours canonical -> compare score lists + validity
            orc.check_topk(idx, sc, V, q.reshape(-1), c["metric"], c["top_k"], tol=tol)
            if name == "duplicate_rows":
                ci, cs = orc.canonical(idx, sc)
                assert np.array_equal(ci, np.asarray(idx)), "ties must come out by ascending index"
        else:
            assert list(np.asarray(idx).ravel()) == list(ref_idx.ravel()), name


# ------------------------------------------------------------------------------------------------
# 3. oracle on seeded inputs at sizes it finishes in seconds (fused sampled-threshold path, N > 8192)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,n,d", [(np.float32, 100_000, 384), (np.float16, 120_000, 384),
                                       (np.float16, 60_000, 768), (np.float64, 30_000, 96),
                                       (np.float32, 50_000, 100), (np.float16, 40_001, 50)])
def test_oracle_parity_medium(ranking, orc, dtype, n, d):
    rng = np.random.default_rng(n + d)
    V = rng.standard_normal((n, d)).astype(np.float32).astype(dtype)
    h = ranking.register_vectors(V)
    try:
        for metric in GPU_METRICS:
            for qi in range(2):
                q = (rng.standard_normal(d) if qi == 0 else V[n // 5].astype(np.float64) + 0.1 * rng.standard_normal(d))
                q = q.astype(dtype)
                idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q.copy(), top_k=100, metric=metric)
                oi, osc = orc.rank(V, q.copy(), top_k=100, metric=metric)
                tol = _tol(dtype, metric)
                orc.check_topk(idx, sc, V, q, metric, 100, tol=tol)
                if metric in ("hamming_distance", "jaccard_similarity"):
                    assert np.allclose(sc, np.sort(osc)[::-1], rtol=1e-6, atol=0), (metric, qi)
                else:
                    assert orc.same_result_modulo_ties(idx, sc, oi, osc, tol), (metric, qi)
        assert h.index.stat("path") in (1, 2)
    finally:
        h.close()


def test_config2_fp32_1m_cosine(ranking, orc):
    """BASELINE config 2: N=1M d=384 fp32 single-query cosine top-100, against the oracle itself."""
    n, d = 1_000_000, 384
    rng = np.random.default_rng(1234)
    V = np.empty((n, d), dtype=np.float32)
    for lo in range(0, n, 100_000):
        V[lo:lo + 100_000] = rng.standard_normal((100_000, d), dtype=np.float32)
    q = np.random.default_rng(4321).standard_normal(d).astype(np.float32)
    h = ranking.register_vectors(V)
    try:
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q, top_k=100, metric="cosine_similarity")
        assert h.index.stat("path") == 1                                   # fused sampled-threshold path
        oi, osc = orc.rank(V, q, top_k=100, metric="cosine_similarity")
        assert list(idx) == list(oi)
        assert np.all(np.abs(sc - osc) <= 1e-5)
        idx2, sc2 = ranking.hyperDB_ranking_algorithm_sort(h, q, top_k=100, metric="euclidean_metric")
        oi2, osc2 = orc.rank(V, q, top_k=100, metric="euclidean_metric")
        assert list(idx2) == list(oi2) and np.all(np.abs(sc2 - osc2) <= 1e-5)
    finally:
        h.close()


def _slices(n, count=10, rows=200_000):
    """`count` row windows of `rows` rows spread evenly over [0, n), the first at row 0 and the last ending at n."""
    rows = min(rows, n)
    if count == 1 or rows == n:
        return [(0, rows)]
    return [(lo, lo + rows) for lo in (int(round(i * (n - rows) / (count - 1))) for i in range(count))]


def _assert_nothing_left_out(orc, V_dev, q, metric, idx_row, sc_row, tol, bias_dev=None, count=10, rows=200_000, row_base=0):
    """Independent omission check over the WHOLE row range: float64 scores (oracle.exact_scores, host) of `count`
    windows spread from the first to the last row of the device matrix; any row that beats the k-th returned score by
    more than the rounding band must be among the returned rows."""
    n = int(V_dev.shape[0])
    kth = float(np.min(sc_row))
    returned = set(int(i) for i in idx_row)
    slack = 2.0 * tol * max(1.0, abs(kth)) if tol else 0.0
    for lo, hi in _slices(n, count, rows):
        b = None if bias_dev is None else bias_dev[lo:hi].cpu().numpy().astype(np.float64)
        ex = orc.exact_scores(V_dev[lo:hi].cpu().numpy(), q, metric, bias=b)
        better = np.nonzero(ex > kth + slack)[0] + lo + row_base
        missing = [int(r) for r in better if int(r) not in returned]
        assert not missing, f"rows {missing[:5]} in [{lo},{hi}) beat the k-th score {kth} but were left out ({metric})"


# ------------------------------------------------------------------------------------------------
# 4. properties at full size (N=10M d=384 fp16, top-100): fused == exact on device, validity of
#    every returned row against float64 scores of those rows, recency, batches
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big_fp16():
    import torch
    n, d, blk = 10_000_000, 384, 250_000
    V = torch.empty((n, d), dtype=torch.float16, device="cuda")
    for b in range(n // blk):
        g = torch.Generator(device="cuda").manual_seed(1234 + b)
        V[b * blk:(b + 1) * blk] = torch.randn((blk, d), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    g = torch.Generator(device="cuda").manual_seed(4321)
    Q = torch.randn((8, d), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    from hyperdb._native import GpuIndex
    ix = GpuIndex(V)
    yield ix, V, Q
    ix.close()


@pytest.mark.parametrize("metric", ["cosine_similarity", "dot_product", "euclidean_metric"])
def test_full_size_fused_equals_exact(big_fp16, orc, metric):
    import torch
    from hyperdb._native import METRIC_IDS
    ix, V, Q = big_fp16
    mid = METRIC_IDS[metric]
    idx, sc, st = ix.topk_device(Q[:4], 100, mid)
    assert int(st.abs().sum().item()) == 0, "sampled threshold failed on iid data"
    assert ix.stat("path") == 1
    eidx, esc, _ = ix.topk_device(Q[:4], 100, mid, exact=True)
    assert ix.stat("path") == 2
    assert torch.equal(idx, eidx) and torch.equal(sc, esc)
    # validity of the returned rows against float64 scores computed on the host from those rows only,
    # and against a float64 threshold count on a 1M-row slice
    idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
    for qi in range(2):
        rows = V[idx[qi]].cpu().numpy()
        ex = orc.exact_scores(rows, Q[qi].cpu().numpy(), metric)
        assert np.all(np.abs(ex - sc_h[qi]) <= 1e-3 * np.maximum(1, np.abs(ex)))
        assert np.all(np.diff(sc_h[qi]) <= 0) and np.unique(idx_h[qi]).size == 100
        # omission check by an independent scorer on 10 windows from row 0 to the LAST row (byte offsets past 4 GiB)
        _assert_nothing_left_out(orc, V, Q[qi].cpu().numpy(), metric, idx_h[qi], sc_h[qi], 1e-3)


def test_full_size_hamming_exact_properties(big_fp16, orc):
    from hyperdb._native import METRIC_IDS
    ix, V, Q = big_fp16
    import torch
    idx, sc, st = ix.topk_device(Q[:2], 100, METRIC_IDS["hamming_distance"])
    assert ix.stat("path") == 1 and int(st.abs().sum().item()) == 0, "random data: the sampled threshold must hold for hamming"
    ei, es, _ = ix.topk_device(Q[:2], 100, METRIC_IDS["hamming_distance"], exact=True)
    assert ix.stat("path") == 2
    assert torch.equal(idx, ei) and torch.equal(sc, es), "sampled-threshold path and exact selection must agree bit for bit"
    idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
    for qi in range(2):
        rows = V[idx[qi]].cpu().numpy()
        ex = orc.exact_scores(rows, Q[qi].cpu().numpy(), "hamming_distance")
        assert np.array_equal(ex, sc_h[qi].astype(np.float64))              # bit-exact integer scores
        assert np.all(np.diff(sc_h[qi]) <= 0)
        # canonical tie order: equal scores by ascending index
        for s in np.unique(sc_h[qi]):
            run = idx_h[qi][sc_h[qi] == s]
            assert np.all(np.diff(run) > 0)
        # boundary: no row of 10 windows spread over the whole matrix beats the k-th score unless returned (bit-exact)
        _assert_nothing_left_out(orc, V, Q[qi].cpu().numpy(), "hamming_distance", idx_h[qi], sc_h[qi], 0.0)


def test_full_size_batch_equals_singles(big_fp16):
    import torch
    from hyperdb._native import METRIC_IDS
    ix, V, Q = big_fp16
    mid = METRIC_IDS["dot_product"]
    bi, bs, st = ix.topk_device(Q, 100, mid)
    assert int(st.abs().sum().item()) == 0
    for qi in (0, 5, 7):
        si, ss, _ = ix.topk_device(Q[qi:qi + 1], 100, mid)
        assert torch.equal(si[0], bi[qi])
        assert torch.allclose(ss[0], bs[qi], rtol=1e-6, atol=1e-6)


def test_full_size_recency_and_mask(big_fp16, orc):
    import torch
    from hyperdb._native import METRIC_IDS
    ix, V, Q = big_fp16
    n = ix.n
    g = torch.Generator(device="cuda").manual_seed(99)
    ts = 1.7e9 + torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 30 * 86400.0
    ix.set_recency(ts, 0.5)
    try:
        mid = METRIC_IDS["cosine_similarity"]
        idx, sc, st = ix.topk_device(Q[:1], 100, mid)
        eidx, esc, _ = ix.topk_device(Q[:1], 100, mid, exact=True)
        assert torch.equal(idx, eidx) and torch.equal(sc, esc)
        # the newest documents must dominate: bias up to 0.5 vs cosine ~ +-0.2
        ts_h = ts[idx[0]].cpu().numpy()
        bias = 0.5 * np.exp(ts_h - float(ts.max().item()))
        rows = V[idx[0]].cpu().numpy()
        ex = orc.exact_scores(rows, Q[0].cpu().numpy(), "cosine_similarity") + bias
        assert np.all(np.abs(ex - sc[0].cpu().numpy()) <= 1e-3)
    finally:
        ix.set_bias(None)
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[::7] = 1
    ix.set_row_mask(mask)
    try:
        idx, sc, st = ix.topk_device(Q[:1], 50, METRIC_IDS["dot_product"])
        if int(st[0].item()) != 0:
            idx, sc, _ = ix.topk_device(Q[:1], 50, METRIC_IDS["dot_product"], exact=True)
        assert torch.all(idx[0] % 7 == 0)
    finally:
        ix.set_row_mask(None)


# ------------------------------------------------------------------------------------------------
# 5. exact path: heavy ties, forced overflow fallback, merge of shards
# ------------------------------------------------------------------------------------------------
def test_massive_float_ties_and_fallback(ranking, orc):
    rng = np.random.default_rng(5)
    base = rng.standard_normal((50, 64)).astype(np.float32)
    V = np.tile(base, (2000, 1))                       # 100k rows, every score value repeated 2000 times
    q = rng.standard_normal(64).astype(np.float32)
    h = ranking.register_vectors(V)
    try:
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q, top_k=100, metric="dot_product")
        ex = orc.exact_scores(V, q, "dot_product")
        best = np.max(ex)
        assert np.allclose(sc, best, rtol=1e-5)         # top-100 all tie at the best score
        want = np.nonzero(np.isclose(ex, best, rtol=1e-6))[0][:100]
        assert np.array_equal(idx, want), "ties must resolve to the lowest row indices"
    finally:
        h.close()


@pytest.mark.parametrize("metric", ["hamming_distance", "jaccard_similarity"])
@pytest.mark.parametrize("d", [8, 16, 40])
def test_bit_metrics_coarse_levels_fall_back(ranking, orc, metric, d):
    """Few bits per row: every score level holds far more rows than the candidate list, so the sampled threshold
    overflows and the call must come back through the exact selection -- same rows and scores as the oracle."""
    rng = np.random.default_rng(d)
    n, k = 150_000, 100
    V = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    h = ranking.register_vectors(V)
    try:
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q.copy(), top_k=k, metric=metric)
        oi, osc = orc.rank(V, q.copy(), top_k=k, metric=metric)
        orc.check_topk(idx, sc, V, q.copy(), metric, k, tol=0.0 if metric == "hamming_distance" else 1e-6)
        assert np.array_equal(np.sort(sc)[::-1], sc)
        assert np.allclose(sc, np.asarray(osc, dtype=np.float64), rtol=0, atol=0 if metric == "hamming_distance" else 1e-6)
        for s in np.unique(sc):                                  # the build's tie rule: equal scores by ascending row
            run = idx[sc == s]
            assert np.all(np.diff(run) > 0)
    finally:
        h.close()


def test_shard_merge_equals_global(orc):
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS, merge_topk, merge_topk_packed, packed_bytes
    rng = np.random.default_rng(11)
    n, d, k, parts = 90_000, 128, 100, 3
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((5, d)).astype(np.float16)
    mid = METRIC_IDS["cosine_similarity"]
    whole = GpuIndex(V)
    gi, gs, _ = whole.topk_device(Q, k, mid)
    per = n // parts
    shards = [GpuIndex(V[p * per:(p + 1) * per], row_base=p * per) for p in range(parts)]
    outs = [s.topk_device(Q, k, mid) for s in shards]
    mi, ms = merge_topk(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), k)
    assert torch.equal(mi, gi) and torch.equal(ms, gs)
    nb = packed_bytes(5, k)
    rec = torch.zeros((parts, nb), dtype=torch.uint8, device="cuda")
    for p, s in enumerate(shards):
        s.topk_packed(Q, k, mid, rec[p])
    pi, ps, pst = merge_topk_packed(rec, parts, 5, k)
    assert torch.equal(pi, gi) and torch.equal(ps, gs) and int(pst.abs().sum().item()) == 0
    # the host flavour of the exchange: every shard's record in host memory, hdb_merge_topk_host
    from hyperdb._native import merge_topk_host
    host = np.concatenate([s.topk_record_host(Q, k, mid)[:nb].copy() for s in shards])
    hi, hs, hst = merge_topk_host(host, parts, 5, k, np.empty(nb, dtype=np.uint8))
    assert np.array_equal(hi, gi.cpu().numpy()) and np.array_equal(hs, gs.cpu().numpy()) and not hst.any()
    for s in shards + [whole]:
        s.close()


def _two_rank_worker(rank, world, port, n, d, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "local-hyperdb_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # bookkeeping collectives on the host; both ranks share cuda:0
    try:
        import torch
        import bench
        from hyperdb._native import GpuIndex, METRIC_IDS
        from hyperdb.sharded import ShardedIndex
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        V, lo, hi = bench.make_shard(n, d, torch.float16, rank, world, dev)
        local = GpuIndex(V, device=dev, row_base=lo)
        sh = ShardedIndex(local, n_total=n, group=dist.group.WORLD, exchange="host")
        assert sh._hx is not None
        Q = bench.make_queries(12, d, torch.float16, dev).float()
        mid = METRIC_IDS["cosine_similarity"]
        res = [sh.query(Q[i:i + 1], 100, mid) for i in range(10)]          # one exchange per query, slots reused
        bi, bs = sh.query(Q[10:12], 37, mid)
        sh.close()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=np.stack([r[0][0] for r in res]), sc=np.stack([r[1][0] for r in res]), bi=bi, bs=bs)
        local.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_host_exchange_equals_single_index(tmp_path):
    """The launch model of bench.py with the product engine: two processes, each with its row shard in HBM (both on this
    one GPU), swap their host records through the shared-memory exchange and merge with hdb_merge_topk_host -- every rank
    gets exactly what one index over all rows returns."""
    import torch
    import torch.multiprocessing as mp
    import bench
    from hyperdb._native import GpuIndex, METRIC_IDS
    n, d = 750_000, 384                                   # three 250k-row blocks: shards of 250k and 500k rows
    port = 36500 + (os.getpid() % 2000)
    mp.spawn(_two_rank_worker, args=(2, port, n, d, str(tmp_path)), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    Q = bench.make_queries(12, d, torch.float16, dev).float()
    whole = GpuIndex(V)
    try:
        mid = METRIC_IDS["cosine_similarity"]
        r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
        for key in ("idx", "sc", "bi", "bs"):
            assert np.array_equal(r0[key], r1[key]), key
        for i in range(10):
            wi, ws = whole.topk(Q[i:i + 1], 100, mid)
            assert np.array_equal(r0["idx"][i], wi[0]) and np.array_equal(r0["sc"][i], ws[0]), i
        wi, ws = whole.topk(Q[10:12], 37, mid)
        assert np.array_equal(r0["bi"], wi) and np.array_equal(r0["bs"], ws)
    finally:
        whole.close()


@pytest.mark.parametrize("parts,k,levels", [(8, 100, 7), (8, 1024, 3), (2, 5, 2), (5, 333, 50), (1, 64, 4)])
def test_merge_ties_padding_and_interleaved_rows(parts, k, levels):
    """hdb_merge_topk against a host merge: heavy score ties across parts, row ids interleaved between parts
    (tie order must follow the row id, not the part number), ragged lists padded with -1."""
    import torch
    from hyperdb._native import merge_topk
    rng = np.random.default_rng(parts * 1000 + k)
    nq = 4
    idx = np.full((parts, nq, k), -1, np.int64)
    sc = np.full((parts, nq, k), -np.inf, np.float32)
    want_i = np.full((nq, k), -1, np.int64)
    want_s = np.full((nq, k), -np.inf, np.float32)
    for q in range(nq):
        rows = rng.permutation(parts * k * 2)[:parts * k].reshape(parts, k)       # interleaved, unique
        allp = []
        for p in range(parts):
            m = k if (p + q) % 3 else int(rng.integers(0, k + 1))                    # ragged
            s = rng.integers(0, levels, size=m).astype(np.float32) * 0.25 - 1.0
            if m and q == 1:
                s[rng.integers(0, m)] = -np.inf                                      # a real -inf score with a valid row
            order = np.lexsort((rows[p, :m], -s))
            idx[p, q, :m], sc[p, q, :m] = rows[p, :m][order], s[order]
            allp += list(zip(-s, rows[p, :m]))
        allp.sort()
        top = allp[:k]
        want_i[q, :len(top)] = [r for _, r in top]
        want_s[q, :len(top)] = [-v for v, _ in top]
    gi, gs = merge_topk(torch.from_numpy(idx).cuda(), torch.from_numpy(sc).cuda(), k)
    assert np.array_equal(gi.cpu().numpy(), want_i)
    assert np.array_equal(gs.cpu().numpy(), want_s)


def test_rank_batch_matches_oracle(ranking, orc):
    rng = np.random.default_rng(21)
    V = rng.standard_normal((30_000, 384)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((37, 384)).astype(np.float16)
    idx, sc = ranking.rank_batch(V, Q, top_k=20, metric="dot_product")
    assert idx.shape == (37, 20) and sc.dtype == np.float64
    for qi in (0, 17, 36):
        oi, osc = orc.rank(V, Q[qi], top_k=20, metric="dot_product")
        assert orc.same_result_modulo_ties(idx[qi], sc[qi], oi, osc, 1e-3)
        orc.check_topk(idx[qi], sc[qi], V, Q[qi], "dot_product", 20, tol=1e-3)


# ------------------------------------------------------------------------------------------------
# 6. MFMA batched path (fp16, d=384): same answers as the VALU scan and as the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric", ["dot_product", "cosine_similarity"])
@pytest.mark.parametrize("nq", [256, 100, 8, 300])
def test_mfma_batch_matches_valu_and_oracle(orc, metric, nq):
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(nq)
    n, d, k = 200_003, 384, 100                       # ragged last tile on purpose
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((nq, d)).astype(np.float16)
    Q[1] = V[n - 1]                                   # exact duplicate of the very last row
    Q[2] = V[777] * 0.5
    ix = GpuIndex(V)
    try:
        mid = METRIC_IDS[metric]
        mi, ms, mst = ix.topk_device(Q, k, mid)
        assert ix.stat("mfma") == 1 and ix.stat("path") == 1
        assert int(mst.abs().sum().item()) == 0
        ix.set_option("use_mfma", 0)
        vi, vs, vst = ix.topk_device(Q, k, mid)
        assert ix.stat("mfma") == 0
        ix.set_option("use_mfma", 1)
        mi_h, ms_h, vi_h, vs_h = mi.cpu().numpy(), ms.cpu().numpy(), vi.cpu().numpy(), vs.cpu().numpy()
        # both paths accumulate exact fp16 products in fp32, in different orders: scores agree to ~1e-5 rel
        assert np.all(np.abs(ms_h - vs_h) <= 2e-5 * np.maximum(1, np.abs(vs_h)))
        for qi in range(nq):
            assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi], vi_h[qi], vs_h[qi], 2e-5), qi
        assert mi_h[1][0] == n - 1 and mi_h[2][0] == 777
        for qi in (0, 1, nq - 1):
            orc.check_topk(mi_h[qi], ms_h[qi], V, Q[qi], metric, k, tol=1e-3)
            oi, osc = orc.rank(V, Q[qi].copy(), top_k=k, metric=metric)
            assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi], oi, osc, 1e-3), qi
        # exact path through the MFMA score writer
        ei, es, _ = ix.topk_device(Q[:16], k, mid, exact=True)
        assert ix.stat("mfma") == 1 and ix.stat("path") == 2
        assert torch.equal(ei, mi[:16]) and torch.equal(es, ms[:16])
    finally:
        ix.close()


def test_mfma_with_recency_bias(orc):
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(3)
    n, d, k, nq = 150_000, 384, 50, 64
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((nq, d)).astype(np.float16)
    ts = 1.7e9 + rng.uniform(0, 86400.0 * 3, size=n)
    ix = GpuIndex(V)
    try:
        ix.set_recency(ts, 0.75)
        mid = METRIC_IDS["cosine_similarity"]
        mi, ms, mst = ix.topk_device(Q, k, mid)
        assert ix.stat("mfma") == 1 and int(mst.abs().sum().item()) == 0
        ix.set_option("use_mfma", 0)
        vi, vs, _ = ix.topk_device(Q, k, mid)
        assert torch.equal(mi, vi)
        assert torch.allclose(ms, vs, rtol=2e-5, atol=2e-5)
        bias = 0.75 * np.exp(ts - ts.max())
        orc.check_topk(mi[0].cpu().numpy(), ms[0].cpu().numpy(), V, Q[0], "cosine_similarity", k, bias=bias, tol=1e-3)
    finally:
        ix.close()


@pytest.mark.parametrize("d,nq,metric,bias", [(768, 64, "euclidean_metric", True), (768, 64, "cosine_similarity", False),
                                              (128, 40, "dot_product", False), (256, 128, "euclidean_metric", False),
                                              (512, 33, "cosine_similarity", True), (640, 16, "dot_product", True),
                                              (384, 96, "euclidean_metric", True), (1024, 48, "cosine_similarity", False),
                                              (1024, 128, "euclidean_metric", True), (1536, 32, "dot_product", False),
                                              (1536, 70, "cosine_similarity", True), (256, 200, "cosine_similarity", False),
                                              (512, 256, "euclidean_metric", True), (640, 129, "dot_product", False),
                                              (128, 256, "dot_product", True),
                                              # every multiple of 128 up to 1536 rides the matrix cores (16-row stages, 7-11 pieces per staging wave)
                                              (896, 40, "cosine_similarity", False), (1152, 33, "dot_product", True),
                                              (1280, 64, "euclidean_metric", False), (1408, 17, "cosine_similarity", True),
                                              (896, 128, "euclidean_metric", True),
                                              # round 4: ANY width that is a multiple of 8 rides the next wider geometry as one K slice (hdb_mfma_anyd.h) --
                                              # chunks past the end of a row are not fetched, the query fragments are zero there
                                              (96, 16, "dot_product", False), (96, 64, "euclidean_metric", True), (200, 64, "cosine_similarity", True),
                                              (200, 16, "euclidean_metric", False), (296, 16, "cosine_similarity", False), (304, 64, "dot_product", True),
                                              (312, 130, "euclidean_metric", False), (1000, 16, "dot_product", False),
                                              (1000, 64, "euclidean_metric", True), (8, 16, "cosine_similarity", False), (520, 40, "dot_product", False)])
def test_mfma_shapes_and_euclidean(orc, d, nq, metric, bias):
    """Config-5 shaped case (d=768, Q=64, euclidean + time decay) and the other MFMA geometries."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(d + nq)
    n, k = 60_000 + 17, 50
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((nq, d)).astype(np.float16)
    Q[0] = V[n - 5]                                   # exact duplicate: euclidean similarity must come out as 1.0
    Q[1] = (V[123].astype(np.float32) + 0.01 * rng.standard_normal(d)).astype(np.float16)   # near duplicate
    ts = 1.7e9 + rng.uniform(0, 30 * 86400.0, size=n)
    ix = GpuIndex(V)
    try:
        b = None
        if bias:
            ix.set_recency(ts, 0.5)
            b = 0.5 * np.exp(ts - ts.max())
        mid = METRIC_IDS[metric]
        mi, ms, mst = ix.topk_device(Q, k, mid)
        assert ix.stat("mfma") == 1 and ix.stat("path") == 1
        assert int(mst.abs().sum().item()) == 0
        ix.set_option("use_mfma", 0)
        vi, vs, _ = ix.topk_device(Q, k, mid)
        ix.set_option("use_mfma", 1)
        mi_h, ms_h, vi_h, vs_h = mi.cpu().numpy(), ms.cpu().numpy(), vi.cpu().numpy(), vs.cpu().numpy()
        for qi in range(nq):
            assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi], vi_h[qi], vs_h[qi], 2e-5), qi
        for qi in (0, 1, nq - 1):
            orc.check_topk(mi_h[qi], ms_h[qi], V, Q[qi], metric, k, bias=b, tol=1e-3)
        if metric == "euclidean_metric" and not bias:
            assert mi_h[0][0] == n - 5 and abs(ms_h[0][0] - 1.0) < 1e-6, "exact duplicate must score exactly 1"
            assert mi_h[1][0] == 123
    finally:
        ix.close()


@pytest.mark.parametrize("metric,scale", [("dot_product", 1e6), ("dot_product", 3e-7), ("dot_product", 8.0),
                                          ("cosine_similarity", 1e6), ("cosine_similarity", 3e-7), ("cosine_similarity", 8.0),
                                          ("euclidean_metric", 8.0)])     # |q| >> |v| makes 1/(1+dist) ill-conditioned in fp32
def test_mfma_query_magnitude(orc, metric, scale):
    """fp32 queries far outside the fp16 range (or deep in its subnormals) on an fp16 matrix: the MFMA scan multiplies
    with a power-of-two scaled fp16 copy and must agree with the VALU scan (exact fp32 queries) and the oracle."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(77)
    n, d, k = 40_000, 384, 30
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = (rng.standard_normal((5, d)) * scale).astype(np.float32)
    Q[:, ::7] *= 1e-3                                          # wide dynamic range inside one query
    mid = METRIC_IDS[metric]
    ix = GpuIndex(V)
    try:
        for sl in (slice(0, 1), slice(0, 5)):                   # single query and a small batch
            mi, ms, mst = ix.topk_device(Q[sl], k, mid)
            assert ix.stat("mfma") == 1 and int(mst.abs().sum().item()) == 0
            ix.set_option("use_mfma", 0)
            vi, vs, _ = ix.topk_device(Q[sl], k, mid)
            ix.set_option("use_mfma", 1)
            assert ix.stat("mfma") == 0
            mi_h, ms_h, vi_h, vs_h = mi.cpu().numpy(), ms.cpu().numpy(), vi.cpu().numpy(), vs.cpu().numpy()
            assert np.isfinite(ms_h).all()
            for qi in range(mi_h.shape[0]):
                ref = float(np.abs(vs_h[qi]).max()) or 1.0        # compare at unit scale whatever the query magnitude
                assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi] / ref, vi_h[qi], vs_h[qi] / ref, 1e-3), (qi, scale)
    finally:
        ix.close()


def test_mfma_degenerate_queries(ranking, orc):
    """All-zero query (every score ties at 0 -> candidate overflow -> exact selection -> the first k rows), a query
    far below fp16's range and one with a single huge element, on an fp16 matrix through the MFMA scan.  (A query of fp32
    subnormals is left out: its squared norm underflows to 0 in float32 -- in numpy as here -- so it ranks as unnormalised.)"""
    rng = np.random.default_rng(123)
    n, d, k = 30_000, 256, 25
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    h = ranking.register_vectors(V)
    try:
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, np.zeros(d, np.float32), top_k=k, metric="cosine_similarity")
        assert list(idx) == list(range(k)) and np.all(sc == 0.0)
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, np.zeros(d, np.float32), top_k=k, metric="dot_product")
        assert list(idx) == list(range(k)) and np.all(sc == 0.0)
        tiny = (rng.standard_normal(d) * 1e-17).astype(np.float32)            # far below fp16's range, ||q||^2 still a normal fp32
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, tiny.copy(), top_k=k, metric="cosine_similarity")
        orc.check_topk(idx, sc, V, tiny.astype(np.float64), "cosine_similarity", k, tol=2e-3)
        spike = rng.standard_normal(d).astype(np.float32)
        spike[7] = 1.0e15                                                       # (1e30 would overflow ||q||^2 in float32, in numpy too)
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, spike.copy(), top_k=k, metric="cosine_similarity")
        assert np.isfinite(sc).all()
        want = np.argsort(-V[:, 7].astype(np.float64) / np.linalg.norm(V.astype(np.float64), axis=1), kind="stable")[:k]
        assert set(idx.tolist()) == set(want.tolist())                        # the spike decides: cos ~ v_7 / ||v||
    finally:
        h.close()


@pytest.mark.parametrize("metric,k", [("cosine_similarity", 5000), ("hamming_distance", 3000), ("dot_product", 60_000)])
def test_large_k_full_sort_path(ranking, orc, metric, k):
    """k above HDB_MAX_K on a matrix larger than the candidate list: all-scores + stable radix sort."""
    rng = np.random.default_rng(k)
    n, d = 50_000, 64
    V = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    h = ranking.register_vectors(V)
    try:
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q.copy(), top_k=k, metric=metric)
        assert h.index.stat("path") == 3
        kk = min(k, n)
        assert idx.shape == (kk,) and sc.shape == (kk,)
        ex = orc.exact_scores(V, q, metric)
        ci, cs = orc.canonical(idx, sc)
        assert np.array_equal(ci, idx), "rows must come out in (score desc, index asc) order"
        tol = 0.0 if metric == "hamming_distance" else 1e-5
        orc.check_topk(idx, sc, V, q, metric, k, tol=tol, exact=ex)
        if metric == "hamming_distance":
            want = np.lexsort((np.arange(n), -ex))[:kk]
            assert np.array_equal(idx, want)
    finally:
        h.close()


# ------------------------------------------------------------------------------------------------
# 7. boundaries of the pipeline: candidate-cap edge, forced threshold failures, lifecycle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [8191, 8192, 8193, 8208])
def test_candidate_cap_boundary(ranking, orc, n):
    rng = np.random.default_rng(n)
    V = rng.standard_normal((n, 32)).astype(np.float32)
    q = rng.standard_normal(32).astype(np.float32)
    h = ranking.register_vectors(V)
    try:
        for k in (1, 100, 2048):
            idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q, top_k=k, metric="dot_product")
            oi, osc = orc.rank(V, q, top_k=k, metric="dot_product")
            assert list(idx) == list(oi) and np.allclose(sc, osc, rtol=1e-5, atol=1e-5), (n, k)
        # k=2048 of ~8200 rows goes straight to the exact path (k > n/32); the smaller k use the sampled threshold
        assert h.index.stat("path") == (0 if n <= 8192 else 2)
        ranking.hyperDB_ranking_algorithm_sort(h, q, top_k=100, metric="dot_product")
        assert h.index.stat("path") == (0 if n <= 8192 else 1)
    finally:
        h.close()


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_fuzz_sampled_threshold_equals_exact_selection(seed):
    """Random shapes through both selection paths of the same kernels: the sampled-threshold path (status 0) must
    return exactly the rows and scores of the exact selection -- fp16 MFMA geometries (all d, 1..300 queries, ragged
    tiles, bias) and the VALU scan (fp32, other d, manhattan)."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(seed)
    mfma_d = [128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1408, 1536]
    for case in range(10):
        use16 = rng.random() < 0.7
        d = int(rng.choice(mfma_d)) if use16 else int(rng.choice([24, 100, 384, 200]))
        n = int(rng.integers(8200, 120_000))
        nq = int(rng.choice([1, 2, 5, 16, 17, 64, 129, 200, 257, 300])) if use16 else int(rng.choice([1, 3, 6]))
        k = int(rng.choice([1, 7, 100, 257]))
        metric = str(rng.choice(["dot_product", "cosine_similarity", "euclidean_metric"] + ([] if use16 else ["manhattan_distance"])))
        V = torch.randn((n, d), generator=torch.Generator().manual_seed(seed * 100 + case))
        V = V.to(torch.float16 if use16 else torch.float32).cuda()
        Q = torch.randn((nq, d), generator=torch.Generator().manual_seed(seed * 100 + case + 50)).to(V.dtype).float().cuda()
        ix = GpuIndex(V)
        try:
            if rng.random() < 0.4:
                ix.set_bias((torch.rand(n, generator=torch.Generator().manual_seed(case)) * 0.3).float().cuda())
            mid = METRIC_IDS[metric]
            fi, fs, fst = ix.topk_device(Q, k, mid)
            path = ix.stat("path")
            ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
            ok = (fst == 0)
            assert bool(ok.all()) or path == 2, (seed, case, "status", fst.cpu().tolist())
            tag = (seed, case, n, d, nq, k, metric, use16)
            if metric == "euclidean_metric" and ix.stat("mfma"):
                # near-duplicates are re-scored AFTER selection: the sampled path re-ranks all its survivors, the exact path
                # only its k, so rows within rounding of the k-th score may swap -- the score lists agree to 1e-6
                sa, sb = torch.sort(fs[ok], dim=-1, descending=True)[0], torch.sort(es[ok], dim=-1, descending=True)[0]
                assert bool(((sa - sb).abs() <= 1e-6 * sa.abs().clamp(min=1e-3)).all()), tag
            else:
                assert torch.equal(fi[ok], ei[ok]), tag
                assert torch.equal(fs[ok], es[ok]), tag
        finally:
            ix.close()
        del V, Q
    torch.cuda.empty_cache()


@pytest.mark.parametrize("keep", [0.5, 0.02, 0.0004])
def test_row_mask_mfma_equals_valu(orc, keep):
    """Row masks (filters) on fp16 matrices: the MFMA scan sees them as a bias of -inf and must return what the VALU
    scan (mask input) returns -- single query and batch, with and without a recency bias, down to fewer than k rows."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(int(keep * 1e6))
    n, d, k = 70_000, 256, 40
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((9, d)).astype(np.float16)
    mask_h = (rng.random(n) < keep).astype(np.uint8)
    mask_h[:3] = 1
    mask = torch.from_numpy(mask_h).cuda()
    bias = torch.from_numpy((rng.random(n) * 0.2).astype(np.float32)).cuda()
    ix = GpuIndex(V)
    try:
        ix.set_row_mask(mask)
        for use_bias in (False, True):
            ix.set_bias(bias if use_bias else None)
            for metric in ("cosine_similarity", "euclidean_metric"):
                mid = METRIC_IDS[metric]
                for sl in (slice(0, 1), slice(0, 9)):
                    mi, ms = ix.topk(Q[sl], k, mid)
                    assert ix.stat("mfma") == 1
                    ix.set_option("use_mfma", 0)
                    vi, vs = ix.topk(Q[sl], k, mid)
                    ix.set_option("use_mfma", 1)
                    assert ix.stat("mfma") == 0
                    for qi in range(mi.shape[0]):
                        live = mi[qi] >= 0
                        assert np.array_equal(live, vi[qi] >= 0)
                        got = mi[qi][live & np.isfinite(ms[qi])]
                        assert mask_h[got].all(), "an excluded row came back with a finite score"
                        assert orc.same_result_modulo_ties(mi[qi], np.nan_to_num(ms[qi], neginf=-1e30), vi[qi],
                                                           np.nan_to_num(vs[qi], neginf=-1e30), 2e-5), (keep, use_bias, metric, qi)
                        if mask_h.sum() >= k:
                            assert np.isfinite(ms[qi]).all() and mask_h[mi[qi]].all()
    finally:
        ix.close()


@pytest.mark.parametrize("d,nq,metric,bias", [(384, 64, "cosine_similarity", False), (384, 130, "dot_product", True),
                                              (128, 5, "euclidean_metric", False), (256, 33, "cosine_similarity", True),
                                              (512, 17, "euclidean_metric", True), (768, 128, "dot_product", False),
                                              (384, 9, "pearson_correlation", False),
                                              # round 4: any width that is a multiple of 4 floats (GloVe 100 / 200 / 300, ...)
                                              (100, 16, "cosine_similarity", False), (300, 64, "dot_product", True), (200, 33, "euclidean_metric", False),
                                              (300, 7, "pearson_correlation", False), (500, 16, "cosine_similarity", True), (700, 20, "euclidean_metric", True)])
def test_fp32_mfma_batches_match_valu_and_oracle(orc, d, nq, metric, bias):
    """float32 matrices (the reference's default fp_precision): batches of 5+ queries ride the fp32 MFMA scan
    (v_mfma_f32_16x16x4_f32, exact fp32 products) and must agree with the VALU scan and the oracle to 1e-5."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(d * 1000 + nq)
    n, k = 50_000 + 13, 50
    V = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q[0] = V[n - 2]                                             # exact duplicate of a row in the ragged last tile
    Q[1] = V[77] + 0.05 * rng.standard_normal(d).astype(np.float32)
    ts = 1.7e9 + rng.uniform(0, 30 * 86400.0, size=n)
    ix = GpuIndex(V)
    try:
        b = None
        if bias:
            ix.set_recency(ts, 0.5)
            b = 0.5 * np.exp(ts - ts.max())
        mid = METRIC_IDS[metric]
        mi, ms, mst = ix.topk_device(Q, k, mid)
        assert ix.stat("mfma") == 1 and ix.stat("path") == 1 and int(mst.abs().sum().item()) == 0
        parts = ix.stat("f32_split") == 1          # (round 4: larger batches and wide rows multiply as bf16 parts, the exact path in float32)
        ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
        import torch
        if parts:
            ptol = 1e-5                           # (relative above 1: same_result_modulo_ties)
            for qi in range(nq):
                assert orc.same_result_modulo_ties(mi[qi].cpu().numpy(), ms[qi].cpu().numpy(), ei[qi].cpu().numpy(), es[qi].cpu().numpy(), ptol), qi
        else:
            assert torch.equal(mi, ei) and torch.equal(ms, es)
        ix.set_option("use_mfma", 0)
        vi, vs, _ = ix.topk_device(Q, k, mid)
        ix.set_option("use_mfma", 1)
        assert ix.stat("mfma") == 0
        mi_h, ms_h, vi_h, vs_h = mi.cpu().numpy(), ms.cpu().numpy(), vi.cpu().numpy(), vs.cpu().numpy()
        vtol = 1e-5
        for qi in range(nq):
            assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi], vi_h[qi], vs_h[qi], vtol), qi
        for qi in (0, 1, nq - 1):
            orc.check_topk(mi_h[qi], ms_h[qi], V, Q[qi], metric, k, bias=b, tol=1e-5)
        if metric == "euclidean_metric" and not bias:
            assert mi_h[0][0] == n - 2 and abs(ms_h[0][0] - 1.0) < 1e-6
            assert mi_h[1][0] == 77
        # up to 4 queries stay on the VALU scan (one pass at HBM speed)
        ix.topk_device(Q[:4], k, mid)
        assert ix.stat("mfma") == 0
    finally:
        ix.close()


@pytest.mark.parametrize("d,nq,metric,bias", [(128, 40, "cosine_similarity", False), (128, 128, "euclidean_metric", True),
                                              (256, 12, "dot_product", False), (256, 100, "cosine_similarity", True),
                                              (384, 33, "cosine_similarity", False), (384, 64, "euclidean_metric", False),
                                              (384, 128, "dot_product", True), (384, 150, "cosine_similarity", False),
                                              (384, 40, "pearson_correlation", False),
                                              (512, 5, "cosine_similarity", False), (512, 64, "euclidean_metric", True), (512, 100, "dot_product", False),
                                              (768, 16, "cosine_similarity", False), (768, 64, "euclidean_metric", False), (768, 130, "dot_product", True),
                                              # widths that ride a padded geometry (hdb_mfma_anyd.h) or two K slices (hdb_mfma_ksplit.hip)
                                              (100, 40, "cosine_similarity", False), (300, 64, "dot_product", True), (500, 16, "cosine_similarity", False),
                                              (700, 70, "euclidean_metric", True), (1024, 16, "cosine_similarity", False), (1536, 40, "euclidean_metric", False)])
def test_fp32_batches_as_bf16_parts_match_the_float32_flavour_and_oracle(orc, d, nq, metric, bias):
    """float32 matrices, larger batches: the rows travel as two bf16 parts (converted in LDS, in place), the queries as three exact
    parts, five bf16 MFMAs per k-step (hdb_mfma_f32s.hip; d = 512 / 768: two waves share the k-steps of a row).  Same answer as
    the v_mfma_f32_16x16x4_f32 flavour and the oracle (hyperdb/ranking_algorithm.py:29,:41,:49 on float32) within 1e-5, in the
    single launch and in the multi-kernel pipeline."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(d * 977 + nq)
    n, k = (70_000 if d <= 768 else 30_000) + 5, 50
    V = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q[0] = V[n - 3]                                             # exact duplicate of a row in the ragged last tile
    Q[1] = V[91] + 0.05 * rng.standard_normal(d).astype(np.float32)
    V[1234] *= 1.0e-3                                           # rows of very different scale
    if metric != "dot_product": V[4321] *= 3.0e3                #   (a dot product against such a row cancels to 1e-2 of |v||q|: float32 itself is no better than 1e-5 of THAT score)
    ts = 1.7e9 + rng.uniform(0, 30 * 86400.0, size=n)
    ix = GpuIndex(V)
    try:
        b = None
        if bias:
            ix.set_recency(ts, 0.5)
            b = 0.5 * np.exp(ts - ts.max())
        mid = METRIC_IDS[metric]
        pi, ps, pst = ix.topk_device(Q, k, mid)
        assert ix.stat("f32_split") == 1 and ix.stat("mfma") == 1 and int(pst.abs().sum().item()) == 0
        fused = ix.stat("fused")
        pi_h, ps_h = pi.cpu().numpy(), ps.cpu().numpy()
        ix.set_option("use_batch1", 0)                          # sample scan + threshold + filter scan + finalize: MODE 0 / 1 of the same kernels
        ki, ks, kst = ix.topk_device(Q, k, mid)
        ix.set_option("use_batch1", 1)
        assert ix.stat("f32_split") == 1 and (fused == 0 or ix.stat("fused") == 0) and int(kst.abs().sum().item()) == 0
        ki_h, ks_h = ki.cpu().numpy(), ks.cpu().numpy()
        ix.set_option("f32_split", 0)
        fi, fs, _ = ix.topk_device(Q, k, mid)
        assert ix.stat("f32_split") == 0 and ix.stat("mfma") == 1
        ix.set_option("f32_split", 1)
        fi_h, fs_h = fi.cpu().numpy(), fs.cpu().numpy()
        tol = 1e-5                                              # (relative above 1: same_result_modulo_ties)
        for qi in range(nq):
            assert orc.same_result_modulo_ties(pi_h[qi], ps_h[qi], fi_h[qi], fs_h[qi], tol), qi
            assert orc.same_result_modulo_ties(pi_h[qi], ps_h[qi], ki_h[qi], ks_h[qi], tol * 0.1), qi
        for qi in (0, 1, nq // 2, nq - 1):
            orc.check_topk(pi_h[qi], ps_h[qi], V, Q[qi], metric, k, bias=b, tol=1e-5)
        if metric == "euclidean_metric" and not bias:
            assert pi_h[0][0] == n - 3 and abs(ps_h[0][0] - 1.0) < 1e-6 and pi_h[1][0] == 91
    finally:
        ix.close()


def test_fp32_bf16_parts_keep_away_from_infinities(orc):
    """The parts of an infinite value cancel to NaN (inf - inf), np.dot keeps the infinity: a matrix with an infinite (or
    overflowing) row stays on the float32 MFMAs, a query with an infinite element is re-run through the exact path."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(8)
    n, d, nq, k = 40_000, 384, 48, 20
    V = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    mid = METRIC_IDS["dot_product"]
    for poison in (np.inf, 3.0e22):                             # 3e22 ** 2 overflows the float32 sum of squares
        W = V.copy(); W[777, 5] = poison
        ix = GpuIndex(W)
        try:
            i1, s1, st = ix.topk_views(Q, k, mid)
            assert ix.stat("f32_split") == 0 and ix.stat("mfma") == 1 and int(np.abs(st).sum()) == 0
            for qi in (0, 7, nq - 1):
                orc.check_topk(i1[qi], s1[qi], W, Q[qi], "dot_product", k, tol=1e-5)
        finally:
            ix.close()
    ix = GpuIndex(V)
    try:
        Q2 = Q.copy(); Q2[3, 11] = np.inf
        with np.errstate(invalid="ignore"):
            i2, s2, st2 = ix.topk_views(Q2, k, mid)
            i2, s2, st2 = i2.copy(), s2.copy(), st2.copy()
            assert int(np.abs(st2).sum()) == 0               # (HDB_Q_UNDERFLOW from the parts flavour, answered by the re-run in hdb_topk_host)
            for qi in (2, 3, 4):
                orc.check_topk(i2[qi], s2[qi], V, Q2[qi], "dot_product", k, tol=1e-5)
        ix.set_option("f32_split", 0)
        i3, s3, _ = ix.topk_views(Q2, k, mid)
        assert np.array_equal(i3[3], i2[3]) and np.array_equal(s3[3], s2[3])
    finally:
        ix.close()


def test_topk_host_pinned_and_pageable_records(orc):
    """hdb_topk_host stores into pinned records from the kernels themselves and copies for pageable ones; same bytes."""
    import ctypes, torch
    from hyperdb import _native
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(3)
    V = rng.standard_normal((50_000, 128)).astype(np.float32)
    Q = rng.standard_normal((4, 128)).astype(np.float32)
    ix = GpuIndex(V)
    try:
        k, mid = 20, METRIC_IDS["cosine_similarity"]
        i1, s1, st1 = ix.topk_views(Q, k, mid)
        assert ix.stat("host_direct") == 1
        i1, s1 = i1.copy(), s1.copy()
        nb = _native.packed_bytes(4, k)
        page = np.zeros(nb, dtype=np.uint8)                       # ordinary pageable memory
        qt = torch.from_numpy(Q).cuda()
        rc = _native._lib.hdb_topk_host(ix._h, ctypes.c_void_p(qt.data_ptr()), 4, k, mid, ctypes.c_void_p(page.ctypes.data),
                                        _native._stream_ptr(ix.device))
        assert rc == 0 and ix.stat("host_direct") == 0
        i2 = page[:4 * k * 8].view(np.int64).reshape(4, k)
        s2 = page[4 * k * 8:4 * k * 12].view(np.float32).reshape(4, k)
        assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
        for qi in range(4):
            orc.check_topk(i1[qi], s1[qi], V, Q[qi], "cosine_similarity", k, tol=1e-5)
    finally:
        ix.close()


@pytest.mark.parametrize("d", [384, 1536])
@pytest.mark.parametrize("n", [8193, 8256, 8257, 12345])
def test_mfma_single_query_small_matrices(orc, d, n):
    """fp16 matrices just above the small-matrix limit go through the MFMA scan even for one query: ragged last
    tiles (16- and 64-row geometries), tile counts below the grid size, sample tiles = all tiles."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(n + d)
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((3, d)).astype(np.float16)
    ix = GpuIndex(V)
    try:
        for metric in ("cosine_similarity", "euclidean_metric"):
            mid = METRIC_IDS[metric]
            for sl, k in ((slice(0, 1), 10), (slice(0, 3), 100)):
                idx, sc = ix.topk(Q[sl], k, mid)
                assert ix.stat("mfma") == 1
                for qi in range(idx.shape[0]):
                    orc.check_topk(idx[qi], sc[qi], V, Q[sl][qi], metric, k, tol=1e-3)
        idx, sc = ix.topk(V[n - 1:n], 5, METRIC_IDS["cosine_similarity"])     # the very last (ragged-tile) row finds itself
        assert idx[0][0] == n - 1 and abs(sc[0][0] - 1.0) < 1e-3
    finally:
        ix.close()


@pytest.mark.parametrize("target", [16, 200_000])
def test_forced_threshold_failure_falls_back_to_exact(orc, target):
    """sample_target=16 makes the threshold far too high (fewer than k survivors -> UNDERFLOW),
    200000 far too low (more than 8192 survivors -> OVERFLOW); both must end in the exact answer."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(target)
    n, d, k = 300_000, 64, 100
    V = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((3, d)).astype(np.float32)
    ix = GpuIndex(V)
    try:
        ix.set_option("sample_target", target)
        mid = METRIC_IDS["cosine_similarity"]
        _, _, st = ix.topk_device(Q, k, mid)
        assert int((st != 0).sum().item()) == 3, "the test did not provoke the failure it is about"
        assert set(st.cpu().tolist()) == ({1} if target == 16 else {2})
        idx, sc = ix.topk(Q, k, mid)                       # host entry: re-runs the failed queries exactly
        for qi in range(3):
            oi, osc = orc.rank(V, Q[qi], top_k=k, metric="cosine_similarity")
            assert list(idx[qi]) == list(oi) and np.allclose(sc[qi], osc, atol=1e-5)
    finally:
        ix.close()


def test_index_update_lifecycle(orc):
    """hdb_index_update: grow / shrink the resident matrix; caches (norms, NaN flag, sign bits) follow."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(9)
    V = rng.standard_normal((20_000, 48)).astype(np.float32)
    q = rng.standard_normal(48).astype(np.float32)
    ix = GpuIndex(V[:9000])
    try:
        for metric in ("cosine_similarity", "hamming_distance", "pearson_correlation"):
            i1, s1 = ix.topk(q.reshape(1, -1), 10, METRIC_IDS[metric])
            o1, os1 = orc.rank(V[:9000], q.copy(), top_k=10, metric=metric)
            assert np.allclose(np.sort(s1[0]), np.sort(os1), atol=1e-5)
        ix.update(V)                                       # grown to 20000 rows
        assert ix.n == 20_000 and not ix.has_nan
        for metric in ("cosine_similarity", "hamming_distance", "pearson_correlation"):
            i2, s2 = ix.topk(q.reshape(1, -1), 10, METRIC_IDS[metric])
            o2, os2 = orc.rank(V, q.copy(), top_k=10, metric=metric)
            assert np.allclose(np.sort(s2[0]), np.sort(os2), atol=1e-5)
            if metric == "cosine_similarity":
                assert list(i2[0]) == list(o2)
        Vn = V[:100].copy(); Vn[37, 5] = np.nan
        ix.update(Vn)
        assert ix.n == 100 and ix.has_nan
    finally:
        ix.close()


def test_fp64_and_query_dtype_promotion(ranking, orc):
    rng = np.random.default_rng(64)
    V = rng.standard_normal((12_000, 40))                  # float64 matrix, float64 accumulate on the device
    q = rng.standard_normal(40)
    for metric in ("dot_product", "euclidean_metric", "manhattan_distance"):
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(V, q, top_k=25, metric=metric)
        oi, osc = orc.rank(V, q, top_k=25, metric=metric)
        assert list(idx) == list(oi) and np.allclose(sc, osc, rtol=1e-6, atol=1e-6)
    d = ranking.euclidean_metric(V[:50], q, get_similarity_score=False)
    assert np.allclose(d, orc.score_euclidean(V[:50], q, get_similarity_score=False), rtol=1e-6)
    # integer matrix / list inputs are accepted like numpy would (promoted to float64)
    Vi = rng.integers(-3, 4, size=(200, 6))
    idx, sc = ranking.hyperDB_ranking_algorithm_sort(Vi.tolist(), [1, 0, 2, 0, 0, 1], top_k=5, metric="dot_product")
    oi, osc = orc.rank(Vi, np.array([1, 0, 2, 0, 0, 1]), top_k=5, metric="dot_product")
    assert np.allclose(sc, osc) and set(sc) == set(osc)


def test_full_size_q256_mfma_equals_single_query_scan(big_fp16, orc):
    """Config 3 at full size: the 256-query MFMA pass against the single-query row scan, query by query."""
    import torch
    from hyperdb._native import METRIC_IDS
    import bench
    ix, V, _ = big_fp16
    Q = bench.make_queries(256, 384, torch.float16, V.device)
    mid = METRIC_IDS["dot_product"]
    bi, bs, st = ix.topk_device(Q, 100, mid)
    assert ix.stat("mfma") == 1 and ix.stat("fused") == 2 and int(st.abs().sum().item()) == 0      # ONE launch for the batch
    ix.set_option("use_fused", 0)                      # the five-kernel pipeline: same rows, same scores, bit for bit
    ui, us, ust = ix.topk_device(Q, 100, mid)
    ix.set_option("use_fused", 1)
    assert ix.stat("fused") == 0 and int(ust.abs().sum().item()) == 0 and torch.equal(bi, ui) and torch.equal(bs, us)
    ix.set_option("use_mfma", 0)
    try:
        for qi in (0, 100, 255):
            si, ss, _ = ix.topk_device(Q[qi:qi + 1], 100, mid)
            assert ix.stat("mfma") == 0
            assert torch.equal(si[0], bi[qi]), qi
            assert torch.allclose(ss[0], bs[qi], rtol=2e-6, atol=2e-5)
    finally:
        ix.set_option("use_mfma", 1)
    # every one of the 256 x 100 returned rows re-scored in float64 from the stored row (as config 5 does) ...
    Qh = Q.float().cpu().numpy().astype(np.float64)
    bi_h, bs_h = bi.cpu().numpy(), bs.cpu().numpy().astype(np.float64)
    for q0 in range(0, 256, 32):
        rows = V[bi[q0:q0 + 32].reshape(-1)].float().cpu().numpy().astype(np.float64).reshape(32, 100, 384)
        ex = np.einsum("qkd,qd->qk", rows, Qh[q0:q0 + 32])
        got = bs_h[q0:q0 + 32]
        assert np.all(np.abs(ex - got) <= 1e-3 * np.maximum(1.0, np.abs(ex))), q0
        assert np.all(np.diff(got, axis=1) <= 0)
    assert all(np.unique(bi_h[q]).size == 100 for q in range(256))
    # ... and an independent omission check over the whole row range for the first, a middle and the last query
    for qi in (0, 127, 255):
        _assert_nothing_left_out(orc, V, Q[qi].float().cpu().numpy(), "dot_product", bi_h[qi], bs_h[qi], 1e-3)


def test_sample_does_not_alias_with_periodic_data(orc):
    """Rows that sit exactly on the sampling period carry 5x larger vectors.  An un-jittered strided sample would
    see only those rows, set a threshold ~5x too high and underflow; the jittered sample must not."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(12)
    n, d, k = 1_000_000, 16, 100
    V = rng.standard_normal((n, d)).astype(np.float32)
    ix = GpuIndex(V)
    try:
        mid = METRIC_IDS["dot_product"]
        q = rng.standard_normal((1, d)).astype(np.float32)
        ix.topk_device(q, k, mid)
        s_rows = ix.stat("sample_rows")
        stride = (n // 16) // (s_rows // 16)
        assert stride > 8
        tiles = np.arange(n) // 16
        V[tiles % stride == 0] *= 5.0
        ix.update(V)
        idx, sc, st = ix.topk_device(q, k, mid)
        assert int(st[0].item()) == 0 and ix.stat("path") == 1
        oi, osc = orc.rank(V, q[0], top_k=k, metric="dot_product")
        assert list(idx[0].cpu().numpy()) == list(oi)
    finally:
        ix.close()


def test_growable_index_append_matches_full_rebuild(orc):
    """hdb_index_rebase / hdb_index_extend: appended rows get their caches incrementally; answers must equal a
    fresh registration of the concatenated matrix, for every cached quantity (norms, sign bits, pearson scales)."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(31)
    V = rng.standard_normal((30_000, 64)).astype(np.float32).astype(np.float16)
    q = rng.standard_normal((3, 64)).astype(np.float16)
    ix = GpuIndex(V[:500])
    try:
        ix.topk(q, 5, METRIC_IDS["hamming_distance"])            # builds the sign-bit cache, must be invalidated later
        for lo, hi in ((500, 501), (501, 9_000), (9_000, 30_000)):      # grows past the capacity several times
            ix.append(V[lo:hi])
            assert ix.n == hi
        fresh = GpuIndex(V)
        for metric in ("cosine_similarity", "euclidean_metric", "hamming_distance", "pearson_correlation", "dot_product"):
            i1, s1 = ix.topk(q, 50, METRIC_IDS[metric])
            i2, s2 = fresh.topk(q, 50, METRIC_IDS[metric])
            assert np.array_equal(i1, i2) and np.array_equal(s1, s2), metric
        fresh.close()
        bad = V[:3].copy(); bad[1, 7] = np.nan
        ix.append(bad)
        assert ix.has_nan
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 9. BASELINE configs 4 and 5 at full size, and a GPU-vs-oracle fuzz that is not a self-comparison
# ------------------------------------------------------------------------------------------------
def test_config4_100m_rows_eight_shards_merge_equals_single_index(orc):
    """Config 4 (N=100M d=384 fp16, 8 row shards, all-gather + merge) on ONE GPU (76.8 GB fits in 288 GB): the eight
    12.5M-row shards are scanned with their row_base, their packed records are laid out as the all-gather would and
    merged by hdb_merge_topk_packed; the result must equal the single 100M-row index bit for bit, every returned row
    re-scores in float64, and windows up to the LAST row (global ids >= 87.5M, byte offsets up to 76.8 GB) hold no
    better row.  Data = bench.make_shard (block-seeded standard normal), queries = bench.make_queries."""
    import torch
    import bench
    from hyperdb._native import GpuIndex, METRIC_IDS, merge_topk_packed, packed_bytes
    n, d, k, parts, nq = 100_000_000, 384, 100, 8, 4
    dev = torch.device("cuda", 0)
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    assert (lo, hi) == (0, n)
    Q = bench.make_queries(nq, d, torch.float16, dev).float()
    whole = GpuIndex(V)
    shards = []
    try:
        per = n // parts
        shards = [GpuIndex(V[p * per:(p + 1) * per], row_base=p * per) for p in range(parts)]
        for metric in ("cosine_similarity", "dot_product"):
            mid = METRIC_IDS[metric]
            gi, gs, gst = whole.topk_device(Q, k, mid)
            assert whole.stat("path") == 1 and whole.stat("mfma") == 1 and int(gst.abs().sum().item()) == 0
            nb = packed_bytes(nq, k)
            rec = torch.zeros((parts, nb), dtype=torch.uint8, device=dev)
            for p, sh in enumerate(shards):
                sh.topk_packed(Q, k, mid, rec[p])
            mi, ms, mst = merge_topk_packed(rec, parts, nq, k)
            assert int(mst.abs().sum().item()) == 0
            assert torch.equal(mi, gi) and torch.equal(ms, gs), metric
            gi_h, gs_h = gi.cpu().numpy(), gs.cpu().numpy()
            assert gi_h.max() < n and gi_h.min() >= 0
            for qi in range(nq):
                rows = V[gi[qi]].cpu().numpy()
                ex = orc.exact_scores(rows, Q[qi].cpu().numpy(), metric)
                assert np.all(np.abs(ex - gs_h[qi]) <= 1e-3 * np.maximum(1, np.abs(ex))), (metric, qi)
                assert np.all(np.diff(gs_h[qi]) <= 0) and np.unique(gi_h[qi]).size == k
            assert (gi_h >= n - per).any(), "no hit from the last shard in 4 x 100 results?"
            for qi in (0, nq - 1):
                _assert_nothing_left_out(orc, V, Q[qi].cpu().numpy(), metric, gi_h[qi], gs_h[qi], 1e-3, count=12)
        # the last shard alone, VALU scan (an independent kernel) == its MFMA scan
        last = shards[-1]
        a_i, a_s, _ = last.topk_device(Q[:2], k, METRIC_IDS["dot_product"])
        last.set_option("use_mfma", 0)
        b_i, b_s, _ = last.topk_device(Q[:2], k, METRIC_IDS["dot_product"])
        assert last.stat("mfma") == 0 and torch.equal(a_i, b_i) and torch.allclose(a_s, b_s, rtol=2e-6, atol=2e-5)
        assert int(a_i.min().item()) >= n - per
    finally:
        for sh in shards + [whole]:
            sh.close()
        del V
        torch.cuda.empty_cache()


@pytest.mark.parametrize("n,d,nq,metric", [(6_000_001, 384, 17, "cosine_similarity"), (3_000_017, 768, 40, "dot_product"),
                                            (1_500_001, 1536, 33, "euclidean_metric")])
def test_batched_filter_pass_tiles_from_a_counter(n, d, nq, metric):
    """Long filter passes over rows of >= 768 bytes with up to 64 queries take their tiles from a global counter
    (hdb_mfma_kernel.h, "tile sequence") instead of the static split: same candidates, so the same result bit for bit,
    with and without a bias, ragged last tile included; and both equal the on-device exact selection."""
    import torch
    import bench
    from hyperdb._native import GpuIndex, METRIC_IDS
    dev = torch.device("cuda", 0)
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    V[n - 1] = V[5]                                         # the ragged last tile holds a row that ties with an early one
    Q = bench.make_queries(nq, d, torch.float16, dev).float()
    Q[1] = V[5].float()
    g = torch.Generator(device=dev).manual_seed(n)
    ix = GpuIndex(V)
    try:
        mid = METRIC_IDS[metric]
        for with_bias in (False, True):
            ix.set_bias((torch.rand(n, generator=g, device=dev) * 0.05).float() if with_bias else None)
            ix.set_option("dyn_tiles", 1)
            di, ds, dst = ix.topk_device(Q, 100, mid)
            assert ix.stat("mfma") == 1 and ix.stat("path") == 1 and int(dst.abs().sum().item()) == 0
            ix.set_option("dyn_tiles", 0)
            si, ss, sst = ix.topk_device(Q, 100, mid)
            ix.set_option("dyn_tiles", 1)
            assert int(sst.abs().sum().item()) == 0 and torch.equal(di, si) and torch.equal(ds, ss), with_bias
            ei, es, _ = ix.topk_device(Q[:6], 100, mid, exact=True)
            if metric == "euclidean_metric":                # near-duplicates are re-scored directly on the filter path only
                sa, sb = torch.sort(ds[:6], dim=-1, descending=True)[0], torch.sort(es, dim=-1, descending=True)[0]
                assert bool(((sa - sb).abs() <= 1e-6 * sa.abs().clamp(min=1e-3)).all())
            else:
                assert torch.equal(di[:6], ei) and torch.equal(ds[:6], es), with_bias
    finally:
        ix.close()
        del V
        torch.cuda.empty_cache()


def test_config5_full_size_euclidean_with_time_decay(orc):
    """Config 5 at full size: N=10M d=768 fp16, 64 queries, euclidean similarity + recency term (timestamps uniform
    over 30 days, recency_bias 0.5).  MFMA pass vs float64 scores of the returned rows (all 64 queries), vs the VALU
    scan (independent kernel) and vs the on-device exact selection; omission windows over the whole row range."""
    import torch
    import bench
    from hyperdb._native import GpuIndex, METRIC_IDS
    n, d, k, nq = 10_000_000, 768, 100, 64
    dev = torch.device("cuda", 0)
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    Q = bench.make_queries(nq, d, torch.float16, dev).float()
    g = torch.Generator(device=dev).manual_seed(99)
    ts = 1.7e9 + torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 30 * 86400.0
    ix = GpuIndex(V)
    try:
        mid = METRIC_IDS["euclidean_metric"]
        for rb in (0.5, 0.02):      # 0.5: the newest rows dominate; 0.02: distance and recency are of the same size
            ix.set_recency(ts, rb)
            bias = (rb * torch.exp(ts - ts.max())).float()
            mi, ms, st = ix.topk_device(Q, k, mid)
            assert ix.stat("mfma") == 1 and ix.stat("path") == 1 and int(st.abs().sum().item()) == 0
            mi_h, ms_h = mi.cpu().numpy(), ms.cpu().numpy()
            for qi in range(nq):
                rows = V[mi[qi]].cpu().numpy()
                ex = orc.exact_scores(rows, Q[qi].cpu().numpy(), "euclidean_metric", bias=bias[mi[qi]].cpu().numpy().astype(np.float64))
                assert np.all(np.abs(ex - ms_h[qi]) <= 1e-3), (rb, qi, float(np.abs(ex - ms_h[qi]).max()))
                assert np.all(np.diff(ms_h[qi]) <= 0) and np.unique(mi_h[qi]).size == k
            ei, es, _ = ix.topk_device(Q[:8], k, mid, exact=True)
            sa, sb = torch.sort(ms[:8], dim=-1, descending=True)[0], torch.sort(es, dim=-1, descending=True)[0]
            assert bool(((sa - sb).abs() <= 1e-6 * sa.abs().clamp(min=1e-3)).all()), "fused vs exact selection"
            ix.set_option("use_mfma", 0)
            vi, vs, _ = ix.topk_device(Q[:3], k, mid)
            ix.set_option("use_mfma", 1)
            vi_h, vs_h = vi.cpu().numpy(), vs.cpu().numpy()
            for qi in range(3):
                assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi], vi_h[qi], vs_h[qi], 2e-5), (rb, qi)
            for qi in (0, nq - 1):
                _assert_nothing_left_out(orc, V, Q[qi].cpu().numpy(), "euclidean_metric", mi_h[qi], ms_h[qi], 1e-3,
                                         bias_dev=bias, count=8, rows=100_000)
    finally:
        ix.close()
        del V
        torch.cuda.empty_cache()


def test_fuzz_gpu_against_oracle(orc):
    """Seeded slice of tests/fuzz_oracle.py inside the suite: random shapes, GPU result vs the oracle's float64
    comparator (not a self-comparison) -- seven metrics x three dtypes x {bias, no bias}, MFMA and odd dimensions,
    single queries and batches, duplicate rows, a query equal to a stored row."""
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(20261004)
    combos = [(m, dt) for m in GPU_METRICS for dt in (np.float16, np.float32, np.float64)]
    cases = combos + combos + [combos[i] for i in rng.permutation(len(combos))[:21]]          # 63 cases
    seen = set()
    for case, (metric, dt) in enumerate(cases):
        d = int(rng.choice([128, 256, 384, 512, 768, 1024, 24, 100, 33, 200]))
        n = int(rng.integers(8200, 40_000)) if rng.random() < 0.8 else int(rng.integers(2, 8192))
        nq = int(rng.choice([1, 1, 2, 5, 9, 33]))
        k = int(rng.choice([1, 5, 40, 100]))
        with_bias = case >= len(combos) and case < 2 * len(combos)
        V = rng.standard_normal((n, d)).astype(np.float32).astype(dt)
        if rng.random() < 0.2:
            V[rng.integers(0, n, size=20)] = V[0]
        Q = rng.standard_normal((nq, d)).astype(np.float32).astype(dt)
        if rng.random() < 0.3:
            Q[0] = V[n // 2]
        ix = GpuIndex(V)
        try:
            bias = None
            if with_bias:
                ts = 1.7e9 + rng.uniform(0, 86400.0, size=n)
                ix.set_recency(ts, 0.4)
                bias = 0.4 * np.exp(ts - ts.max())
            idx, sc = ix.topk(Q, min(k, n), METRIC_IDS[metric])
            tol = _tol(dt, metric)
            if metric == "hamming_distance" and bias is not None:
                tol = 1e-5
            for qi in range(nq):
                orc.check_topk(idx[qi], sc[qi], V, Q[qi].copy(), metric, k, bias=bias, tol=tol)
            seen.add((metric, np.dtype(dt).name, with_bias, bool(ix.stat("mfma"))))
        except AssertionError as e:
            raise AssertionError(f"case {case}: n={n} d={d} {np.dtype(dt).name} nq={nq} k={k} {metric} bias={with_bias} "
                                 f"mfma={ix.stat('mfma')} path={ix.stat('path')}: {e}") from e
        finally:
            ix.close()
    assert len({(m, t) for m, t, _, _ in seen}) == 21 and any(b for _, _, b, _ in seen) and any(f for _, _, _, f in seen)


# ------------------------------------------------------------------------------------------------
# 10. the single-launch pipeline (hdb_mfma_fused.h): 1-4 dot / cosine queries on fp16 matrices
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d", [(8193, 384), (8200 + 63, 256), (50_000, 768), (400_000, 256), (1_300_001, 384),
                                 (60_001, 1024), (300_000, 1536), (100_003, 1408), (150_000, 1152)])
def test_single_launch_pipeline_equals_multi_kernel_and_oracle(orc, n, d):
    """Every call shape the fused kernel takes (1-4 queries, dot / cosine / pearson, one euclidean query, k <= 128, bias, row mask, ragged last tile,
    grids smaller than the CU count, static and counter-fed tile chunks) returns exactly what the five-kernel pipeline
    and the on-device exact selection return; one case per shape is checked against the oracle's float64 scores."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(n + d)
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    V[n - 1] = V[7]                                        # duplicate rows across the matrix ends: tie -> lower row first
    Q = rng.standard_normal((4, d)).astype(np.float16).astype(np.float32)
    Q[1] = V[n // 3].astype(np.float32)
    Q[2] = rng.standard_normal(d).astype(np.float32) * 37.5          # a genuinely float32 query: norm sums must round alike in both pipelines
    ix = GpuIndex(V); ix.set_option("fused_max_q", 4)      # (by rule 2-4 fp16 queries reach this kernel from 1.5M rows on: forced here)
    try:
        bias = torch.rand(n, generator=torch.Generator().manual_seed(3)).float().cuda() * 0.2
        mask = (torch.rand(n, generator=torch.Generator().manual_seed(4)) < 0.3).to(torch.uint8).cuda()
        for metric in ("cosine_similarity", "dot_product", "euclidean_metric", "pearson_correlation"):
            mid = METRIC_IDS[metric]
            for setup in ("plain", "bias", "mask", "mask+bias"):
                ix.set_bias(bias if "bias" in setup else None)
                ix.set_row_mask(mask if "mask" in setup else None)
                for nq, k in ((1, 100), (2, 1), (3, 128), (4, 37)):
                    ix.set_option("use_fused", 1)
                    fi, fs, fst = ix.topk_device(Q[:nq], k, mid)
                    single = nq <= (4 if d <= 768 else 2)      # beyond d = 768 the query fragments live in LDS: two queries
                    if metric == "euclidean_metric" and (d == 768 or nq > 1):
                        single = False                         # (d = 768: three registers short; 2-4 queries: the batched launch is faster)
                    # (what this kernel does not take goes to the batched single launch, stat 2: hdb_mfma_kernel.h MODE 2)
                    other = 2                                               # (what this kernel does not take: the batched single launch)
                    assert ix.stat("fused") == (1 if single else other) and ix.stat("path") == 1 and int(fst.abs().sum().item()) == 0, (metric, setup, nq, k)
                    ix.set_option("use_fused", 0)
                    ui, us, ust = ix.topk_device(Q[:nq], k, mid)
                    assert ix.stat("fused") == 0
                    assert torch.equal(fi, ui) and torch.equal(fs, us), (metric, setup, nq, k)
                    if metric != "euclidean_metric":           # (euclidean: the exact selection ranks before the near-duplicate re-score)
                        ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                        assert torch.equal(fi, ei) and torch.equal(fs, es), (metric, setup, nq, k)
            ix.set_bias(None); ix.set_row_mask(None); ix.set_option("use_fused", 1)
            idx, sc = ix.topk(Q[:2], 100, mid)
            assert ix.stat("fused") == (2 if metric == "euclidean_metric" else 1)
            for qi in range(2 if (n < 1_000_000 or metric in ("cosine_similarity", "euclidean_metric")) else 0):      # (see test_single_launch_pipeline_float32)
                orc.check_topk(idx[qi], sc[qi], V, Q[qi], metric, 100, tol=1e-3)
            if metric == "euclidean_metric":
                assert idx[1][0] == n // 3 and abs(sc[1][0] - 1.0) < 1e-6, "Q[1] is a stored row: it must score exactly 1"
        # k > 128: the multi-kernel pipeline; five queries, euclidean: the batched single launch (stat 2)
        ix.topk_device(Q[:1], 200, METRIC_IDS["dot_product"]); assert ix.stat("fused") == 0
        ix.topk_device(np.concatenate([Q, Q[:1]]), 10, METRIC_IDS["dot_product"]); assert ix.stat("fused") == 2
        ix.topk_device(Q[:1], 10, METRIC_IDS["euclidean_metric"]); assert ix.stat("fused") == (2 if d == 768 else 1)
    finally:
        ix.close()


def test_single_launch_pipeline_failure_paths(ranking, orc):
    """What must come back through the exact selection: a NaN query (status bit), massive ties that overflow the
    candidate list, and an exchange that gives up (spin timeout forced to 10 ns) -- the caller still gets the right rows."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS, Q_NAN
    rng = np.random.default_rng(99)
    base = rng.standard_normal((40, 384)).astype(np.float32).astype(np.float16)
    V = np.tile(base, (1500, 1))                           # 60k rows, every score repeated 1500 times
    q = rng.standard_normal(384).astype(np.float16).astype(np.float32)
    ix = GpuIndex(V); ix.set_option("fused_max_q", 4)      # (by rule 2-4 fp16 queries reach this kernel from 1.5M rows on: forced here)
    try:
        mid = METRIC_IDS["dot_product"]
        idx, sc = ix.topk(q.reshape(1, -1), 100, mid)      # overflow -> hdb_topk_host re-runs the exact path
        ex = orc.exact_scores(V, q, "dot_product")
        want = np.nonzero(np.isclose(ex, ex.max(), rtol=1e-6))[0][:100]
        assert np.array_equal(idx[0], want) and np.allclose(sc[0], ex.max(), rtol=1e-3)
        qn = q.copy(); qn[5] = np.nan
        _, _, st = ix.topk_device(np.stack([q, qn]), 10, mid)
        assert ix.stat("fused") == 1 and (int(st[1].item()) & Q_NAN) and not (int(st[0].item()) & Q_NAN)
    finally:
        ix.close()
    # 700k rows = 43 tiles per workgroup: more than the parking area holds (a selector wave that serves two queries
    # gave up on the first only, the second kept its NaN threshold and wave 0 parked past the end of its buffer)
    V2 = torch.randn((700_000, 384), generator=torch.Generator(device="cuda").manual_seed(3), device="cuda").to(torch.float16)
    ix = GpuIndex(V2); ix.set_option("fused_max_q", 4)      # (by rule 2-4 fp16 queries reach this kernel from 1.5M rows on: forced here)
    try:
        mid = METRIC_IDS["cosine_similarity"]
        for nq in (1, 2, 3, 4):
            Qn = torch.randn((nq, 384), generator=torch.Generator(device="cuda").manual_seed(nq), device="cuda").float()
            ei, es, _ = ix.topk_device(Qn, 50, mid, exact=True)
            ix.set_option("fused_timeout_us", 1)           # (the option floor; the kernel compares 100 MHz ticks)
            for _ in range(3):
                fi, fs, st = ix.topk_device(Qn, 50, mid)   # device status: a give-up shows as UNDERFLOW, never as a stray NaN bit
                assert ix.stat("fused") == 1
                for qi in range(nq):
                    sq = int(st[qi].item())
                    assert not (sq & Q_NAN), (nq, qi, sq)
                    if sq == 0:                            # (on an idle GPU the first sweep may well succeed in time)
                        assert torch.equal(fi[qi], ei[qi]) and torch.equal(fs[qi], es[qi]), (nq, qi)
            idx, sc = ix.topk(Qn, 50, mid)                 # host entry: falls back when the kernel gave up
            assert np.array_equal(idx, ei.cpu().numpy()) and np.array_equal(sc, es.cpu().numpy()), nq
            ix.set_option("fused_timeout_us", 2000)
            fi, fs, st = ix.topk_device(Qn, 50, mid)       # and the next call is clean again
            assert int(st.abs().sum().item()) == 0 and torch.equal(fi, ei) and torch.equal(fs, es), nq
    finally:
        ix.close()


@pytest.mark.parametrize("dt,n,d,nq", [(np.float16, 50_282, 512, 4), (np.float16, 19_201, 256, 1), (np.float32, 43_279, 256, 2),
                                        (np.float16, 700_001, 384, 3)])
def test_single_launch_pipeline_threshold_minus_inf(dt, n, d, nq):
    """A row mask that leaves 5 % of the rows (or a bias of -inf on the rest) leaves fewer than 8 finite scores in the row
    sample: the sampled threshold is -inf, every unmasked row is a candidate, and every workgroup -- the one that computes
    the threshold and the ones that are told -- must filter with exactly that (a -inf that came back as NaN from the
    exchange once parked every tile of the other workgroups for good)."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(n + d)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    Q = torch.randn((nq, d), generator=g, device="cuda").to(V.dtype).float()
    ix = GpuIndex(V); ix.set_option("fused_max_q", 4)      # (by rule 2-4 fp16 queries reach this kernel from 1.5M rows on: forced here)
    try:
        for keep in (0.05, 0.004):
            ix.set_row_mask((torch.rand(n, generator=g, device="cuda") < keep).to(torch.uint8))
            for with_bias in (False, True):
                ix.set_bias((torch.rand(n, generator=g, device="cuda") * 0.3).float() if with_bias else None)
                for metric in ("cosine_similarity", "dot_product"):
                    for k in (1, 5, 100):
                        mid = METRIC_IDS[metric]
                        fi, fs, st = ix.topk_device(Q, k, mid)
                        assert ix.stat("fused") == 1
                        ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
                        for q in range(nq):
                            if int(st[q].item()) == 0:
                                assert torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q]), (keep, with_bias, metric, k, q)
                        hi, hs = ix.topk(Q, k, mid)
                        assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy()), (keep, with_bias, metric, k)
    finally:
        ix.close()


def test_fuzz_single_launch_against_exact():
    """tools/fuzz_fused.py inside the suite: 1 500 random shapes the single-launch pipeline takes (1-4 queries fp16, 1-2
    float32, k <= 128, bias, masks, duplicate rows, clusters around a query, 8 200 .. 2.5 M rows) -- a call that reports
    status 0 must equal the on-device exact selection bit for bit (the run that found the two exchange bugs of round 2)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_fused.py"), "1500", "21", "240"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0 and "0 mismatches" in tail, (r.stdout[-1500:], r.stderr[-500:])


@pytest.mark.parametrize("dt,n,d", [(np.float32, 549_592, 128), (np.float16, 700_000, 384)])
def test_single_launch_pipeline_cluster_in_a_sample_tile(dt, n, d):
    """300 near-copies of one row, queried with that row: when the cluster falls into a tile of the row sample the
    threshold lands inside the cluster and fewer than k rows pass it.  Workgroups that filtered some tiles with an
    earlier, lower threshold collect extra rows, so the candidate list can still hold k entries -- the call must report
    UNDERFLOW from the rows above the LAST threshold, not from the length of the list (found by tools/fuzz_fused.py:
    72 rows above, 100+ collected, status 0, ranks 72.. wrong)."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(n)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    noise = torch.randn((300, d), generator=g, device="cuda")
    mid = METRIC_IDS["cosine_similarity"]
    flagged = 0
    ix = GpuIndex(V); ix.set_option("fused_max_q", 4)      # (by rule 2-4 fp16 queries reach this kernel from 1.5M rows on: forced here)
    try:
        for trial in range(150):
            c0 = (trial * 9001 + 17) % (n - 300)
            saved = V[c0:c0 + 300].clone()
            V[c0:c0 + 300] = (saved[0:1].float() + 0.05 * noise).to(V.dtype)
            ix.update(V)
            q = V[c0].float().reshape(1, -1)
            fi, fs, st = ix.topk_device(q, 100, mid)
            assert ix.stat("fused") == 1
            ei, es, _ = ix.topk_device(q, 100, mid, exact=True)
            if int(st[0].item()) == 0:
                assert torch.equal(fi, ei) and torch.equal(fs, es), (trial, c0)
            else:
                flagged += 1
                hi, hs = ix.topk(q, 100, mid)
                assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy()), (trial, c0)
            V[c0:c0 + 300] = saved
    finally:
        ix.close()
    assert flagged >= 1, "no trial put the cluster into a sample tile: move the clusters"


@pytest.mark.parametrize("dt", [np.float16, np.float32])
def test_single_launch_pipeline_winners_in_parked_tiles(orc, dt):
    """The first filter tiles of every workgroup are scored before any threshold exists and parked in LDS
    (hdb_mfma_fused.h, "parking").  (a) every winner sits in such a tile: the 128 best rows are scattered over the first
    65 536 rows (the first chunk of each workgroup), the rest of the matrix is in random order, so the row sample stays
    representative and the call must succeed in the single launch; (b) rows sorted by descending score: the strided tile
    sample then sees the very best rows, the threshold lands above the k-th best, the call reports UNDERFLOW and the host
    entry re-runs it exactly -- either way the caller gets the exact top-k."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(2024)
    n, d = (600_000, 384) if dt == np.float16 else (300_000, 384)
    V = rng.standard_normal((n, d)).astype(np.float32).astype(dt)
    q = rng.standard_normal(d).astype(np.float32).astype(dt).astype(np.float32)
    order = np.argsort(-(V.astype(np.float32) @ q), kind="stable")
    head = np.sort(rng.choice(65_536, size=128, replace=False))
    rest = np.setdiff1d(np.arange(n), head, assume_unique=True)
    scattered = np.empty_like(V)
    scattered[head] = V[order[:128]]
    scattered[rest] = V[order[128:][rng.permutation(n - 128)]]
    clean = 0
    for M in (scattered, np.ascontiguousarray(V[order])):
        ix = GpuIndex(M); ix.set_option("fused_max_q", 4)      # (by rule 2-4 fp16 queries reach this kernel from 1.5M rows on: forced here)
        try:
            for metric in ("dot_product", "cosine_similarity"):
                mid = METRIC_IDS[metric]
                for k in (1, 100, 128):
                    fi, fs, st = ix.topk_device(q.reshape(1, -1), k, mid)
                    assert ix.stat("fused") == 1
                    ei, es, _ = ix.topk_device(q.reshape(1, -1), k, mid, exact=True)
                    if int(st.abs().sum().item()) == 0:
                        assert torch.equal(fi, ei) and torch.equal(fs, es), (metric, k)
                        clean += M is scattered
                    hi, hs = ix.topk(q.reshape(1, -1), k, mid)          # host entry: re-runs what the launch could not settle
                    assert np.array_equal(hi[0], ei[0].cpu().numpy()) and np.array_equal(hs[0], es[0].cpu().numpy()), (metric, k)
            idx, sc = ix.topk(q.reshape(1, -1), 100, METRIC_IDS["dot_product"])
            orc.check_topk(idx[0], sc[0], M, q, "dot_product", 100, tol=1e-3 if dt == np.float16 else 1e-5)
        finally:
            ix.close()
    assert clean == 6, "the scattered layout must be settled by the single launch itself"


@pytest.mark.parametrize("n,d", [(8193, 384), (70_001, 128), (250_000, 256), (400_003, 384), (1_600_001, 512), (90_003, 768), (300_000, 768)])
def test_single_launch_pipeline_float32(orc, n, d):
    """float32 matrices (the reference's default fp_precision, BASELINE config 2): 1-2 dot / cosine / euclidean queries run as one
    launch whose float32 sums are computed in the VALU from the staged tiles -- bit-identical to the five-kernel VALU
    pipeline and the exact selection (same accumulation order, same rounding steps), and within 1e-5 of the oracle."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(n * 3 + d)
    V = rng.standard_normal((n, d)).astype(np.float32)
    V[n - 1] = V[11]
    Q = rng.standard_normal((3, d)).astype(np.float32)
    Q[1] = V[n // 2] + 0.01 * rng.standard_normal(d).astype(np.float32)
    ix = GpuIndex(V)
    try:
        bias = torch.rand(n, generator=torch.Generator().manual_seed(5)).float().cuda() * 0.2
        mask = (torch.rand(n, generator=torch.Generator().manual_seed(6)) < 0.4).to(torch.uint8).cuda()
        for metric in ("cosine_similarity", "dot_product", "euclidean_metric", "pearson_correlation"):      # euclidean: the direct sum of (v - q)^2, as hdb_scan.hip
            mid = METRIC_IDS[metric]
            for setup in ("plain", "bias", "mask+bias"):
                ix.set_bias(bias if "bias" in setup else None)
                ix.set_row_mask(mask if "mask" in setup else None)
                for nq, k in ((1, 100), (2, 7), (1, 128)):
                    ix.set_option("use_fused", 1)
                    fi, fs, fst = ix.topk_device(Q[:nq], k, mid)
                    single = nq == 1 or d <= 384          # two float32 queries fit the registers up to d = 384 (16-row tiles beyond: one;
                                                          # d = 512 only from 1.5 M rows on, where the single launch wins)
                    assert ix.stat("fused") == int(single) and ix.stat("mfma") == 0 and int(fst.abs().sum().item()) == 0, (metric, setup, nq, k)
                    ix.set_option("use_fused", 0)
                    ui, us, _ = ix.topk_device(Q[:nq], k, mid)
                    assert ix.stat("fused") == 0
                    ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                    assert torch.equal(fi, ei) and torch.equal(fs, es) and torch.equal(fi, ui) and torch.equal(fs, us), (metric, setup, nq, k)
            ix.set_bias(None); ix.set_row_mask(None); ix.set_option("use_fused", 1)
            # (beyond a million rows a float64 pass of the oracle takes seconds: there one metric speaks for the others, which the
            # assertions above tie to it bit for bit through the exact selection)
            for qi in range(2 if (n < 1_000_000 or metric == "cosine_similarity") else 0):
                idx, sc = ix.topk(Q[qi:qi + 1], 100, mid)
                assert ix.stat("fused") == 1
                orc.check_topk(idx[0], sc[0], V, Q[qi], metric, 100, tol=1e-5)
        ix.topk_device(Q[:3], 10, METRIC_IDS["dot_product"])      # three float32 queries: the VALU five kernels; d <= 384 from 300k rows on: the batched launch
        assert ix.stat("fused") == (2 if (d <= 384 and n >= 300_000) else 0)
        if not (d == 512 and n < 1_500_000):
            idx, sc = ix.topk(V[11:12].copy(), 3, METRIC_IDS["euclidean_metric"])                 # an exact duplicate of two stored rows
            assert ix.stat("fused") == 1 and idx[0][0] == 11 and idx[0][1] == n - 1 and sc[0][0] == 1.0 and sc[0][1] == 1.0
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 12. the single-launch BATCHED pipeline (hdb_mfma_kernel.h, MODE 2): 5-256 dot / cosine queries, 1-256 euclidean
#     ones, fp16 and float32 matrices -- one launch per <= 256 queries
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,n,d,nqs", [(np.float16, 200_003, 384, (5, 16, 100, 256, 300)), (np.float16, 60_017, 768, (1, 7, 64, 130)),
                                         (np.float16, 30_000, 128, (40, 256)), (np.float16, 90_001, 512, (33, 200)),
                                         (np.float16, 50_000, 1536, (3, 32)), (np.float16, 8_193, 384, (9,)),
                                         (np.float16, 1_300_001, 384, (17,)), (np.float32, 120_000, 384, (5, 64, 130)),
                                         (np.float32, 70_000, 768, (8, 128))])
def test_batched_single_launch_equals_multi_kernel_and_oracle(orc, dt, n, d, nqs):
    """Every call shape the batched single launch takes -- query counts on both sides of a launch's capacity (chunks of 128 /
    256), dot / cosine / euclidean, k = 1 .. 128, bias, row mask, ragged last tile, grids smaller than the CU count, a
    float32 query far outside the fp16 range -- returns exactly what the five-kernel pipeline returns (and, for dot /
    cosine, the on-device exact selection), bit for bit; one query per shape is checked against the oracle's float64 scores."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(n + d)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    V[n - 1] = V[7]                                        # duplicate rows across the matrix ends: tie -> lower row first
    ix = GpuIndex(V)
    try:
        bias = (torch.rand(n, generator=g, device="cuda") * 0.2).float()
        mask = (torch.rand(n, generator=g, device="cuda") < 0.3).to(torch.uint8)
        for nq in nqs:
            Q = torch.randn((nq, d), generator=g, device="cuda").float()
            Q[0] = V[n // 3].float()                       # an exact duplicate of a stored row (euclidean: re-scored directly)
            if nq > 2:
                Q[2] = Q[2] * 37.5                         # a genuinely float32 query: norm sums must round alike in both pipelines
            for metric in ("cosine_similarity", "dot_product", "euclidean_metric", "pearson_correlation"):     # (pearson: the kernel centres its queries)
                if metric != "euclidean_metric" and nq <= (4 if d <= 768 else 2) and dt == np.float16:
                    continue                               # hdb_mfma_fused_kernel's calls
                if dt == np.float32 and nq < 5:
                    continue                               # the VALU pipelines' calls
                mid = METRIC_IDS[metric]
                for setup in ("plain", "bias", "mask+bias"):
                    ix.set_bias(bias if "bias" in setup else None)
                    ix.set_row_mask(mask if "mask" in setup else None)
                    for k in (100, 1, 128):
                        ix.set_option("use_fused", 1)
                        fi, fs, fst = ix.topk_device(Q, k, mid)
                        assert ix.stat("fused") == 2 and ix.stat("path") == 1 and ix.stat("mfma") == 1, (nq, metric, setup, k)
                        assert int(fst.abs().sum().item()) == 0, (nq, metric, setup, k)
                        ix.set_option("use_fused", 0)
                        ui, us, ust = ix.topk_device(Q, k, mid)
                        assert ix.stat("fused") == 0 and int(ust.abs().sum().item()) == 0
                        ix.set_option("use_fused", 1)
                        assert torch.equal(fi, ui) and torch.equal(fs, us), (nq, metric, setup, k)
                        if metric != "euclidean_metric" and k == 100 and nq <= 64:      # (euclidean: the exact selection ranks before the re-score)
                            ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
                            assert torch.equal(fi, ei) and torch.equal(fs, es), (nq, metric, setup, k)
                ix.set_bias(None); ix.set_row_mask(None)
                idx, sc = ix.topk(Q[:min(nq, 3)], 100, mid)
                Vh = V.cpu().numpy()
                tol = 1e-3 if dt == np.float16 else 1e-5
                for qi in range(min(nq, 3)):
                    if qi == 2 and metric == "euclidean_metric":
                        continue                           # |q| >> |v|: 1/(1+dist) is ill-conditioned in float32 (test_mfma_query_magnitude)
                    orc.check_topk(idx[qi], sc[qi], Vh, Q[qi].cpu().numpy(), metric, 100, tol=tol)
                if metric == "euclidean_metric":
                    assert idx[0][0] == n // 3 and abs(sc[0][0] - 1.0) < 1e-6, "an exact duplicate must score exactly 1"
    finally:
        ix.close()


def test_batched_single_launch_failure_paths(orc):
    """What must come back through the exact selection: a NaN query (status bit), massive ties that overflow the candidate
    lists, a row mask that leaves fewer than 8 sampled rows (threshold -inf), and an exchange that gives up (spin timeout
    forced to its floor) -- the host entry still returns the right rows, and the next call is clean."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS, Q_NAN
    rng = np.random.default_rng(77)
    base = rng.standard_normal((40, 384)).astype(np.float32).astype(np.float16)
    V = np.tile(base, (1500, 1))                           # 60k rows, every score repeated 1500 times
    Q = rng.standard_normal((9, 384)).astype(np.float16).astype(np.float32)
    ix = GpuIndex(V)
    try:
        mid = METRIC_IDS["dot_product"]
        ix.topk_device(Q, 100, mid)
        assert ix.stat("fused") == 2
        idx, sc = ix.topk(Q, 100, mid)                     # hdb_topk_host re-runs overflowing lists through the exact path
        for qi in (0, 8):
            ex = orc.exact_scores(V, Q[qi], "dot_product")
            want = np.nonzero(np.isclose(ex, ex.max(), rtol=1e-6))[0][:100]
            assert np.array_equal(idx[qi], want) and np.allclose(sc[qi], ex.max(), rtol=1e-3)
    finally:
        ix.close()
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 300_000
    V2 = torch.randn((n, 384), generator=g, device="cuda").to(torch.float16)
    Q2 = torch.randn((20, 384), generator=g, device="cuda").float()
    ix = GpuIndex(V2)
    try:
        mid = METRIC_IDS["cosine_similarity"]
        Qn = Q2.clone(); Qn[5, 17] = float("nan")
        _, _, st = ix.topk_device(Qn, 10, mid)
        assert ix.stat("fused") == 2 and (int(st[5].item()) & Q_NAN) and not any(int(st[q].item()) & Q_NAN for q in range(20) if q != 5)
        # fewer than 8 unmasked rows in the sample: the threshold is -inf, every unmasked row is a candidate
        ix.set_row_mask((torch.rand(n, generator=g, device="cuda") < 0.004).to(torch.uint8))
        fi, fs, st = ix.topk_device(Q2, 50, mid)
        ei, es, _ = ix.topk_device(Q2, 50, mid, exact=True)
        for q in range(20):
            if int(st[q].item()) == 0:
                assert torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q]), q
        hi, hs = ix.topk(Q2, 50, mid)
        assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy())
        ix.set_row_mask(None)
        # forced timeout: statuses tell, the host entry falls back, the next call is clean
        ei, es, _ = ix.topk_device(Q2, 50, mid, exact=True)
        ix.set_option("fused_timeout_us", 1)
        fi, fs, st = ix.topk_device(Q2, 50, mid)
        assert ix.stat("fused") == 2
        for q in range(20):
            if int(st[q].item()) == 0:
                assert torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q]), q
        hi, hs = ix.topk(Q2, 50, mid)
        assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy())
        ix.set_option("fused_timeout_us", 2000)
        fi, fs, st = ix.topk_device(Q2, 50, mid)
        assert int(st.abs().sum().item()) == 0 and torch.equal(fi, ei) and torch.equal(fs, es)
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 13. the single-launch pipeline of the bit metrics (hdb_bits_fused.hip): 1-4 hamming / jaccard queries per launch
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,n,d", [(np.float16, 200_003, 384), (np.float32, 70_001, 100), (np.float16, 1_000_000, 768),
                                     (np.float32, 8_193, 384), (np.float16, 300_000, 1536)])
def test_bits_single_launch_equals_multi_kernel_and_exact(orc, dt, n, d):
    """hamming / jaccard through ONE launch (stat 3) against the six-launch pipeline and the on-device exact selection, bit for
    bit (integer scores, ties by ascending row): 1-7 queries (more than four go four at a time), bias, row mask, k up to 128,
    d not a multiple of 32, a NaN query's status bit, and scores checked against the oracle's float64 integers."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS, Q_NAN
    g = torch.Generator(device="cuda").manual_seed(n + d)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    V[n - 1] = V[7]
    ix = GpuIndex(V); ix.set_option("bits_max_q", 16)      # (by rule more than four queries take the six launches in one go: forced here)
    try:
        bias = (torch.rand(n, generator=g, device="cuda") * 3.0).float()
        mask = (torch.rand(n, generator=g, device="cuda") < 0.3).to(torch.uint8)
        Q = torch.randn((7, d), generator=g, device="cuda").float()
        Q[1] = V[n // 2].float()
        for metric in ("hamming_distance", "jaccard_similarity"):
            mid = METRIC_IDS[metric]
            for setup in ("plain", "bias", "mask+bias"):
                ix.set_bias(bias if "bias" in setup else None)
                ix.set_row_mask(mask if "mask" in setup else None)
                for nq, k in ((1, 100), (2, 1), (3, 128), (4, 37), (7, 10)):
                    ix.set_option("use_fused", 1)
                    fi, fs, fst = ix.topk_device(Q[:nq], k, mid)
                    assert ix.stat("fused") == 3 and ix.stat("path") == 1, (metric, setup, nq, k)
                    ix.set_option("use_fused", 0)
                    ui, us, ust = ix.topk_device(Q[:nq], k, mid)
                    assert ix.stat("fused") == 0
                    ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                    for q in range(nq):
                        if int(fst[q].item()) == 0:      # (coarse integer levels may overflow a list: those go through the exact path)
                            assert torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q]), (metric, setup, nq, k, q)
                        assert int(fst[q].item()) == int(ust[q].item()) or int(fst[q].item()) == 0 or int(ust[q].item()) == 0
                    ix.set_option("use_fused", 1)
                    hi, hs = ix.topk(Q[:nq], k, mid)     # host entry: overflowing lists re-run inside
                    assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy()), (metric, setup, nq, k)
            ix.set_bias(None); ix.set_row_mask(None)
            idx, sc = ix.topk(Q[:2], 100, mid)
            Vh = V.cpu().numpy()
            for qi in range(2):
                orc.check_topk(idx[qi], sc[qi], Vh, Q[qi].cpu().numpy(), metric, 100, tol=0.0 if metric == "hamming_distance" else 1e-6)
            if metric == "hamming_distance":
                assert idx[1][0] == n // 2 and sc[1][0] == d
        ix.set_option("bits_fused", 3)                   # one query on a large matrix through the six launches (comparison setting)
        fi, fs, fst = ix.topk_device(Q[:1], 50, METRIC_IDS["hamming_distance"])
        assert ix.stat("fused") == (0 if n >= 1_000_000 else 3)
        ei, es, _ = ix.topk_device(Q[:1], 50, METRIC_IDS["hamming_distance"], exact=True)
        assert ix.stat("path") == 2 and (int(fst[0].item()) != 0 or (torch.equal(fi, ei) and torch.equal(fs, es)))
        ix.set_option("bits_fused", 1)
        Qn = Q[:3].clone(); Qn[2, 5] = float("nan")
        _, _, st = ix.topk_device(Qn, 10, METRIC_IDS["hamming_distance"])
        assert ix.stat("fused") == 3 and (int(st[2].item()) & Q_NAN) and not (int(st[0].item()) & Q_NAN)
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 13a. which pipeline serves a call of a few queries (tools/sweep_dispatch.py, profiles/r3_dispatch_few_queries.txt)
# ------------------------------------------------------------------------------------------------
def test_dispatch_rules_for_few_queries(orc):
    """One query: the 1-4-query single launch.  2-4 fp16 queries: the batched single launch below 1.5M rows (its eight multiplying
    waves beat the one multiplying wave of the 1-4-query kernel until the pass dominates), the 1-4-query kernel above (2-3
    queries).  float32, d <= 384, 3-4 queries: the batched launch from 300k rows on.  Hamming: up to four queries in one launch,
    more through the six launches in one go.  Every choice returns the same rows."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(77)
    cos, ham = METRIC_IDS["cosine_similarity"], METRIC_IDS["hamming_distance"]
    for dt, n, expect in ((torch.float16, 200_000, {1: 1, 2: 2, 3: 2, 4: 2, 5: 2}), (torch.float16, 1_600_000, {1: 1, 2: 1, 3: 1, 4: 2, 5: 2}),
                          (torch.float32, 100_000, {1: 1, 2: 1, 3: 0, 4: 0, 5: 2}), (torch.float32, 400_000, {1: 1, 2: 1, 3: 2, 4: 2, 5: 2})):
        V = torch.randn((n, 384), generator=g, device="cuda").to(dt)
        ix = GpuIndex(V)
        try:
            Q = torch.randn((8, 384), generator=g, device="cuda").to(dt).float()
            for nq, kind in expect.items():
                fi, fs, st = ix.topk_device(Q[:nq], 50, cos)
                assert ix.stat("fused") == kind, (dt, n, nq)
                ei, es, _ = ix.topk_device(Q[:nq], 50, cos, exact=True)
                assert int(st.abs().sum().item()) != 0 or (torch.equal(fi, ei) and torch.equal(fs, es)), (dt, n, nq)
            if dt == torch.float16:
                for nq, kind in ((1, 3), (4, 3), (5, 0), (16, 0)):
                    hi, hs = ix.topk(Q[:nq], 50, ham)
                    assert ix.stat("fused") == kind, (n, nq)
                    ei, es, _ = ix.topk_device(Q[:nq], 50, ham, exact=True)
                    assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy()), (n, nq)
        finally:
            ix.close()


# ------------------------------------------------------------------------------------------------
# 13b. appended rows: the lazy per-row caches (sign bits, pearson scales) are extended, not rebuilt
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,d", [(np.float16, 384), (np.float32, 100)])
def test_appended_rows_extend_sign_bits_and_pearson_scales(orc, dt, d):
    """GpuIndex.append after hamming / jaccard / pearson queries: the next such query packs / scales the appended rows only
    (hdb_index_extend keeps the 256-row blocks of sign bits and the scales of the old rows) -- results equal a fresh index over
    all rows bit for bit, across several appends, a growth of the bit buffer and block boundaries."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(d)
    total = 30_011
    V = rng.standard_normal((total, d)).astype(np.float32).astype(dt)
    Q = rng.standard_normal((3, d)).astype(np.float32)
    Q[1] = V[total - 5].astype(np.float32)                       # its best match arrives with the last append
    cuts = [9_000, 9_001, 9_256, 20_000, total]                  # one row, up to a block boundary, past the 25 % head room, the rest
    ix = GpuIndex(V[:cuts[0]].copy())
    try:
        metrics = ("hamming_distance", "jaccard_similarity", "pearson_correlation")
        for m in metrics:
            ix.topk(Q, 10, METRIC_IDS[m])                        # builds the lazy caches on the first rows
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            ix.append(V[lo:hi])
            fresh = GpuIndex(V[:hi].copy())
            try:
                for m in metrics:
                    for nq in (1, 3):
                        gi, gs = ix.topk(Q[:nq], 50, METRIC_IDS[m])
                        fi, fs = fresh.topk(Q[:nq], 50, METRIC_IDS[m])
                        assert np.array_equal(gi, fi) and np.array_equal(gs, fs), (m, nq, hi)
            finally:
                fresh.close()
        idx, sc = ix.topk(Q[1:2], 5, METRIC_IDS["hamming_distance"])
        assert idx[0][0] == total - 5 and sc[0][0] == d
        orc.check_topk(idx[0], sc[0], V, Q[1], "hamming_distance", 5, tol=0.0)
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 14. wide rows on the matrix cores through K slices (hdb_mfma_ksplit.hip): float32 d = 1024 / 1536, fp16 d = 2048 / 3072 / 4096
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,n,d,nq", [(np.float32, 40_017, 1536, 64), (np.float32, 30_000, 1024, 17), (np.float16, 50_003, 2048, 64),
                                        (np.float16, 20_000, 3072, 130), (np.float16, 20_017, 4096, 33)])
def test_wide_rows_k_slices_match_valu_scan_and_oracle(orc, dt, n, d, nq):
    """Batches on rows too wide for one wave's query fragments ride the matrix cores in K slices (partial sums in a float32
    buffer between the launches): same rows as the VALU scan (4 queries per pass) for dot / cosine / euclidean, with bias and
    row mask, ragged last tile, an exact duplicate (euclidean: re-scored directly) -- and the oracle's float64 scores."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(n + d)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    Q = torch.randn((nq, d), generator=g, device="cuda").to(V.dtype).float()       # (fp16 matrices multiply with fp16 copies of the queries)
    Q[0] = V[n - 3].float()
    ix = GpuIndex(V)
    try:
        bias = (torch.rand(n, generator=g, device="cuda") * 0.2).float()
        mask = (torch.rand(n, generator=g, device="cuda") < 0.3).to(torch.uint8)
        tol = 1e-3 if dt == np.float16 else 1e-5
        Vh = V.cpu().numpy()
        for metric in ("cosine_similarity", "dot_product", "euclidean_metric"):
            mid = METRIC_IDS[metric]
            for setup in ("plain", "mask+bias"):
                ix.set_bias(bias if "bias" in setup else None)
                ix.set_row_mask(mask if "mask" in setup else None)
                ix.set_option("use_mfma", 1)
                mi, ms, mst = ix.topk_device(Q, 50, mid)
                assert ix.stat("mfma") == 1 and ix.stat("path") == 1 and ix.stat("fused") == 0 and int(mst.abs().sum().item()) == 0
                ix.set_option("use_mfma", 0)
                vi, vs, vst = ix.topk_device(Q, 50, mid)
                assert ix.stat("mfma") == 0 and int(vst.abs().sum().item()) == 0
                ix.set_option("use_mfma", 1)
                mi_h, ms_h, vi_h, vs_h = mi.cpu().numpy(), ms.cpu().numpy(), vi.cpu().numpy(), vs.cpu().numpy()
                for qi in range(nq):
                    assert orc.same_result_modulo_ties(mi_h[qi], ms_h[qi], vi_h[qi], vs_h[qi], 5e-5 if dt == np.float16 else 1e-5), (metric, setup, qi)
                if setup == "plain":
                    for qi in (0, 1, nq - 1):
                        orc.check_topk(mi_h[qi], ms_h[qi], Vh, Q[qi].cpu().numpy(), metric, 50, tol=tol)
                    if metric == "euclidean_metric":
                        assert mi_h[0][0] == n - 3 and abs(ms_h[0][0] - 1.0) < 1e-6
            ix.set_bias(None); ix.set_row_mask(None)
            ei, es, _ = ix.topk_device(Q[:8], 50, mid, exact=True)       # the exact selection through the K-slice score writer
            assert ix.stat("mfma") == 1 and ix.stat("path") == 2
            fi, fs, _ = ix.topk_device(Q[:8], 50, mid)
            if metric != "euclidean_metric":
                assert torch.equal(ei, fi) and torch.equal(es, fs)
        ix.topk_device(Q[:3], 10, METRIC_IDS["dot_product"]); assert ix.stat("mfma") == 0      # up to 4 queries: one VALU pass
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 15. manhattan batches through the LDS-staged tile kernel (hdb_l1_tile.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,n,d", [(np.float16, 120_003, 384), (np.float16, 60_000, 256), (np.float16, 40_017, 128),
                                     (np.float32, 70_001, 384), (np.float32, 50_000, 128),
                                     (np.float16, 50_001, 768), (np.float16, 40_000, 640), (np.float16, 60_003, 512),
                                     (np.float32, 30_000, 768), (np.float32, 40_001, 512)])
def test_manhattan_tile_kernel_matches_scan_and_oracle(orc, dt, n, d):
    """manhattan_distance for 1-20 queries per call: one pass over V with the queries in registers against the 4-query VALU
    scan (same rows; fp16 data subtracts in fp16 like the reference, so scores agree to fp16 rounding of the differences) and the
    oracle's float64 scores; bias, row mask, ragged last tile, an exact duplicate (distance 0 -> similarity exactly 1)."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(n + d)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    Q = torch.randn((20, d), generator=g, device="cuda").to(V.dtype).float()
    Q[0] = V[n - 2].float()
    ix = GpuIndex(V)
    try:
        mid = METRIC_IDS["manhattan_distance"]
        bias = (torch.rand(n, generator=g, device="cuda") * 0.01).float()
        mask = (torch.rand(n, generator=g, device="cuda") < 0.3).to(torch.uint8)
        Vh = V.cpu().numpy()
        rel = 2e-4 if dt == np.float16 else 1e-6
        for setup in ("plain", "bias", "mask+bias"):
            ix.set_bias(bias if "bias" in setup else None)
            ix.set_row_mask(mask if "mask" in setup else None)
            for nq, k in ((1, 100), (5, 10), (8, 128), (20, 50)):
                ix.set_option("use_l1_tile", 1)
                ti, ts, tst = ix.topk_device(Q[:nq], k, mid)
                ix.set_option("use_l1_tile", 0)
                si, ss, sst = ix.topk_device(Q[:nq], k, mid)
                ix.set_option("use_l1_tile", 1)
                assert int(tst.abs().sum().item()) == 0 and int(sst.abs().sum().item()) == 0
                ti_h, ts_h, si_h, ss_h = ti.cpu().numpy(), ts.cpu().numpy(), si.cpu().numpy(), ss.cpu().numpy()
                for qi in range(nq):
                    a_, b_ = np.sort(ts_h[qi])[::-1], np.sort(ss_h[qi])[::-1]
                    assert np.all(np.abs(a_ - b_) <= rel * np.maximum(np.abs(b_), 1e-6)), (setup, nq, k, qi)
                    common = len(set(ti_h[qi].tolist()) & set(si_h[qi].tolist()))
                    assert common >= k - max(1, k // 20), (setup, nq, k, qi, common)          # near-ties at the k-th place may swap
            if setup == "plain":
                idx, sc = ix.topk(Q[:3], 100, mid)
                for qi in range(3):
                    orc.check_topk(idx[qi], sc[qi], Vh, Q[qi].cpu().numpy(), "manhattan_distance", 100, tol=1e-3 if dt == np.float16 else 1e-5)
                assert idx[0][0] == n - 2 and sc[0][0] == 1.0
        ix.set_bias(None); ix.set_row_mask(None)
        ei, es, _ = ix.topk_device(Q[:6], 50, mid, exact=True)            # exact selection through the tile kernel's score writer
        fi, fs, _ = ix.topk_device(Q[:6], 50, mid)
        assert torch.equal(ei, fi) and torch.equal(es, fs)
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------
# 14. round 4: the LOCAL flavour of the single launch (short matrices: no row sample, no exchange)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,n,d", [(np.float16, 151, 384), (np.float16, 1000, 384), (np.float16, 8192, 384), (np.float16, 20_011, 384),
                                    (np.float16, 100_000, 384), (np.float16, 262_144, 384), (np.float16, 262_209, 384),
                                    (np.float16, 30_000, 256), (np.float16, 70_001, 640), (np.float16, 40_003, 1024),
                                    (np.float32, 500, 384), (np.float32, 60_000, 384), (np.float32, 45_001, 768), (np.float32, 9_000, 128)])
def test_local_flavour_equals_the_other_pipelines_and_exact(orc, dt, n, d):
    """Matrices whose tiles fit the workgroups' parking areas take the single launch without any row sample or exchange: every
    workgroup emits the rows at or above ITS OWN local_m-th best and the last one checks those thresholds against the k-th best
    of the union (hdb_mfma_fused.h).  Same answers, bit for bit, as the pipelines it replaces (use_local = 0: the sampled
    threshold, or the three launches of n <= 8192) and as the on-device exact selection -- every metric, 1-4 queries, bias,
    row mask, k from 1 to 128, ragged last tiles, grids smaller than the CU count; 262 209 x 384 is one tile too many (17 per
    workgroup) and must fall back to the exchange by itself."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    rng = np.random.default_rng(n + d)
    V = rng.standard_normal((n, d)).astype(np.float32).astype(dt)
    V[n - 1] = V[7]                                        # duplicate rows at the matrix ends: tie -> lower row first
    Q = rng.standard_normal((4, d)).astype(dt).astype(np.float32)
    Q[1] = V[n // 3].astype(np.float32)
    Q[2] = rng.standard_normal(d).astype(np.float32) * 37.5
    ix = GpuIndex(V); ix.set_option("local_max_q", 4); ix.set_option("local_max_tiles", 16); ix.set_option("local_small", 1)
    # (by rule the local flavour takes calls of up to two queries on more than 8192 rows and up to 4 tiles per workgroup: every
    # shape it CAN take is forced here)
    maxq = (2 if d <= 384 else 1) if dt == np.float32 else (4 if d <= 768 else 2)
    rows = next(r for r in (64, 32, 16) if r * d * np.dtype(dt).itemsize <= 48 * 1024)
    tiles = (n + rows - 1) // rows
    per_wg = (tiles + min(tiles, 256) - 1) // min(tiles, 256)

    def fits(metric, nq):                                  # mirrors hdb_mfma_fused_local_tiles: tiles a workgroup can park
        if dt == np.float32:
            cap = 16 if (nq <= 1 or d > 384) else 8
        elif (d <= 640 and not (metric == "euclidean_metric" and d > 512)) or d > 768:
            cap = 16 if nq <= 2 else 32 // nq
        else:
            cap = 0
        return per_wg <= cap
    try:
        bias = torch.rand(n, generator=torch.Generator().manual_seed(3)).float().cuda() * 0.2
        mask = (torch.rand(n, generator=torch.Generator().manual_seed(4)) < 0.3).to(torch.uint8).cuda()
        for metric in ("cosine_similarity", "dot_product", "euclidean_metric", "pearson_correlation"):
            mid = METRIC_IDS[metric]
            for setup in ("plain", "bias", "mask+bias"):
                ix.set_bias(bias if "bias" in setup else None)
                ix.set_row_mask(mask if "mask" in setup else None)
                for nq, k in ((1, 100), (2, 1), (3, 128), (4, 37), (1, 5)):
                    if nq > maxq or (metric == "euclidean_metric" and dt == np.float16 and nq > 1):
                        continue
                    if "mask" in setup and int(mask.sum().item()) < min(k, n):
                        continue                               # (fewer rows than k pass the mask: every pipeline reports that, another test's subject)
                    ix.set_option("use_local", 1)
                    li, ls, lst = ix.topk_device(Q[:nq], k, mid)
                    if n > 8192 or fits(metric, nq):
                        assert ix.stat("fused") in (1, 2) and ix.stat("local") == (1 if fits(metric, nq) else 0), (metric, setup, nq, k)
                    assert int(lst.abs().sum().item()) == 0, (metric, setup, nq, k, lst.tolist())
                    ix.set_option("use_local", 0)
                    oi, os_, ost = ix.topk_device(Q[:nq], k, mid)
                    assert ix.stat("local") == 0 and int(ost.abs().sum().item()) == 0
                    if n <= 8192 and dt == np.float16:
                        # up to 8192 rows the other pipeline is the VALU scan: float32 queries as they are, where the matrix cores
                        # multiply with fp16 copies of them (Q[2] is a genuine float32 query) -- the fp16 contract, 1e-3, applies
                        for qi in range(nq):
                            if qi == 2:
                                continue                       # (Q[2] is no fp16 vector: its fp16 copy differs from it by 2^-11 per element, outside any score tolerance)
                            assert orc.same_result_modulo_ties(li[qi].cpu().numpy(), ls[qi].cpu().numpy().astype(np.float64),
                                                               oi[qi].cpu().numpy(), os_[qi].cpu().numpy().astype(np.float64), 1e-3), (metric, setup, nq, k, qi)
                        continue
                    assert torch.equal(li, oi) and torch.equal(ls, os_), (metric, setup, nq, k)
                    if metric != "euclidean_metric" or dt == np.float32:     # (fp16 euclidean: the exact selection ranks before the near-duplicate re-score)
                        ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                        assert torch.equal(li, ei) and torch.equal(ls, es), (metric, setup, nq, k)
            ix.set_bias(None); ix.set_row_mask(None); ix.set_option("use_local", 1)
            idx, sc = ix.topk(Q[:1], min(100, n), mid)
            orc.check_topk(idx[0], sc[0], V, Q[0], metric, min(100, n), tol=1e-3 if dt == np.float16 else 1e-5)
    finally:
        ix.close()


@pytest.mark.parametrize("dt", [np.float16, np.float32])
def test_local_flavour_cluster_in_one_tile_is_reported_and_rerun(dt):
    """The local flavour is exact only while no workgroup's own threshold reaches the k-th best of the union.  60 near-copies of one
    row inside ONE tile, queried with that row: that workgroup emits its local_m best only, its threshold lies above the k-th best,
    the last workgroup must report UNDERFLOW (never a silently short cluster) and the host call must come back, through the exact
    selection, with the true top-k.  A cluster spread over many tiles passes the check and needs no re-run."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    n, d = 90_000, 384
    g = torch.Generator(device="cuda").manual_seed(11)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    noise = torch.randn((60, d), generator=g, device="cuda")
    mid = METRIC_IDS["cosine_similarity"]
    ix = GpuIndex(V); ix.set_option("local_max_tiles", 16)
    try:
        for trial, c0 in enumerate((64 * 100, 64 * 777 + 1, n - 64)):
            saved = V[c0:c0 + 60].clone()
            V[c0:c0 + 60] = (saved[0:1].float() + 0.05 * noise).to(V.dtype)
            ix.update(V)
            q = V[c0].float().reshape(1, -1)
            li, ls, st = ix.topk_device(q, 100, mid)
            assert ix.stat("local") == 1
            assert int(st[0].item()) & 1, "a cluster of 60 in one tile must fail the check (UNDERFLOW)"
            ei, es, _ = ix.topk_device(q, 100, mid, exact=True)
            hi, hs = ix.topk(q, 100, mid)
            assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy()), (trial, c0)
            V[c0:c0 + 60] = saved
        rows = torch.arange(60, device="cuda") * 997 + 13                     # the same cluster, one copy per tile
        saved = V[rows].clone()
        V[rows] = (saved[0:1].float() + 0.05 * noise).to(V.dtype)
        ix.update(V)
        q = V[13].float().reshape(1, -1)
        li, ls, st = ix.topk_device(q, 100, mid)
        ei, es, _ = ix.topk_device(q, 100, mid, exact=True)
        assert int(st[0].item()) == 0 and torch.equal(li, ei) and torch.equal(ls, es)
    finally:
        ix.close()


@pytest.mark.parametrize("n,d", [(1_100_000, 384), (2_000_003, 128), (1_048_576, 100)])
def test_bits_local_flavour_equals_the_exchange_and_reports_clusters(n, d):
    """Round 4: hamming / jaccard without row sample and exchange (BitsArgs::local): same answers as the exchange flavour
    (bits_local = 0) and the exact selection on random rows, and 60 copies of one row inside one chunk of a workgroup -- whose own
    threshold then lies above the k-th best -- come back as UNDERFLOW from the device call and exact from the host call."""
    import torch
    from hyperdb._native import GpuIndex, METRIC_IDS
    g = torch.Generator(device="cuda").manual_seed(n + d)
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16)
    Q = torch.randn((4, d), generator=g, device="cuda").float()
    ix = GpuIndex(V)
    try:
        for metric in ("hamming_distance", "jaccard_similarity"):
            mid = METRIC_IDS[metric]
            for nq, k in ((1, 100), (4, 17), (2, 128)):
                ix.set_option("bits_local", 1)
                li, ls, lst = ix.topk_device(Q[:nq], k, mid)
                assert ix.stat("fused") == 3 and ix.stat("local") == 1 and int(lst.abs().sum().item()) == 0, (metric, nq, k, lst.tolist())
                ix.set_option("max_blocks", 150); ix.topk_device(Q[:nq], 100, mid)      # fewer than 2 k workgroups: the exchange flavour
                assert ix.stat("local") == 0
                ix.set_option("max_blocks", 0)
                ix.set_option("bits_local", 0)
                oi, os_, ost = ix.topk_device(Q[:nq], k, mid)
                assert ix.stat("local") == 0
                ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                assert torch.equal(li, ei) and torch.equal(ls, es), (metric, nq, k)
                for q in range(nq):
                    if int(ost[q].item()) == 0:
                        assert torch.equal(li[q], oi[q]) and torch.equal(ls[q], os_[q]), (metric, nq, k, q)
        ix.set_option("bits_local", 1)
        # 60 near-copies of one row, copy j with j signs flipped (scores d, d-1, ..., d-59), 64 rows apart inside ONE 4096-row chunk =
        # one workgroup: it emits its eight best only, the other fifty (all far above the k-th best of the union) stay behind
        c0 = 4096 * 3
        base = V[c0].clone()
        for j in range(60):
            row = base.clone()
            row[:j] = -row[:j]
            V[c0 + 64 * j] = row
        ix.update(V)
        q = base.float().reshape(1, -1)
        mid = METRIC_IDS["hamming_distance"]
        li, ls, st = ix.topk_device(q, 100, mid)
        assert int(st[0].item()) & 1, "sixty graded near-copies inside one chunk must fail the owner's check"
        ei, es, _ = ix.topk_device(q, 100, mid, exact=True)
        hi, hs = ix.topk(q, 100, mid)
        assert np.array_equal(hi, ei.cpu().numpy()) and np.array_equal(hs, es.cpu().numpy())
        assert float(hs[0][0]) == d and float(hs[0][20]) == d - 20
    finally:
        ix.close()
