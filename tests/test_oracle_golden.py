"""Pin the CPU oracle (oracle/ranking_oracle.py) to the reference.

Everything here runs on CPU.  The golden files were produced by the real reference
(tests/golden/make_golden.py); the oracle must reproduce them bit for bit, because it is
the same numpy op sequence.  The 19 cases of the reference's tests/test_ranking_algorithm.py
are restated against the oracle as well.
"""
import json
import os

import numpy as np
import pytest

from oracle import ranking_oracle as orc


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    return z, json.loads(str(z["manifest"]))


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


# ---------------------------------------------------------------- reference KATs (restated)
class TestReferenceKnownAnswers:
    """Same inputs/assertions as reference tests/test_ranking_algorithm.py:6-123."""

    def test_euclidean_shape_and_values(self):
        r = orc.score_euclidean(np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]), np.array([1, 1, 1]))
        assert r.shape == (3,) and np.all(r > 0)

    def test_euclidean_empty(self):
        with pytest.raises(ValueError):
            orc.score_euclidean(np.array([]), np.array([]))

    def test_cosine_values(self):
        assert np.array_equal(orc.score_cosine(np.array([[1, 0], [0, 1]]), np.array([1, 0])), [1.0, 0.0])

    def test_manhattan(self):
        assert np.allclose(orc.score_manhattan(np.array([[1, 0], [0, 1]]), np.array([1, 0])), [1.0, 1 / 3])

    def test_jaccard(self):
        assert np.array_equal(orc.score_jaccard(np.array([[1, 1], [1, 0], [0, 0]]), np.array([1, 1])), [1.0, 0.5, 0.0])
        assert np.array_equal(orc.score_jaccard(np.array([[2, 2], [2, 0], [0, 0]]), np.array([1, 1])), [1.0, 0.5, 0.0])

    def test_pearson(self):
        r = orc.score_pearson(np.array([[1, 1], [0, 1], [1, 0]]), np.array([1, 1]))
        assert np.isnan(r[0]) and r[1] != 0.0 and r[2] != 0.0
        assert np.all(np.isnan(orc.score_pearson(np.array([[1, 1], [0, 0], [1, 1]]), np.array([1, 1]))))

    def test_hamming(self):
        assert np.array_equal(orc.score_hamming(np.array([[1, 1], [0, 1], [1, 0]]), np.array([1, 1])), [2, 1, 1])

    @pytest.mark.parametrize("metric, rb, expected", [
        ("cosine_similarity", 0, [0, 2, 1]), ("cosine_similarity", 1, [2, 0, 1]),
        ("euclidean_metric", 0, [0, 2, 1]), ("manhattan_distance", 0, [0, 2, 1]),
        ("jaccard_similarity", 0, [0, 2, 1]), ("pearson_correlation", 0, [0, 1, 2]),
        ("hamming_distance", 0, [0, 2, 1])])
    def test_sort(self, metric, rb, expected):
        V = np.array([[1, 0], [0, 1], [0.5, 0.5]])
        idx, _ = orc.rank(V, np.array([1, 0]), metric=metric,
                          timestamps=[1627825200.0, 1627911600.0, 1627998000.0], recency_bias=rb)
        assert list(idx) == expected

    def test_unknown_metric(self):
        with pytest.raises(ValueError):
            orc.rank(np.array([[1, 0], [0, 1]]), np.array([1, 0]), metric="unknown_metric")

    def test_1d_vectors(self):
        with pytest.raises(ValueError):
            orc.rank(np.array([1, 0]), np.array([1, 0]), metric="euclidean_metric")

    def test_nan(self):
        with pytest.raises(ValueError):
            orc.rank(np.array([[1, 0], [0, 1], [np.nan, np.nan]]), np.array([1, 0]))


# ---------------------------------------------------------------- golden: outputs of the real reference
def test_kat_golden(golden_dir):
    z, cases = _load(golden_dir, "kat.npz")
    fn = {"euclidean_metric": orc.score_euclidean, "cosine_similarity": orc.score_cosine,
          "manhattan_distance": orc.score_manhattan, "jaccard_similarity": orc.score_jaccard,
          "pearson_correlation": orc.score_pearson, "hamming_distance": orc.score_hamming}
    n = 0
    for c in cases:
        if c["kind"] == "metric":
            got = fn[c["fn"]](z[c["name"] + ".V"].copy(), z[c["name"] + ".q"].copy())
            assert _same(got, z[c["name"] + ".out"]), c["name"]
            n += 1
        elif c["kind"] == "sort":
            idx, sc = orc.rank(z["sort.V"].copy(), z["sort.q"].copy(), metric=c["metric"],
                               timestamps=list(z["sort.ts"]), recency_bias=c["recency_bias"])
            assert _same(idx, z[c["name"] + ".idx"]) and _same(sc, z[c["name"] + ".scores"]), c["name"]
            n += 1
    assert n == 16


def test_sweep_golden(golden_dir):
    """560 (matrix, query, metric, k, recency) cases: indices AND float64 scores identical."""
    z, cases = _load(golden_dir, "sweep.npz")
    for c in cases:
        V, q = z[c["mat"] + ".V"], z[f"{c['mat']}.{c['query']}"]
        ts = {"none": None, "unix": z[c["mat"] + ".ts"], "small": z[c["mat"] + ".ts_small"]}[c["recency"]]
        with np.errstate(all="ignore"):
            idx, sc = orc.rank(V.copy(), q.copy(), top_k=c["top_k"], metric=c["metric"], timestamps=ts,
                               recency_bias=c["recency_bias"])
        assert _same(idx, z[c["name"] + ".idx"]), c["name"]
        assert _same(sc, z[c["name"] + ".scores"]), c["name"]


def test_sweep_full_vectors(golden_dir):
    z, cases = _load(golden_dir, "sweep.npz")
    seen = set()
    for c in cases:
        key = (c["mat"], c["query"], c["metric"])
        if key in seen:
            continue
        seen.add(key)
        V, q = z[c["mat"] + ".V"].copy(), z[f"{c['mat']}.{c['query']}"].copy()
        with np.errstate(all="ignore"):
            got = orc._SCORERS[c["metric"]](V, q)
        want = z[f"{c['mat']}.{c['query']}.{c['metric']}.full"]
        assert got.dtype == want.dtype and _same(got, want), key


def test_edge_golden(golden_dir, capsys):
    z, cases = _load(golden_dir, "edge.npz")
    for c in cases:
        name = c["name"]
        kw = {k: c[k] for k in ("top_k", "metric", "recency_bias") if k in c}
        if name + ".ts" in z.files:
            kw["timestamps"] = z[name + ".ts"]
        q = z[name + ".q"].copy()
        with np.errstate(all="ignore"):
            idx, sc = orc.rank(z[name + ".V"].copy(), q, **kw)
        printed = capsys.readouterr().out
        assert _same(idx, z[name + ".idx"]), name
        assert _same(sc, z[name + ".scores"]), name
        assert _same(q, z[name + ".q_after"]), name + " (in-place query mutation)"
        assert printed == c["printed"], name


# ---------------------------------------------------------------- exact arbitration + comparator self-checks
def test_exact_scores_close_to_reference(golden_dir):
    z, cases = _load(golden_dir, "sweep.npz")
    for mat in ("f16_2048x384", "f32_1024x384", "f64_256x96", "f32_300x100"):
        V, q = z[mat + ".V"], z[mat + ".q0"]
        for metric in ("dot_product", "cosine_similarity", "euclidean_metric", "hamming_distance",
                       "manhattan_distance"):
            ref = z[f"{mat}.q0.{metric}.full"].astype(np.float64)
            ex = orc.exact_scores(V, q, metric)
            tol = 0.0 if metric == "hamming_distance" else (
                2e-3 if V.dtype == np.float16 else 1e-4 if metric == "dot_product" else 1e-5)
            assert np.all(np.abs(ref - ex) <= tol * np.maximum(1, np.abs(ex))), (mat, metric)


def test_comparator_accepts_reference_and_rejects_wrong(golden_dir):
    z, cases = _load(golden_dir, "sweep.npz")
    for c in cases:
        if c["recency"] != "none" or c["metric"] in ("jaccard_similarity", "pearson_correlation"):
            continue
        if c["mat"] not in ("f16_2048x384", "f32_1024x384", "f32_64x8"):
            continue
        V, q = z[c["mat"] + ".V"], z[f"{c['mat']}.{c['query']}"]
        idx, sc = z[c["name"] + ".idx"], z[c["name"] + ".scores"]
        tol = 0.0 if c["metric"] == "hamming_distance" else (2e-3 if V.dtype == np.float16 else 1e-4)
        orc.check_topk(*orc.canonical(idx, sc), V, q, c["metric"], c["top_k"], tol=tol)
        if 1 < len(idx) < V.shape[0] and c["metric"] != "hamming_distance":
            exact = orc.exact_scores(V, q, c["metric"])
            worst = int(np.argmin(exact))
            if worst not in idx:
                bad = np.array(idx).copy()
                bad[0] = worst          # replace the best hit by the worst row
                with pytest.raises(AssertionError):
                    orc.check_topk(*orc.canonical(bad, exact[bad]), V, q, c["metric"], c["top_k"], tol=tol)


def test_config1_plumbing_golden(golden_dir):
    """BASELINE config 1 (151 documents, d=384, cosine top-5 via numpy on the CPU, no GPU): the oracle reproduces the
    reference's answer on the committed fixture bit for bit (tests/golden/c1.npz, made by make_golden.py c1)."""
    g = np.load(os.path.join(golden_dir, "c1.npz"))
    idx, sc = orc.rank(g["V"].copy(), g["q"].copy(), top_k=5, metric="cosine_similarity")
    assert np.array_equal(idx, g["ref_idx"]) and np.array_equal(sc, g["ref_scores"])
    assert idx[0] == 142 and sc.dtype == np.float64
