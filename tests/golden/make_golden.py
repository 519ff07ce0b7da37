#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference ranking module.

Run in the build container only (it needs /root/reference, which does not exist on the
GPU box):   python tests/golden/make_golden.py

The reference module is loaded *by file path* (its package __init__ pulls in onnxruntime
etc., which are absent -- SURVEY.md section 8c).  Nothing from the reference is copied:
the outputs written here are data (inputs + the reference's answers).

Files written:
  kat.npz      the reference's own 19 unit-test cases (tests/test_ranking_algorithm.py)
               re-run through the reference, inputs and outputs recorded;
  sweep.npz    seeded random matrices x {7 metrics} x {k} x {recency on/off}: full score
               vectors of every metric function and (indices, scores) of the sort entry;
  c1.npz       BASELINE config 1: 151 x 384 seeded vectors (the size of demo/pokemon.jsonl with MiniLM-sized
               embeddings), query = row 142 + noise, cosine top-5 by the reference on the CPU;
  edge.npz     N==1, top_k==0, top_k>N, zero rows, (1,d) query, ties, in-place query
               binarisation, raw-vs-double recency.
"""
import importlib.util
import io
import json
import contextlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/hyperdb/ranking_algorithm.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_ranking", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


METRIC_FUNCS = {
    "dot_product": "dot_product",
    "cosine_similarity": "cosine_similarity",
    "euclidean_metric": "euclidean_metric",
    "manhattan_distance": "manhattan_distance",
    "jaccard_similarity": "jaccard_similarity",
    "pearson_correlation": "pearson_correlation",
    "hamming_distance": "hamming_distance",
}


def quiet(fn, *a, **kw):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


def kat(ref):
    """The reference's own test inputs (tests/test_ranking_algorithm.py), answers recorded."""
    out, cases = {}, []

    def rec(name, fn, V, q):
        res = getattr(ref, fn)(V.copy(), q.copy())
        out[f"{name}.V"], out[f"{name}.q"], out[f"{name}.out"] = V, q, np.asarray(res)
        cases.append({"name": name, "kind": "metric", "fn": fn})

    rec("euclid_shape", "euclidean_metric", np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]), np.array([1, 1, 1]))  # :7-14
    rec("cosine_values", "cosine_similarity", np.array([[1, 0], [0, 1]]), np.array([1, 0]))                    # :24-29
    rec("manhattan_basic", "manhattan_distance", np.array([[1, 0], [0, 1]]), np.array([1, 0]))                 # :32-37
    rec("jaccard_basic", "jaccard_similarity", np.array([[1, 1], [1, 0], [0, 0]]), np.array([1, 1]))           # :40-45
    rec("jaccard_nonbinary", "jaccard_similarity", np.array([[2, 2], [2, 0], [0, 0]]), np.array([1, 1]))       # :47-52
    rec("pearson_basic", "pearson_correlation", np.array([[1, 1], [0, 1], [1, 0]]), np.array([1, 1]))          # :55-62
    rec("pearson_constant", "pearson_correlation", np.array([[1, 1], [0, 0], [1, 1]]), np.array([1, 1]))       # :64-71
    rec("hamming_basic", "hamming_distance", np.array([[1, 1], [0, 1], [1, 0]]), np.array([1, 1]))             # :74-79

    V = np.array([[1, 0], [0, 1], [0.5, 0.5]])                                                                  # :93
    q = np.array([1, 0])                                                                                        # :94
    ts = [1627825200.0, 1627911600.0, 1627998000.0]                                                             # :95
    out["sort.V"], out["sort.q"], out["sort.ts"] = V, q, np.array(ts)
    for metric, rb in [("cosine_similarity", 0), ("cosine_similarity", 1), ("euclidean_metric", 0),
                       ("manhattan_distance", 0), ("jaccard_similarity", 0), ("pearson_correlation", 0),
                       ("hamming_distance", 0), ("dot_product", 0)]:
        idx, sc = ref.hyperDB_ranking_algorithm_sort(V.copy(), q.copy(), metric=metric, timestamps=ts, recency_bias=rb)
        name = f"sort.{metric}.rb{rb}"
        out[f"{name}.idx"], out[f"{name}.scores"] = np.asarray(idx), np.asarray(sc)
        cases.append({"name": name, "kind": "sort", "metric": metric, "recency_bias": rb})
    # error cases (:16-21, :100-105, :107-114, :116-123) are behavioural: recorded as names only
    cases += [{"name": "euclid_empty", "kind": "raises"}, {"name": "unknown_metric", "kind": "raises"},
              {"name": "euclid_1d", "kind": "raises"}, {"name": "nan_input", "kind": "raises"}]
    out["manifest"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)
    print("kat.npz:", len(cases), "cases")


def sweep(ref):
    rng = np.random.default_rng(20261003)
    mats = {
        "f16_2048x384": (rng.standard_normal((2048, 384)).astype(np.float32)).astype(np.float16),
        "f32_1024x384": rng.standard_normal((1024, 384)).astype(np.float32),
        "f16_768x768": (rng.standard_normal((768, 768)).astype(np.float32)).astype(np.float16),
        "f64_256x96": rng.standard_normal((256, 96)),
        "f32_300x100": rng.standard_normal((300, 100)).astype(np.float32),   # d not a multiple of 8
        "f16_513x50": rng.standard_normal((513, 50)).astype(np.float16),     # ragged everything
        "f32_64x8": rng.standard_normal((64, 8)).astype(np.float32),
        "f32_7x2": rng.standard_normal((7, 2)).astype(np.float32),
    }
    out, cases = {}, []
    for tag, V in mats.items():
        n, d = V.shape
        # two queries per matrix: an independent one and a noisy copy of a stored row
        q0 = rng.standard_normal(d).astype(V.dtype)
        q1 = (V[n // 3].astype(np.float64) + 0.05 * rng.standard_normal(d)).astype(V.dtype)
        ts = 1.7e9 + rng.uniform(0, 30 * 86400.0, size=n)          # unix seconds, 30-day window
        ts_small = rng.uniform(0, 5.0, size=n)                      # decay visible in top-k
        out[f"{tag}.V"], out[f"{tag}.q0"], out[f"{tag}.q1"] = V, q0, q1
        out[f"{tag}.ts"], out[f"{tag}.ts_small"] = ts, ts_small
        for qi, q in (("q0", q0), ("q1", q1)):
            for metric, fn in METRIC_FUNCS.items():
                with np.errstate(all="ignore"):
                    full = np.asarray(getattr(ref, fn)(V.copy(), q.copy()))
                out[f"{tag}.{qi}.{metric}.full"] = full
                for k in sorted({1, 5, 100, n + 3}):
                    for rec_name, tsv, rb in (("none", None, 0), ("unix", ts, 0.5), ("small", ts_small, 0.25)):
                        if rec_name != "none" and (k != 5 or qi != "q0"):
                            continue
                        with np.errstate(all="ignore"):
                            (idx, sc), _ = quiet(ref.hyperDB_ranking_algorithm_sort, V.copy(), q.copy(), top_k=k,
                                                 metric=metric, timestamps=tsv, recency_bias=rb)
                        name = f"{tag}.{qi}.{metric}.k{k}.{rec_name}"
                        out[f"{name}.idx"], out[f"{name}.scores"] = np.asarray(idx), np.asarray(sc)
                        cases.append({"name": name, "mat": tag, "query": qi, "metric": metric, "top_k": k,
                                      "recency": rec_name, "recency_bias": rb})
    out["manifest"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "sweep.npz"), **out)
    print("sweep.npz:", len(cases), "cases")


def edge(ref):
    rng = np.random.default_rng(7)
    out, cases = {}, []

    def sort_case(name, V, q, **kw):
        qq = q.copy()
        with np.errstate(all="ignore"):
            (idx, sc), printed = quiet(ref.hyperDB_ranking_algorithm_sort, V.copy(), qq, **kw)
        out[f"{name}.V"], out[f"{name}.q"] = V, q
        out[f"{name}.q_after"] = qq                      # in-place mutation of the query (hamming/jaccard)
        out[f"{name}.idx"], out[f"{name}.scores"] = np.asarray(idx), np.asarray(sc)
        if kw.get("timestamps") is not None:
            out[f"{name}.ts"] = np.asarray(kw["timestamps"], dtype=np.float64)
        meta = {k: v for k, v in kw.items() if k != "timestamps"}
        cases.append({"name": name, "printed": printed, **meta})

    d = 16
    V1 = rng.standard_normal((1, d)).astype(np.float32)
    sort_case("single_row", V1, V1[0] * 0.5, top_k=3, metric="cosine_similarity")
    V = rng.standard_normal((40, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    sort_case("topk_zero", V, q, top_k=0, metric="dot_product")
    sort_case("topk_gt_n", V, q, top_k=1000, metric="dot_product")
    sort_case("topk_eq_n", V, q, top_k=40, metric="euclidean_metric")
    Vz = V.copy(); Vz[[3, 17]] = 0
    sort_case("zero_rows_cosine", Vz, q, top_k=40, metric="cosine_similarity")
    sort_case("zero_query_cosine", V, np.zeros(d, np.float32), top_k=5, metric="cosine_similarity")
    sort_case("query_1xd", V, q.reshape(1, d), top_k=5, metric="cosine_similarity")
    sort_case("hamming_inplace", V, q, top_k=7, metric="hamming_distance")
    sort_case("jaccard_inplace", V, q, top_k=7, metric="jaccard_similarity")
    Vi = np.eye(4)
    sort_case("recency_identity", Vi, Vi[0].copy(), top_k=4, metric="dot_product",
              timestamps=[0.0, 1.0, 2.0, 3.0], recency_bias=0.5)
    # recency_bias without timestamps has no effect (reference :180-183)
    sort_case("recency_no_ts", V, q, top_k=5, metric="dot_product", recency_bias=3.0)
    # empty timestamp list behaves like None
    sort_case("recency_empty_ts", V, q, top_k=5, metric="dot_product", timestamps=[], recency_bias=3.0)
    # duplicated rows -> exactly tied float scores
    Vd = np.concatenate([V[:10], V[:10], V[:10]]).astype(np.float32)
    sort_case("duplicate_rows", Vd, q, top_k=12, metric="dot_product")
    # double recency as applied through HyperDB.query (hyperdb.py:1344 then ranking :183)
    ts = 1.7e9 + rng.uniform(0, 86400.0, size=40)
    first = 0.5 * np.exp(-np.max(ts) + ts)
    sort_case("double_recency", V, q, top_k=5, metric="cosine_similarity", timestamps=first, recency_bias=0.5)
    # hamming with many ties, larger N
    Vh = rng.standard_normal((4096, 64)).astype(np.float32)
    qh = rng.standard_normal(64).astype(np.float32)
    sort_case("hamming_ties", Vh, qh, top_k=100, metric="hamming_distance")
    out["hamming_ties.full"] = np.asarray(ref.hamming_distance(Vh.copy(), qh.copy()))
    out["manifest"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "edge.npz"), **out)
    print("edge.npz:", len(cases), "cases")


def c1(ref):
    """BASELINE config 1 (plumbing): the reference's numpy path on a pokemon-sized store."""
    rng = np.random.default_rng(151)
    V = rng.standard_normal((151, 384)).astype(np.float32)
    q = (V[142] + 0.05 * rng.standard_normal(384)).astype(np.float32)
    (idx, sc), _ = quiet(ref.hyperDB_ranking_algorithm_sort, V.copy(), q.copy(), top_k=5, metric="cosine_similarity")
    np.savez_compressed(os.path.join(HERE, "c1.npz"), V=V, q=q, ref_idx=np.asarray(idx), ref_scores=np.asarray(sc))
    print("c1.npz: top-5", list(idx))


if __name__ == "__main__":
    import sys
    ref = load_reference()
    if sys.argv[1:] == ["c1"]:          # only the config-1 fixture (the other files are unchanged)
        c1(ref)
        raise SystemExit(0)
    kat(ref)
    sweep(ref)
    edge(ref)
    c1(ref)
    import numpy, scipy
    with open(os.path.join(HERE, "PROVENANCE.txt"), "w") as f:
        f.write("Generated by tests/golden/make_golden.py from the reference ranking module loaded by path\n"
                f"numpy {numpy.__version__}, scipy {scipy.__version__} (reference pins numpy==1.26.3)\n")
