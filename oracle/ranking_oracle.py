"""CPU oracle for the brute-force ranking path of HyperDB.  TEST INFRASTRUCTURE ONLY.

This file is a numpy *restatement* of the algorithm in the reference's
``hyperdb/ranking_algorithm.py`` (204 lines).  It exists so that the HIP path can be
checked on the GPU box, where ``/root/reference`` does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it;
the product (``local-hyperdb_amd/``) never does and fails loudly without its HIP library.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here
against (a) the 19 known-answer cases of the reference's own
``tests/test_ranking_algorithm.py`` and (b) ``tests/golden/*.npz``, which were produced by
running the real reference (loaded by file path) in the build container with
``tests/golden/make_golden.py``.

Three layers live here:

* ``score_*`` / ``rank``      -- op-for-op restatement (same numpy calls, same dtype
                                 promotion, same quirks) of the reference.
* ``exact_scores``            -- float64 arithmetic on the *stored* values, used to
                                 arbitrate GPU-vs-oracle disagreements inside the fp16
                                 rounding band (SURVEY.md section 8a rule 3).
* ``canonical`` / ``check_topk`` -- the parity comparator (ties, tolerance bands).
"""
from __future__ import annotations

import numpy as np

METRICS = (
    "dot_product",
    "cosine_similarity",
    "euclidean_metric",
    "manhattan_distance",
    "jaccard_similarity",
    "pearson_correlation",
    "hamming_distance",
)

NAN_MESSAGE = "Vectors and query_vector should not contain NaN values."


# --------------------------------------------------------------------------------------
# op-for-op restatement
# --------------------------------------------------------------------------------------
def unit_rows(x):
    """x / ||x||_2 along the last axis, rows of norm 0 left untouched (divided by 1).

    Follows reference ranking_algorithm.py:8-21.  The norm is taken in the dtype of
    ``x`` (an fp16 matrix gets fp16 norms), exactly like the reference.
    """
    length = np.linalg.norm(x, axis=-1, keepdims=True)          # :9
    if np.where(length == 0)[0].size > 0:                        # :11,:14
        length[length == 0] = 1                                  # :15
    bad = np.where(np.isnan(x))                                  # :12
    if bad[0].size > 0:                                          # :17
        print(f"Warning: Vectors at indices {bad} contain NaN values.")  # :18
    return x / length                                            # :20


def score_dot(V, q):
    """V . q^T -- reference ranking_algorithm.py:24-30."""
    return np.dot(V, q.T)


def score_cosine(V, q):
    """dot of unit rows with the unit query, flattened -- reference :32-42."""
    return np.dot(unit_rows(V), unit_rows(q).T).flatten()


def score_euclidean(V, q, get_similarity_score=True):
    """1 / (1 + ||v - q||_2) per row (or the raw distance) -- reference :44-52."""
    dist = np.linalg.norm(V - q, axis=1)
    return 1 / (1 + dist) if get_similarity_score else dist


def score_manhattan(V, q):
    """1 / (1 + sum |v - q|) per row -- reference :54-61."""
    return 1 / (1 + np.sum(np.abs(V - q), axis=1))


def binarize_inplace(x):
    """Map x to {0,1} by ``x > 0`` IN PLACE unless it already is -- reference :116-126.

    The in-place write is a reference quirk that callers can observe on their query
    array ("semantics quirks").
    """
    seen = np.unique(x)
    for ok in ([0, 1], [0], [1]):
        if np.array_equal(seen, ok):
            return x
    x[x > 0] = 1
    x[x <= 0] = 0
    return x


def score_jaccard(V, q):
    """|v AND q| / |v OR q| on binarised uint8 rows -- reference :63-75."""
    vb = binarize_inplace(V).astype(np.uint8)
    qb = binarize_inplace(q).astype(np.uint8)
    both = np.bitwise_and(vb, qb)
    either = np.bitwise_or(vb, qb)
    return np.sum(both, axis=1) / np.sum(either, axis=1)


def score_pearson(V, q):
    """Pearson r per row with the reference's NaN-on-constant rule -- reference :77-113."""
    q = q.flatten()
    q_mu, v_mu = np.mean(q), np.mean(V, axis=1)
    q_sd, v_sd = np.std(q), np.std(V, axis=1)
    cov = np.sum((V - v_mu[:, np.newaxis]) * (q - q_mu), axis=1)
    den = v_sd * q_sd * V.shape[1]
    out = np.zeros(V.shape[0])
    nz = den != 0
    out[nz] = cov[nz] / den[nz]
    q_const, v_const = (q_sd == 0), (v_sd == 0)
    out[q_const & v_const] = np.nan
    out[q_const ^ v_const] = np.nan
    return out


def score_hamming(V, q):
    """d - popcount(bin(v) XOR bin(q)) per row, uint64 -- reference :128-147."""
    vb = binarize_inplace(V).astype(np.uint8)
    qb = binarize_inplace(q).astype(np.uint8)
    flips = np.sum(np.unpackbits(np.bitwise_xor(vb, qb), axis=1), axis=1)
    return V.shape[-1] - flips


_SCORERS = {
    "dot_product": score_dot,
    "cosine_similarity": score_cosine,
    "euclidean_metric": score_euclidean,
    "manhattan_distance": score_manhattan,
    "jaccard_similarity": score_jaccard,
    "pearson_correlation": score_pearson,
    "hamming_distance": score_hamming,
}


def recency_term(n, timestamps, recency_bias):
    """rb * exp(ts - max ts), zeros when no timestamps -- reference :180-183."""
    term = np.zeros(n)
    if timestamps is not None and len(timestamps) > 0:
        term = recency_bias * np.exp(-np.max(timestamps) + timestamps)
    return term


def rank(vectors, query_vector, top_k=5, metric="cosine_similarity", timestamps=None, recency_bias=0):
    """Restatement of ``hyperDB_ranking_algorithm_sort`` -- reference :149-204.

    Returns (int64 indices, float64 scores) sorted by score descending, with the
    reference's special cases: NaN -> ValueError (:150-151), unknown metric ->
    ValueError (:166), a single row -> (array([0]), array([scores])) plus an Info print
    (:189-191), top_k == 0 -> ([], []) (:202).
    """
    if np.isnan(vectors).any() or np.isnan(query_vector).any():   # :150
        raise ValueError(NAN_MESSAGE)                               # :151
    vectors = np.array(vectors)                                     # :153 (defensive copy)
    scorer = _SCORERS.get(metric)
    if scorer is None:
        raise ValueError(f"Unknown metric: {metric}")               # :166
    sims = scorer(vectors, query_vector).astype(float)              # :168,:171
    sims[np.isnan(sims)] = -np.inf                                  # :174
    scores = sims + recency_term(len(sims), timestamps, recency_bias)  # :180-186
    if np.array(scores).shape == () or (len(scores) == 1 and np.array(scores).ndim == 1):
        print("Info: Only one document left.")                      # :190
        return np.array([0]), np.array([scores])                    # :191
    if len(scores) > 0:                                             # :194
        k = max(0, min(top_k, len(scores)))                         # :195
        if k <= 0:
            return [], []                                           # :202
        scores = scores.flatten()                                   # :198
        top = np.argpartition(scores, -k)[-k:]                      # :199
        top = top[np.argsort(-scores[top])]                         # :200
    return top, scores[top]                                         # :204


# --------------------------------------------------------------------------------------
# float64 arbitration
# --------------------------------------------------------------------------------------
def _exact_block(V, q, metric):
    if metric == "dot_product":
        return V @ q
    if metric == "cosine_similarity":
        vn = np.sqrt((V * V).sum(axis=1))
        qn = np.sqrt((q * q).sum())
        vn[vn == 0] = 1.0
        qn = qn if qn != 0 else 1.0
        return (V @ q) / (vn * qn)
    if metric == "euclidean_metric":
        diff = V - q
        return 1.0 / (1.0 + np.sqrt((diff * diff).sum(axis=1)))
    if metric == "manhattan_distance":
        return 1.0 / (1.0 + np.abs(V - q).sum(axis=1))
    if metric == "hamming_distance":
        return (V.shape[1] - ((V > 0) != (q > 0)).sum(axis=1)).astype(np.float64)
    if metric == "jaccard_similarity":
        vb, qb = V > 0, q > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            s = (vb & qb).sum(axis=1) / (vb | qb).sum(axis=1)
        s[np.isnan(s)] = -np.inf
        return s
    if metric == "pearson_correlation":
        s = score_pearson(V, q)
        s[np.isnan(s)] = -np.inf
        return s
    raise ValueError(f"Unknown metric: {metric}")


def exact_scores(vectors, query_vector, metric, bias=None, block=1 << 16):
    """Scores in float64 arithmetic on the stored (fp16/fp32/fp64) values.

    Not the reference's rounding behaviour -- the mathematically exact value of the
    metric on the data as stored (NaN scores become -inf like reference :174).
    ``bias`` (length N) is added as-is.  Works blockwise so N = 1M rows stays small.
    """
    vectors = np.asarray(vectors)
    q = np.asarray(query_vector, dtype=np.float64).reshape(-1)
    n = vectors.shape[0]
    out = np.empty(n, dtype=np.float64)
    for lo in range(0, n, block):
        out[lo:lo + block] = _exact_block(vectors[lo:lo + block].astype(np.float64), q, metric)
    if bias is not None:
        out = out + np.asarray(bias, dtype=np.float64)
    return out


# --------------------------------------------------------------------------------------
# parity comparator
# --------------------------------------------------------------------------------------
def canonical(indices, scores):
    """Reorder a (indices, scores) result to (score descending, index ascending)."""
    indices = np.asarray(indices).reshape(-1).astype(np.int64)
    scores = np.asarray(scores, dtype=np.float64).reshape(-1)
    order = np.lexsort((indices, -scores))
    return indices[order], scores[order]


def tolerance_for(dtype, metric):
    """(abs/rel band) from BASELINE.json north_star: 1e-3 for fp16, 1e-5 for fp32/fp64,
    applied as tol * max(1, |s|) (SURVEY.md section 8a rules 2-3)."""
    return 1e-3 if np.dtype(dtype) == np.float16 else 1e-5


def check_topk(got_idx, got_scores, vectors, query_vector, metric, top_k, *, bias=None,
               tol=None, exact=None):
    """Assert that (got_idx, got_scores) is a valid top-k of the metric within tolerance.

    Size-independent validity check used at every N:
      * k = min(top_k, N) distinct in-range indices, scores sorted descending;
      * every returned score is within ``tol * max(1,|s|)`` of the float64-exact score of
        the row it names;
      * no row left out beats the k-th returned row by more than ``2*tol*|s_k|`` (relative,
        because euclidean/manhattan scores are ~1e-2 and an absolute band would accept
        anything), so an index difference from the reference can only be a swap inside the
        rounding band;
      * for integer metrics (hamming) tol == 0 makes all of this bit-exact.
    Returns the float64-exact score vector for reuse.
    """
    V = np.asarray(vectors)
    n = V.shape[0]
    if tol is None:
        tol = 0.0 if metric == "hamming_distance" else tolerance_for(V.dtype, metric)
    if exact is None:
        exact = exact_scores(V, query_vector, metric, bias=bias)
    got_idx = np.asarray(got_idx).reshape(-1)
    got_scores = np.asarray(got_scores, dtype=np.float64).reshape(-1)
    k = max(0, min(int(top_k), n))
    assert got_idx.shape[0] == k, f"expected {k} results, got {got_idx.shape[0]}"
    assert got_scores.shape[0] == k
    if k == 0:
        return exact
    assert got_idx.min() >= 0 and got_idx.max() < n, "index out of range"
    assert np.unique(got_idx).shape[0] == k, "duplicate indices in top-k"
    assert np.all(got_scores[1:] <= got_scores[:-1]), "scores not sorted descending"   # (-inf, -inf) pairs allowed
    ref_at = exact[got_idx]
    finite = np.isfinite(ref_at)
    band = tol * np.maximum(1.0, np.abs(ref_at))
    err = np.abs(got_scores[finite] - ref_at[finite])
    assert np.all(err <= band[finite] + 0.0), (
        f"score error {err.max():.3e} exceeds band (tol={tol}) for metric {metric}")
    assert np.array_equal(got_scores[~finite], ref_at[~finite]), "non-finite score mismatch"
    # nothing outside the returned set may beat the weakest returned row by > band
    kth = ref_at.min()
    mask = np.ones(n, dtype=bool)
    mask[got_idx] = False
    if mask.any():
        best_out = exact[mask].max()
        slack = 2.0 * tol * abs(kth) if np.isfinite(kth) else 0.0     # relative: scores may be tiny
        assert best_out <= kth + slack, (
            f"row with exact score {best_out!r} was left out while {kth!r} was returned")
    return exact


def same_result_modulo_ties(idx_a, sc_a, idx_b, sc_b, tol):
    """True when two top-k results agree: equal after canonicalisation, or differing only
    where the scores involved sit within ``tol*max(1,|s|)`` of each other."""
    ia, sa = canonical(idx_a, sc_a)
    ib, sb = canonical(idx_b, sc_b)
    if ia.shape != ib.shape:
        return False
    if not np.all(np.abs(sa - sb) <= tol * np.maximum(1.0, np.abs(sa)) + 0.0):
        return False
    if np.array_equal(ia, ib):
        return True
    only_a = np.setdiff1d(ia, ib)
    only_b = np.setdiff1d(ib, ia)
    if only_a.size != only_b.size:
        return False
    # the symmetric difference must sit at the bottom band of the list
    kth = min(sa.min(), sb.min())
    band = 2.0 * tol * abs(kth)
    sa_map = dict(zip(ia.tolist(), sa.tolist()))
    sb_map = dict(zip(ib.tolist(), sb.tolist()))
    return all(sa_map[i] <= kth + band for i in only_a) and all(sb_map[i] <= kth + band for i in only_b)
