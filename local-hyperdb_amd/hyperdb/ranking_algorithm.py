"""MI355X-native drop-in for the reference's ``hyperdb/ranking_algorithm.py``.

Same public names and signatures as the reference module (``get_norm_vector``, ``dot_product``,
``cosine_similarity``, ``euclidean_metric``, ``hamming_distance``, ``check_and_binarize_vectors``,
``hyperDB_ranking_algorithm_sort``; reference ranking_algorithm.py:8,24,32,44,116,128,149), so
``import hyperdb.ranking_algorithm as ranking`` (reference hyperdb.py:13) keeps working.  The
arithmetic runs in hand-written HIP kernels on gfx950 through the C ABI in
``include/hyperdb_hip.h``; this file only validates arguments, moves data and restores the
reference's return types (int64 indices, float64 scores, ValueError on NaN / unknown metric).

``vectors`` may be, in every function:
  * a numpy array / list of rows  - uploaded for the call (PCIe-inclusive, like a cold start);
  * a CUDA ``torch.Tensor``       - borrowed in place, row caches built per call;
  * a :class:`ResidentVectors`    - registered once with :func:`register_vectors`; the N x d
    matrix and its 1/||v|| cache stay in HBM between calls (what ``HyperDB`` uses).

Additive surface (the reference has no batched call): :func:`rank_batch`.

There is NO CPU fallback: without the HIP library importing this module fails, and without a GPU every
call raises ``HyperDBNativeError`` instead of silently computing on the host.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _native
from ._native import GpuIndex, METRIC_IDS, NAN_MESSAGE
from .group import GpuGroup

__all__ = [
    "get_norm_vector", "dot_product", "cosine_similarity", "euclidean_metric", "manhattan_distance",
    "jaccard_similarity", "pearson_correlation", "check_and_binarize_vectors", "hamming_distance",
    "hyperDB_ranking_algorithm_sort", "rank_batch", "register_vectors", "ResidentVectors",
]

_GPU_METRICS = tuple(METRIC_IDS)      # all seven metrics of the reference's dispatch table have kernels
_ALL_METRICS = tuple(METRIC_IDS)


class ResidentVectors:
    """Handle for a matrix registered on the GPU; pass it wherever ``vectors`` is expected."""

    def __init__(self, vectors, device=None, devices=None):
        self.index = GpuGroup(vectors, devices) if devices else GpuIndex(vectors, device=device)
        arr_dtype = vectors.dtype if hasattr(vectors, "dtype") else None
        self.np_dtype = _np_dtype_of(arr_dtype, self.index)

    @property
    def shape(self):
        return (self.index.n, self.index.d)

    def __len__(self):
        return self.index.n

    def close(self):
        self.index.close()


def register_vectors(vectors, device=None, devices=None):
    """Upload ``vectors`` once; returns a handle usable in place of ``vectors``.  ``devices=[...]`` row-shards the matrix
    over several GPUs behind the same handle (single process, see hyperdb/group.py)."""
    return ResidentVectors(vectors, device=device, devices=devices)


_TORCH2NP = {torch.float16: np.float16, torch.float32: np.float32, torch.float64: np.float64}


def _np_dtype_of(dt, index):
    if isinstance(dt, torch.dtype):
        return np.dtype(_TORCH2NP.get(dt, np.float64))
    if dt is not None:
        d = np.dtype(dt)
        return d
    return np.dtype(_TORCH2NP[(index.shards[0] if isinstance(index, GpuGroup) else index).V.dtype])


def _resolve(vectors):
    """-> (GpuIndex, numpy dtype of the caller's data, owned flag)."""
    if isinstance(vectors, ResidentVectors):
        return vectors.index, vectors.np_dtype, False
    if isinstance(vectors, GpuIndex):
        return vectors, np.dtype(_TORCH2NP[vectors.V.dtype]), False
    if isinstance(vectors, GpuGroup):
        return vectors, np.dtype(_TORCH2NP[vectors.shards[0].V.dtype]), False
    if isinstance(vectors, torch.Tensor):
        ix = GpuIndex(vectors)
        return ix, np.dtype(_TORCH2NP.get(vectors.dtype, np.float64)), True
    arr = np.asarray(vectors)
    ix = GpuIndex(arr)
    return ix, arr.dtype, True


def _query_host(query_vector):
    if isinstance(query_vector, torch.Tensor):
        return query_vector.detach().cpu().numpy()
    return np.asarray(query_vector)


def _result_dtype(vdt, q):
    qdt = q.dtype if isinstance(q, np.ndarray) else np.asarray(q).dtype
    try:
        return np.result_type(vdt, qdt)
    except TypeError:
        return np.dtype(np.float64)


def _scores(vectors, query_vector, metric_id, out_dtype=None):
    ix, vdt, owned = _resolve(vectors)
    try:
        qh = _query_host(query_vector)
        s = ix.scores(qh, metric_id).cpu().numpy()
        if out_dtype is None:
            out_dtype = _result_dtype(vdt, qh)
            if not np.issubdtype(out_dtype, np.floating):
                out_dtype = np.dtype(np.float64)
        return s.astype(out_dtype)
    finally:
        if owned:
            ix.close()


# ---------------------------------------------------------------------------------------------
# per-metric functions (reference ranking_algorithm.py:8-147)
# ---------------------------------------------------------------------------------------------
def get_norm_vector(vector):
    """vector / ||vector|| along the last axis, zero norm -> divide by 1 (reference :8-21).

    Host helper: the reference calls it on ONE vector (the ANN query, hyperdb.py:207,:1361); the
    resident matrix never goes through it - its 1/||v|| cache is built on the GPU at registration.
    """
    vector = np.asarray(vector)
    norms = np.linalg.norm(vector, axis=-1, keepdims=True)
    norms = np.where(norms == 0, 1, norms)
    if np.isnan(vector).any():
        print(f"Warning: Vectors at indices {np.where(np.isnan(vector))} contain NaN values.")
    return vector / norms


def dot_product(vectors, query_vector):
    """V . q for every row (reference :24-30) on the GPU; returns an (N,) numpy array."""
    return _scores(vectors, query_vector, METRIC_IDS["dot_product"])


def cosine_similarity(vectors, query_vector):
    """cos(v, q) for every row, zero-norm rows score 0 (reference :32-42)."""
    return _scores(vectors, query_vector, METRIC_IDS["cosine_similarity"])


def euclidean_metric(vectors, query_vector, get_similarity_score=True):
    """1/(1+||v-q||) per row, or the raw distance (reference :44-52)."""
    mid = METRIC_IDS["euclidean_metric"] if get_similarity_score else _native.EUCLIDEAN_DIST
    return _scores(vectors, query_vector, mid)


def check_and_binarize_vectors(vectors):
    """x > 0 -> 1 else 0, IN PLACE unless already binary (reference :116-126).  Host helper used
    to reproduce the reference's visible side effect on the caller's query array."""
    # (the reference asks np.unique for exactly the value sets [0, 1], [0], [1]: "every element is 0 or 1", without the sort)
    if isinstance(vectors, np.ndarray) and vectors.size and bool(((vectors == 0) | (vectors == 1)).all()):
        return vectors
    if not isinstance(vectors, np.ndarray):
        unique_values = np.unique(vectors)
        if any(np.array_equal(unique_values, ok) for ok in ([0, 1], [0], [1])):
            return vectors
    if isinstance(vectors, np.ndarray) and vectors.dtype.kind == "f":
        # the reference's two masked assignments (x > 0 -> 1, x <= 0 -> 0; a NaN is neither and stays) in one masked copy
        np.copyto(vectors, vectors > 0, casting="unsafe", where=(vectors == vectors))
        return vectors
    vectors[vectors > 0] = 1
    vectors[vectors <= 0] = 0
    return vectors


def hamming_distance(vectors, query_vector):
    """d - popcount(sign(v) xor sign(q)) per row as unsigned integers (reference :128-147).

    Like the reference, a numpy ``query_vector`` is binarised in place."""
    out = _scores(vectors, query_vector, METRIC_IDS["hamming_distance"], out_dtype=np.uint64)
    if isinstance(query_vector, np.ndarray) and query_vector.flags.writeable:
        check_and_binarize_vectors(query_vector)
    return out


def manhattan_distance(vectors, query_vector):
    """1/(1+sum|v-q|) per row (reference :54-61)."""
    return _scores(vectors, query_vector, METRIC_IDS["manhattan_distance"])


def jaccard_similarity(vectors, query_vector):
    """|v AND q| / |v OR q| on the x>0 bits, NaN where both are empty (reference :63-75); float64 like numpy's
    true division.  A numpy ``query_vector`` is binarised in place, as in the reference."""
    out = _scores(vectors, query_vector, METRIC_IDS["jaccard_similarity"], out_dtype=np.float64)
    if isinstance(query_vector, np.ndarray) and query_vector.flags.writeable:
        check_and_binarize_vectors(query_vector)
    return out


def pearson_correlation(vectors, query_vector):
    """Pearson r per row, NaN when the row or the query is constant (reference :77-113); float64."""
    return _scores(vectors, np.asarray(_query_host(query_vector)).reshape(-1), METRIC_IDS["pearson_correlation"],
                   out_dtype=np.float64)


# ---------------------------------------------------------------------------------------------
# the sort entry (reference ranking_algorithm.py:149-204)
# ---------------------------------------------------------------------------------------------
def _validate_metric(metric):
    if metric not in _ALL_METRICS:
        raise ValueError(f"Unknown metric: {metric}")            # reference :166



def _apply_recency(ix, timestamps, recency_bias):
    """reference :180-183: zeros unless timestamps are given and non-empty."""
    if timestamps is not None and len(timestamps) > 0:
        ix.set_recency(timestamps, recency_bias)
        return True
    ix.set_bias(None)
    return False


def hyperDB_ranking_algorithm_sort(vectors, query_vector, top_k=5, metric='cosine_similarity', timestamps=None,
                                   recency_bias=0):
    """Top-k rows of ``vectors`` for one query (reference ranking_algorithm.py:149-204).

    Returns ``(indices int64 (k,), scores float64 (k,))`` sorted by score descending, ties by index
    ascending.  Behaviour kept from the reference: NaN anywhere -> ValueError (:150-151); unknown
    metric -> ValueError (:166); one stored row -> ``(array([0]), array([[s]]))`` and an Info line
    (:189-191); ``top_k == 0`` -> ``([], [])`` (:202); ``top_k > N`` -> all N rows; hamming binarises
    a numpy query in place (:123-124).
    """
    ix, _, owned = _resolve(vectors)
    try:
        # a CUDA tensor query stays on the device (its NaN check is the kernels' status word, which ix.topk turns into the
        # same ValueError); a host query is checked here and staged once in a pinned buffer by the index
        on_device = isinstance(query_vector, torch.Tensor) and query_vector.is_cuda and ix.n > 1
        qh = query_vector.detach() if on_device else _query_host(query_vector)
        # (one host query: a NaN anywhere makes q.q a NaN -- a dot product instead of an isnan pass and a reduction)
        if ix.has_nan or (not on_device and (math.isnan(float(np.dot(qh, qh))) if (qh.ndim == 1 and qh.dtype.char in "fd") else np.isnan(qh).any())):
            raise ValueError(NAN_MESSAGE)
        _validate_metric(metric)
        if ix.n == 0:
            raise ValueError("vectors is empty")
        had_bias = _apply_recency(ix, timestamps, recency_bias)
        try:
            if ix.n == 1:
                s = ix.scores(qh, METRIC_IDS[metric]).cpu().numpy().astype(np.float64)
                if had_bias:
                    s = s + ix._bias.cpu().numpy().astype(np.float64)
                s[np.isnan(s)] = -np.inf
                print("Info: Only one document left.")
                return np.array([0]), np.array([s])
            k = max(0, min(int(top_k), ix.n))
            if k == 0:
                return [], []
            # views of the index's pinned result record: converted once into the arrays handed back (int64 / float64 like the reference)
            idx_v, sc_v, st_v = ix.topk_views(qh.reshape(1, -1), k, METRIC_IDS[metric])
            if st_v[0] & _native.Q_NAN:
                raise ValueError(NAN_MESSAGE)
            idx, sc = idx_v[0].copy(), sc_v[0].astype(np.float64)
        finally:
            if had_bias:
                ix.set_bias(None)
        if metric in ("hamming_distance", "jaccard_similarity") and isinstance(query_vector, np.ndarray) \
                and query_vector.flags.writeable:
            check_and_binarize_vectors(query_vector)
        return idx, sc
    finally:
        if owned:
            ix.close()


def rank_batch(vectors, query_vectors, top_k=5, metric='cosine_similarity', timestamps=None, recency_bias=0):
    """Additive: top-k for a (Q, d) batch of independent queries in one pass over the matrix.

    Semantics are exactly "Q calls of hyperDB_ranking_algorithm_sort" (the reference has no batched
    entry: a 2-D query raises or returns garbage there).  Returns (int64 (Q,k), float64 (Q,k))."""
    ix, _, owned = _resolve(vectors)
    try:
        qh = _query_host(query_vectors)
        if qh.ndim == 1:
            qh = qh.reshape(1, -1)
        if ix.has_nan or np.isnan(qh).any():
            raise ValueError(NAN_MESSAGE)
        _validate_metric(metric)
        if ix.n == 0:
            raise ValueError("vectors is empty")
        k = max(0, min(int(top_k), ix.n))
        if k == 0:
            return np.zeros((qh.shape[0], 0), np.int64), np.zeros((qh.shape[0], 0), np.float64)
        had_bias = _apply_recency(ix, timestamps, recency_bias)
        try:
            idx, sc = ix.topk(qh, k, METRIC_IDS[metric])
        finally:
            if had_bias:
                ix.set_bias(None)
        return idx.astype(np.int64), sc.astype(np.float64)
    finally:
        if owned:
            ix.close()
