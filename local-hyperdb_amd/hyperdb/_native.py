"""ctypes binding of libhyperdb_hip.so (include/hyperdb_hip.h) + the resident-matrix handle.

This is the only place where Python touches the C ABI.  PyTorch-ROCm is used for device memory,
streams and H2D/D2H copies only; every score and every top-k comes out of the hand-written HIP
kernels.  There is no CPU fallback: if the library is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch  # must be imported before the .so so that ONE libamdhip64 (torch's) serves both

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HYPERDB_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libhyperdb_hip.so")

HDB_F16, HDB_F32, HDB_F64 = 0, 1, 2
METRIC_IDS = {
    "dot_product": 0,
    "cosine_similarity": 1,
    "euclidean_metric": 2,
    "hamming_distance": 3,
    "manhattan_distance": 4,
    "jaccard_similarity": 5,
    "pearson_correlation": 6,
}
EUCLIDEAN_DIST = 7
Q_UNDERFLOW, Q_OVERFLOW, Q_NAN = 1, 2, 4
HDB_MAX_K = 2048
MERGE_DEVICE_CAP = 8192      # hdb_merge_topk / hdb_merge_topk_packed rank parts*k entries per query in LDS; beyond: hdb_merge_topk_host

NAN_MESSAGE = "Vectors and query_vector should not contain NaN values."   # reference ranking_algorithm.py:151

EXPORTS = (
    "hdb_version", "hdb_last_error", "hdb_index_create", "hdb_index_update", "hdb_index_rebase", "hdb_index_extend",
    "hdb_index_gather", "hdb_index_set_row_base", "hdb_index_destroy",
    "hdb_group_create", "hdb_group_topk_host", "hdb_group_destroy",
    "hdb_index_has_nan", "hdb_index_set_bias", "hdb_index_set_row_mask", "hdb_scores", "hdb_topk",
    "hdb_topk_exact", "hdb_merge_topk", "hdb_set_option", "hdb_get_stat", "hdb_recency_bias", "hdb_recency_bias_twice",
    "hdb_packed_bytes", "hdb_merge_topk_packed", "hdb_merge_topk_host", "hdb_host_exchange_merge", "hdb_topk_host",
)


class HyperDBNativeError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP extension has not been built. "
            "Run `python __graft_entry__.py` (or local-hyperdb_amd/csrc/build.sh). "
            "There is deliberately no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, cp = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_char_p
    lib.hdb_version.restype = ctypes.c_int
    lib.hdb_last_error.restype = cp
    lib.hdb_index_create.argtypes = [ctypes.POINTER(vp), vp, i64, i32, ctypes.c_int, ctypes.c_int, i64, vp]
    lib.hdb_index_update.argtypes = [vp, vp, i64, vp]
    lib.hdb_index_rebase.argtypes = [vp, vp]
    lib.hdb_index_extend.argtypes = [vp, i64, vp]
    lib.hdb_index_gather.argtypes = [vp, vp, i64, vp, vp]
    lib.hdb_index_set_row_base.argtypes = [vp, i64]
    lib.hdb_group_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), i32]
    lib.hdb_group_topk_host.argtypes = [vp, vp, i32, i32, ctypes.c_int, vp]
    lib.hdb_group_destroy.argtypes = [vp]
    lib.hdb_group_destroy.restype = None
    lib.hdb_index_destroy.argtypes = [vp]
    lib.hdb_index_destroy.restype = None
    lib.hdb_index_has_nan.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    lib.hdb_index_set_bias.argtypes = [vp, vp]
    lib.hdb_index_set_row_mask.argtypes = [vp, vp]
    lib.hdb_scores.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    lib.hdb_topk.argtypes = [vp, vp, i32, i32, ctypes.c_int, vp, vp, vp, vp]
    lib.hdb_topk_exact.argtypes = [vp, vp, i32, i32, ctypes.c_int, vp, vp, vp, vp]
    lib.hdb_merge_topk.argtypes = [vp, vp, i32, i32, i32, vp, vp, ctypes.c_int, vp]
    lib.hdb_set_option.argtypes = [vp, cp, i64]
    lib.hdb_get_stat.argtypes = [vp, cp, ctypes.POINTER(i64)]
    lib.hdb_recency_bias.argtypes = [vp, i64, ctypes.c_double, ctypes.c_double, vp, ctypes.c_int, vp]
    lib.hdb_recency_bias_twice.argtypes = [vp, vp, i64, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, ctypes.c_int, vp]
    lib.hdb_topk_host.argtypes = [vp, vp, i32, i32, ctypes.c_int, vp, vp]
    lib.hdb_packed_bytes.argtypes = [i32, i32]
    lib.hdb_merge_topk_packed.argtypes = [vp, i32, i32, i32, vp, vp, vp, ctypes.c_int, vp]
    lib.hdb_merge_topk_host.argtypes = [vp, i32, i32, i32, vp]
    lib.hdb_host_exchange_merge.argtypes = [vp, i64, i32, i32, ctypes.c_uint64, vp, i32, i32, vp, ctypes.c_double]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("hdb_last_error", "hdb_index_destroy", "hdb_group_destroy", "hdb_packed_bytes"):
            fn.restype = ctypes.c_int
    lib.hdb_packed_bytes.restype = i64
    return lib


_lib = _load()


def lib():
    return _lib


def _check(rc, what):
    if rc == 0:
        return
    msg = _lib.hdb_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    if rc == -3:
        raise NotImplementedError(f"{what}: {msg}")
    raise HyperDBNativeError(f"{what}: {msg} (status {rc})")


def require_gpu():
    if not torch.cuda.is_available():
        raise HyperDBNativeError(
            "no MI355X visible: the HyperDB ranking path runs only on the GPU (no CPU fallback).")


_NP2HDB = {np.dtype(np.float16): HDB_F16, np.dtype(np.float32): HDB_F32, np.dtype(np.float64): HDB_F64}
_TORCH2HDB = {torch.float16: HDB_F16, torch.float32: HDB_F32, torch.float64: HDB_F64}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the current stream's handle without building a Stream object (0.3 vs 1.5 us)


def _stream_int(device):
    if _raw_stream is not None:
        return _raw_stream(device.index or 0)
    return torch.cuda.current_stream(device).cuda_stream


def _stream_ptr(device):
    return ctypes.c_void_p(_stream_int(device))


def to_device_matrix(vectors, device):
    """numpy / list / torch -> C-contiguous 2-D torch tensor on `device` in a supported dtype.

    Dtype policy mirrors HyperDB.fp_precision (hyperdb.py:65-66): float16/32/64 stay as they are,
    anything else numeric (ints, bools) is widened to float64 like numpy would promote it."""
    if isinstance(vectors, torch.Tensor):
        t = vectors
        if t.dtype not in _TORCH2HDB:
            t = t.to(torch.float64)
        return t.to(device).contiguous()
    arr = np.asarray(vectors)
    if arr.dtype not in _NP2HDB:
        if not (np.issubdtype(arr.dtype, np.number) or arr.dtype == np.bool_):
            raise ValueError("vectors must be numeric")
        arr = arr.astype(np.float64)
    arr = np.ascontiguousarray(arr)
    return torch.from_numpy(arr).to(device, non_blocking=False)


class GpuIndex:
    """A resident N x d matrix on one MI355X plus its per-row caches (hdb_index handle).

    Replaces the per-query work of reference ranking_algorithm.py:150 (NaN scan), :153 (copy) and
    :37 (re-normalising every row): those happen once here, at registration.
    """

    def __init__(self, vectors, device=None, row_base=0):
        require_gpu()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        t = to_device_matrix(vectors, self.device)
        if t.dim() != 2:
            # reference: non-2-D vectors make the metric functions raise (tests/test_ranking_algorithm.py:107-114)
            raise ValueError(f"vectors must be 2-D (N x d), got shape {tuple(t.shape)}")
        self.V = t
        self.n, self.d = int(t.shape[0]), int(t.shape[1])
        if self.d == 0:
            raise ValueError("vectors must have d > 0")
        self.dtype = _TORCH2HDB[t.dtype]
        self.row_base = int(row_base)
        self._h = ctypes.c_void_p()
        self._bias = None
        self._mask = None
        self._nan = None
        self._buf = None
        self._qstage = OrderedDict()           # (nq, dtype) -> (pinned host tensor, its numpy view, device tensor or None, address the kernels read)
        self._host_records = OrderedDict()     # (nq, k) -> pinned record + views, owned by THIS index (small LRU)
        with torch.cuda.device(self.device):
            _check(_lib.hdb_index_create(ctypes.byref(self._h), ctypes.c_void_p(t.data_ptr()), self.n, self.d,
                                         self.dtype, self.device.index or 0, self.row_base,
                                         _stream_ptr(self.device)), "hdb_index_create")

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.hdb_index_destroy(self._h)
            self._h = ctypes.c_void_p()
        if getattr(self, "_host_records", None):
            self._host_records.clear()                 # releases the pinned host memory
        if getattr(self, "_qstage", None):
            self._qstage.clear()
        self._buf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def append(self, rows):
        """Append rows behind the stored ones (HyperDB.add): amortised O(new rows).  The device allocation grows
        by doubling; the library re-points (hdb_index_rebase) and extends its row caches (hdb_index_extend)."""
        t = to_device_matrix(rows, self.device)
        if t.dim() == 1:
            t = t.reshape(1, -1)
        if t.dim() != 2 or int(t.shape[1]) != self.d:
            raise ValueError(f"append: rows must have {self.d} columns")
        if t.dtype != self.V.dtype:
            t = t.to(self.V.dtype)
        m = int(t.shape[0])
        if m == 0:
            return
        buf = self._buf
        if buf is None or buf.data_ptr() != self.V.data_ptr():
            buf = self.V                                    # first append: the registered tensor is the buffer
        cap = int(buf.shape[0])
        if self.n + m > cap:
            new_cap = max(self.n + m, 2 * cap, 1024)
            nbuf = torch.empty((new_cap, self.d), dtype=self.V.dtype, device=self.device)
            nbuf[:self.n].copy_(self.V)
            buf = nbuf
            _check(_lib.hdb_index_rebase(self._h, ctypes.c_void_p(buf.data_ptr())), "hdb_index_rebase")
        buf[self.n:self.n + m].copy_(t)
        self._buf = buf
        self.n += m
        self.V = buf[:self.n]
        self._bias = self._mask = None
        self._nan = None
        _check(_lib.hdb_index_extend(self._h, self.n, _stream_ptr(self.device)), "hdb_index_extend")

    def update(self, vectors):
        """Point the handle at a new matrix (after add/remove) and rebuild the row caches."""
        t = to_device_matrix(vectors, self.device)
        if t.dim() != 2 or int(t.shape[1]) != self.d or _TORCH2HDB[t.dtype] != self.dtype:
            raise ValueError("update: matrix must keep d and dtype")
        self.V, self.n = t, int(t.shape[0])
        self._bias = self._mask = self._nan = None
        self._buf = None                                    # the old capacity buffer is released with the old matrix
        _check(_lib.hdb_index_update(self._h, ctypes.c_void_p(t.data_ptr()), self.n, _stream_ptr(self.device)),
               "hdb_index_update")

    def set_row_base(self, row_base):
        self.row_base = int(row_base)
        _check(_lib.hdb_index_set_row_base(self._h, self.row_base), "hdb_index_set_row_base")

    def host_matrix(self):
        """The stored rows as a host array (one D2H copy; the resident copy stays the only one kept)."""
        return self.V.cpu().numpy()

    def compact(self, keep_rows):
        """Keep only the rows `keep_rows` (ascending local row ids): device-side gather into a fresh allocation, the
        row caches travel with their rows (hdb_index_gather) -- the matrix part of HyperDB.remove_document
        (hyperdb.py:691-766) without a host round trip or a cache rebuild."""
        rows = torch.from_numpy(np.ascontiguousarray(np.asarray(keep_rows, dtype=np.int64))).to(self.device)
        m = int(rows.numel())
        out = torch.empty((max(m, 1), self.d), dtype=self.V.dtype, device=self.device)
        _check(_lib.hdb_index_gather(self._h, ctypes.c_void_p(rows.data_ptr()), m, ctypes.c_void_p(out.data_ptr()),
                                     _stream_ptr(self.device)), "hdb_index_gather")
        self._buf = out
        self.V, self.n = out[:m], m
        self._bias = self._mask = self._nan = None

    @property
    def has_nan(self):
        if self._nan is None:
            flag = ctypes.c_int(0)
            _check(_lib.hdb_index_has_nan(self._h, ctypes.byref(flag)), "hdb_index_has_nan")
            self._nan = bool(flag.value)
        return self._nan

    # -- per-row additive term / row subset ----------------------------------------------------
    def set_bias(self, bias):
        """bias: None, or N floats (numpy / torch) added to every score before top-k."""
        if bias is None:
            if self._bias is not None:                     # (nothing set: nothing to clear -- saves a C call per query)
                self._bias = None
                _check(_lib.hdb_index_set_bias(self._h, None), "hdb_index_set_bias")
            return
        if isinstance(bias, torch.Tensor):
            b = bias.to(self.device, torch.float32).contiguous()
        else:
            b = torch.from_numpy(np.ascontiguousarray(np.asarray(bias, dtype=np.float32))).to(self.device)
        if b.numel() != self.n:
            raise ValueError(f"bias must have {self.n} entries, got {b.numel()}")
        self._bias = b
        _check(_lib.hdb_index_set_bias(self._h, ctypes.c_void_p(b.data_ptr())), "hdb_index_set_bias")

    def set_recency(self, timestamps, recency_bias, ts_max=None, valid=None):
        """bias = recency_bias * exp(ts - max ts) (reference ranking_algorithm.py:183), computed on the
        device in float64 from float64 timestamps, stored as float32.

        ts_max: the maximum to normalise by when it is not the maximum of `timestamps` itself -- the GLOBAL maximum of
        a row-sharded matrix (ShardedIndex.set_recency all-reduces it), or the maximum over the rows that take part
        when a row mask is set.  valid: optional boolean array; the maximum is then taken over those rows only (the
        reference ranks the FILTERED rows, hyperdb.py:1556, so rows a filter removed must not move the maximum)."""
        if timestamps is None or len(timestamps) == 0:
            self.set_bias(None)
            return
        if isinstance(timestamps, torch.Tensor):
            ts = timestamps.to(self.device, torch.float64).contiguous()
            if ts_max is None:
                sel = ts if valid is None else ts[torch.as_tensor(np.asarray(valid, dtype=bool), device=self.device)]
                ts_max = float(sel.max().item()) if sel.numel() else 0.0
        else:
            ts_h = np.ascontiguousarray(np.asarray(timestamps, dtype=np.float64))
            if ts_max is None:
                sel = ts_h if valid is None else ts_h[np.asarray(valid, dtype=bool)]
                ts_max = float(np.max(sel)) if sel.size else 0.0
            ts = torch.from_numpy(ts_h).to(self.device)
        if ts.numel() != self.n:
            raise ValueError(f"operands could not be broadcast together with shapes ({self.n},) ({ts.numel()},)")
        out = torch.empty(self.n, dtype=torch.float32, device=self.device)
        _check(_lib.hdb_recency_bias(ctypes.c_void_p(ts.data_ptr()), self.n, float(recency_bias), float(ts_max),
                                     ctypes.c_void_p(out.data_ptr()), self.device.index or 0,
                                     _stream_ptr(self.device)), "hdb_recency_bias")
        self._bias = out
        _check(_lib.hdb_index_set_bias(self._h, ctypes.c_void_p(out.data_ptr())), "hdb_index_set_bias")

    def recency_twice(self, ts_dev, mask_dev, recency_bias, ts_max, ts_min):
        """Both decays of a HyperDB.query() call (reference hyperdb.py:1344, then ranking_algorithm.py:183) for the rows
        `mask_dev` keeps, from a resident float64 timestamp column: -> float32 CUDA tensor (hdb_recency_bias_twice).
        ts_max / ts_min: newest / oldest timestamp among the kept rows."""
        out = torch.empty(self.n, dtype=torch.float32, device=self.device)
        _check(_lib.hdb_recency_bias_twice(ctypes.c_void_p(ts_dev.data_ptr()),
                                           ctypes.c_void_p(mask_dev.data_ptr()) if mask_dev is not None else None, self.n,
                                           float(recency_bias), float(ts_max), float(ts_min), ctypes.c_void_p(out.data_ptr()),
                                           self.device.index or 0, _stream_ptr(self.device)), "hdb_recency_bias_twice")
        return out

    def set_row_mask(self, mask):
        if mask is None:
            if self._mask is not None:
                self._mask = None
                _check(_lib.hdb_index_set_row_mask(self._h, None), "hdb_index_set_row_mask")
            return
        if isinstance(mask, torch.Tensor) and mask.dtype == torch.uint8 and mask.device == self.device and mask.is_contiguous():
            m = mask                                       # a resident 0/1 mask (HyperDB caches them per filter): borrowed as it is
        elif isinstance(mask, torch.Tensor):
            m = (mask != 0).to(self.device, torch.uint8).contiguous()
        else:
            m = torch.from_numpy(np.ascontiguousarray((np.asarray(mask) != 0).astype(np.uint8))).to(self.device)
        if m.numel() != self.n:
            raise ValueError("mask must have one entry per row")
        self._mask = m
        _check(_lib.hdb_index_set_row_mask(self._h, ctypes.c_void_p(m.data_ptr())), "hdb_index_set_row_mask")

    # -- options / stats -----------------------------------------------------------------------
    def set_option(self, name, value):
        _check(_lib.hdb_set_option(self._h, name.encode(), int(value)), "hdb_set_option")

    def stat(self, name):
        v = ctypes.c_int64(0)
        _check(_lib.hdb_get_stat(self._h, name.encode(), ctypes.byref(v)), "hdb_get_stat")
        return int(v.value)

    # -- queries -------------------------------------------------------------------------------
    def _stage_host_query(self, q):
        """A host query batch for a SYNCHRONOUS call: converted once into a cached pinned buffer.  Up to 4 queries are read by
        the kernels straight from that buffer (pinned host memory is device-visible: 1.5 KB per query over PCIe costs less
        than a copy on the stream ahead of the launch); larger batches go through one asynchronous copy into a cached device
        tensor.  (The reference hands numpy queries to numpy: hyperdb.py:1556.)"""
        npdt = np.float64 if self.dtype == HDB_F64 else np.float32
        a = np.asarray(q)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        if a.ndim != 2 or a.shape[1] != self.d:
            raise ValueError(f"shapes ({self.n},{self.d}) and {tuple(np.asarray(q).shape)} not aligned")
        nq = int(a.shape[0])
        slot = self._stage_slot(nq, npdt)
        np.copyto(slot[1], a, casting="unsafe")
        if slot[2] is None:
            return slot[0]
        slot[2].copy_(slot[0], non_blocking=True)
        return slot[2]

    def _stage_slot(self, nq, npdt):
        key = (nq, npdt)
        slot = self._qstage.get(key)
        if slot is None:
            pin = torch.empty((nq, self.d), dtype=torch.float64 if npdt is np.float64 else torch.float32, pin_memory=True)
            dev = None if nq <= 4 else torch.empty((nq, self.d), dtype=pin.dtype, device=self.device)
            slot = self._qstage[key] = (pin, pin.numpy(), dev, (pin if dev is None else dev).data_ptr())
            while len(self._qstage) > HOST_RECORD_SLOTS:
                self._qstage.popitem(last=False)
        else:
            self._qstage.move_to_end(key)
        return slot

    def _query_tensor(self, q, batched, staged=False):
        qdt = torch.float64 if self.dtype == HDB_F64 else torch.float32
        if staged and batched and not isinstance(q, torch.Tensor):
            return self._stage_host_query(q)
        if isinstance(q, torch.Tensor):
            if batched and q.dtype == qdt and q.dim() == 2 and q.shape[1] == self.d and q.is_contiguous() and \
                    (q.device == self.device or (staged and q.device.type == "cpu" and q.is_pinned())):
                return q                                 # the per-query fast path: nothing to convert (or: already staged)
            t = q.to(self.device, qdt)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(q, dtype=np.float64))).to(self.device, qdt)
        if batched:
            if t.dim() == 1:
                t = t.reshape(1, -1)
            if t.dim() != 2 or t.shape[1] != self.d:
                raise ValueError(f"shapes ({self.n},{self.d}) and {tuple(t.shape)} not aligned")
        else:
            t = t.reshape(-1)
            if t.numel() != self.d:
                raise ValueError(f"shapes ({self.n},{self.d}) and ({t.numel()},) not aligned: "
                                 f"{self.d} (dim 1) != {t.numel()} (dim 0)")
        return t.contiguous()

    def scores(self, q, metric_id):
        """All N scores of one query as a float32 CUDA tensor (no bias, no mask)."""
        qt = self._query_tensor(q, batched=False)
        out = torch.empty(self.n, dtype=torch.float32, device=self.device)
        _check(_lib.hdb_scores(self._h, ctypes.c_void_p(qt.data_ptr()), int(metric_id),
                               ctypes.c_void_p(out.data_ptr()), _stream_ptr(self.device)), "hdb_scores")
        return out

    def topk_device(self, Q, k, metric_id, exact=False):
        """Enqueue the top-k of a (nq, d) query batch; returns CUDA tensors (idx, score, status)."""
        qt = self._query_tensor(Q, batched=True)
        nq = int(qt.shape[0])
        idx = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        sc = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        st = torch.empty((nq,), dtype=torch.int32, device=self.device)
        fn = _lib.hdb_topk_exact if exact else _lib.hdb_topk
        _check(fn(self._h, ctypes.c_void_p(qt.data_ptr()), nq, int(k), int(metric_id),
                  ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
                  ctypes.c_void_p(st.data_ptr()), _stream_ptr(self.device)), "hdb_topk")
        return idx, sc, st

    def topk_packed(self, Q, k, metric_id, record, exact=False):
        """Enqueue the top-k of a batch straight into a packed exchange record (uint8 CUDA tensor of
        hdb_packed_bytes(nq, k) bytes): [idx int64][score f32][status i32]."""
        qt = self._query_tensor(Q, batched=True)
        nq = int(qt.shape[0])
        base = record.data_ptr()
        fn = _lib.hdb_topk_exact if exact else _lib.hdb_topk
        _check(fn(self._h, ctypes.c_void_p(qt.data_ptr()), nq, int(k), int(metric_id), ctypes.c_void_p(base),
                  ctypes.c_void_p(base + nq * k * 8), ctypes.c_void_p(base + nq * k * 12),
                  _stream_ptr(self.device)), "hdb_topk")

    def topk_views(self, Q, k, metric_id):
        """One C call (hdb_topk_host): kernels + D2H of the packed record into a cached pinned buffer + sync + the
        rare exact re-run.  Returns numpy VIEWS (idx int64 [nq,k], score float32 [nq,k], status int32 [nq]) that are
        overwritten by the next call of THIS index with the same (nq, k); the record belongs to the index (two
        indices, devices or shard groups never share one) and at most HOST_RECORD_SLOTS shapes are kept."""
        if type(Q) is np.ndarray and Q.ndim <= 2 and Q.shape[-1] == self.d and Q.size <= 4 * self.d and Q.size:
            # the per-query path of the drop-in entry points (a host query of 1-4 rows): straight into the cached pinned staging
            # buffer the kernels read, no tensor objects, no per-call ctypes wrappers (tools/time_host_path.py)
            nq = Q.size // self.d
            qs = self._stage_slot(nq, np.float64 if self.dtype == HDB_F64 else np.float32)
            np.copyto(qs[1], Q if Q.ndim == 2 else Q.reshape(1, -1), casting="unsafe")
            qptr = qs[3]
        else:
            qt = self._query_tensor(Q, batched=True, staged=True)
            nq, qptr = int(qt.shape[0]), qt.data_ptr()
        k = int(k)
        slot = self._host_records.get((nq, k))
        if slot is None:                                 # pinned record + its numpy views, built once per (nq, k)
            host = torch.empty(packed_bytes(nq, k), dtype=torch.uint8, pin_memory=True)
            h = host.numpy()
            slot = (host, host.data_ptr(), h[:nq * k * 8].view(np.int64).reshape(nq, k),
                    h[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k), h[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32), h)
            self._host_records[(nq, k)] = slot
            while len(self._host_records) > HOST_RECORD_SLOTS:
                self._host_records.popitem(last=False)
        elif len(self._host_records) > 1:
            self._host_records.move_to_end((nq, k))
        rc = _lib.hdb_topk_host(self._h, qptr, nq, k, int(metric_id), slot[1], _stream_int(self.device))
        if rc:
            _check(rc, "hdb_topk_host")
        return slot[2], slot[3], slot[4]

    def topk_record_host(self, Q, k, metric_id):
        """topk_views, but returns the whole packed record as ONE uint8 numpy view (what a shard hands to the exchange)."""
        qt = self._query_tensor(Q, batched=True, staged=True)
        nq = int(qt.shape[0])
        self.topk_views(qt, k, metric_id)
        return self._host_records[(nq, int(k))][5]

    def topk(self, Q, k, metric_id):
        """Top-k of a query batch on the host: (int64 [nq,k], float32 [nq,k]) (copies)."""
        idx, sc, st = self.topk_views(Q, k, metric_id)
        if (st[0] & Q_NAN) if len(st) == 1 else (st & Q_NAN).any():      # (one query: a scalar test instead of two array operations)
            raise ValueError(NAN_MESSAGE)
        return idx.copy(), sc.copy()


def packed_bytes(nq, k):
    return int(_lib.hdb_packed_bytes(int(nq), int(k)))


HOST_RECORD_SLOTS = 8      # pinned result records kept per index / per staging owner (LRU)


def record_to_host(record, nq, k, cache):
    """ONE D2H copy of a packed record (pinned staging buffer from `cache`, an OrderedDict owned by the caller and
    keyed by size) -> numpy views (idx int64 [nq,k], score float32 [nq,k], status int32 [nq]).  Views alias the
    staging buffer: callers copy what they keep."""
    nb = record.numel()
    host = cache.get(nb)
    if host is None:
        host = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
        cache[nb] = host
        while len(cache) > HOST_RECORD_SLOTS:
            cache.popitem(last=False)
    else:
        cache.move_to_end(nb)
    host.copy_(record, non_blocking=True)
    torch.cuda.current_stream(record.device).synchronize()
    h = host.numpy()
    idx = h[:nq * k * 8].view(np.int64).reshape(nq, k)
    sc = h[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k)
    st = h[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32)
    return idx, sc, st


def record_views(h, nq, k):
    """numpy views (idx int64 [nq,k], score float32 [nq,k], status int32 [nq]) of a packed record held in a uint8 array."""
    return (h[:nq * k * 8].view(np.int64).reshape(nq, k), h[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k),
            h[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32))


def merge_topk_host(records, parts, nq, k, out):
    """hdb_merge_topk_host: `records` = parts packed records back to back in HOST memory (uint8 numpy), `out` = uint8
    numpy of packed_bytes(nq, k).  Pure host code (no GPU call).  Returns the views of `out`."""
    nb = packed_bytes(nq, k)
    if records.dtype != np.uint8 or records.size < parts * nb or out.dtype != np.uint8 or out.size < nb:
        raise ValueError("merge_topk_host: records / out must be uint8 arrays of parts * packed_bytes / packed_bytes")
    if not (records.flags["C_CONTIGUOUS"] and out.flags["C_CONTIGUOUS"]):
        raise ValueError("merge_topk_host: contiguous arrays only")
    _check(_lib.hdb_merge_topk_host(ctypes.c_void_p(records.ctypes.data), int(parts), int(nq), int(k),
                                    ctypes.c_void_p(out.ctypes.data)), "hdb_merge_topk_host")
    return record_views(out, nq, k)


def host_exchange_merge(shm_addr, stride, world, rank, seq, record_addr, nq, k, out_addr, timeout_s):
    """hdb_host_exchange_merge on raw addresses (the caller caches them: this sits on the per-query path)."""
    _check(_lib.hdb_host_exchange_merge(shm_addr, stride, world, rank, seq, record_addr, nq, k, out_addr, timeout_s),
           "hdb_host_exchange_merge")


def merge_topk_packed_into(gathered, parts, nq, k, out_record):
    """Merge `parts` gathered records into another packed record (same layout)."""
    base = out_record.data_ptr()
    device = gathered.device
    _check(_lib.hdb_merge_topk_packed(ctypes.c_void_p(gathered.data_ptr()), int(parts), int(nq), int(k),
                                      ctypes.c_void_p(base), ctypes.c_void_p(base + nq * k * 8),
                                      ctypes.c_void_p(base + nq * k * 12), device.index or 0, _stream_ptr(device)),
           "hdb_merge_topk_packed")


def merge_topk_packed(gathered, parts, nq, k):
    """Merge `parts` gathered exchange records (one uint8 CUDA tensor) -> (idx, score, status) CUDA tensors."""
    device = gathered.device
    idx = torch.empty((nq, k), dtype=torch.int64, device=device)
    sc = torch.empty((nq, k), dtype=torch.float32, device=device)
    st = torch.empty((nq,), dtype=torch.int32, device=device)
    _check(_lib.hdb_merge_topk_packed(ctypes.c_void_p(gathered.data_ptr()), int(parts), int(nq), int(k),
                                      ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
                                      ctypes.c_void_p(st.data_ptr()), device.index or 0, _stream_ptr(device)),
           "hdb_merge_topk_packed")
    return idx, sc, st


def merge_topk(idx_parts, score_parts, k):
    """Merge all-gathered per-shard lists: idx_parts [parts, nq, k] int64, score_parts float32."""
    parts, nq, kk = (int(x) for x in idx_parts.shape)
    assert kk == k and score_parts.shape == idx_parts.shape
    device = idx_parts.device
    idx = torch.empty((nq, k), dtype=torch.int64, device=device)
    sc = torch.empty((nq, k), dtype=torch.float32, device=device)
    ip, sp = idx_parts.contiguous(), score_parts.contiguous()
    _check(_lib.hdb_merge_topk(ctypes.c_void_p(ip.data_ptr()), ctypes.c_void_p(sp.data_ptr()), parts, nq, k,
                               ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
                               device.index or 0, _stream_ptr(device)), "hdb_merge_topk")
    return idx, sc
