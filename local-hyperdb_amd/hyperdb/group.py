"""Single-process multi-GPU index: the stored N x d matrix row-sharded over several MI355X behind ONE Python object.

``HyperDB.query()`` is a single-process call (reference hyperdb/hyperdb.py:1584); ``GpuGroup`` lets it reach all GPUs of
a node without a launcher: shard p is an ordinary :class:`GpuIndex` on ``devices[p]`` with ``row_base`` = its first
global row, and one C call (``hdb_group_topk_host``, include/hyperdb_hip.h) uploads the queries to every device, runs
the per-shard pipelines concurrently (one worker thread + stream per shard inside the library), merges the per-shard
top-k records on the first device and hands back the merged packed record in pinned host memory.  The exchange unit is
the same packed record as the multi-process path (``sharded.ShardedIndex`` + one RCCL all-gather under torchrun, which
is what ``bench.py --gpus N`` measures); here it travels through a pinned, portable host buffer every device writes.

The class mirrors the part of the :class:`GpuIndex` interface the drop-in modules use, so ``HyperDB(devices=[...])``
and ``register_vectors(..., devices=[...])`` are the only new surface.  A device may be listed more than once (several
logical shards on one GPU) -- that is how the path is tested on a one-GPU box.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _native
from ._native import GpuIndex, _check, _lib, packed_bytes, Q_NAN, NAN_MESSAGE


def _bounds(n, parts):
    return [(p * n // parts, (p + 1) * n // parts) for p in range(parts)]


class GpuGroup:
    def __init__(self, vectors, devices):
        _native.require_gpu()
        devices = [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        if isinstance(vectors, torch.Tensor):
            mat = vectors
        else:
            mat = np.asarray(vectors)
            if mat.dtype not in _native._NP2HDB:
                mat = mat.astype(np.float64)
        if mat.ndim != 2:
            raise ValueError(f"vectors must be 2-D (N x d), got shape {tuple(mat.shape)}")
        n = int(mat.shape[0])
        parts = min(len(devices), max(n, 1))
        self.devices = devices[:parts]
        self.shards = []
        try:
            for (lo, hi), dev in zip(_bounds(n, parts), self.devices):
                self.shards.append(GpuIndex(mat[lo:hi], device=dev, row_base=lo))
        except Exception:
            self.close()
            raise
        self.d, self.dtype = self.shards[0].d, self.shards[0].dtype
        self.device = self.devices[0]
        self._g = ctypes.c_void_p()
        self._records = OrderedDict()
        self._make_group()

    # -- plumbing --------------------------------------------------------------------------------
    def _make_group(self):
        if self._g.value:
            _lib.hdb_group_destroy(self._g)
            self._g = ctypes.c_void_p()
        arr = (ctypes.c_void_p * len(self.shards))(*[s._h.value for s in self.shards])
        _check(_lib.hdb_group_create(ctypes.byref(self._g), arr, len(self.shards)), "hdb_group_create")

    def _rebase(self):
        base = 0
        for s in self.shards:
            s.set_row_base(base)
            base += s.n

    @property
    def n(self):
        return sum(s.n for s in self.shards)

    @property
    def bounds(self):
        out, base = [], 0
        for s in self.shards:
            out.append((base, base + s.n))
            base += s.n
        return out

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            _lib.hdb_group_destroy(self._g)
            self._g = ctypes.c_void_p()
        for s in getattr(self, "shards", []):
            s.close()
        if getattr(self, "_records", None):
            self._records.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def host_matrix(self):
        return np.concatenate([s.host_matrix() for s in self.shards], axis=0)

    @property
    def has_nan(self):
        return any(s.has_nan for s in self.shards)

    @property
    def _bias(self):
        if any(s._bias is None for s in self.shards):
            return None
        return torch.cat([s._bias.cpu() for s in self.shards])

    # -- per-row inputs: global arrays are cut at the shard bounds ---------------------------------
    def _split(self, arr):
        a = arr.cpu().numpy() if isinstance(arr, torch.Tensor) else np.asarray(arr)
        if a.shape[0] != self.n:
            raise ValueError(f"operands could not be broadcast together with shapes ({self.n},) ({a.shape[0]},)")
        return [a[lo:hi] for lo, hi in self.bounds]

    def _settle(self):
        """Per-row inputs are produced on torch's current stream of each shard's device (uploads, the decay kernel); the
        group's workers query on private non-blocking streams that do not order against it.  Called once per set_* with
        data, never per query."""
        for s in self.shards:
            if s.n:
                torch.cuda.current_stream(s.device).synchronize()

    def set_bias(self, bias):
        for s, part in zip(self.shards, [None] * len(self.shards) if bias is None else self._split(bias)):
            s.set_bias(part)
        if bias is not None:
            self._settle()

    def set_row_mask(self, mask):
        for s, part in zip(self.shards, [None] * len(self.shards) if mask is None else self._split(mask)):
            s.set_row_mask(part)
        if mask is not None:
            self._settle()

    def set_recency(self, timestamps, recency_bias, ts_max=None, valid=None):
        """The maximum of reference ranking_algorithm.py:183 is over ALL rows: computed once here, passed to every shard."""
        if timestamps is None or len(timestamps) == 0:
            self.set_bias(None)
            return
        ts = timestamps.cpu().numpy() if isinstance(timestamps, torch.Tensor) else np.asarray(timestamps, dtype=np.float64)
        if ts.shape[0] != self.n:
            raise ValueError(f"operands could not be broadcast together with shapes ({self.n},) ({ts.shape[0]},)")
        if ts_max is None:
            sel = ts if valid is None else ts[np.asarray(valid, dtype=bool)]
            ts_max = float(np.max(sel)) if sel.size else 0.0
        for s, (lo, hi) in zip(self.shards, self.bounds):
            if s.n:
                s.set_recency(ts[lo:hi], recency_bias, ts_max=ts_max)
        self._settle()

    def set_option(self, name, value):
        for s in self.shards:
            s.set_option(name, value)

    def stat(self, name):
        return self.shards[0].stat(name)

    # -- lifecycle -------------------------------------------------------------------------------
    def append(self, rows):
        """New rows go behind the last shard's rows, so global row ids keep following insertion order."""
        self.shards[-1].append(rows)

    def compact(self, keep_rows):
        keep_rows = np.asarray(keep_rows, dtype=np.int64)
        for s, (lo, hi) in zip(self.shards, self.bounds):
            mine = keep_rows[(keep_rows >= lo) & (keep_rows < hi)] - lo
            if mine.size != s.n:
                s.compact(mine)
        self._rebase()

    # -- queries ---------------------------------------------------------------------------------
    def scores(self, q, metric_id):
        return torch.cat([s.scores(q, metric_id).to(self.device) for s in self.shards if s.n])

    def topk_views(self, Q, k, metric_id):
        qdt = np.float64 if self.dtype == _native.HDB_F64 else np.float32
        qh = Q.detach().cpu().numpy() if isinstance(Q, torch.Tensor) else np.asarray(Q)
        qh = np.ascontiguousarray(qh.reshape(1, -1) if qh.ndim == 1 else qh, dtype=qdt)
        if qh.ndim != 2 or qh.shape[1] != self.d:
            raise ValueError(f"shapes ({self.n},{self.d}) and {tuple(qh.shape)} not aligned")
        nq, k = int(qh.shape[0]), int(k)
        slot = self._records.get((nq, k))
        if slot is None:
            host = np.empty(packed_bytes(nq, k), dtype=np.uint8)
            slot = (host, host[:nq * k * 8].view(np.int64).reshape(nq, k), host[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k),
                    host[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32))
            self._records[(nq, k)] = slot
            while len(self._records) > _native.HOST_RECORD_SLOTS:
                self._records.popitem(last=False)
        else:
            self._records.move_to_end((nq, k))
        _check(_lib.hdb_group_topk_host(self._g, qh.ctypes.data_as(ctypes.c_void_p), nq, k, int(metric_id),
                                        slot[0].ctypes.data_as(ctypes.c_void_p)), "hdb_group_topk_host")
        return slot[1], slot[2], slot[3]

    def topk(self, Q, k, metric_id):
        idx, sc, st = self.topk_views(Q, k, metric_id)
        if (st & Q_NAN).any():
            raise ValueError(NAN_MESSAGE)
        return idx.copy(), sc.copy()
