"""Row-sharded index: one process per GPU, one RCCL all-gather per query batch.

The stored N x d matrix is split into contiguous row shards (rank r holds rows [lo_r, hi_r)).  Scores
of a row depend only on that row, the query and the row's bias, and top-k is an associative merge,
so the only exchange step is the gather of each shard's k best:

    local scan + top-k  ->  packed record [idx int64 | score f32 | status i32]   (device, no host hop)
    torch.distributed.all_gather_into_tensor (backend "nccl" == RCCL over xGMI; payload = nq*k*12 B per rank)
    hdb_merge_topk_packed on every rank   ->  identical global top-k everywhere

Ordering is the build's total order (score descending, global row ascending), so the result does not
depend on the number of shards.  Queries whose sampled threshold failed on ANY shard (status != 0 in
the gathered records, seen identically by all ranks) are re-run through the exact path collectively.

The compute engine is injectable so that the sharding / exchange / merge bookkeeping can be exercised
with the gloo backend on CPU in tests; the product engine is the HIP one and nothing else.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


class HipEngine:
    """Adapter: GpuIndex + the packed-record merge kernel."""

    def __init__(self, index):
        self.index = index
        self.device = index.device
        self._records = {}
        self._host = OrderedDict()        # pinned staging records of THIS engine, bounded (see _native.HOST_RECORD_SLOTS)

    def packed_bytes(self, nq, k):
        from . import _native
        return _native.packed_bytes(nq, k)

    def new_record(self, nbytes, slot=0):
        """Device scratch for one exchange record; cached per (slot, size) so a query allocates nothing."""
        key = (slot, int(nbytes))
        buf = self._records.get(key)
        if buf is None:
            buf = self._records[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return buf

    def topk_packed(self, Q, k, metric_id, record, exact=False):
        self.index.topk_packed(Q, k, metric_id, record, exact=exact)

    def merge_packed_into(self, gathered, parts, nq, k, out_record):
        from . import _native
        _native.merge_topk_packed_into(gathered, parts, nq, k, out_record)

    def record_to_host(self, record, nq, k):
        from . import _native
        return _native.record_to_host(record, nq, k, self._host)

    def set_recency(self, timestamps, recency_bias, ts_max):
        self.index.set_recency(timestamps, recency_bias, ts_max=ts_max)

    def local_ts_max(self, timestamps):
        if timestamps is None or len(timestamps) == 0:
            return float("-inf")
        if isinstance(timestamps, torch.Tensor):
            return float(timestamps.max().item())
        return float(np.max(np.asarray(timestamps, dtype=np.float64)))

    def merge_packed_to_host(self, gathered, parts, nq, k):
        """Merge straight into a pinned host record (the merge kernel stores over PCIe itself: no D2H copy)."""
        from . import _native
        nb = _native.packed_bytes(nq, k)
        host = self._host.get(("merge", nb))
        if host is None:
            host = self._host[("merge", nb)] = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
            while len(self._host) > _native.HOST_RECORD_SLOTS:
                self._host.popitem(last=False)
        _native.merge_topk_packed_into(gathered, parts, nq, k, host)
        torch.cuda.current_stream(self.device).synchronize()
        h = host.numpy()
        return (h[:nq * k * 8].view(np.int64).reshape(nq, k), h[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k),
                h[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32))

    def topk_host(self, Q, k, metric_id):
        return self.index.topk_views(Q, k, metric_id)

    def select_queries(self, Q, which):
        return Q.index_select(0, torch.as_tensor(which, device=Q.device))


def shard_bounds(n_total, world, granule=1):
    """Contiguous, granule-aligned row ranges: rank r -> [lo, hi)."""
    nblk = (n_total + granule - 1) // granule
    out = []
    for r in range(world):
        lo = (r * nblk // world) * granule
        hi = min(((r + 1) * nblk // world) * granule, n_total)
        out.append((lo, hi))
    return out


class ShardedIndex:
    def __init__(self, local, n_total=None, group=None, engine=None, force_exchange=False):
        self.engine = engine if engine is not None else HipEngine(local)
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None and dist is not None) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n_total = n_total
        self.force_exchange = force_exchange and group is not None     # run gather+merge even with one rank (tests)

    # -- helpers -------------------------------------------------------------------------------
    def _gather_merge(self, Q, k, metric_id, exact):
        """-> host views (idx, score, status) of the merged global result."""
        eng = self.engine
        nq = int(Q.shape[0])
        nb = eng.packed_bytes(nq, k)
        rec = eng.new_record(nb, 0)
        eng.topk_packed(Q, k, metric_id, rec, exact=exact)
        if self.world == 1 and not self.force_exchange:
            return eng.record_to_host(rec, nq, k)               # already the global answer
        gathered = eng.new_record(nb * self.world, 1)
        dist.all_gather_into_tensor(gathered, rec, group=self.group)
        if hasattr(eng, "merge_packed_to_host"):
            return eng.merge_packed_to_host(gathered, self.world, nq, k)
        merged = eng.new_record(nb, 2)
        eng.merge_packed_into(gathered, self.world, nq, k, merged)
        return eng.record_to_host(merged, nq, k)

    # -- public --------------------------------------------------------------------------------
    def set_recency(self, timestamps, recency_bias):
        """Recency term of THIS rank's rows, normalised by the GLOBAL newest timestamp: bias_i = rb * exp(ts_i - max
        over ALL shards) (reference ranking_algorithm.py:183 takes the maximum over every stored row).  One all-reduce
        (MAX) of a scalar when the term is set -- nothing is exchanged per query.  `timestamps` = this shard's rows."""
        ts_max = self.engine.local_ts_max(timestamps)
        if self.world > 1:
            t = torch.tensor([ts_max], dtype=torch.float64, device=self.engine.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            ts_max = float(t.item())
        self.engine.set_recency(timestamps, recency_bias, ts_max)

    def query(self, Q, k, metric_id):
        """Global top-k of a (nq, d) query batch: (int64 [nq,k], float32 [nq,k]) numpy, same on all ranks."""
        if Q.dim() == 1:
            Q = Q.reshape(1, -1)
        if self.world == 1 and not self.force_exchange and hasattr(self.engine, "topk_host"):
            idx_v, sc_v, st_v = self.engine.topk_host(Q, k, metric_id)      # one C call, fallback included
            if (st_v & 4).any():
                raise ValueError("Vectors and query_vector should not contain NaN values.")
            return idx_v.copy(), sc_v.copy()
        idx_v, sc_v, st_v = self._gather_merge(Q, k, metric_id, exact=False)
        if (st_v & 4).any():
            raise ValueError("Vectors and query_vector should not contain NaN values.")
        idx_h, sc_h = idx_v.copy(), sc_v.copy()
        bad = np.nonzero(st_v & 3)[0]
        if bad.size:   # every rank sees the same OR-ed status -> the same collective re-run
            Qb = self.engine.select_queries(Q, bad)
            i2, s2, _ = self._gather_merge(Qb, k, metric_id, exact=True)
            idx_h[bad] = i2
            sc_h[bad] = s2
        return idx_h, sc_h
