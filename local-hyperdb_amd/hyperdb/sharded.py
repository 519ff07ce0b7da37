"""Row-sharded index: one process per GPU, one RCCL all-gather per query batch.

The stored N x d matrix is split into contiguous row shards (rank r holds rows [lo_r, hi_r)).  Scores
of a row depend only on that row, the query and the row's bias, and top-k is an associative merge,
so the only exchange step is the gather of each shard's k best:

    local scan + top-k  ->  packed record [idx int64 | score f32 | status i32]   (device, no host hop)
    torch.distributed.all_gather_into_tensor (backend "nccl" == RCCL over xGMI; payload = nq*k*12 B per rank)
    hdb_merge_topk_packed on every rank   ->  identical global top-k everywhere

That all-gather is the default exchange.  On ONE node (the launch model of bench.py: one process per GPU of a node) there is a
second transport for records of up to 64 KiB per rank, chosen with exchange="host" / HDB_EXCHANGE=host (or "auto"): every
rank's hdb_topk_host leaves its record in host memory anyway (the answer is for the host), so the ranks swap records through a
shared-memory segment (HostExchange: sequence-tagged slots, two parities) and merge with hdb_merge_topk_host -- ~5 us on the
build container against a collective + a merge launch + another stream synchronisation (>= 30 us at 1.2 KB on one GPU).  No
multi-GPU run has compared the two yet: bench.py times both in the same run and uses the faster one for `value`.

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


class HostExchangeUnavailable(RuntimeError):
    """Raised on EVERY rank of the group when the shared-memory segment could not be set up on some rank."""


class HostExchange:
    """All-gather of small host records between the ranks of ONE node through a file in /dev/shm that every rank maps.

    Layout: [parity 0 | parity 1] x [rank] x (64-byte header {seq u64} + slot_bytes).  Exchange number s uses parity s & 1:
    a rank copies its record into its slot, then stores seq = s (release), then spins until every slot of that parity
    carries s (acquire) and merges the records straight out of the segment -- all inside hdb_host_exchange_merge.  A slot
    is rewritten two exchanges later, and nobody can be two exchanges ahead of a rank that has not published the one in
    between, so readers never see a slot change under them.
    The file is unlinked as soon as every rank has mapped it: nothing is left behind, whatever happens to the processes."""
    HEADER = 64

    def __init__(self, group, rank, world, device, slot_bytes=65536, timeout_s=120.0):
        import mmap
        import os
        import time
        self.rank, self.world, self.slot_bytes, self.timeout_s = rank, world, int(slot_bytes), float(timeout_s)
        self._time = time
        stride = self.HEADER + self.slot_bytes
        total = 2 * world * stride
        # the file name travels as a number in a tensor broadcast (plain collectives only: they are what RCCL / gloo do best)
        tag = torch.zeros(1, dtype=torch.int64, device=device)
        if rank == 0:
            tag[0] = (time.time_ns() ^ (os.getpid() << 20)) & 0x7FFFFFFFFFFFFFFF
        src = dist.get_global_rank(group, 0) if (group is not None and hasattr(dist, "get_global_rank")) else 0
        dist.broadcast(tag, src=src, group=group)
        name = f"/dev/shm/hdb_x_{int(tag.item()):x}"

        def agreed(ok):                               # every rank learns whether the step worked everywhere (also the barrier)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return bool(flag.item())

        fd, ok = -1, True
        if rank == 0:
            try:
                fd = os.open(name, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.ftruncate(fd, total)
            except OSError:
                ok = False
        if not agreed(ok):                            # the file exists -- or nobody goes on
            if fd >= 0:
                os.close(fd)
                os.unlink(name)
            raise HostExchangeUnavailable("rank 0 could not create the shared-memory segment")
        self._mm = None
        try:
            if rank != 0:
                fd = os.open(name, os.O_RDWR)
            self._mm = mmap.mmap(fd, total)
        except (OSError, ValueError):
            ok = False
        if fd >= 0:
            os.close(fd)
        everybody = agreed(ok)                        # everybody has mapped it -- or nobody uses it
        if rank == 0:
            os.unlink(name)
        if not everybody:
            if self._mm is not None:
                self._mm.close()
            raise HostExchangeUnavailable("a rank could not map the shared-memory segment")
        import ctypes
        self._stride = stride
        self._cobj = ctypes.c_char.from_buffer(self._mm)
        self._addr = ctypes.addressof(self._cobj)      # for hdb_host_exchange_merge
        self._out = {}                                # (nq, k) -> (uint8 array, its address, its views)
        self._rec_addr = {}
        self._count = 0

    def exchange_merge(self, record, nq, k):
        """One C call (hdb_host_exchange_merge): publish this rank's packed record (uint8 numpy, host memory), wait for the
        others, merge all of them -> views (idx, score, status) of the merged record (overwritten by the next call with the
        same shape)."""
        from . import _native
        slot = self._out.get((nq, k))
        if slot is None:
            out = np.empty(_native.packed_bytes(nq, k), dtype=np.uint8)
            slot = self._out[(nq, k)] = (out, out.ctypes.data, _native.record_views(out, nq, k))
        self._count += 1
        addr = self._rec_addr.get(id(record))             # the records are long-lived pinned buffers: look their address up once
        if addr is None:
            if len(self._rec_addr) > 64:
                self._rec_addr.clear()
            addr = self._rec_addr[id(record)] = (record.ctypes.data, record)     # (keeps the array alive: ids are not reused)
        _native.host_exchange_merge(self._addr, self._stride, self.world, self.rank, self._count, addr[0], nq, k,
                                    slot[1], self.timeout_s)
        return slot[2]

    def close(self):
        self._cobj = None
        self._addr = None
        try:
            self._mm.close()
        except BufferError:                           # an exported pointer is still alive: the mapping goes with the process
            pass


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
        if parts * k > _native.MERGE_DEVICE_CAP:                  # (see merge_packed_to_host)
            out = np.empty(_native.packed_bytes(nq, k), dtype=np.uint8)
            _native.merge_topk_host(gathered.cpu().numpy(), parts, nq, k, out)
            out_record.copy_(torch.from_numpy(out))
            return
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
        h = host.numpy()
        if parts * k > _native.MERGE_DEVICE_CAP:
            # the merge kernel ranks parts*k entries per query in LDS (<= 8192): beyond that the gathered records come to the
            # host in one copy and hdb_merge_topk_host (any parts*k) merges them -- the transport stays the all-gather
            _native.merge_topk_host(gathered.cpu().numpy(), parts, nq, k, h)
        else:
            _native.merge_topk_packed_into(gathered, parts, nq, k, host)
            torch.cuda.current_stream(self.device).synchronize()
        return (h[:nq * k * 8].view(np.int64).reshape(nq, k), h[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k),
                h[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32))

    def topk_host(self, Q, k, metric_id):
        return self.index.topk_views(Q, k, metric_id)

    def topk_record_host(self, Q, k, metric_id, exact=False):
        """This shard's packed record in HOST memory (uint8 numpy view, overwritten by the next call)."""
        if not exact:
            return self.index.topk_record_host(Q, k, metric_id)      # one C call; a failed threshold is re-run inside it
        nq = int(Q.shape[0])                             # (rare: only when another shard still reported a failure)
        rec = self.new_record(self.packed_bytes(nq, k), 0)
        self.index.topk_packed(Q, k, metric_id, rec, exact=True)
        return rec.cpu().numpy()

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
    def __init__(self, local, n_total=None, group=None, engine=None, force_exchange=False, exchange=None):
        """exchange: "host" (shared-memory swap of host records + host merge; one node only), "collective" (all-gather of
        device records + merge kernel) or None = $HDB_EXCHANGE, else "host" when every rank runs on this node."""
        import os
        self.engine = engine if engine is not None else HipEngine(local)
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None and dist is not None) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n_total = n_total
        self.force_exchange = force_exchange and group is not None     # run gather+merge even with one rank (tests)
        self._hx = None
        self._host_out = {}               # (nq, k) -> merged record of the host-side merge of an all-gather (world*k > 8192)
        # default: the RCCL all-gather (what the north star names).  The shared-memory swap is chosen explicitly -- exchange="host" /
        # HDB_EXCHANGE=host -- or by a caller that has timed both on its node (bench.py calibrates and reports both); "auto" =
        # the swap wherever every rank sits on one node.
        mode = exchange or os.environ.get("HDB_EXCHANGE") or "collective"
        if mode not in ("host", "collective", "auto"):
            raise ValueError("exchange must be 'host', 'collective' or None")
        if self.world > 1 and mode != "collective" and hasattr(self.engine, "topk_record_host"):
            import socket
            import zlib
            dev = self._coll_device()
            mine = torch.tensor([zlib.crc32(socket.gethostname().encode())], dtype=torch.int64, device=dev)
            hosts = torch.zeros(self.world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(hosts, mine, group=group)
            if bool((hosts == hosts[0]).all().item()):
                try:
                    self._hx = HostExchange(group, self.rank, self.world, dev)
                except HostExchangeUnavailable:       # raised on every rank alike: all of them keep the collective
                    if mode == "host":
                        raise
            elif mode == "host":
                raise ValueError("exchange='host' needs every rank on one node")

    def _coll_device(self):
        """Where the tensors of the small bookkeeping collectives live: the engine's device under RCCL, the host under gloo."""
        try:
            backend = str(dist.get_backend(self.group))
        except Exception:
            backend = "nccl"
        return torch.device("cpu") if backend == "gloo" else self.engine.device

    def close(self):
        if self._hx is not None:
            self._hx.close()
            self._hx = None

    # -- helpers -------------------------------------------------------------------------------
    POISON = 1 << 30          # status bit of a record published in place of a shard's answer: that rank's local top-k raised

    @staticmethod
    def _poison_record(nb, nq, k):
        rec = np.zeros(nb, dtype=np.uint8)
        rec[:nq * k * 8].view(np.int64)[:] = -1
        rec[nq * k * 8:nq * k * 12].view(np.float32)[:] = -np.inf
        rec[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32)[:] = ShardedIndex.POISON
        return rec

    def _gather_merge(self, Q, k, metric_id, exact):
        """-> host views (idx, score, status) of the merged global result.

        A rank whose local top-k raises still takes part in the exchange, with a poison record (status bit 30 on every
        query): its peers would otherwise wait for it until the timeout, and the sequence counters of the shared-memory
        swap (or the collective) would differ between the ranks for good.  The failing rank re-raises its own error, the
        others raise a RuntimeError when they see the bit."""
        eng = self.engine
        nq = int(Q.shape[0])
        nb = eng.packed_bytes(nq, k)
        err = None
        if self._hx is not None and nb <= self._hx.slot_bytes:
            try:
                mine = eng.topk_record_host(Q, k, metric_id, exact=exact)
            except Exception as e:                               # noqa: BLE001 -- re-raised below, after the exchange
                err, mine = e, self._poison_record(nb, nq, k)
            out = self._hx.exchange_merge(mine, nq, k)
        else:
            rec = eng.new_record(nb, 0)
            try:
                eng.topk_packed(Q, k, metric_id, rec, exact=exact)
            except Exception as e:                               # noqa: BLE001
                err = e
                if self.world > 1 or self.force_exchange:
                    # (a sticky device error makes this copy raise too: the rank can then not reach the exchange at all)
                    try:
                        rec.copy_(torch.from_numpy(self._poison_record(nb, nq, k)))
                    except Exception as e2:                      # noqa: BLE001
                        self._leave_group(e, e2)
                        raise e
            if self.world == 1 and not self.force_exchange:
                if err is not None:
                    raise err
                return eng.record_to_host(rec, nq, k)           # already the global answer
            try:
                gathered = eng.new_record(nb * self.world, 1)
                dist.all_gather_into_tensor(gathered, rec, group=self.group)
            except Exception as e2:                              # noqa: BLE001
                if err is None:
                    raise
                self._leave_group(err, e2)
                raise err
            if self.world * k > self._merge_device_cap():
                # the device merge ranks world*k entries per query in LDS (<= 8192; 8 ranks: k <= 1024).  Larger results keep the
                # same transport -- the all-gather above -- and merge on the host (hdb_merge_topk_host: any world*k)
                out = self._merge_gathered_on_host(gathered, nq, k)
            elif hasattr(eng, "merge_packed_to_host"):
                out = eng.merge_packed_to_host(gathered, self.world, nq, k)
            else:
                merged = eng.new_record(nb, 2)
                eng.merge_packed_into(gathered, self.world, nq, k, merged)
                out = eng.record_to_host(merged, nq, k)
        if err is not None:
            raise err
        if (out[2] & self.POISON).any():
            raise RuntimeError("another rank failed while computing its shard's top-k (see that rank's error)")
        return out

    @staticmethod
    def _merge_device_cap():
        from . import _native
        return _native.MERGE_DEVICE_CAP

    def _merge_gathered_on_host(self, gathered, nq, k):
        """All-gathered packed records (one uint8 tensor, device or host) -> views of the merged record, merged by
        hdb_merge_topk_host: one D2H copy of world * record bytes, then host code."""
        from . import _native
        g = gathered.cpu().numpy()
        out = self._host_out.get((nq, k))
        if out is None:
            if len(self._host_out) >= _native.HOST_RECORD_SLOTS:
                self._host_out.clear()
            out = self._host_out[(nq, k)] = np.empty(_native.packed_bytes(nq, k), dtype=np.uint8)
        return _native.merge_topk_host(g, self.world, nq, k, out)

    def _leave_group(self, err, err2):
        """This rank's top-k raised AND it cannot publish a poison record (the device is gone): its peers are, or soon will be,
        inside the collective of this query and would wait for it until the backend's timeout.  Abort the communicator where
        torch offers that, else end the process with a non-zero code -- under torchrun either one ends the peers' wait."""
        import os
        import sys
        print(f"hyperdb.sharded: rank {self.rank} cannot take part in the exchange any more ({err!r}; then {err2!r}); "
              "leaving the process group so that the other ranks do not wait", file=sys.stderr, flush=True)
        abort = getattr(getattr(dist, "distributed_c10d", None), "_abort_process_group", None)
        try:
            if abort is None:
                raise RuntimeError("no abort entry point")
            abort(self.group)
        except Exception:                                        # noqa: BLE001
            os._exit(70)

    # -- public --------------------------------------------------------------------------------
    def set_recency(self, timestamps, recency_bias):
        """Recency term of THIS rank's rows, normalised by the GLOBAL newest timestamp: bias_i = rb * exp(ts_i - max
        over ALL shards) (reference ranking_algorithm.py:183 takes the maximum over every stored row).  One all-reduce
        (MAX) of a scalar when the term is set -- nothing is exchanged per query.  `timestamps` = this shard's rows."""
        ts_max = self.engine.local_ts_max(timestamps)
        if self.world > 1:
            t = torch.tensor([ts_max], dtype=torch.float64, device=self._coll_device())
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
