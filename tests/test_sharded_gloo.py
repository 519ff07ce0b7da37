"""Row-sharded index over torch.distributed with the gloo backend, world_size 2 and 8, on CPU.

What is under test is the HOST logic of hyperdb/sharded.py: shard bounds, the packed exchange record
([idx int64 | score f32 | status i32]), ONE exchange per batch -- the all-gather of device records AND the shared-memory
swap of host records with the library's host merge (hdb_merge_topk_host: host code, runs here) --, merge ordering, global
row ids, and the collective exact re-run when any shard reports a failed threshold.  The compute engine is a
CPU stand-in defined HERE, in tests/, on top of the oracle -- the product engine (HipEngine) needs a
GPU and is covered by tests/test_gpu_parity.py::test_shard_merge_equals_global.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METRICS = {0: "dot_product", 1: "cosine_similarity", 2: "euclidean_metric", 3: "hamming_distance"}


class OracleEngine:
    """CPU stand-in with the HipEngine interface (test infrastructure only)."""

    def __init__(self, V_local, row_base, fail_query=None):
        from oracle import ranking_oracle as orc
        from hyperdb import _native
        self.orc, self.nat = orc, _native
        self.V, self.row_base, self.fail_query = V_local, row_base, fail_query
        self.device = torch.device("cpu")
        self.exact_calls = 0
        self.bias = None
        self.ts_max_seen = None

    def local_ts_max(self, timestamps):
        return float(np.max(np.asarray(timestamps, dtype=np.float64)))

    def set_recency(self, timestamps, recency_bias, ts_max):
        self.ts_max_seen = ts_max
        self.bias = (recency_bias * np.exp(np.asarray(timestamps, dtype=np.float64) - ts_max)).astype(np.float32)

    def packed_bytes(self, nq, k):
        return self.nat.packed_bytes(nq, k)        # the C function: layout comes from the library

    def new_record(self, nbytes, slot=0):
        return torch.zeros(nbytes, dtype=torch.uint8)

    def _views(self, record, nq, k):
        h = record.numpy()
        return (h[:nq * k * 8].view(np.int64).reshape(nq, k), h[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k),
                h[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32))

    def topk_packed(self, Q, k, metric_id, record, exact=False):
        nq = Q.shape[0]
        idx, sc, st = self._views(record, nq, k)
        idx[:], sc[:], st[:] = -1, -np.inf, 0
        if exact:
            self.exact_calls += 1
        for qi in range(nq):
            ex = self.orc.exact_scores(self.V, Q[qi].numpy(), METRICS[metric_id]).astype(np.float32)
            if self.bias is not None:
                ex = ex + self.bias
            order = np.lexsort((np.arange(len(ex)), -ex))[:k]
            idx[qi, :len(order)] = order + self.row_base
            sc[qi, :len(order)] = ex[order]
            if not exact and self.fail_query is not None and qi == self.fail_query:
                st[qi] = 1                               # pretend the sampled threshold underflowed here
                idx[qi], sc[qi] = -1, -np.inf

    def merge_packed_into(self, gathered, parts, nq, k, out_record):
        nb = self.packed_bytes(nq, k)
        oi, osc, ost = self._views(out_record, nq, k)
        ost[:] = 0
        cand_i, cand_s = [], []
        for p in range(parts):
            i, s, st = self._views(gathered[p * nb:(p + 1) * nb], nq, k)
            cand_i.append(i.copy()); cand_s.append(s.copy()); ost |= st
        ci, cs = np.concatenate(cand_i, axis=1), np.concatenate(cand_s, axis=1)
        for qi in range(nq):
            ok = ci[qi] >= 0
            ii, ss = ci[qi][ok], cs[qi][ok]
            order = np.lexsort((ii, -ss))[:k]
            oi[qi], osc[qi] = -1, -np.inf
            oi[qi, :len(order)], osc[qi, :len(order)] = ii[order], ss[order]

    def record_to_host(self, record, nq, k):
        return self._views(record, nq, k)

    # host-record flavour (HostExchange -> hdb_host_exchange_merge)
    def topk_record_host(self, Q, k, metric_id, exact=False):
        rec = self.new_record(self.packed_bytes(Q.shape[0], k))
        self.topk_packed(Q, k, metric_id, rec, exact=exact)
        return rec.numpy()

    def select_queries(self, Q, which):
        return Q[torch.as_tensor(which)]


def _worker(rank, world, port, n, d, k, metric_id, fail_rank, out_dir, recency=False, exchange="collective"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hyperdb.sharded import ShardedIndex, shard_bounds
        rng = np.random.default_rng(77)
        V = rng.standard_normal((n, d)).astype(np.float32)
        V[n // 2 + 3] = V[5]                              # a duplicate row across shards: tie -> lower global row first
        Q = torch.from_numpy(rng.standard_normal((4, d)).astype(np.float32))
        lo, hi = shard_bounds(n, world, granule=16)[rank]
        eng = OracleEngine(V[lo:hi], lo, fail_query=2 if rank == fail_rank else None)
        sh = ShardedIndex(None, n_total=n, group=dist.group.WORLD, engine=eng, exchange=exchange)
        assert (sh._hx is not None) == (exchange == "host")
        if recency:      # rows get newer with the row id: every shard has a different local maximum
            ts = 1.7e9 + np.arange(n, dtype=np.float64) * 0.002
            sh.set_recency(ts[lo:hi], 5.0)
        idx, sc = sh.query(Q, k, metric_id)
        for rep in range(3):                              # later exchanges reuse the slots (two parities): same answer
            i2, s2 = sh.query(Q, k, metric_id)
            assert np.array_equal(i2, idx) and np.array_equal(s2, sc)
        sh.close()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, sc=sc, exact_calls=eng.exact_calls, lo=lo, hi=hi,
                 ts_max=(eng.ts_max_seen if eng.ts_max_seen is not None else np.nan))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["collective", "host"])
@pytest.mark.parametrize("metric_id,fail_rank", [(1, None), (0, 1), (3, None)])
def test_two_rank_gloo_matches_global(tmp_path, metric_id, fail_rank, exchange):
    from oracle import ranking_oracle as orc
    world, n, d, k = 2, 1000, 24, 10
    port = 29500 + (os.getpid() % 2000) + metric_id + (10 if exchange == "host" else 0)
    mp.spawn(_worker, args=(world, port, n, d, k, metric_id, fail_rank, str(tmp_path), False, exchange), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    V = rng.standard_normal((n, d)).astype(np.float32)
    V[n // 2 + 3] = V[5]
    Q = rng.standard_normal((4, d)).astype(np.float32)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 496, 496, 1000)
    assert np.array_equal(r0["idx"], r1["idx"]) and np.array_equal(r0["sc"], r1["sc"]), "ranks must agree"
    for qi in range(4):
        ex = orc.exact_scores(V, Q[qi], METRICS[metric_id]).astype(np.float32)
        want = np.lexsort((np.arange(n), -ex))[:k]
        assert np.array_equal(r0["idx"][qi], want), (qi, r0["idx"][qi], want)
        assert np.array_equal(r0["sc"][qi], ex[want])
    # a failed threshold on ONE rank triggers the exact re-run on BOTH (collective consistency)
    expect_exact = 4 if fail_rank is not None else 0        # (the worker asks four times)
    assert int(r0["exact_calls"]) == expect_exact and int(r1["exact_calls"]) == expect_exact


@pytest.mark.parametrize("exchange", ["collective", "host"])
def test_eight_rank_gloo_matches_global(tmp_path, exchange):
    """The node-sized case (8 ranks, uneven granule-aligned shards, a failed threshold on rank 5) on CPU."""
    from oracle import ranking_oracle as orc
    from hyperdb.sharded import shard_bounds
    world, n, d, k, metric_id, fail_rank = 8, 1000, 24, 10, 1, 5
    port = 31500 + (os.getpid() % 2000) + (10 if exchange == "host" else 0)
    mp.spawn(_worker, args=(world, port, n, d, k, metric_id, fail_rank, str(tmp_path), False, exchange), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    V = rng.standard_normal((n, d)).astype(np.float32)
    V[n // 2 + 3] = V[5]
    Q = rng.standard_normal((4, d)).astype(np.float32)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [(int(r["lo"]), int(r["hi"])) for r in ranks] == shard_bounds(n, world, granule=16)
    for r in ranks[1:]:
        assert np.array_equal(r["idx"], ranks[0]["idx"]) and np.array_equal(r["sc"], ranks[0]["sc"]), "ranks must agree"
        assert int(r["exact_calls"]) == 4                    # one rank's failed threshold -> every rank re-runs (four calls)
    for qi in range(4):
        ex = orc.exact_scores(V, Q[qi], METRICS[metric_id]).astype(np.float32)
        want = np.lexsort((np.arange(n), -ex))[:k]
        assert np.array_equal(ranks[0]["idx"][qi], want) and np.array_equal(ranks[0]["sc"][qi], ex[want])


def test_recency_uses_the_global_newest_timestamp(tmp_path):
    """Shards whose timestamp maxima differ must normalise exp(ts - max) by the GLOBAL maximum (one all-reduce in
    ShardedIndex.set_recency); with per-shard maxima the merged top-k would differ from the single-matrix answer
    (reference ranking_algorithm.py:183)."""
    from oracle import ranking_oracle as orc
    world, n, d, k, metric_id = 2, 1000, 24, 10, 1
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n, d, k, metric_id, None, str(tmp_path), True), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    V = rng.standard_normal((n, d)).astype(np.float32)
    V[n // 2 + 3] = V[5]
    Q = rng.standard_normal((4, d)).astype(np.float32)
    ts = 1.7e9 + np.arange(n, dtype=np.float64) * 0.002
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert float(r0["ts_max"]) == float(r1["ts_max"]) == ts.max()
    assert np.array_equal(r0["idx"], r1["idx"])
    for qi in range(4):
        oi, osc = orc.rank(V, Q[qi], top_k=k, metric=METRICS[metric_id], timestamps=ts, recency_bias=5.0)
        assert np.array_equal(r0["idx"][qi], oi), (qi, r0["idx"][qi], oi)
        assert np.allclose(r0["sc"][qi], osc, atol=1e-5)
    # with per-shard maxima rank 0's rows would get the bias of rank 1's newest rows: a different answer
    wrong = np.concatenate([5.0 * np.exp(ts[:496] - ts[:496].max()), 5.0 * np.exp(ts[496:] - ts.max())])
    ex = orc.exact_scores(V, Q[0], METRICS[metric_id]) + wrong
    assert not np.array_equal(np.argsort(-ex, kind="stable")[:k], r0["idx"][0])


def test_shard_bounds_cover_everything():
    from hyperdb.sharded import shard_bounds
    for n, w, g in [(10_000_000, 8, 250_000), (1000, 3, 16), (7, 4, 1), (100_000_000, 8, 250_000)]:
        b = shard_bounds(n, w, g)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(lo % g == 0 for lo, _ in b)


def _timeout_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hyperdb.sharded import HostExchange
        from hyperdb import _native
        hx = HostExchange(dist.group.WORLD, rank, world, torch.device("cpu"), timeout_s=1.0)
        nq, k = 2, 5
        rec = np.zeros(_native.packed_bytes(nq, k), dtype=np.uint8)
        idx, sc, st = _native.record_views(rec, nq, k)
        idx[:], sc[:] = rank * 100 + np.arange(k), -np.arange(k, dtype=np.float32) - rank * 0.5
        i1, s1, _ = hx.exchange_merge(rec, nq, k)                       # a normal exchange first
        ok = bool(np.array_equal(i1[0], [0, 100, 1, 101, 2]))
        err = ""
        if rank == 0:
            try:
                hx.exchange_merge(rec, nq, k)                           # rank 1 never comes
            except RuntimeError as e:
                err = str(e)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(f"{int(ok)}|{err}")
        dist.barrier()
        hx.close()
    finally:
        dist.destroy_process_group()


def test_host_exchange_times_out_when_a_rank_does_not_publish(tmp_path):
    """The shared-memory swap merges what both ranks published; a rank that waits for a record that never comes gets a
    RuntimeError after its timeout instead of spinning for ever."""
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_timeout_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = open(tmp_path / "rank0.txt").read().split("|")
    r1 = open(tmp_path / "rank1.txt").read().split("|")
    assert r0[0] == "1" and r1[0] == "1"
    assert "did not publish" in r0[1] and r1[1] == ""


def _poison_worker(rank, world, port, out_dir, exchange):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hyperdb.sharded import ShardedIndex, shard_bounds
        rng = np.random.default_rng(5)
        n, d, k = 600, 16, 7
        V = rng.standard_normal((n, d)).astype(np.float32)
        Q = torch.from_numpy(rng.standard_normal((3, d)).astype(np.float32))
        lo, hi = shard_bounds(n, world, granule=16)[rank]

        class Flaky(OracleEngine):
            boom = False

            def topk_packed(self, Q_, k_, metric_id, record, exact=False):
                if self.boom:
                    raise ValueError("shard exploded")
                return super().topk_packed(Q_, k_, metric_id, record, exact=exact)
        eng = Flaky(V[lo:hi], lo)
        sh = ShardedIndex(None, n_total=n, group=dist.group.WORLD, engine=eng, exchange=exchange)
        good, _ = sh.query(Q, k, 1)
        eng.boom = rank == 1                                 # the next call fails on rank 1 only
        seen = ""
        try:
            sh.query(Q, k, 1)
        except Exception as e:                               # noqa: BLE001
            seen = type(e).__name__ + ":" + str(e)
        eng.boom = False
        again, _ = sh.query(Q, k, 1)                         # counters stayed aligned: the next exchange works
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(f"{seen}|{int(np.array_equal(good, again))}")
        sh.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["collective", "host"])
def test_a_failing_rank_poisons_the_exchange_instead_of_hanging_it(tmp_path, exchange):
    """a rank that raises before publishing left its peers spinning until the timeout and the sequence counters
    apart for good.  It now publishes a poison record: every rank raises at once, and the next query works."""
    port = 37500 + (os.getpid() % 2000) + (10 if exchange == "host" else 0)
    mp.spawn(_poison_worker, args=(2, port, str(tmp_path), exchange), nprocs=2, join=True)
    r0 = open(tmp_path / "rank0.txt").read().split("|")
    r1 = open(tmp_path / "rank1.txt").read().split("|")
    assert r1[0] == "ValueError:shard exploded" and r0[0].startswith("RuntimeError:another rank failed")
    assert r0[1] == "1" and r1[1] == "1"


def _bigk_worker(rank, world, port, out_dir, exchange, nq):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hyperdb.sharded import ShardedIndex, shard_bounds
        rng = np.random.default_rng(123)
        n, d, k = 4000, 8, 3000
        V = rng.standard_normal((n, d)).astype(np.float32)
        V[2100] = V[7]                                     # a tie across shards
        Q = torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32))
        lo, hi = shard_bounds(n, world, granule=16)[rank]

        class NoDeviceMerge(OracleEngine):
            """The device merge kernels rank at most 8192 entries per query: the product path must never ask for more."""

            def merge_packed_into(self, gathered, parts, nq_, k_, out_record):
                assert parts * k_ <= 8192, "world*k above the device merge's cap reached the device merge"
                return super().merge_packed_into(gathered, parts, nq_, k_, out_record)
        eng = NoDeviceMerge(V[lo:hi], lo)
        sh = ShardedIndex(None, n_total=n, group=dist.group.WORLD, engine=eng, exchange=exchange)
        idx, sc = sh.query(Q, k, 1)                        # every shard holds < k rows: its record is padded with -1
        i2, s2 = sh.query(Q, k, 1)
        assert np.array_equal(idx, i2) and np.array_equal(sc, s2)
        small_i, _ = sh.query(Q, 10, 1)                    # world*k = 80: the device-merge branch still serves small k
        assert np.array_equal(small_i, idx[:, :10])
        used_host_swap = sh._hx is not None and eng.packed_bytes(nq, k) <= sh._hx.slot_bytes
        sh.close()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, sc=sc, swap=used_host_swap)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,nq", [("collective", 2), ("host", 1), ("host", 6)])
def test_eight_ranks_k_3000_merges_on_the_host(tmp_path, exchange, nq):
    """8 ranks x k = 3000 = 24 000 entries per query exceeds what the device merge ranks in LDS (8192).  The
    all-gather path now merges the gathered records with hdb_merge_topk_host; the shared-memory swap always did (one query: the
    36-KB record fits a slot; six queries: 216 KB does not, so that call takes the all-gather and the host merge)."""
    from oracle import ranking_oracle as orc
    world, n, d, k = 8, 4000, 8, 3000
    port = 39500 + (os.getpid() % 2000) + (10 if exchange == "host" else 0) + nq
    mp.spawn(_bigk_worker, args=(world, port, str(tmp_path), exchange, nq), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    V = rng.standard_normal((n, d)).astype(np.float32)
    V[2100] = V[7]
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert bool(ranks[0]["swap"]) == (exchange == "host" and nq == 1)
    for r in ranks[1:]:
        assert np.array_equal(r["idx"], ranks[0]["idx"]) and np.array_equal(r["sc"], ranks[0]["sc"])
    for qi in range(nq):
        ex = orc.exact_scores(V, Q[qi], METRICS[1]).astype(np.float32)
        want = np.lexsort((np.arange(n), -ex))[:k]
        assert np.array_equal(ranks[0]["idx"][qi], want) and np.array_equal(ranks[0]["sc"][qi], ex[want])


def test_a_rank_that_cannot_publish_leaves_the_group(monkeypatch):
    """when the local top-k raised AND the poison record cannot be written (a sticky device error), the rank must not
    simply re-raise -- its peers are on their way into the collective.  _gather_merge calls _leave_group (abort the
    communicator / exit non-zero) before re-raising the ORIGINAL error."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
    from hyperdb.sharded import ShardedIndex

    class Dead(OracleEngine):
        def topk_packed(self, Q_, k_, metric_id, record, exact=False):
            raise ValueError("device lost")

        def new_record(self, nbytes, slot=0):
            class Rec:                                       # a record whose upload fails like the kernels did
                def copy_(self_inner, _src):
                    raise RuntimeError("hipErrorLaunchFailure")
            return Rec()
    sh = ShardedIndex(None, n_total=10, group=None, engine=Dead(np.zeros((10, 4), dtype=np.float32), 0))
    sh.world, sh.force_exchange = 2, True                    # as if a second rank were waiting
    left = []
    monkeypatch.setattr(sh, "_leave_group", lambda e1, e2: left.append((str(e1), str(e2))))
    with pytest.raises(ValueError, match="device lost"):
        sh._gather_merge(torch.zeros((1, 4)), 3, 1, exact=False)
    assert left == [("device lost", "hipErrorLaunchFailure")]
