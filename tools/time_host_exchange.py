"""Latency of the shared-memory record swap + host merge (hyperdb/sharded.py HostExchange, hdb_merge_topk_host) for W ranks on
this host: no GPU involved.  python tools/time_host_exchange.py [W] [nq] [k]"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(rank, world, port, nq, k):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperdb.sharded import HostExchange
    from hyperdb import _native
    hx = HostExchange(dist.group.WORLD, rank, world, torch.device("cpu"))
    nb = _native.packed_bytes(nq, k)
    rec = np.zeros(nb, dtype=np.uint8)
    idx, sc, st = _native.record_views(rec, nq, k)
    rng = np.random.default_rng(rank)
    for q in range(nq):
        s = np.sort(rng.standard_normal(k).astype(np.float32))[::-1]
        idx[q], sc[q] = rank * 1_250_000 + np.arange(k), s
    dist.barrier()
    for _ in range(200): hx.exchange_merge(rec, nq, k)
    dist.barrier()
    one = []
    for _ in range(3000):
        t0 = time.perf_counter(); hx.exchange_merge(rec, nq, k); one.append(time.perf_counter() - t0)
    if rank == 0:
        o = np.array(one) * 1e6
        print(f"world={world} nq={nq} k={k}: hdb_host_exchange_merge (one C call) p50 {np.median(o):.1f} us p99 {np.percentile(o, 99):.1f}", flush=True)
    hx.close(); dist.destroy_process_group()

if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    mp.spawn(worker, args=(w, 37000 + os.getpid() % 2000, nq, k), nprocs=w, join=True)
