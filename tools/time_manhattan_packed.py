"""manhattan_distance on fp16 rows: packed fp16 differences (round 4, l1_packed = 1) against the float32 arithmetic of round 3
(l1_packed = 0) in the same tile kernel, interleaved on one GPU; fp16-valued queries (what an fp16 store is asked with)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
for n, d in ((5_000_000, 384), (2_000_000, 384), (2_500_000, 768), (4_000_000, 512), (5_000_000, 128)):
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    mid = METRIC_IDS["manhattan_distance"]
    for nq in (2, 5, 16, 64):
        Q = bench.make_queries(nq, d, torch.float16, dev).float()
        res = {1: [], 0: []}
        for rnd in range(2):
            for pk in (1, 0):
                ix.set_option("l1_packed", pk)
                for _ in range(3): ix.topk_views(Q, 100, mid)
                ts = []
                for _ in range(12):
                    t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
                res[pk].append(np.median(ts) * 1e6)
        ix.set_option("l1_packed", 1)
        print(f"fp16 n={n} d={d} nq={nq}: packed {' / '.join(f'{x:.0f}' for x in res[1])} us, float32 arithmetic {' / '.join(f'{x:.0f}' for x in res[0])} us "
              f"[one pass over V at 7 TB/s = {n * d * 2 / 7e6:.0f} us]", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
