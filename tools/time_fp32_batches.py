"""fp32 matrices: batch latency through the fp32 MFMA scan vs the VALU scan (4 queries per pass)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for (n, d) in ((1_000_000, 384), (4_000_000, 384), (1_000_000, 768)):
    V, lo, hi = bench.make_shard(n, d, torch.float32, 0, 1, dev)
    ix = GpuIndex(V)
    mid = METRIC_IDS['cosine_similarity']
    for q in (4, 5, 8, 16, 64, 128, 256):
        Q = bench.make_queries(q, d, torch.float32, dev)
        res = []
        for use in (1, 0):
            ix.set_option('use_mfma', use)
            for _ in range(2): ix.topk_views(Q, 100, mid)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): ix.topk_views(Q, 100, mid)
            res.append(((time.perf_counter() - t0) / 5 * 1e3, ix.stat('mfma')))
        print(f"fp32 N={n} d={d} q={q}: default {res[0][0]:.3f} ms (mfma={res[0][1]})  valu-only {res[1][0]:.3f} ms", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
