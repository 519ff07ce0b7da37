"""Latency of the bit-metric path (hamming / jaccard) at N=10M, d=384: sampled threshold first, exact selection on overflow."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d = 10_000_000, 384
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
for metric in ('hamming_distance', 'jaccard_similarity'):
    mid = METRIC_IDS[metric]
    for q in (1, 8):
        Q = bench.make_queries(q, d, torch.float16, dev).to(torch.float32)
        for _ in range(3): ix.topk_views(Q, 100, mid)
        lat = []
        for _ in range(30):
            t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); lat.append(time.perf_counter() - t0)
        print(f"{metric} q={q}: p50 {np.median(lat)*1e3:.3f} ms  path={ix.stat('path')}", flush=True)
