"""End-to-end single-query latency vs finalize block size (N=10M d=384 fp16 cosine top-100)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d = 10_000_000, 384
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(256, d, torch.float16, dev).to(torch.float32)
mid = METRIC_IDS['cosine_similarity']
for rep in range(2):
    for t in (1024, 512, 256):
        ix.set_option('finalize_threads', t)
        for i in range(5): ix.topk(Q[i:i+1], 100, mid)
        lat = []
        for i in range(100):
            t0 = time.perf_counter(); ix.topk(Q[i:i+1], 100, mid); lat.append(time.perf_counter() - t0)
        print(f"finalize_threads={t}: p50 {1e3*np.median(lat):.4f} ms  mean {1e3*np.mean(lat):.4f}", flush=True)
