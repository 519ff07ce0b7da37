"""Kernel time of the float32 single-query call (single-launch pipeline) against the number of rows: fixed cost vs streaming rate."""
import sys, os
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for dt, es in ((torch.float32, 4), (torch.float16, 2)):
    for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000):
        V, lo, hi = bench.make_shard(n, 384, dt, 0, 1, dev)
        ix = GpuIndex(V)
        Q = bench.make_queries(8, 384, dt, dev).to(torch.float32)
        for i in range(5): ix.topk_device(Q[i % 8:i % 8 + 1], 100, mid)
        ts = []
        for rep in range(3):
            ix.set_option('profile', 1); torch.cuda.synchronize()
            for i in range(20): ix.topk_device(Q[i % 8:i % 8 + 1], 100, mid)
            torch.cuda.synchronize()
            ts.append(ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3); ix.set_option('profile', 0)
        t = min(ts)
        print(f"{'fp32' if es == 4 else 'fp16'} n={n}: kernel {t:.1f} us, {n*384*es/t/1e3:.0f} GB/s, fused={ix.stat('fused')}", flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
