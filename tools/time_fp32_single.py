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
for d in (512, 768):
    for n in (500_000, 1_000_000, 2_000_000):
        V, lo, hi = bench.make_shard(n, d, torch.float32, 0, 1, dev)
        ix = GpuIndex(V)
        Q = bench.make_queries(8, d, torch.float32, dev)
        res = {}
        for fused in (1, 0):
            ix.set_option('use_fused', fused)
            for i in range(5): ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid)
            import time, numpy as np
            lat = []
            for i in range(100):
                t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid); lat.append(time.perf_counter() - t0)
            res[fused] = float(np.median(lat)) * 1e6
        print(f"fp32 d={d} n={n}: host call p50 single launch {res[1]:.1f} us, five kernels {res[0]:.1f} us, {n*d*4/res[1]/1e3:.0f} GB/s end to end", flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
