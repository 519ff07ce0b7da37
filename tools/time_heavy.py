"""Kernel time of the 'heavy' MFMA passes (more than 64 queries per pass: all eight waves multiply) on several shapes,
for A/B runs of two libraries:  HYPERDB_HIP_LIB=... python tools/time_heavy.py"""
import sys, os
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
tag = os.environ.get('HYPERDB_HIP_LIB', 'product')
for (n, d, q, metric) in ((10_000_000, 384, 256, 'dot_product'), (10_000_000, 384, 128, 'cosine_similarity'), (10_000_000, 384, 200, 'euclidean_metric'),
                          (5_000_000, 768, 128, 'dot_product'), (6_000_000, 512, 256, 'cosine_similarity'), (10_000_000, 256, 256, 'dot_product'),
                          (3_000_000, 1024, 100, 'dot_product')):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(q, d, torch.float16, dev)
    mid = METRIC_IDS[metric]
    for _ in range(5): ix.topk_device(Q, 100, mid)
    ts = []
    for rep in range(3):
        ix.set_option('profile', 1); torch.cuda.synchronize()
        for _ in range(10): ix.topk_device(Q, 100, mid)
        torch.cuda.synchronize()
        ts.append(ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3); ix.set_option('profile', 0)
    print(f"{os.path.basename(tag)} n={n} d={d} q={q} {metric}: kernel {min(ts):.1f} us (min of 3x10), {n*d*2/min(ts)/1e3:.0f} GB/s, {2*q*n*d/min(ts)/1e6:.0f} TFLOP/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
