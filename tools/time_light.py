"""Kernel time of the HBM-bound MFMA passes (5..64 queries) for A/B runs of two libraries."""
import sys, os
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
tag = os.path.basename(os.environ.get('HYPERDB_HIP_LIB', 'product'))
for (n, d, q, metric, bias) in ((10_000_000, 384, 8, 'cosine_similarity', False), (10_000_000, 384, 64, 'cosine_similarity', False),
                                (10_000_000, 768, 64, 'euclidean_metric', True), (10_000_000, 768, 16, 'dot_product', False),
                                (2_500_000, 1536, 32, 'cosine_similarity', False), (10_000_000, 128, 48, 'dot_product', False),
                                (1_250_000, 384, 16, 'cosine_similarity', False)):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    if bias:
        g = torch.Generator(device=dev).manual_seed(99)
        ix.set_recency(1.7e9 + torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 30 * 86400.0, 0.5)
    Q = bench.make_queries(q, d, torch.float16, dev)
    mid = METRIC_IDS[metric]
    for _ in range(5): ix.topk_device(Q, 100, mid)
    ts = []
    for rep in range(3):
        ix.set_option('profile', 1); torch.cuda.synchronize()
        for _ in range(10): ix.topk_device(Q, 100, mid)
        torch.cuda.synchronize()
        ts.append(ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3); ix.set_option('profile', 0)
    print(f"{tag} n={n} d={d} q={q} {metric}{' +bias' if bias else ''}: kernel {min(ts):.1f} us (min of 3x10), {n*d*2/min(ts)/1e3:.0f} GB/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
