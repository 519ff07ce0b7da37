"""Kernel time of the MFMA scan on the batched configs (HIP events recorded by the library)."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for (n, d, q, metric) in ((10_000_000, 384, 256, 'dot_product'), (10_000_000, 384, 64, 'cosine_similarity'), (10_000_000, 768, 64, 'euclidean_metric')):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(q, d, torch.float16, dev)
    mid = METRIC_IDS[metric]
    for rep in range(3):
        for _ in range(2): ix.topk_device(Q, 100, mid)
        ix.set_option('profile', 1); torch.cuda.synchronize()
        for _ in range(5): ix.topk_device(Q, 100, mid)
        torch.cuda.synchronize()
        ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
        print(f"d={d} q={q} {metric}: kernel {ns/l/1e3:.1f} us -> {n*d*2/(ns/l):.1f} GB/s, {2*q*n*d/(ns/l)/1e3:.1f} TFLOP/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
