"""Kernel time of the single-launch pipeline over its geometries (tile bytes 16-48 KiB): GB/s of the whole call."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for dt, d, n in ((torch.float16, 256, 10_000_000), (torch.float16, 384, 10_000_000), (torch.float16, 512, 5_000_000),
                 (torch.float16, 640, 5_000_000), (torch.float16, 768, 5_000_000), (torch.float16, 1024, 4_000_000), (torch.float16, 1536, 2_500_000),
                 (torch.float32, 128, 5_000_000), (torch.float32, 256, 5_000_000), (torch.float32, 384, 4_000_000), (torch.float32, 768, 2_000_000),
                 (torch.float16, 384, 1_250_000), (torch.float16, 256, 1_250_000), (torch.float16, 1024, 500_000)):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(8, d, dt, dev).float()
    for i in range(5): ix.topk_device(Q[i % 8:i % 8 + 1], 100, mid)
    assert ix.stat('fused') == 1
    ts = []
    for rep in range(3):
        ix.set_option('profile', 1); torch.cuda.synchronize()
        for i in range(20): ix.topk_device(Q[i % 8:i % 8 + 1], 100, mid)
        torch.cuda.synchronize()
        ts.append(ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3); ix.set_option('profile', 0)
    es = 2 if dt == torch.float16 else 4
    print(f"{'fp16' if es == 2 else 'fp32'} d={d} n={n}: kernel {min(ts):.1f} us, {n*d*es/min(ts)/1e3:.0f} GB/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
