"""Small matrices: single launch vs five kernels (host call p50), fp16 and float32 d=384."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for dt in (torch.float16, torch.float32):
    for n in (9_000, 20_000, 50_000, 120_000, 250_000, 500_000):
        V, lo, hi = bench.make_shard(n, 384, dt, 0, 1, dev)
        ix = GpuIndex(V)
        Q = bench.make_queries(8, 384, dt, dev).float()
        res = {}
        for fused in (1, 0, 1, 0):
            ix.set_option('use_fused', fused)
            for i in range(10): ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid)
            assert ix.stat('fused') == fused
            lat = []
            for i in range(200):
                t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid); lat.append(time.perf_counter() - t0)
            res.setdefault(fused, []).append(float(np.median(lat)) * 1e6)
        print(f"{'fp16' if dt == torch.float16 else 'fp32'} n={n}: single launch {min(res[1]):.1f} us, five kernels {min(res[0]):.1f} us", flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
