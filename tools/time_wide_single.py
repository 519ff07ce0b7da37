"""Single-query calls on wide fp16 rows: the single-launch pipeline (query fragments in LDS) against the five-kernel pipeline."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for d, n in ((896, 2_000_000), (1024, 500_000), (1024, 2_000_000), (1024, 5_000_000), (1536, 500_000), (1536, 2_500_000), (1280, 2_000_000)):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(8, d, torch.float16, dev).float()
    res = {}
    for fused in (1, 0, 1, 0):
        ix.set_option('use_fused', fused)
        for i in range(5): ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid)
        assert ix.stat('fused') == fused
        lat = []
        for i in range(100):
            t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid); lat.append(time.perf_counter() - t0)
        res.setdefault(fused, []).append(float(np.median(lat)) * 1e6)
    print(f"fp16 d={d} n={n}: host call p50 single launch {min(res[1]):.1f} us, five kernels {min(res[0]):.1f} us, {n*d*2/min(res[1])/1e3:.0f} GB/s end to end", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
