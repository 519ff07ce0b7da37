"""Single-query calls on narrow fp16 / float32 rows: the single-launch pipeline against the five-kernel pipeline (host call p50)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for dt, d, n in ((torch.float16, 128, 1_000_000), (torch.float16, 128, 10_000_000), (torch.float16, 256, 1_000_000), (torch.float16, 256, 10_000_000),
                 (torch.float32, 128, 1_000_000), (torch.float32, 128, 5_000_000), (torch.float32, 256, 1_000_000), (torch.float32, 256, 5_000_000),
                 (torch.float16, 512, 1_000_000), (torch.float16, 640, 1_000_000)):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(8, d, dt, dev).float()
    res = {}
    for fused in (1, 0, 1, 0):
        ix.set_option('use_fused', fused)
        for i in range(5): ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid)
        assert ix.stat('fused') == fused
        lat = []
        for i in range(100):
            t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + 1], 100, mid); lat.append(time.perf_counter() - t0)
        res.setdefault(fused, []).append(float(np.median(lat)) * 1e6)
    es = 2 if dt == torch.float16 else 4
    print(f"{'fp16' if es == 2 else 'fp32'} d={d} n={n}: single launch {min(res[1]):.1f} us, five kernels {min(res[0]):.1f} us", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
