"""2-4 query calls: single launch vs five kernels (host call p50)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for dt, d, n, qs in ((torch.float16, 384, 10_000_000, (2, 4)), (torch.float16, 384, 1_250_000, (2, 4)), (torch.float16, 768, 4_000_000, (2, 4)),
                     (torch.float16, 1536, 2_000_000, (2,)), (torch.float32, 384, 1_000_000, (2,)), (torch.float32, 256, 2_000_000, (2,))):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(16, d, dt, dev).float()
    for nq in qs:
        res = {}
        for fused in (1, 0, 1, 0):
            ix.set_option('use_fused', fused)
            for i in range(5): ix.topk_views(Q[i:i + nq], 100, mid)
            assert ix.stat('fused') == fused
            lat = []
            for i in range(60):
                t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + nq], 100, mid); lat.append(time.perf_counter() - t0)
            res.setdefault(fused, []).append(float(np.median(lat)) * 1e6)
        print(f"{'fp16' if dt == torch.float16 else 'fp32'} d={d} n={n} nq={nq}: single launch {min(res[1]):.1f} us, five kernels {min(res[0]):.1f} us", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
