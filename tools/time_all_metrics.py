"""Every metric of the reference, 1 / 2 / 5 queries per call: host call p50 and the GB/s of V it implies (anomaly hunt)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for dt, d, n in ((torch.float16, 384, 5_000_000), (torch.float32, 384, 2_000_000)):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(16, d, dt, dev).float()
    es = 2 if dt == torch.float16 else 4
    for metric in ("dot_product", "cosine_similarity", "euclidean_metric", "manhattan_distance", "pearson_correlation", "hamming_distance", "jaccard_similarity"):
        mid = METRIC_IDS[metric]
        out = []
        for nq in (1, 2, 5):
            for i in range(4): ix.topk_views(Q[i:i + nq], 100, mid)
            lat = []
            for i in range(30):
                t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + nq], 100, mid); lat.append(time.perf_counter() - t0)
            t = float(np.median(lat)) * 1e6
            out.append(f"nq={nq} {t:.0f} us (path {ix.stat('path')}, mfma {ix.stat('mfma')}, fused {ix.stat('fused')})")
        print(f"{'fp16' if es == 2 else 'fp32'} n={n} {metric}: " + "; ".join(out) + f"   [one pass over V at 7 TB/s = {n*d*es/7e6:.0f} us]", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
