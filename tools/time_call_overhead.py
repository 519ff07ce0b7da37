"""Per-call latency of the single-query path at shard-sized N (what each rank of an 8-GPU run holds)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
short = len(sys.argv) > 1 and sys.argv[1] == 'short'      # one size, few calls: for rocprofv3 --kernel-trace
for n in ((1_250_000,) if short else (1_250_000, 10_000_000)):
    V, lo, hi = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(400, 384, torch.float16, dev).to(torch.float32)
    mid = METRIC_IDS['cosine_similarity']
    for direct in ((1,) if short else (0, 1, 0, 1)):
        ix.set_option('host_direct', direct)
        for i in range(20): ix.topk_views(Q[i:i + 1], 100, mid)
        lat = []
        for i in range(20, 120 if short else 400):
            t0 = time.perf_counter(); ix.topk_views(Q[i:i + 1], 100, mid); lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e6
        print(f"n={n} host_direct={direct}: p50 {np.median(lat):.1f} us  mean {lat.mean():.1f}  p99 {np.percentile(lat, 99):.1f}", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
