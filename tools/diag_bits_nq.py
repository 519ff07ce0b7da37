import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for n in (250_000, 1_250_000, 2_500_000):
    V, _, _ = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    mid = METRIC_IDS['hamming_distance']
    Qall = bench.make_queries(8, 384, torch.float16, dev).float()
    for nq in (1, 2, 3, 4):
        for off in (0, 4):
            Q = Qall[off:off + nq].contiguous()
            for _ in range(5): ix.topk_views(Q, 100, mid)
            ts = []
            for _ in range(100):
                t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
            _, _, st = ix.topk_device(Q, 100, mid)
            print(f"n={n} nq={nq} off={off}: p50 {np.median(ts)*1e6:.1f} us, status of the sampled pass {st.cpu().numpy().tolist()}, path {ix.stat('path')} fused {ix.stat('fused')} sample rows {ix.stat('sample_rows')} m {ix.stat('sample_m')}", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
