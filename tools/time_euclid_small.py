"""Per-call latency of euclidean / cosine single-query calls at shard size (N=1.25M d=384 fp16): single launch against the
five-kernel pipeline, interleaved."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
for n, d in ((1_250_000, 384), (10_000_000, 384), (1_250_000, 768)):
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    for metric, nq in (("euclidean_metric", 1), ("cosine_similarity", 1), ("euclidean_metric", 4), ("cosine_similarity", 8), ("euclidean_metric", 16)):
        Q = bench.make_queries(nq, d, torch.float16, dev).float()
        mid = METRIC_IDS[metric]
        res = {0: [], 1: []}
        for rnd in range(3):
            for fused in (1, 0):
                ix.set_option("use_fused", fused)
                for _ in range(10):
                    ix.topk_views(Q, 100, mid)
                kind = ix.stat("fused")
                ts = []
                for _ in range(100):
                    t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
                res[fused].append((np.median(ts) * 1e6, kind))
        print(f"n={n} d={d} {metric} nq={nq}: single launch (kind {res[1][0][1]}) p50 {' / '.join(f'{a:.1f}' for a, _ in res[1])} us; five kernels {' / '.join(f'{a:.1f}' for a, _ in res[0])} us", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
