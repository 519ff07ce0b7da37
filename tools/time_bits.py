"""hamming / jaccard per-call latency: the single launch (hdb_bits_fused.hip) against the six-launch pipeline, interleaved."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
for n, d in ((250_000, 384), (1_250_000, 384), (5_000_000, 384), (10_000_000, 384), (2_000_000, 768)):
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    ix.set_option('bits_fused', 2)
    for metric, nq in (("hamming_distance", 1), ("hamming_distance", 4), ("jaccard_similarity", 1)):
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
        print(f"n={n} d={d} {metric} nq={nq}: single launch (kind {res[1][0][1]}) p50 {' / '.join(f'{a:.1f}' for a, _ in res[1])} us; six launches {' / '.join(f'{a:.1f}' for a, _ in res[0])} us", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
