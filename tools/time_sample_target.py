"""Whole-call time of batched top-k vs the expected number of survivors per query (option sample_target)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for (n, d, q, metric) in ((10_000_000, 384, 256, 'dot_product'), (10_000_000, 384, 64, 'cosine_similarity'), (10_000_000, 384, 16, 'cosine_similarity')):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(q, d, torch.float16, dev).float()
    mid = METRIC_IDS[metric]
    for rep in range(2):
        for T in (0, 1536, 1024, 768, 512):
            ix.set_option('sample_target', T)
            for _ in range(3): ix.topk_views(Q, 100, mid)
            ix.set_option('profile', 1); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): ix.topk_views(Q, 100, mid)
            el = (time.perf_counter() - t0) / 20 * 1e6
            ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
            print(f"d={d} q={q} {metric} T={T or 2048}: call {el:.1f} us, filter kernel {ns/l/1e3:.1f} us, sample rows {ix.stat('sample_rows')}", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
