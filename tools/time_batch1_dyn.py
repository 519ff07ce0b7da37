"""Static split against dynamic tile hand-out in the single-launch batched call (and in the five-kernel pipeline's filter pass):
interleaved rounds on one GPU, kernel time from the library's HIP events + host p50 per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
cases = [(1_250_000, 384, 8, "cosine_similarity"), (1_250_000, 384, 1, "euclidean_metric"), (2_500_000, 384, 32, "dot_product"),
         (10_000_000, 384, 256, "dot_product"), (10_000_000, 384, 16, "cosine_similarity"), (5_000_000, 768, 64, "euclidean_metric"),
         (1_250_000, 384, 256, "dot_product"), (5_000_000, 128, 48, "dot_product")]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1:] or str(c[2]) in sys.argv[1:]]
variants = [("one launch, static", dict(use_fused=1, dyn_tiles=0)), ("one launch, dyn >= 16 MiB, not heavy", dict(use_fused=1, dyn_tiles=1, dyn_min_mb=16, dyn_heavy=0)),
            ("one launch, dyn always", dict(use_fused=1, dyn_tiles=1, dyn_min_mb=0, dyn_heavy=1)),
            ("five kernels, dyn >= 16 MiB, not heavy", dict(use_fused=0, dyn_tiles=1, dyn_min_mb=16, dyn_heavy=0)),
            ("five kernels, dyn always", dict(use_fused=0, dyn_tiles=1, dyn_min_mb=0, dyn_heavy=1))]
for n, d, nq, metric in cases:
    V, _, _ = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    Q = bench.make_queries(nq, d, torch.float16, dev).float()
    ix = GpuIndex(V)
    mid = METRIC_IDS[metric]
    res = {name: [] for name, _ in variants}
    for rnd in range(3):
        for name, opts in variants:
            for k, v in opts.items():
                ix.set_option(k, v)
            for _ in range(5):
                ix.topk_views(Q, 100, mid)
            ix.set_option("profile", 1)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
            kern = ix.stat("scan_time_ns") / max(1, ix.stat("scan_launches")) / 1e3
            ix.set_option("profile", 0)
            res[name].append((np.median(ts) * 1e6, kern))
    print(f"n={n} d={d} nq={nq} {metric}")
    for name, _ in variants:
        print(f"    {name:40s} p50 per call {' / '.join(f'{a:.1f}' for a, _ in res[name])} us; dominant kernel {' / '.join(f'{b:.1f}' for _, b in res[name])} us", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
