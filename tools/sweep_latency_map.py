"""Per-call p50 over (dtype, d, metric, rows, queries): a map to spot calls whose time is far from 'one pass over V + fixed cost'."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(9)
rows = (20_000, 100_000, 500_000, 2_000_000)
for dt, d in ((torch.float16, 384), (torch.float32, 384), (torch.float16, 768), (torch.float32, 768), (torch.float16, 128), (torch.float32, 1536), (torch.float16, 1024)):
    for n in rows:
        V = torch.randn((n, d), generator=g, device='cuda').to(dt)
        ix = GpuIndex(V)
        passus = n * d * V.element_size() / 7e6
        for metric in ("cosine_similarity", "euclidean_metric", "pearson_correlation", "manhattan_distance", "hamming_distance"):
            out = []
            for nq in (1, 4, 16, 64):
                Q = torch.randn((nq, d), generator=g, device='cuda').to(dt).float()
                mid = METRIC_IDS[metric]
                for _ in range(3): ix.topk_views(Q, 100, mid)
                ts = []
                for _ in range(30):
                    t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
                out.append(f"nq={nq}: {np.median(ts)*1e6:.0f} us (k{ix.stat('fused')}{'m' if ix.stat('mfma') else ''})")
            print(f"{str(dt)[6:]:8s} d={d:5d} n={n:8d} {metric[:9]:9s} pass {passus:6.1f} us | " + "  ".join(out), flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
