"""Odd row widths on the matrix cores (hdb_mfma_anyd.h) against the VALU scan: p50 per call, 16 and 64 queries, 1M rows."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(5)
def p50(ix, Q, mid, reps=30):
    for _ in range(3): ix.topk_views(Q, 100, mid)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for dt, d in ((torch.float16, 96), (torch.float16, 200), (torch.float16, 304), (torch.float16, 1000), (torch.float32, 100), (torch.float32, 300)):
    n = 1_000_000
    V = torch.randn((n, d), generator=g, device='cuda').to(dt)
    ix = GpuIndex(V)
    passus = n * d * V.element_size() / 7e6
    for metric in ("cosine_similarity", "euclidean_metric"):
        out = []
        for nq in (16, 64):
            Q = torch.randn((nq, d), generator=g, device='cuda').to(dt).float()
            mid = METRIC_IDS[metric]
            ix.set_option("use_mfma", 1); a = p50(ix, Q, mid); m = ix.stat("mfma")
            ix.set_option("use_mfma", 0); b = p50(ix, Q, mid)
            ix.set_option("use_mfma", 1)
            out.append(f"nq={nq}: {a:.0f} us (mfma={m}) | VALU scan {b:.0f} us")
        print(f"{str(dt)[6:]:8s} d={d:5d} n={n} {metric[:9]:9s} one pass at 7 TB/s {passus:6.1f} us | " + "   ".join(out), flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
