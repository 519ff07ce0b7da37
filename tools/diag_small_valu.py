"""Small matrices through the VALU scan pipeline (odd d, manhattan, fp64): per-call time and kernel time of the filter pass vs survivors."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
dev = torch.device('cuda', 0)
g = torch.Generator(device='cuda').manual_seed(5)
for dt, n, d, metric in ((torch.float32, 200_000, 100, 'cosine_similarity'), (torch.float32, 1_000_000, 100, 'cosine_similarity'),
                         (torch.float16, 200_000, 384, 'manhattan_distance'), (torch.float32, 200_000, 300, 'dot_product'),
                         (torch.float64, 200_000, 96, 'cosine_similarity')):
    V = torch.randn((n, d), generator=g, device='cuda').to(dt)
    ix = GpuIndex(V)
    for nq in (1, 4, 16):
        Q = torch.randn((nq, d), generator=g, device='cuda').float()
        mid = METRIC_IDS[metric]
        for _ in range(5): ix.topk_views(Q, 100, mid)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
        ix.set_option('profile', 1)
        for _ in range(10): ix.topk_views(Q, 100, mid)
        torch.cuda.synchronize()
        ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches')
        ix.set_option('profile', 0)
        print(f"{str(dt)[6:]} n={n} d={d} {metric} nq={nq}: p50 {np.median(ts)*1e6:.1f} us; scan launches {l/10:.1f} per call, {ns/10/1e3:.1f} us of scan kernels per call; "
              f"one pass over V at 7 TB/s = {n*d*V.element_size()/7e6:.1f} us; path {ix.stat('path')} mfma {ix.stat('mfma')} fused {ix.stat('fused')}", flush=True)
    ix.close(); del V
