"""Kernel time of the HBM-bound MFMA passes (5..64 queries): static tile split vs dynamic hand-out, same library, interleaved."""
import sys, os
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for (n, d, q, metric, bias) in ((10_000_000, 384, 8, 'cosine_similarity', False), (10_000_000, 384, 64, 'cosine_similarity', False),
                                (10_000_000, 768, 64, 'euclidean_metric', True), (10_000_000, 768, 16, 'dot_product', False),
                                (2_500_000, 1536, 32, 'cosine_similarity', False), (10_000_000, 128, 48, 'dot_product', False),
                                (1_250_000, 384, 16, 'cosine_similarity', False), (1_250_000, 384, 64, 'cosine_similarity', False),
                                (2_500_000, 384, 32, 'dot_product', False), (5_000_000, 256, 40, 'dot_product', False), (5_000_000, 512, 64, 'cosine_similarity', False)):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    if bias:
        g = torch.Generator(device=dev).manual_seed(99)
        ix.set_recency(1.7e9 + torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 30 * 86400.0, 0.5)
    Q = bench.make_queries(q, d, torch.float16, dev)
    mid = METRIC_IDS[metric]
    for _ in range(5): ix.topk_device(Q, 100, mid)
    res = {0: [], 1: []}
    ref = None
    for rep in range(4):
        for dyn in (0, 1):
            ix.set_option('dyn_tiles', dyn)
            out = ix.topk_device(Q, 100, mid)
            if ref is None: ref = [o.clone() for o in out[:2]]
            else: assert all(torch.equal(a, b) for a, b in zip(ref, out[:2])), 'results differ between static and dynamic'
            ix.set_option('profile', 1); torch.cuda.synchronize()
            for _ in range(10): ix.topk_device(Q, 100, mid)
            torch.cuda.synchronize()
            res[dyn].append(ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3); ix.set_option('profile', 0)
    print(f"n={n} d={d} q={q} {metric}{' +bias' if bias else ''}: static {min(res[0]):.1f} us, dynamic {min(res[1]):.1f} us "
          f"({(min(res[1]) / min(res[0]) - 1) * 100:+.1f} %), {n*d*2/min(res[1])/1e3:.0f} GB/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
