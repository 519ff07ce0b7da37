"""Kernel time of the batched MFMA scan with 1, 2, 4, 8 workgroups per CU (option max_blocks = -mult): the hardware
dispatcher balances CUs of different streaming speed when there are more workgroups than CUs."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
for (n, d, q, metric, bias) in ((10_000_000, 384, 256, 'dot_product', False), (10_000_000, 384, 64, 'cosine_similarity', False),
                                (10_000_000, 384, 8, 'cosine_similarity', False), (10_000_000, 768, 64, 'euclidean_metric', True)):
    V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    if bias:
        g = torch.Generator(device=dev).manual_seed(99)
        ix.set_recency(1.7e9 + torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 30 * 86400.0, 0.5)
    Q = bench.make_queries(q, d, torch.float16, dev)
    mid = METRIC_IDS[metric]
    for rep in range(2):
        for mult in (0, -2, -4, -8, -16):
            ix.set_option('max_blocks', mult)
            for _ in range(3): ix.topk_device(Q, 100, mid)
            ix.set_option('profile', 1); torch.cuda.synchronize()
            for _ in range(10): ix.topk_device(Q, 100, mid)
            torch.cuda.synchronize()
            ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
            print(f"d={d} q={q} {metric} wg/CU={max(1, -mult)}: kernel {ns/l/1e3:.1f} us -> {n*d*2/(ns/l):.1f} GB/s, {2*q*n*d/(ns/l)/1e3:.1f} TFLOP/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
