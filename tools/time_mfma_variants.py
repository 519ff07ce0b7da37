"""A/B the two d=384 Q=256 MFMA variants on the same box (kernel time from the library's HIP events)."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d = 10_000_000, 384
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
for q, metric in ((256, 'dot_product'), (256, 'cosine_similarity'), (192, 'dot_product')):
    Q = bench.make_queries(q, d, torch.float16, dev)
    mid = METRIC_IDS[metric]
    ref = None
    for rep in range(2):
        for variant in (32, 16):
            ix.set_option('mfma_variant', variant)
            for _ in range(3): out = ix.topk_device(Q, 100, mid)
            ix.set_option('profile', 1); torch.cuda.synchronize()
            for _ in range(8): ix.topk_device(Q, 100, mid)
            torch.cuda.synchronize()
            ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
            same = ''
            if ref is None: ref = out
            else: same = f" same_idx={bool(torch.equal(ref[0], out[0]))} status={int(out[2].abs().sum())}"
            print(f"q={q} {metric} variant={variant}: kernel {ns/l/1e3:.1f} us, {2*q*n*d/(ns/l)/1e3:.1f} TFLOP/s{same}", flush=True)
