import sys, os, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
n, d = 10_000_000, 384
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, torch.device('cuda', 0))
ix = GpuIndex(V)
Q = bench.make_queries(256, d, torch.float16, torch.device('cuda', 0))
mid = METRIC_IDS['dot_product']
for flags in (0, 16, 4, 20):
    ix.set_option('debug_flags', flags)
    for _ in range(2): ix.topk_device(Q, 100, mid)
    ix.set_option('profile', 1)
    torch.cuda.synchronize()
    for _ in range(5): ix.topk_device(Q, 100, mid)
    torch.cuda.synchronize()
    ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches')
    ix.set_option('profile', 0)
    print(f"flags={flags} (noDMA={flags&1} noMFMA={(flags>>1)&1} noFilter={(flags>>2)&1}): scan kernel {ns/l/1e3:.1f} us", flush=True)
