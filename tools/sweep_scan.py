"""Row-scan kernel time (HIP events recorded by the library) for a few shapes and grid sizes."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for (n, d, dt, name) in ((10_000_000, 384, torch.float16, 'fp16 d384 N=10M'), (1_000_000, 384, torch.float32, 'fp32 d384 N=1M'), (4_000_000, 768, torch.float16, 'fp16 d768 N=4M')):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(64, d, dt, dev).to(torch.float32)
    es = 2 if dt == torch.float16 else 4
    for mb in (512, 768, 1024, 1536, 2048):
        ix.set_option('max_blocks', mb)
        for i in range(3): ix.topk_device(Q[i:i+1], 100, mid)
        ix.set_option('profile', 1); torch.cuda.synchronize()
        for i in range(20): ix.topk_device(Q[i:i+1], 100, mid)
        torch.cuda.synchronize()
        ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
        print(f"{name} max_blocks={mb}: {ns/l/1e3:.1f} us -> {n*d*es/(ns/l):.1f} GB/s", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
