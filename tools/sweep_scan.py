import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d = 10_000_000, 384
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(64, d, torch.float16, dev)
mid = METRIC_IDS['cosine_similarity']
def run(tag):
    for i in range(3): ix.topk_device(Q[i:i+1], 100, mid)
    ix.set_option('profile', 1); torch.cuda.synchronize()
    for i in range(20): ix.topk_device(Q[i:i+1], 100, mid)
    torch.cuda.synchronize()
    ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
    print(f"{tag}: {ns/l/1e3:.1f} us -> {n*d*2/(ns/l):.1f} GB/s", flush=True)
for rep in range(2):
    for mb in (256, 512, 1024):
        for nt in (0,):
            ix.set_option('max_blocks', mb); ix.set_option('debug_flags', nt)
            run(f"fp16 d384 rep{rep} max_blocks={mb} plain_loads={1 if nt else 0}")
ix.close(); del V; torch.cuda.empty_cache()
for (n, d, dt, name) in ((1_000_000, 384, torch.float32, 'fp32 d384 N=1M'), (4_000_000, 768, torch.float16, 'fp16 d768 N=4M'), (4_000_000, 384, torch.float32, 'fp32 d384 N=4M')):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(64, d, dt, dev)
    es = 2 if dt == torch.float16 else 4
    def run2(tag):
        for i in range(3): ix.topk_device(Q[i:i+1], 100, mid)
        ix.set_option('profile', 1); torch.cuda.synchronize()
        for i in range(20): ix.topk_device(Q[i:i+1], 100, mid)
        torch.cuda.synchronize()
        ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches'); ix.set_option('profile', 0)
        print(f"{tag}: {ns/l/1e3:.1f} us -> {n*d*es/(ns/l):.1f} GB/s", flush=True)
    for mb in (256, 512, 1024, 2048):
        for nt in (0,):
            ix.set_option('max_blocks', mb); ix.set_option('debug_flags', nt)
            run2(f"{name} max_blocks={mb} plain_loads={1 if nt else 0}")
    ix.close(); del V; torch.cuda.empty_cache()
