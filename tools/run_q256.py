"""The 256-query MFMA pass alone (N=10M d=384 fp16 dot top-100), for rocprofv3 passes: put this program directly after
`--` (rocprofv3 --pmc ... -- python3 tools/run_q256.py [launches]).  Prints the HIP-event kernel time."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
n, d, q = 10_000_000, 384, 256
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(q, d, torch.float16, dev)
mid = METRIC_IDS['dot_product']
for _ in range(5): ix.topk_device(Q, 100, mid)
ix.set_option('profile', 1); torch.cuda.synchronize()
for _ in range(launches): ix.topk_device(Q, 100, mid)
torch.cuda.synchronize()
ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches')
print(f"q256 kernel {ns / l / 1e3:.1f} us = {2 * q * n * d / (ns / l) / 1e3:.0f} TFLOP/s over {l} launches", flush=True)
