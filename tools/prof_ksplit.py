"""One wide-rows configuration for rocprofv3 --kernel-trace --stats: float32 d=1536 (or argv: dtype d n nq), cosine top-100, 20 calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "fp16" else torch.float32
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda", 0)
V, _, _ = bench.make_shard(n, d, dt, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(nq, d, dt, dev).float()
for _ in range(20): ix.topk_views(Q, 100, METRIC_IDS["cosine_similarity"])
torch.cuda.synchronize()
print("path", ix.stat("path"), "mfma", ix.stat("mfma"), "launches", ix.stat("scan_launches"))
