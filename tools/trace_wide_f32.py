"""One shape under rocprofv3 --kernel-trace: float32 d = 1536 (two K slices), 16 queries, cosine -- which launch takes what."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
n, d, nq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device='cuda').manual_seed(12)
V = torch.randn((n, d), generator=g, device='cuda')
ix = GpuIndex(V)
Q = torch.randn((nq, d), generator=g, device='cuda')
mid = METRIC_IDS["cosine_similarity"]
for _ in range(12): ix.topk_views(Q, 100, mid)
print("split", ix.stat("f32_split"), "fused", ix.stat("fused"), "sample rows", ix.stat("sample_rows"))
