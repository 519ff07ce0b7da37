"""Statuses of the local flavour over a grid of (metric, setup, queries, k, local_m) on one seeded matrix (diagnostic)."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 384)
rng = np.random.default_rng(n + d)
V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
V[n - 1] = V[7]
Q = rng.standard_normal((4, d)).astype(np.float16).astype(np.float32)
Q[1] = V[n // 3].astype(np.float32)
Q[2] = rng.standard_normal(d).astype(np.float32) * 37.5
ix = GpuIndex(V); ix.set_option("local_max_q", 4); ix.set_option("local_max_tiles", 16); ix.set_option("local_small", 1)
bias = torch.rand(n, generator=torch.Generator().manual_seed(3)).float().cuda() * 0.2
for metric in ("dot_product", "cosine_similarity"):
    mid = METRIC_IDS[metric]
    for setup in ("plain", "bias"):
        ix.set_bias(bias if setup == "bias" else None)
        for nq in (1, 2, 3, 4):
            for k in (100, 128):
                for m in (0, 8, 16, 32):
                    ix.set_option("local_m", m)
                    out = []
                    for rep in range(3):
                        li, ls, st = ix.topk_device(Q[:nq], k, mid)
                        ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                        out.append((st.tolist(), bool(torch.equal(li, ei))))
                    print(metric, setup, "nq", nq, "k", k, "m", m, "local", ix.stat("local"), out, flush=True)
ix.close()
