"""float32 matrices, 1-8 queries per call: VALU scan vs fp32 MFMA scan (five-kernel pipeline) vs the single launch (1-2 queries)."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for d, n in ((384, 1_000_000), (768, 1_000_000)):
    V, lo, hi = bench.make_shard(n, d, torch.float32, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(16, d, torch.float32, dev)
    for nq in (1, 2, 3, 4, 5, 8):
        out = []
        for label, fused, minq in (("single launch", 1, 5), ("VALU five kernels", 0, 99), ("fp32 MFMA five kernels", 0, 1)):
            ix.set_option('use_fused', fused); ix.set_option('mfma_min_q', minq)
            for i in range(5): ix.topk_views(Q[i:i + nq], 100, mid)
            if label == "single launch" and ix.stat('fused') == 0: continue
            lat = []
            for i in range(40):
                t0 = time.perf_counter(); ix.topk_views(Q[i % 8:i % 8 + nq], 100, mid); lat.append(time.perf_counter() - t0)
            out.append(f"{label} {np.median(lat)*1e6:.0f} us (mfma={ix.stat('mfma')})")
        print(f"fp32 d={d} n={n} nq={nq}: " + ", ".join(out), flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
