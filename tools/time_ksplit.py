"""Wide rows: the K-slice MFMA scan (hdb_mfma_ksplit.hip) against the VALU scan (4 queries per pass), batches of 16 / 64 / 128."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
for dt, n, d in ((torch.float32, 1_000_000, 1536), (torch.float32, 1_000_000, 1024), (torch.float16, 1_000_000, 2048), (torch.float16, 1_000_000, 4096)):
    V, _, _ = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    es = 2 if dt == torch.float16 else 4
    for nq in (16, 64, 128):
        Q = bench.make_queries(nq, d, dt, dev).float()
        mid = METRIC_IDS["cosine_similarity"]
        res = {}
        for mf in (1, 0):
            ix.set_option("use_mfma", mf)
            for _ in range(2): ix.topk_views(Q, 100, mid)
            ts = []
            for _ in range(8):
                t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
            res[mf] = np.median(ts) * 1e3
        print(f"{'fp16' if es == 2 else 'fp32'} n={n} d={d} nq={nq} cosine top-100: K slices on the matrix cores {res[1]:.2f} ms, VALU scan {res[0]:.2f} ms "
              f"[one pass over V at 7 TB/s = {n * d * es / 7e9:.2f} ms]", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
