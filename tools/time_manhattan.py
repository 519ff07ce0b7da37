"""manhattan_distance: the LDS-staged tile kernel (hdb_l1_tile.hip) against the 4-query VALU scan, interleaved on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
for dt, n, d in ((torch.float16, 5_000_000, 384), (torch.float16, 2_500_000, 768), (torch.float16, 3_000_000, 640), (torch.float16, 4_000_000, 512), (torch.float16, 5_000_000, 128), (torch.float32, 2_000_000, 384), (torch.float32, 1_250_000, 768), (torch.float32, 2_000_000, 512), (torch.float32, 4_000_000, 128)):
    V, _, _ = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    es = 2 if dt == torch.float16 else 4
    mid = METRIC_IDS["manhattan_distance"]
    for nq in (1, 2, 5, 8, 16):
        Q = bench.make_queries(nq, d, dt, dev).float()
        res = {1: [], 0: []}
        for rnd in range(2):
            for tile in (1, 0):
                ix.set_option("use_l1_tile", tile)
                for _ in range(3): ix.topk_views(Q, 100, mid)
                ts = []
                for _ in range(15):
                    t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
                res[tile].append(np.median(ts) * 1e6)
        print(f"{'fp16' if es == 2 else 'fp32'} n={n} d={d} nq={nq}: tile kernel {' / '.join(f'{x:.0f}' for x in res[1])} us, 4-query scan {' / '.join(f'{x:.0f}' for x in res[0])} us "
              f"[one pass over V at 7 TB/s = {n * d * es / 7e6:.0f} us]", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
