"""Time hdb_merge_topk for the exchange shapes of the sharded index (parts x nq x k)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
from hyperdb._native import merge_topk

for parts, nq, k in [(8, 1, 100), (8, 256, 100), (8, 64, 100), (2, 1, 100), (8, 1, 1024), (4, 1, 2048)]:
    rng = np.random.default_rng(0)
    sc = -np.sort(-rng.standard_normal((parts, nq, k)).astype(np.float32), axis=-1)
    idx = rng.permutation(parts * nq * k).reshape(parts, nq, k).astype(np.int64)
    I, S = torch.from_numpy(idx).cuda(), torch.from_numpy(sc).cuda()
    for _ in range(5):
        merge_topk(I, S, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        merge_topk(I, S, k)
    e1.record(); torch.cuda.synchronize()
    print(f"parts={parts} nq={nq} k={k}: {e0.elapsed_time(e1) / 200 * 1000:.1f} us/call", flush=True)
