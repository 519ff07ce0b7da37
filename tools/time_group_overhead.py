"""What the single-process group costs per call: GpuGroup with ONE shard against the plain GpuIndex on the same matrix
(hdb_group_topk_host - hdb_topk_host), shard-sized and headline-sized, interleaved rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
from hyperdb.group import GpuGroup
import bench
dev = torch.device("cuda", 0)
for n in (1_250_000, 10_000_000):
    V, _, _ = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
    Qh = bench.make_queries(64, 384, torch.float16, dev).float().cpu().numpy()
    one, grp = GpuIndex(V), GpuGroup(V, [0])
    mid = METRIC_IDS["cosine_similarity"]
    res = {"index": [], "group": []}
    for rnd in range(3):
        for name, ix in (("index", one), ("group", grp)):
            for i in range(10):
                ix.topk_views(Qh[i:i + 1], 100, mid)
            ts = []
            for i in range(200):
                t0 = time.perf_counter(); ix.topk_views(Qh[i % 64:i % 64 + 1], 100, mid); ts.append(time.perf_counter() - t0)
            res[name].append(np.median(ts) * 1e6)
    a, b = one.topk_views(Qh[:1], 100, mid), grp.topk_views(Qh[:1], 100, mid)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    print(f"n={n} d=384 fp16 cosine top-100, numpy query: GpuIndex p50 {' / '.join(f'{x:.1f}' for x in res['index'])} us; "
          f"GpuGroup(1 shard) p50 {' / '.join(f'{x:.1f}' for x in res['group'])} us; difference {np.median(res['group']) - np.median(res['index']):.1f} us", flush=True)
    one.close(); grp.close(); del V; torch.cuda.empty_cache()
