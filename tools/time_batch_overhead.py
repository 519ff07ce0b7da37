"""Where a 256-query call spends its time outside the pass over V: host call time with / without the profiling events, with /
without the result copies, status polling on / off; kernel time from the events."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d, q = 10_000_000, 384, 256
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(4 * q, d, torch.float16, dev).float()
mid = METRIC_IDS['dot_product']
def run(label, fn, reps=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    lat = []
    for i in range(reps):
        t0 = time.perf_counter(); fn(i); lat.append(time.perf_counter() - t0)
    print(f"{label}: p50 {np.median(lat)*1e6:.1f} us  mean {np.mean(lat)*1e6:.1f}", flush=True)
for rnd in range(2):
    ix.set_option('profile', 0); ix.set_option('host_poll', 1)
    run("views, no events, poll", lambda i: ix.topk_views(Q[(i % 4) * q:(i % 4 + 1) * q], 100, mid))
    ix.set_option('host_poll', 0)
    run("views, no events, stream sync", lambda i: ix.topk_views(Q[(i % 4) * q:(i % 4 + 1) * q], 100, mid))
    ix.set_option('host_poll', 1); ix.set_option('profile', 1)
    run("views, events, poll", lambda i: ix.topk_views(Q[(i % 4) * q:(i % 4 + 1) * q], 100, mid))
    print(f"   kernel {ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3:.1f} us", flush=True)
    ix.set_option('profile', 0)
    run("topk (copies), no events, poll", lambda i: ix.topk(Q[(i % 4) * q:(i % 4 + 1) * q], 100, mid))
