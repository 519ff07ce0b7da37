"""A/B of two libraries on one GPU: the single-query call at N=10M and N=1.25M (p50 of the host call, kernel time), interleaved.
python tools/time_ab_single.py product path/to/other.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
mid = METRIC_IDS['cosine_similarity']
for n in (10_000_000, 1_250_000):
    V, lo, hi = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(64, 384, torch.float16, dev).float()
    for i in range(20): ix.topk_views(Q[i:i+1], 100, mid)
    ix.set_option('profile', 1)
    lat = []
    for i in range(300):
        t0 = time.perf_counter(); ix.topk_views(Q[i % 64:i % 64 + 1], 100, mid); lat.append(time.perf_counter() - t0)
    print(f"n={n}: p50 {np.median(lat)*1e6:.1f} us, kernel {ix.stat('scan_time_ns')/ix.stat('scan_launches')/1e3:.1f} us", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
'''
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != 'product': env['HYPERDB_HIP_LIB'] = os.path.join(ROOT, lib)
        print(f"--- {lib} (round {rnd})", flush=True)
        subprocess.run([sys.executable, '-c', CHILD], env=env, cwd=ROOT, timeout=300, stderr=subprocess.DEVNULL)
