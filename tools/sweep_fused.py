"""Single-launch top-k (hdb_mfma_fused.h) against the multi-kernel pipeline and the exact selection, small to large N,
then per-call latency of both.  Run under `timeout` on a GPU box."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
bad = 0
for (n, d, dt) in ((8200, 384, torch.float16), (20_000, 384, torch.float16), (70_001, 128, torch.float16), (300_000, 768, torch.float16),
                   (1_250_000, 384, torch.float16), (2_000_003, 256, torch.float16), (1_000_000, 512, torch.float16), (500_000, 640, torch.float16),
                   (8200, 384, torch.float32), (1_000_000, 384, torch.float32), (300_001, 128, torch.float32), (200_000, 256, torch.float32)):
    V, lo, hi = bench.make_shard(n, d, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(8, d, dt, dev).to(torch.float32)
    g = torch.Generator(device=dev).manual_seed(5)
    bias = torch.rand(n, generator=g, device=dev) * 0.2
    for metric in ('cosine_similarity', 'dot_product'):
        for with_bias in (False, True):
            ix.set_bias(bias if with_bias else None)
            for nq in ((1, 2) if dt == torch.float32 else (1, 2, 3, 4)):
                for k in (1, 100, 128):
                    mid = METRIC_IDS[metric]
                    ix.set_option('use_fused', 1)
                    fi, fs, fst = ix.topk_device(Q[:nq], k, mid)
                    fused = ix.stat('fused')
                    torch.cuda.synchronize()
                    ix.set_option('use_fused', 0)
                    ui, us, ust = ix.topk_device(Q[:nq], k, mid)
                    ei, es, _ = ix.topk_device(Q[:nq], k, mid, exact=True)
                    ok = fused == 1 and int(fst.abs().sum()) == 0 and torch.equal(fi, ei) and torch.equal(fs, es) and torch.equal(fi, ui) and torch.equal(fs, us)
                    if not ok:
                        bad += 1
                        print('MISMATCH', n, d, metric, with_bias, nq, k, 'fused', fused, 'status', fst.tolist(), ust.tolist(),
                              'idx_eq', torch.equal(fi, ei), 'f==e', torch.equal(fs, es), 'f==u', torch.equal(fs, us), 'u==e', torch.equal(us, es), 'maxdiff', float((fs - es).abs().max()), flush=True)
    print(f"n={n} d={d} {dt}: ok so far, bad={bad}", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
print('parity done, bad =', bad, flush=True)
for (n, dt) in ((1_250_000, torch.float16), (10_000_000, torch.float16), (1_000_000, torch.float32)):
    V, lo, hi = bench.make_shard(n, 384, dt, 0, 1, dev)
    ix = GpuIndex(V)
    Q = bench.make_queries(400, 384, dt, dev).to(torch.float32)
    mid = METRIC_IDS['cosine_similarity']
    for fused in (0, 1, 0, 1):
        ix.set_option('use_fused', fused)
        for i in range(20): ix.topk_views(Q[i:i + 1], 100, mid)
        lat = []
        for i in range(20, 400):
            t0 = time.perf_counter(); ix.topk_views(Q[i:i + 1], 100, mid); lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e6
        print(f"n={n} {dt} fused={fused} (stat {ix.stat('fused')}): p50 {np.median(lat):.1f} us  mean {lat.mean():.1f}  p99 {np.percentile(lat, 99):.1f}", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
