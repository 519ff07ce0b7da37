"""Round 4 latency map of the few-query calls: p50 per call (us, top-100, ix.topk_views with a device query batch) over rows x
(dtype, d) x metric x queries, with the local flavour on (the product) and off (round 3's pipelines) side by side.
"pass" = one pass over V at 7 TB/s.  usage: python tools/sweep_latency_r4.py [quick]"""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(9)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
rows = (1_000, 8_192, 20_000, 100_000, 250_000, 500_000, 1_250_000, 10_000_000) if not quick else (20_000, 100_000, 1_250_000, 10_000_000)
shapes = ((torch.float16, 384), (torch.float32, 384), (torch.float16, 768), (torch.float32, 768)) if not quick else ((torch.float16, 384),)
def p50(ix, Q, mid, reps=60):
    for _ in range(5): ix.topk_views(Q, 100, mid)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for dt, d in shapes:
    for n in rows:
        if n * d * (2 if dt == torch.float16 else 4) > 9e9: continue
        V = torch.randn((n, d), generator=g, device='cuda').to(dt)
        ix = GpuIndex(V)
        passus = n * d * V.element_size() / 7e6
        for metric in ("cosine_similarity", "euclidean_metric", "hamming_distance"):
            out = []
            for nq in (1, 2, 4):
                Q = torch.randn((nq, d), generator=g, device='cuda').to(dt).float()
                mid = METRIC_IDS[metric]
                ix.set_option("use_local", 1); ix.set_option("bits_local", 1); a = p50(ix, Q, mid); tag = f"k{ix.stat('fused')}{'L' if ix.stat('local') else ''}"
                ix.set_option("use_local", 0); ix.set_option("bits_local", 0); b = p50(ix, Q, mid); tag0 = f"k{ix.stat('fused')}"
                ix.set_option("use_local", 1); ix.set_option("bits_local", 1)
                out.append(f"nq={nq}: {a:.0f} ({tag}) | r3 {b:.0f} ({tag0})")
            print(f"{str(dt)[6:]:8s} d={d:4d} n={n:8d} {metric[:9]:9s} pass {passus:6.1f} us | " + "   ".join(out), flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
