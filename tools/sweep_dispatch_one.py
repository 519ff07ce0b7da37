"""One query: the 1-4-query single launch vs the batched single launch (fused_max_q = 0), p50 per call."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(9)
def p50(ix, Q, mid, reps=60):
    for _ in range(5): ix.topk_views(Q, 100, mid)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6, ix.stat('fused')
for dt, d in ((torch.float16, 384), (torch.float16, 768), (torch.float16, 1536), (torch.float32, 384), (torch.float32, 768)):
    for n in (20_000, 50_000, 100_000, 200_000, 400_000, 800_000, 1_250_000, 2_500_000):
        V = torch.randn((n, d), generator=g, device='cuda').to(dt); ix = GpuIndex(V)
        Qd = torch.randn((1, d), generator=g, device='cuda').to(dt).float()
        Qh = Qd.cpu().numpy()
        out = []
        for name, Q in (("device query", Qd), ("host query", Qh)):
            r = []
            for knob, mq in ((-1, -1), (0, 1)):
                ix.set_option('fused_max_q', knob); ix.set_option('f32_min_q', mq)
                t, k = p50(ix, Q, METRIC_IDS['cosine_similarity']); r.append(f"{t:.0f} (k{k})")
            ix.set_option('fused_max_q', -1); ix.set_option('f32_min_q', -1)
            out.append(f"{name}: single {r[0]} vs batched {r[1]}")
        print(f"{str(dt)[6:]} d={d} n={n}: " + "   ".join(out), flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
