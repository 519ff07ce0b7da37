"""float32 rows that ride a padded geometry (d = 300, 700) or two K slices (d = 1024, 1536): bf16 parts against float32 MFMAs."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(12)
def p50(ix, Q, mid, reps=12):
    for _ in range(3): ix.topk_views(Q, 100, mid)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for n, d in ((1_000_000, 300), (1_000_000, 700), (1_000_000, 1024), (500_000, 1536)):
    V = torch.randn((n, d), generator=g, device='cuda')
    ix = GpuIndex(V)
    passus = n * d * 4 / 7e6
    mid = METRIC_IDS["cosine_similarity"]
    out = []
    for nq in (8, 16, 64, 128):
        Q = torch.randn((nq, d), generator=g, device='cuda')
        r = {}
        for rnd in range(2):
            for sp in (1, 0):
                ix.set_option("f32_split", sp); ix.set_option("f32_split_min_q", 1); r.setdefault(sp, []).append(p50(ix, Q, mid))
                if rnd == 0:
                    i_, s_, st_ = ix.topk_views(Q, 100, mid); r[("res", sp)] = (np.array(i_), np.array(s_), ix.stat("f32_split"))
        (i1, s1, f1), (i0, s0, f0) = r[("res", 1)], r[("res", 0)]
        out.append(f"nq={nq}: parts {min(r[1]):.0f} (split={f1}) | f32 {min(r[0]):.0f} | idx {float((i1 == i0).mean()):.4f} dscore {float(np.abs(s1 - s0).max()):.1e}")
    print(f"n={n} d={d} cosine pass@7TB/s {passus:6.1f} us | " + "   ".join(out), flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
