"""Which pipeline for few queries?  p50 per call with the dispatch knobs (fused_max_q, f32_min_q, bits_max_q) forced either way."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(9)
def p50(ix, Q, mid, reps=40):
    for _ in range(4): ix.topk_views(Q, 100, mid)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6, ix.stat('fused'), ix.stat('mfma')
which = sys.argv[1] if len(sys.argv) > 1 else "abc"
rows = (20_000, 100_000, 300_000, 500_000, 1_000_000, 2_000_000, 5_000_000)
if "a" in which:
    for dt, d in ((torch.float16, 384), (torch.float16, 768), (torch.float16, 1024)):
        for n in rows:
            V = torch.randn((n, d), generator=g, device='cuda').to(dt); ix = GpuIndex(V)
            out = []
            for nq in (2, 3, 4) if d <= 768 else (2,):
                Q = torch.randn((nq, d), generator=g, device='cuda').to(dt).float()
                r = []
                for knob in (-1, 1):
                    ix.set_option('fused_max_q', knob)
                    t, k, m = p50(ix, Q, METRIC_IDS['cosine_similarity']); r.append(f"{t:.0f} (k{k})")
                ix.set_option('fused_max_q', -1)
                out.append(f"nq={nq}: fused {r[0]} vs batch {r[1]}")
            print(f"A fp16 d={d} n={n}: " + "   ".join(out), flush=True)
            ix.close(); del V; torch.cuda.empty_cache()
if "b" in which:
    for dt, d in ((torch.float32, 384), (torch.float32, 768), (torch.float32, 256)):
        for n in rows[:6]:
            V = torch.randn((n, d), generator=g, device='cuda').to(dt); ix = GpuIndex(V)
            out = []
            for nq in (2, 3, 4):
                Q = torch.randn((nq, d), generator=g, device='cuda').float()
                r = []
                for fm, mq in ((-1, -1), (1, 2)):
                    ix.set_option('fused_max_q', fm); ix.set_option('f32_min_q', mq)
                    t, k, m = p50(ix, Q, METRIC_IDS['cosine_similarity']); r.append(f"{t:.0f} (k{k}{'m' if m else ''})")
                ix.set_option('fused_max_q', -1); ix.set_option('f32_min_q', -1)
                out.append(f"nq={nq}: rule {r[0]} vs batch {r[1]}")
            print(f"B fp32 d={d} n={n}: " + "   ".join(out), flush=True)
            ix.close(); del V; torch.cuda.empty_cache()
if "c" in which:
    for n in (20_000, 100_000, 500_000, 2_000_000, 10_000_000):
        V = torch.randn((n, 384), generator=g, device='cuda').to(torch.float16); ix = GpuIndex(V)
        out = []
        for nq in (5, 8, 12, 16, 64):
            Q = torch.randn((nq, 384), generator=g, device='cuda').float()
            r = []
            for knob in (-1, 4):
                ix.set_option('bits_max_q', knob)
                t, k, m = p50(ix, Q, METRIC_IDS['hamming_distance'], 20); r.append(f"{t:.0f} (k{k})")
            ix.set_option('bits_max_q', -1)
            out.append(f"nq={nq}: single x{(nq + 3) // 4} {r[0]} vs six {r[1]}")
        print(f"C hamming d=384 n={n}: " + "   ".join(out), flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
