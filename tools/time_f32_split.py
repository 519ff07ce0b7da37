"""float32 batches on the matrix cores: rows multiplied as bf16 parts (hdb_mfma_f32s.hip, option f32_split = 1: every launch) against
v_mfma_f32_16x16x4_f32 (0): p50 per call, interleaved, plus the agreement of the two answers and of both with float64."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(11)
def p50(ix, Q, mid, reps=20):
    for _ in range(3): ix.topk_views(Q, 100, mid)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
shapes = [(2_000_000, 384), (2_000_000, 128), (2_000_000, 256), (500_000, 384), (1_000_000, 768), (1_000_000, 512)]
if len(sys.argv) > 1: shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for n, d in shapes:
    V = torch.randn((n, d), generator=g, device='cuda')
    ix = GpuIndex(V)
    passus = n * d * 4 / 7e6
    for metric in ("cosine_similarity", "euclidean_metric"):
        out = []
        for nq in (16, 32, 48, 64, 96, 128):
            Q = torch.randn((nq, d), generator=g, device='cuda')
            mid = METRIC_IDS[metric]
            r = {}
            for rnd in range(2):
                for sp in (1, 0):
                    ix.set_option("f32_split", sp); ix.set_option("f32_split_min_q", 1); r.setdefault(sp, []).append(p50(ix, Q, mid))
                    if rnd == 0:
                        i_, s_, st_ = ix.topk_views(Q, 100, mid); r[("res", sp)] = (np.array(i_), np.array(s_), ix.stat("f32_split"), ix.stat("fused"), int(np.abs(np.array(st_)).max()))
            ix.set_option("f32_split", 1)
            (i1, s1, f1, fu1, st1), (i0, s0, f0, fu0, st0) = r[("res", 1)], r[("res", 0)]
            same = float((i1 == i0).mean()); ds = float(np.abs(s1 - s0).max())
            # float64 scores of the rows the split flavour returned (first 4 queries)
            qq = min(nq, 4); Vd = V[torch.as_tensor(i1[:qq].reshape(-1), device='cuda')].double().reshape(qq, 100, d); Qd = Q[:qq].double()
            dots = (Vd * Qd[:, None, :]).sum(-1)
            if metric == "cosine_similarity": ref = dots / Vd.norm(dim=-1) / Qd.norm(dim=-1)[:, None]
            elif metric == "dot_product": ref = dots
            else: ref = 1.0 / (1.0 + (Vd - Qd[:, None, :]).norm(dim=-1))
            err = float((torch.as_tensor(s1[:qq], device='cuda').double() - ref).abs().max())
            out.append(f"nq={nq}: parts {min(r[1]):.0f} (split={f1} fused={fu1} status={st1}) | f32 {min(r[0]):.0f} | same idx {same:.4f} dscore {ds:.1e} vs f64 {err:.1e}")
        print(f"n={n} d={d} {metric[:9]:9s} pass@7TB/s {passus:6.1f} us | " + "   ".join(out), flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
