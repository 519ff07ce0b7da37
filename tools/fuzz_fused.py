"""Fuzz of the single-launch pipeline: random shapes it takes (1-4 dot / cosine queries on fp16, 1-2 on float32, k <= 128, bias, row
mask, duplicate rows, clustered rows) against the exact selection of the same library.  A call that reports a status must be one
the exact re-run settles (counted); a clean call must equal the exact result bit for bit."""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
rng = np.random.default_rng(seed)
bad = fallbacks = fused_calls = 0
t0 = time.time()
for case in range(cases):
    if time.time() - t0 > budget: break
    f16 = rng.random() < 0.65
    d = int(rng.choice([256, 384, 512, 640, 768, 1024, 1152, 1280, 1408, 1536])) if f16 else int(rng.choice([128, 256, 384, 512, 768]))
    size = rng.random()
    n = int(rng.integers(8200, 60_000)) if size < 0.5 else int(rng.integers(60_000, 600_000)) if size < 0.85 else int(rng.integers(600_000, 2_500_000))
    if not f16: n = min(n, 1_200_000)
    nq = (int(rng.integers(1, 5)) if d <= 768 else int(rng.integers(1, 3))) if f16 else (int(rng.integers(1, 3)) if d <= 384 else 1)
    if d > 768: n = min(n, 1_200_000)
    k = int(rng.choice([1, 5, 37, 100, 128]))
    metric = str(rng.choice(["dot_product", "cosine_similarity"]))
    g = torch.Generator(device='cuda').manual_seed(seed * 100000 + case)
    V = torch.randn((n, d), generator=g, device='cuda').to(torch.float16 if f16 else torch.float32)
    style = rng.random()
    if style < 0.15: V[torch.from_numpy(rng.integers(1, n, size=50)).cuda()] = V[0].clone()          # duplicate rows
    elif style < 0.30:                                                                                # a cluster near the first query
        c0 = int(rng.integers(0, n - 300)); V[c0:c0 + 300] = (V[c0:c0 + 1].float() + 0.05 * torch.randn((300, d), generator=g, device='cuda')).to(V.dtype)
    Q = torch.randn((nq, d), generator=g, device='cuda').to(V.dtype).float()
    if style >= 0.15 and style < 0.30: Q[0] = V[c0].float()
    ix = GpuIndex(V)
    if rng.random() < 0.4: ix.set_bias((torch.rand(n, generator=g, device='cuda') * 0.3).float())
    if rng.random() < 0.25: ix.set_row_mask((torch.rand(n, generator=g, device='cuda') < float(rng.choice([0.5, 0.05]))).to(torch.uint8))
    mid = METRIC_IDS[metric]
    fi, fs, fst = ix.topk_device(Q, k, mid)
    fused_calls += ix.stat('fused')                  # (float32 d = 512 below 1.5 M rows takes the five-kernel pipeline: still compared)
    ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
    st = fst.cpu().numpy()
    ok = True
    for q in range(nq):
        if st[q] != 0: fallbacks += 1; continue
        if not (torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q])): ok = False
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} d={d} f16={f16} nq={nq} k={k} {metric} fused={ix.stat('fused')} status={st.tolist()}", flush=True)
    ix.close(); del V
    if case % 100 == 99: print(f"{case + 1} cases, {fused_calls} single-launch calls, {fallbacks} queries with a status, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {case + 1} cases, {fused_calls} single-launch calls, {fallbacks} queries with a status, {bad} mismatches")
sys.exit(1 if bad else 0)
