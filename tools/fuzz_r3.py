"""Fuzz of the round-3 single-launch paths against the multi-kernel pipeline of the same library (and the exact selection where the
two rank alike): 1-4 dot / cosine / pearson queries and one euclidean query (fp16), 1-2 of each on float32 (hdb_mfma_fused.h);
5-300 dot / cosine / euclidean queries (hdb_mfma_kernel.h MODE 2); 1-7 hamming / jaccard queries (hdb_bits_fused.hip).
Random shapes, k, bias, row mask, duplicate rows, a cluster around the first query.  A query that reports a status is one the
exact re-run settles (counted); a clean one must equal the reference bit for bit.
  python tools/fuzz_r3.py CASES SEED BUDGET_SECONDS [case,case,...: replay only these]"""
import sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
only = set(int(x) for x in sys.argv[4].split(',')) if len(sys.argv) > 4 else None
rng = np.random.default_rng(seed)
bad = fallbacks = 0
kinds = {0: 0, 1: 0, 2: 0, 3: 0}
t0 = time.time()
case = -1
for case in range(cases):
    if time.time() - t0 > budget: break
    family = rng.choice(["few", "batch", "bits"], p=[0.45, 0.25, 0.30])
    f16 = rng.random() < 0.6
    if family == "bits":
        d = int(rng.choice([50, 100, 128, 384, 768, 1000, 1536]))
        metric = str(rng.choice(["hamming_distance", "jaccard_similarity"]))
        nq = int(rng.integers(1, 8))
    elif family == "batch":
        d = int(rng.choice([128, 256, 384, 512, 768, 1024, 1536])) if f16 else int(rng.choice([128, 256, 384, 512, 768]))
        metric = str(rng.choice(["dot_product", "cosine_similarity", "euclidean_metric", "pearson_correlation"]))
        nq = int(rng.choice([5, 8, 16, 33, 64, 100, 130, 256, 300]))
    else:
        d = int(rng.choice([256, 384, 512, 640, 768, 1024, 1152, 1280, 1408, 1536])) if f16 else int(rng.choice([128, 256, 384, 512, 768]))
        metric = str(rng.choice(["dot_product", "cosine_similarity", "euclidean_metric", "pearson_correlation"]))
        nq = (int(rng.integers(1, 5)) if d <= 768 else int(rng.integers(1, 3))) if f16 else (int(rng.integers(1, 3)) if d <= 384 else 1)
    size = rng.random()
    n = int(rng.integers(8200, 60_000)) if size < 0.5 else int(rng.integers(60_000, 600_000)) if size < 0.85 else int(rng.integers(600_000, 2_500_000))
    n = min(n, int(1.2e9 // (d * (2 if f16 else 4))))
    if family == "batch": n = min(n, 400_000)
    k = int(rng.choice([1, 5, 37, 100, 128]))
    # every random draw of the case first (so that a list of case numbers can be replayed: argv[4])
    style = rng.random()
    dup_rows, c0, mask_p = None, 0, 0.0
    if style < 0.15: dup_rows = rng.integers(1, n, size=50)
    elif style < 0.30: c0 = int(rng.integers(0, n - 300))
    with_bias = rng.random() < 0.4
    with_mask = rng.random() < 0.25
    if with_mask: mask_p = float(rng.choice([0.5, 0.05]))
    if only is not None and case not in only: continue
    g = torch.Generator(device='cuda').manual_seed(seed * 100000 + case)
    V = torch.randn((n, d), generator=g, device='cuda').to(torch.float16 if f16 else torch.float32)
    if style < 0.15: V[torch.from_numpy(dup_rows).cuda()] = V[0].clone()                              # duplicate rows
    elif style < 0.30:                                                                                # a cluster near the first query
        V[c0:c0 + 300] = (V[c0:c0 + 1].float() + 0.05 * torch.randn((300, d), generator=g, device='cuda')).to(V.dtype)
    Q = torch.randn((nq, d), generator=g, device='cuda').to(V.dtype).float()
    if 0.15 <= style < 0.30: Q[0] = V[c0].float()
    ix = GpuIndex(V)
    if with_bias: ix.set_bias((torch.rand(n, generator=g, device='cuda') * 0.3).float())
    if with_mask: ix.set_row_mask((torch.rand(n, generator=g, device='cuda') < mask_p).to(torch.uint8))
    mid = METRIC_IDS[metric]
    ix.set_option("use_fused", 1)
    fi, fs, fst = ix.topk_device(Q, k, mid)
    kind = ix.stat('fused')
    kinds[kind] += 1
    ix.set_option("use_fused", 0)
    ri, rs, rst = ix.topk_device(Q, k, mid)                     # the multi-kernel pipeline
    # (euclidean through the matrix cores -- fp16, and float32 batches: the exact selection ranks before the near-duplicate re-score)
    use_exact = metric != "euclidean_metric" or (not f16 and family == "few")
    if use_exact: ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
    st, st0 = fst.cpu().numpy(), rst.cpu().numpy()
    why = []
    for q in range(nq):
        if st[q] != 0: fallbacks += 1; continue
        if st0[q] == 0 and not (torch.equal(fi[q], ri[q]) and torch.equal(fs[q], rs[q])): why.append(f"q{q}: differs from the multi-kernel pipeline")
        if use_exact and not (torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q])): why.append(f"q{q}: differs from the exact selection")
    if why:
        bad += 1
        print(f"MISMATCH case {case}: {family} n={n} d={d} f16={f16} nq={nq} k={k} {metric} kind={kind} style={style:.2f} bias={with_bias} mask={with_mask} status={st.tolist()[:8]} {why[:3]}", flush=True)
    ix.close(); del V
    if case % 100 == 99: print(f"{case + 1} cases, launches by kind {kinds}, {fallbacks} queries with a status, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {case + 1} cases, launches by kind {kinds} (1 = hdb_mfma_fused, 2 = batched single launch, 3 = bit metrics, 0 = multi-kernel), "
      f"{fallbacks} queries with a status, {bad} mismatches")
sys.exit(1 if bad else 0)
