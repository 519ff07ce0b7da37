import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
seed = 7
want = set(int(x) for x in sys.argv[1:])
rng = np.random.default_rng(seed)
for case in range(max(want) + 1):
    f16 = rng.random() < 0.65
    d = int(rng.choice([128, 256, 384, 512, 640, 768])) if f16 else int(rng.choice([128, 256, 384]))
    size = rng.random()
    n = int(rng.integers(8200, 60_000)) if size < 0.5 else int(rng.integers(60_000, 600_000)) if size < 0.85 else int(rng.integers(600_000, 2_500_000))
    if not f16: n = min(n, 1_200_000)
    nq = int(rng.integers(1, 5)) if f16 else int(rng.integers(1, 3))
    k = int(rng.choice([1, 5, 37, 100, 128]))
    metric = str(rng.choice(["dot_product", "cosine_similarity"]))
    style = None
    if case in want:
        g = torch.Generator(device='cuda').manual_seed(seed * 100000 + case)
        V = torch.randn((n, d), generator=g, device='cuda').to(torch.float16 if f16 else torch.float32)
    style = rng.random()
    c0 = None
    if style < 0.15:
        ii = rng.integers(1, n, size=50)
        if case in want: V[torch.from_numpy(ii).cuda()] = V[0].clone()
    elif style < 0.30:
        c0 = int(rng.integers(0, n - 300))
        if case in want: V[c0:c0 + 300] = (V[c0:c0 + 1].float() + 0.05 * torch.randn((300, d), generator=g, device='cuda')).to(V.dtype)
    if case in want:
        Q = torch.randn((nq, d), generator=g, device='cuda').to(V.dtype).float()
        if c0 is not None: Q[0] = V[c0].float()
        ix = GpuIndex(V)
    r1 = rng.random(); r2 = rng.random()
    has_bias = r1 < 0.4; has_mask = r2 < 0.25
    frac = float(rng.choice([0.5, 0.05])) if has_mask else None
    if case in want:
        if has_bias: ix.set_bias((torch.rand(n, generator=g, device='cuda') * 0.3).float())
        if has_mask: ix.set_row_mask((torch.rand(n, generator=g, device='cuda') < frac).to(torch.uint8))
        mid = METRIC_IDS[metric]
        print(f"case {case}: n={n} d={d} f16={f16} nq={nq} k={k} {metric} style={style:.2f} c0={c0} bias={has_bias} mask={has_mask}/{frac}")
        for rep in range(3):
            fi, fs, fst = ix.topk_device(Q, k, mid)
            print("  fused", ix.stat('fused'), "status", fst.tolist(), "sample_rows", ix.stat('sample_rows'))
            ei, es, _ = ix.topk_device(Q, k, mid, exact=True)
            ix.set_option('use_fused', 0); ui, us, ust = ix.topk_device(Q, k, mid); ix.set_option('use_fused', 1)
            for q in range(nq):
                if not (torch.equal(fi[q], ei[q]) and torch.equal(fs[q], es[q])):
                    dif = (fi[q] != ei[q]).nonzero().flatten().tolist()
                    fset, eset = set(fi[q].tolist()), set(ei[q].tolist())
                    print(f"   only in fused {sorted(fset - eset)[:6]} only in exact {sorted(eset - fset)[:6]} n_dif {len(dif)}")
                    print(f"   q{q}: differs at ranks {dif[:8]} fused idx {fi[q][dif[:4]].tolist()} sc {fs[q][dif[:4]].tolist()} | exact idx {ei[q][dif[:4]].tolist()} sc {es[q][dif[:4]].tolist()} | unfused equal exact: {torch.equal(ui[q], ei[q])}")
        ix.close(); del V
