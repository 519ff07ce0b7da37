"""The single-launch BATCHED pipeline (hdb_mfma_kernel.h, MODE 2) against the five-kernel pipeline and the on-device exact
selection, bit for bit, over a list of shapes; then timings of both.  usage: python tools/check_batch1.py [quick]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd"))
import numpy as np
import torch
from hyperdb._native import GpuIndex, METRIC_IDS

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
g = torch.Generator(device="cuda").manual_seed(5)
bad = 0
shapes = [(np.float16, 200_003, 384, [5, 16, 100, 256, 300]), (np.float16, 60_017, 768, [1, 7, 64, 128, 130]),
          (np.float16, 30_000, 128, [40, 256]), (np.float16, 90_001, 512, [33, 200]), (np.float16, 50_000, 1536, [3, 32]),
          (np.float16, 8_193, 384, [9]), (np.float16, 1_300_001, 384, [17]), (np.float32, 120_000, 384, [5, 64, 100, 130]),
          (np.float32, 70_000, 768, [8, 128])]
if quick:
    shapes = shapes[:2]
for dt, n, d, nqs in shapes:
    V = torch.randn((n, d), generator=g, device="cuda").to(torch.float16 if dt == np.float16 else torch.float32)
    V[n - 1] = V[7]
    ix = GpuIndex(V)
    bias = (torch.rand(n, generator=g, device="cuda") * 0.2).float()
    mask = (torch.rand(n, generator=g, device="cuda") < 0.3).to(torch.uint8)
    for nq in nqs:
        Q = torch.randn((nq, d), generator=g, device="cuda").float()
        Q[0] = V[n // 3].float()
        if nq > 2:
            Q[2] = Q[2] * 37.5
        for metric in ("cosine_similarity", "dot_product", "euclidean_metric"):
            if metric != "euclidean_metric" and nq <= 4 and dt == np.float16 and d <= 768:
                continue
            if dt == np.float32 and nq < 5:
                continue
            mid = METRIC_IDS[metric]
            for setup in ("plain", "bias", "mask+bias"):
                ix.set_bias(bias if "bias" in setup else None)
                ix.set_row_mask(mask if "mask" in setup else None)
                for k in (100, 1, 128):
                    ix.set_option("use_fused", 1)
                    fi, fs, fst = ix.topk_device(Q, k, mid)
                    kind = ix.stat("fused")
                    ix.set_option("use_fused", 0)
                    ui, us, ust = ix.topk_device(Q, k, mid)
                    ix.set_option("use_fused", 1)
                    ok_st = int(fst.abs().sum().item()) == 0
                    same = torch.equal(fi, ui) and torch.equal(fs, us)
                    if kind != 2 or not ok_st or not same:
                        bad += 1
                        nbad = int((fi != ui).any(dim=1).sum().item())
                        print(f"MISMATCH {np.dtype(dt).name} n={n} d={d} nq={nq} {metric} {setup} k={k}: kind={kind} status={fst.tolist()[:8]} "
                              f"queries differing={nbad} ust={int(ust.abs().sum().item())}", flush=True)
        print(f"ok so far: {np.dtype(dt).name} n={n} d={d} nq={nq} bad={bad}", flush=True)
    ix.close()
    del V
print("MISMATCHES:", bad)

# ---- timings: one launch against five
for dt, n, d, nq, metric in [(np.float16, 10_000_000, 384, 256, "dot_product"), (np.float16, 10_000_000, 384, 16, "cosine_similarity"),
                             (np.float16, 1_250_000, 384, 8, "cosine_similarity"), (np.float16, 1_250_000, 384, 1, "euclidean_metric"),
                             (np.float16, 5_000_000, 768, 64, "euclidean_metric")]:
    if quick and n > 2_000_000:
        continue
    V = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float16)
    Q = torch.randn((nq, d), generator=g, device="cuda").float()
    ix = GpuIndex(V)
    mid = METRIC_IDS[metric]
    res = {}
    for fused in (1, 0, 1, 0):
        ix.set_option("use_fused", fused)
        for _ in range(5):
            ix.topk(Q, 100, mid)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            ix.topk(Q, 100, mid)
            ts.append(time.perf_counter() - t0)
        res.setdefault(fused, []).append(np.median(ts) * 1e6)
    print(f"n={n} d={d} nq={nq} {metric}: one launch {res[1][0]:.1f} / {res[1][1]:.1f} us, five kernels {res[0][0]:.1f} / {res[0][1]:.1f} us (p50 per call, host record)", flush=True)
    ix.close()
    del V
sys.exit(1 if bad else 0)
