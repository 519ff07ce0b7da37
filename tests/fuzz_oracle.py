"""GPU result vs the numpy oracle (float64-exact comparator) on random shapes: all seven metrics, fp16/fp32/fp64,
MFMA and odd dimensions, single queries and batches, optional recency bias.  Test infrastructure (imports oracle/)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'local-hyperdb_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
from oracle import ranking_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
metrics = list(METRIC_IDS)
bad = 0
t0 = time.time()
for case in range(cases):
    dt = rng.choice([np.float16, np.float32, np.float64], p=[0.5, 0.4, 0.1])
    d = int(rng.choice([128, 256, 384, 512, 768, 1024, 24, 100, 33, 200]))
    n = int(rng.integers(8200, 40_000)) if rng.random() < 0.8 else int(rng.integers(2, 8192))
    nq = int(rng.choice([1, 1, 2, 5, 9, 33]))
    k = int(rng.choice([1, 5, 40, 100]))
    metric = str(rng.choice(metrics))
    V = rng.standard_normal((n, d)).astype(np.float32).astype(dt)
    if rng.random() < 0.2: V[rng.integers(0, n, size=20)] = V[0]
    Q = rng.standard_normal((nq, d)).astype(np.float32).astype(dt)
    if rng.random() < 0.3: Q[0] = V[n // 2]
    bias = None
    ix = GpuIndex(V)
    try:
        if rng.random() < 0.3:
            ts = 1.7e9 + rng.uniform(0, 86400.0, size=n)
            ix.set_recency(ts, 0.4)
            bias = 0.4 * np.exp(ts - ts.max())
        idx, sc = ix.topk(Q, min(k, n), METRIC_IDS[metric])
        tol = 0.0 if (metric == "hamming_distance" and bias is None) else (1e-3 if dt == np.float16 else 1e-5)
        if metric == "hamming_distance" and bias is not None: tol = 1e-5
        for qi in range(nq):
            orc.check_topk(idx[qi], sc[qi], V, Q[qi].copy(), metric, k, bias=bias, tol=tol)
    except AssertionError as e:
        bad += 1
        print("FAIL", dict(case=case, n=n, d=d, dt=np.dtype(dt).name, nq=nq, k=k, metric=metric, bias=bias is not None,
                           mfma=ix.stat("mfma"), path=ix.stat("path")), str(e)[:200], flush=True)
    finally:
        ix.close()
    if case % 25 == 24: print(f"{case + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
print(f"done: {cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
