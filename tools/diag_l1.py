"""Replays one case of tests/test_gpu_parity.py::test_fuzz_gpu_against_oracle with the manhattan tile kernel's two flavours (diagnostic)."""
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np
from hyperdb._native import GpuIndex, METRIC_IDS
from oracle import ranking_oracle as orc
GPU_METRICS = ["dot_product", "cosine_similarity", "euclidean_metric", "hamming_distance", "manhattan_distance", "jaccard_similarity", "pearson_correlation"]
import re
src = open('tests/test_gpu_parity.py').read()
m = re.search(r"^GPU_METRICS = (\[.*?\])", src, re.S | re.M)
if m: GPU_METRICS = eval(m.group(1))
want = int(sys.argv[1]) if len(sys.argv) > 1 else 33
rng = np.random.default_rng(20261004)
combos = [(m_, dt) for m_ in GPU_METRICS for dt in (np.float16, np.float32, np.float64)]
cases = combos + combos + [combos[i] for i in rng.permutation(len(combos))[:21]]
for case, (metric, dt) in enumerate(cases):
    d = int(rng.choice([128, 256, 384, 512, 768, 1024, 24, 100, 33, 200]))
    n = int(rng.integers(8200, 40_000)) if rng.random() < 0.8 else int(rng.integers(2, 8192))
    nq = int(rng.choice([1, 1, 2, 5, 9, 33]))
    k = int(rng.choice([1, 5, 40, 100]))
    with_bias = case >= len(combos) and case < 2 * len(combos)
    V = rng.standard_normal((n, d)).astype(np.float32).astype(dt)
    if rng.random() < 0.2:
        V[rng.integers(0, n, size=20)] = V[0]
    Q = rng.standard_normal((nq, d)).astype(np.float32).astype(dt)
    if rng.random() < 0.3:
        Q[0] = V[n // 2]
    bias = None
    if with_bias:
        ts = 1.7e9 + rng.uniform(0, 86400.0, size=n)
        bias = 0.4 * np.exp(ts - ts.max())
    if case != want:
        continue
    print("case", case, metric, np.dtype(dt).name, "n", n, "d", d, "nq", nq, "k", k, "bias", with_bias, flush=True)
    ix = GpuIndex(V)
    if with_bias: ix.set_recency(ts, 0.4)
    for pk in (0, 1):
        ix.set_option("l1_packed", pk)
        idx, sc = ix.topk(Q, min(k, n), METRIC_IDS[metric])
        for qi in range(nq):
            ex = orc.exact_scores(V[idx[qi]], Q[qi].copy(), metric) + (bias[idx[qi]] if bias is not None else 0)
            bad = np.nonzero(np.abs(ex - sc[qi]) > 1e-3)[0]
            print(" packed", pk, "query", qi, "rows", idx[qi][:6], "scores", sc[qi][:6], "exact", ex[:6], "bad positions", bad[:10], flush=True)
    ix.close()
