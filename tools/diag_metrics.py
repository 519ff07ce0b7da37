"""Every metric once on a seeded float32 / fp16 matrix, one line per call (flushed): which call a crash belongs to."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np
import hyperdb.ranking_algorithm as ranking
dtype = np.float32 if len(sys.argv) < 2 or sys.argv[1] == "fp32" else np.float16
n, d = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100_000, 384)
rng = np.random.default_rng(n + d)
V = rng.standard_normal((n, d)).astype(np.float32).astype(dtype)
h = ranking.register_vectors(V)
for metric in ("dot_product", "cosine_similarity", "euclidean_metric", "hamming_distance", "manhattan_distance", "jaccard_similarity", "pearson_correlation"):
    for qi in range(2):
        q = (rng.standard_normal(d) if qi == 0 else V[n // 5].astype(np.float64) + 0.1 * rng.standard_normal(d)).astype(dtype)
        print("call", metric, qi, flush=True)
        idx, sc = ranking.hyperDB_ranking_algorithm_sort(h, q.copy(), top_k=100, metric=metric)
        print("  ok", idx[:3], sc[:3], "fused", h.index.stat("fused"), "path", h.index.stat("path"), flush=True)
h.close()
