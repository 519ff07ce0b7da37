"""Where the time of ONE query call goes on the host side of the single-launch pipelines: Python wrapper, C entry before the
launch, the launch call, launch -> record complete (kernel + poll), against the kernel's own duration (HIP events).
usage: python tools/time_host_path.py [n ...]   (fp16 d=384 cosine, one query, top-100)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb import _native
from hyperdb._native import GpuIndex, METRIC_IDS
import hyperdb.ranking_algorithm as ranking
import bench

dev = torch.device("cuda", 0)
lib = _native.lib()
sizes = [int(x) for x in sys.argv[1:]] or [20_000, 100_000, 1_250_000]
for metric, dt in (("cosine_similarity", torch.float16), ("hamming_distance", torch.float16), ("cosine_similarity", torch.float32)):
    for n in sizes:
        V, lo, hi = bench.make_shard(n, 384, dt, 0, 1, dev)
        ix = GpuIndex(V)
        Q = bench.make_queries(64, 384, dt, dev).to(torch.float32)
        Qh = [Q[i:i + 1].cpu().numpy() for i in range(64)]
        mid = METRIC_IDS[metric]
        for i in range(20):
            ix.topk_views(Qh[i], 100, mid)
        reps = 400
        # 1) python wrapper, host numpy query (what hyperDB_ranking_algorithm_sort(handle, q) does underneath)
        t = np.empty(reps)
        for i in range(reps):
            t0 = time.perf_counter(); ix.topk(Qh[i % 64], 100, mid); t[i] = time.perf_counter() - t0
        p_topk = np.median(t) * 1e6
        for i in range(reps):
            t0 = time.perf_counter(); ix.topk_views(Qh[i % 64], 100, mid); t[i] = time.perf_counter() - t0
        p_views = np.median(t) * 1e6
        # 2) the raw C call with everything prepared
        qt = ix._stage_host_query(Qh[0])
        slot = ix._host_records[(1, 100)]
        args = (ix._h, ctypes.c_void_p(qt.data_ptr()), 1, 100, mid, slot[1], _native._stream_ptr(dev))
        ix.set_option("host_timing_reset", 0)
        for i in range(reps):
            t0 = time.perf_counter(); lib.hdb_topk_host(*args); t[i] = time.perf_counter() - t0
        p_raw = np.median(t) * 1e6
        calls = max(ix.stat("host_calls"), 1)
        pre, launch, wait, attr = (ix.stat(k) / calls / 1e3 for k in ("host_pre_ns", "host_launch_ns", "host_wait_ns", "host_attr_ns"))
        # 3) kernel duration by HIP events
        ix.set_option("profile", 1)
        for i in range(100):
            lib.hdb_topk_host(*args)
        kern = ix.stat("scan_time_ns") / max(ix.stat("scan_launches"), 1) / 1e3
        ix.set_option("profile", 0)
        # 4) the drop-in entry
        h = ranking.register_vectors(V)
        for i in range(20):
            ranking.hyperDB_ranking_algorithm_sort(h, Qh[i][0], top_k=100, metric=metric)
        for i in range(reps):
            t0 = time.perf_counter(); ranking.hyperDB_ranking_algorithm_sort(h, Qh[i % 64][0], top_k=100, metric=metric); t[i] = time.perf_counter() - t0
        p_sort = np.median(t) * 1e6
        h.close()
        print(f"{metric[:7]} {str(dt)[6:]} n={n}: sort() {p_sort:.1f} | topk {p_topk:.1f} | topk_views {p_views:.1f} | raw C call {p_raw:.1f} = pre {pre:.1f} "
              f"(attr {attr:.1f}) + launch {launch:.1f} + wait {wait:.1f} | kernel (HIP events) {kern:.1f} | fused {ix.stat('fused')}", flush=True)
        ix.close(); del V; torch.cuda.empty_cache()
