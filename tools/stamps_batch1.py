"""Phase timeline of the single-launch BATCHED top-k (hdb_mfma_kernel.h MODE 2, diagnostic build HDB_BATCH_STAMPS=1): where
the fixed cost of a call goes.  build: needs hipcc; run: on an MI355X."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'stamps')

def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(CSRC, 'obj', f) for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_mfma_d384.o']
    o = os.path.join(OUT, 'mfma.o')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed', '-DHDB_BATCH_STAMPS=1',
                           '-c', os.path.join(CSRC, 'hdb_mfma_d384.hip'), '-o', o])
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'lib_batch.so'), o] + objs)
    os.remove(o)
    print('built', os.path.join(OUT, 'lib_batch.so'))

def run():
    os.environ['HYPERDB_HIP_LIB'] = os.path.join(OUT, 'lib_batch.so')
    sys.path.insert(0, os.path.join(ROOT, 'local-hyperdb_amd')); sys.path.insert(0, ROOT)
    import ctypes, time
    import numpy as np, torch
    from hyperdb import _native
    from hyperdb._native import GpuIndex, METRIC_IDS
    import bench
    dev = torch.device('cuda', 0)
    lib = _native.lib()
    lib.hdb_debug_read_batch_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    names = ['start(max)', 'prepared', 'sample done', 'published', 'owner done', 'thr known', 'pass done', 'arrived', 'all arrived', 'sorted', 'left']
    for (n, nq, metric) in ((1_250_000, 8, 'cosine_similarity'), (1_250_000, 5, 'euclidean_metric'), (10_000_000, 256, 'dot_product'), (10_000_000, 16, 'cosine_similarity')):
        V, lo, hi = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
        ix = GpuIndex(V)
        Q = bench.make_queries(nq, 384, torch.float16, dev).to(torch.float32)
        mid = METRIC_IDS[metric]
        rows = []
        for i in range(30):
            t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); wall = (time.perf_counter() - t0) * 1e6
            assert ix.stat('fused') == 2
            if i < 10: continue
            buf = (ctypes.c_uint64 * (16 * 256))()
            lib.hdb_debug_read_batch_stamps(buf, 256)
            a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
            rel = (a - a[:, 0].min()) / 100.0          # us
            if i == 29:
                for j in (5, 6, 7, 8, 9):
                    print(f"   {names[j]}: percentiles 0/10/25/50/75/90/100:", [round(float(np.percentile(rel[:, j], p)), 1) for p in (0, 10, 25, 50, 75, 90, 100)],
                          " by blockIdx % 8:", [round(float(np.median(rel[x::8, j])), 1) for x in range(8)], flush=True)
            if i == 29:
                own = np.nonzero(a[:, 13] > 0)[0][:nq]
                if own.size:
                    print("   owners: first sweep back", [round(float(x), 1) for x in rel[own[:6], 11]], "attempts", a[own[:6], 12].tolist(), "all tagged", [round(float(x), 1) for x in rel[own[:6], 13]],
                          "extracted", [round(float(x), 1) for x in rel[own[:6], 14]], "owner done", [round(float(x), 1) for x in rel[own[:6], 4]], flush=True)
            rows.append([wall] + [rel[:, 0].max()] + [np.median(rel[:, j]) for j in range(1, 11)] + [rel[:, j].max() for j in range(1, 11)])
        r = np.median(np.array(rows), axis=0)
        print(f"n={n} nq={nq} {metric}: host wall {r[0]:.1f} us; median over workgroups: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(names, r[1:12])), flush=True)
        print("      max over workgroups: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(names[1:], r[12:])), flush=True)
        ix.close()
        del V; torch.cuda.empty_cache()

if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
