"""Phase timeline of the single-launch top-k kernel (diagnostic build HDB_FUSED_STAMPS=1): where the fixed cost of a
call goes.  build: needs hipcc; run: on an MI355X."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'stamps')

def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(CSRC, 'obj', f) for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_mfma_fused.o']
    o = os.path.join(OUT, 'fused.o')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed', '-DHDB_FUSED_STAMPS=1',
                           '-c', os.path.join(CSRC, 'hdb_mfma_fused.hip'), '-o', o])
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'lib.so'), o] + objs)
    os.remove(o)
    print('built', os.path.join(OUT, 'lib.so'))

def run():
    os.environ['HYPERDB_HIP_LIB'] = os.path.join(OUT, 'lib.so')
    sys.path.insert(0, os.path.join(ROOT, 'local-hyperdb_amd')); sys.path.insert(0, ROOT)
    import ctypes, time
    import numpy as np, torch
    from hyperdb import _native
    from hyperdb._native import GpuIndex, METRIC_IDS
    import bench
    dev = torch.device('cuda', 0)
    lib = _native.lib()
    lib.hdb_debug_read_fused_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    sizes = [(int(x), torch.float16) for x in sys.argv[2:]] or [(1_250_000, torch.float16), (10_000_000, torch.float16), (1_000_000, torch.float32)]
    for (n, dt) in sizes:
      V, lo, hi = bench.make_shard(n, 384, dt, 0, 1, dev)
      for unc in (0,):
        ix = GpuIndex(V)
        Q = bench.make_queries(64, 384, dt, dev).to(torch.float32)
        mid = METRIC_IDS['cosine_similarity']
        rows = []
        for i in range(40):
            t0 = time.perf_counter(); ix.topk_views(Q[i:i + 1], 100, mid); wall = (time.perf_counter() - t0) * 1e6
            if i < 10: continue
            buf = (ctypes.c_uint64 * (16 * 256))()
            lib.hdb_debug_read_fused_stamps(buf, 256)
            a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
            t_start = a[:, 0].min()
            rel = (a - t_start) / 100.0          # us
            last = int(np.argmax(a[:, 6]))
            if i == 39:
                print('   thr known: WG0', round(float(rel[0, 3]), 1), 'percentiles', [round(float(np.percentile(rel[1:, 3], p)), 1) for p in (0, 10, 50, 90, 100)], 'published WG0', round(float(rel[0, 2]), 1), flush=True)
                order = np.argsort(rel[:, 4])
                print('   per-WG loop-done percentiles (us):', [round(float(np.percentile(rel[:, 4], p)), 1) for p in (0, 10, 50, 90, 99, 100)],
                      ' tiles generated min/med/max:', int(a[:, 7].min()), int(np.median(a[:, 7])), int(a[:, 7].max()),
                      ' tiles of the 5 last finishers:', a[order[-5:], 7].tolist(), ' of the 5 first:', a[order[:5], 7].tolist(), flush=True)
            rows.append([wall, rel[:, 0].max(), np.median(rel[:, 1]), np.median(rel[:, 2]), rel[:, 2].max(), np.median(rel[:, 3]), rel[:, 3].max(),
                         np.median(rel[:, 4]), rel[:, 4].max(), rel[:, 5].max(), rel[last, 6], rel[last, 5], rel[last, 8], rel[last, 9], rel[last, 10],
                         np.median(rel[:, 11]), np.median(rel[:, 12]), np.median(rel[:, 13]), np.median(rel[:, 14]), np.median(rel[:, 15])])
        r = np.median(np.array(rows), axis=0)
        names = ['host wall', 'last start', 'prologue done (med)', 'published (med)', 'published (max)', 'thr known (med)', 'thr known (max)',
                 'loop done (med)', 'loop done (max)', 'ticket (max)', 'finalize done', 'last WG: ticket', 'fin start', 'cands loaded', 'preselected', 'filter tile 1 done (med)', 'tile 4', 'tile 8', 'tile 16', 'tile 32']
        print(f"n={n} {dt}: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(names, r)), flush=True)
        ix.close()
      del V; torch.cuda.empty_cache()

if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
