"""Phase timeline of the single-launch hamming / jaccard top-k (hdb_bits_fused.hip, diagnostic build HDB_BITS_STAMPS=1)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'stamps')

def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    names = [f[:-2] for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_bits_fused.o']
    objs = [os.path.join(CSRC, 'obj', n + '.o') for n in names]
    o = os.path.join(OUT, 'bits.o')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed', '-DHDB_BITS_STAMPS=1',
                           '-c', os.path.join(CSRC, 'hdb_bits_fused.hip'), '-o', o])
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'lib_bits.so'), o] + objs)
    os.remove(o)
    print('built', os.path.join(OUT, 'lib_bits.so'))

def run():
    os.environ['HYPERDB_HIP_LIB'] = os.path.join(OUT, 'lib_bits.so')
    sys.path.insert(0, os.path.join(ROOT, 'local-hyperdb_amd')); sys.path.insert(0, ROOT)
    import ctypes, time
    import numpy as np, torch
    from hyperdb import _native
    from hyperdb._native import GpuIndex, METRIC_IDS
    import bench
    dev = torch.device('cuda', 0)
    lib = _native.lib()
    lib.hdb_debug_read_bits_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    names = ['start(max)', 'prologue', 'sample done', 'published', 'thr known', 'filter done', 'all waves done', 'all arrived', 'sorted', 'fenced', 'counted']
    for (n, nq) in ((10_000_000, 1), (10_000_000, 4), (1_250_000, 1)):
        V, lo, hi = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
        ix = GpuIndex(V)
        ix.set_option('bits_fused', 2)
        Q = bench.make_queries(nq, 384, torch.float16, dev).to(torch.float32)
        mid = METRIC_IDS['hamming_distance']
        for loc in (1, 0):
            ix.set_option('bits_local', loc)
            print(f"--- bits_local = {loc}")
            rows = []
            for i in range(30):
                t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); wall = (time.perf_counter() - t0) * 1e6
                assert ix.stat('fused') == 3
                if i < 10: continue
                buf = (ctypes.c_uint64 * (16 * 256))()
                lib.hdb_debug_read_bits_stamps(buf, 256)
                a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
                rel = (a - a[:, 0].min()) / 100.0
                if i == 29:
                    print('   filter done percentiles 0/10/50/90/100:', [round(float(np.percentile(rel[:, 5], p)), 1) for p in (0, 10, 50, 90, 100)],
                          ' by blockIdx % 8:', [round(float(np.median(rel[x::8, 5])), 1) for x in range(8)], flush=True)
                rows.append([wall, rel[:, 0].max()] + [np.median(rel[:, j]) for j in range(1, 11)] + [rel[:, j].max() for j in range(1, 11)])
            r = np.median(np.array(rows), axis=0)
            print(f"n={n} nq={nq} hamming: host wall {r[0]:.1f} us; median over workgroups: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(names, r[1:12])), flush=True)
            print("      max over workgroups: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(names[1:], r[12:])), flush=True)
        ix.close(); del V; torch.cuda.empty_cache()

if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
