"""Variants of the single-launch bit-metric kernel (hdb_bits_fused.hip): waves per workgroup (HDB_BITS_THREADS) and quarter
chunks for the tail of the pass (HDB_BITS_TAIL), each as a library of its own, timed in interleaved rounds in one job.
  python tools/exp_bits_variants.py build      # hipcc only
  python tools/exp_bits_variants.py run        # on an MI355X
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'bitsvar')
# name: (threads per workgroup, tail chunks on/off, chunk / tail chunk, tail share in %, prefetch point, poll, non-temporal loads of the sign bits)
VARIANTS = {'nt_loads': (1024, 1, 4, 15, 1, 0, 1), 'plain_loads': (1024, 1, 4, 15, 1, 0, 0)}
CHILD = r'''
import sys, time, json
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
out = {}
for n in (10_000_000, 5_000_000, 1_250_000):
    V, lo, hi = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
    ix = GpuIndex(V)
    mid = METRIC_IDS['hamming_distance']
    for nq in (1, 4):
        Q = bench.make_queries(nq, 384, torch.float16, dev).to(torch.float32)
        for kind in ('single', 'six'):
            ix.set_option('use_fused', 1 if kind == 'single' else 0); ix.set_option('bits_fused', 2 if kind == 'single' else 1)
            for _ in range(20): ix.topk_views(Q, 100, mid)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter(); ix.topk_views(Q, 100, mid); ts.append(time.perf_counter() - t0)
            out[f'{n} nq={nq} {kind}'] = round(float(np.median(ts)) * 1e6, 1)
    ix.close(); del V; torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
'''

def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(CSRC, 'obj', f) for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_bits_fused.o']
    procs = []
    for name, (thr, tail, tdiv, tpct, pref, poll, nt) in VARIANTS.items():
        o = os.path.join(OUT, name + '.o')
        procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed', f'-DHDB_BITS_THREADS={thr}',
                                       f'-DHDB_BITS_TAIL={tail}', f'-DHDB_BITS_TAILDIV={tdiv}', f'-DHDB_BITS_TAILPCT={tpct}', f'-DHDB_BITS_PREF={pref}', f'-DHDB_BITS_POLL={poll}', f'-DHDB_BITS_NT={nt}',
                                       '-c', os.path.join(CSRC, 'hdb_bits_fused.hip'), '-o', o]))
    for p in procs:
        if p.wait(): raise SystemExit('hipcc failed')
    for name in VARIANTS:
        o = os.path.join(OUT, name + '.o')
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, f'lib_{name}.so'), o] + objs)
        os.remove(o)
    print('built', sorted(os.listdir(OUT)))

def run():
    import json
    for rnd in range(3):
        for name in VARIANTS:
            env = dict(os.environ, HYPERDB_HIP_LIB=os.path.join(OUT, f'lib_{name}.so'))
            r = subprocess.run([sys.executable, '-c', CHILD], env=env, cwd=ROOT, timeout=400, stderr=subprocess.DEVNULL, stdout=subprocess.PIPE, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith('{')]
            print(name, 'round', rnd, line[-1] if line else r.stdout[-300:], flush=True)

if __name__ == '__main__':
    build() if sys.argv[1] == 'build' else run()
