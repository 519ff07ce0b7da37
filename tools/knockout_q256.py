"""Where the time of the 256-query MFMA pass goes: knock-out builds of hdb_mfma.hip (HDB_MFMA_KNOCKOUT) timed against the
product library on the same GPU.

  python tools/knockout_q256.py build     # anywhere with hipcc (no GPU needed): local-hyperdb_amd/lib/knockout/lib_<v>.so
  python tools/knockout_q256.py run       # on an MI355X: kernel time of N=10M d=384 Q=256 dot per library
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'knockout')
VARIANTS = {1: "no survivor append", 2: "no LDS-DMA after priming", 3: "no append, no DMA", 7: "no append, no DMA, no barrier", 8: "half the fragment reads", 11: "no append, no DMA, half the fragment reads"}
CHILD = r'''
import sys
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d, q = 10_000_000, 384, 256
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(q, d, torch.float16, dev)
mid = METRIC_IDS['dot_product']
for _ in range(10): ix.topk_device(Q, 100, mid)
ix.set_option('profile', 1); torch.cuda.synchronize()
for _ in range(20): ix.topk_device(Q, 100, mid)
torch.cuda.synchronize()
ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches')
print(f"kernel {ns/l/1e3:.1f} us = {2*q*n*d/(ns/l)/1e3:.0f} TFLOP/s", flush=True)
'''

def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(CSRC, 'obj', f) for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_mfma_d384.o']
    procs = []
    for v in VARIANTS:
        o = os.path.join(OUT, f'mfma_{v}.o')
        procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed',
                                       f'-DHDB_MFMA_KNOCKOUT={v}', '-c', os.path.join(CSRC, 'hdb_mfma_d384.hip'), '-o', o]))
    for p in procs:
        if p.wait(): raise SystemExit('hipcc failed')
    for v in VARIANTS:
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, f'lib_{v}.so'),
                               os.path.join(OUT, f'mfma_{v}.o')] + objs)
        os.remove(os.path.join(OUT, f'mfma_{v}.o'))
    print('built', sorted(os.listdir(OUT)))

def run():
    for v in [0] + list(VARIANTS):
        env = dict(os.environ)
        if v: env['HYPERDB_HIP_LIB'] = os.path.join(OUT, f'lib_{v}.so')
        print(f"{'product' if v == 0 else VARIANTS[v]}: ", end='', flush=True)
        subprocess.run([sys.executable, '-c', CHILD], env=env, cwd=ROOT, timeout=300, stderr=subprocess.DEVNULL)

if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1] if len(sys.argv) > 1 else 'run']()
