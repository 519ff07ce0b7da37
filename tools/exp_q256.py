"""A/B timing of experimental builds of the 256-query MFMA pass (HDB_MFMA_EXP, whatever hdb_mfma_kernel.h currently hangs on it):
one library per variant, timed in interleaved rounds on one GPU (kernel time from HIP events, N=10M d=384 Q=256 dot),
each run also checked against the single-query VALU scan for three queries.

  python tools/exp_q256.py build 0 1 2 3 5     # needs hipcc only
  python tools/exp_q256.py run OUT.json 0 1 2 3 5
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'exp')
CHILD = r'''
import sys, json
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d, q = 10_000_000, 384, 256
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(q, d, torch.float16, dev)
mid = METRIC_IDS['dot_product']
bi, bs, st = ix.topk_device(Q, 100, mid)
ok = int(st.abs().sum().item()) == 0
ix.set_option('use_mfma', 0)
for qi in (0, 100, 255):
    si, ss, _ = ix.topk_device(Q[qi:qi + 1], 100, mid)
    ok = ok and torch.equal(si[0], bi[qi]) and torch.allclose(ss[0], bs[qi], rtol=2e-6, atol=2e-5)
ix.set_option('use_mfma', 1)
for _ in range(30): ix.topk_device(Q, 100, mid)
times = []
for rep in range(4):
    ix.set_option('profile', 1); torch.cuda.synchronize()
    for _ in range(25): ix.topk_device(Q, 100, mid)
    torch.cuda.synchronize()
    times.append(ix.stat('scan_time_ns') / ix.stat('scan_launches') / 1e3)
    ix.set_option('profile', 0)
print(json.dumps({"parity_ok": bool(ok), "kernel_us": times}), flush=True)
'''

def build(variants):
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(CSRC, 'obj', f) for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_mfma_d384.o']
    procs = []
    for v in variants:
        o = os.path.join(OUT, f'mfma_{v}.o')
        procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed',
                                       f'-DHDB_MFMA_EXP={v}', '-c', os.path.join(CSRC, 'hdb_mfma_d384.hip'), '-o', o]))
        if len(procs) % 4 == 0:
            for p in procs[-4:]:
                if p.wait(): raise SystemExit('hipcc failed')
    for p in procs:
        if p.wait(): raise SystemExit('hipcc failed')
    for v in variants:
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, f'lib_{v}.so'),
                               os.path.join(OUT, f'mfma_{v}.o')] + objs)
        os.remove(os.path.join(OUT, f'mfma_{v}.o'))
    print('built', sorted(os.listdir(OUT)))

def run(dst, variants):
    res = []
    for rnd in range(2):
        for v in variants:
            env = dict(os.environ, HYPERDB_HIP_LIB=os.path.join(OUT, f'lib_{v}.so'))
            r = subprocess.run([sys.executable, '-c', CHILD], env=env, cwd=ROOT, timeout=300, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith('{')]
            rec = json.loads(line[-1]) if line else {"error": (r.stdout + r.stderr)[-400:]}
            rec.update(exp=v, round=rnd)
            res.append(rec)
            print(rec, flush=True)
    json.dump(res, open(dst, 'w'), indent=1)

if __name__ == '__main__':
    if sys.argv[1] == 'build': build([int(x) for x in sys.argv[2:]])
    else: run(sys.argv[2], [int(x) for x in sys.argv[3:]])
