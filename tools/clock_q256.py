"""In-kernel clock of the 256-query MFMA pass (N=10M d=384 fp16 dot): diagnostic builds of hdb_mfma.hip that stamp
s_memtime / s_memrealtime around the tile loop (HDB_MFMA_CLOCK=1), alone and combined with the knock-outs of
tools/knockout_q256.py, so that "clock or stall" can be read off the numbers:

  clock_ghz   = d(s_memtime) / d(s_memrealtime) * 0.1          (median over workgroups, last launch after >= 2 s of
                                                                back-to-back launches on random data)
  loop_us     = d(s_memrealtime) / 100                          (the tile loop alone, without launch ramp and flush)
  mfma_cycles = tiles per workgroup * 96 MFMAs per wave and tile * 2 waves per SIMD * 16 cycles
  mfma_busy   = mfma_cycles / d(s_memtime)                      (share of shader cycles the SIMD's matrix pipe is issuing)

  python tools/clock_q256.py build     # needs hipcc only
  python tools/clock_q256.py run OUT.json   # on an MI355X
The stamped builds are measurement builds: their stamps go to a buffer nothing reads; the knock-out variants compute
wrong results by design.
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'local-hyperdb_amd', 'csrc')
OUT = os.path.join(ROOT, 'local-hyperdb_amd', 'lib', 'clock')
VARIANTS = {0: "shipped kernel", 1: "no survivor append", 2: "no LDS-DMA after priming", 3: "no append, no DMA"}
CHILD = r'''
import sys, ctypes, json, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb import _native
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device('cuda', 0)
n, d, q = 10_000_000, 384, 256
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(q, d, torch.float16, dev)
mid = METRIC_IDS['dot_product']
import os
if os.environ.get('HDB_Q256_PIPELINE'):            # "16" / "64": the five-kernel pipeline's filter pass with that MFMA variant
    ix.set_option('use_fused', 0); ix.set_option('dyn_tiles', 0); ix.set_option('mfma_variant', int(os.environ['HDB_Q256_PIPELINE']))
t_end = time.time() + 2.5
calls = 0
while time.time() < t_end:                      # >= 2 s of back-to-back launches before the launch that is read
    for _ in range(20): ix.topk_device(Q, 100, mid)
    torch.cuda.synchronize(); calls += 20
ix.set_option('profile', 1); torch.cuda.synchronize()
for _ in range(20): ix.topk_device(Q, 100, mid)
torch.cuda.synchronize()
ns, l = ix.stat('scan_time_ns'), ix.stat('scan_launches')
lib = _native.lib()
wgs = 256
buf = (ctypes.c_uint64 * (4 * wgs))()
lib.hdb_debug_read_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.hdb_debug_read_clock(buf, wgs)
a = np.frombuffer(buf, dtype=np.uint64).reshape(wgs, 4).astype(np.float64)
dc, dr = a[:, 1] - a[:, 0], a[:, 3] - a[:, 2]
ok = dr > 0
clock = dc[ok] / dr[ok] * 0.1
tiles = (n // 64 + wgs - 1) // wgs
mfma_cycles = tiles * 96 * 2 * 16
print(json.dumps({"rc": rc, "warm_calls": calls, "kernel_us_hip_events": ns / l / 1e3,
                  "clock_ghz_median": float(np.median(clock)), "clock_ghz_min": float(clock.min()), "clock_ghz_max": float(clock.max()),
                  "loop_us_median": float(np.median(dr[ok]) / 100.0), "loop_cycles_median": float(np.median(dc[ok])),
                  "mfma_issue_cycles_per_simd": mfma_cycles, "mfma_busy_frac": float(mfma_cycles / np.median(dc[ok])),
                  "workgroups": int(ok.sum())}), flush=True)
'''

def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(CSRC, 'obj', f) for f in sorted(os.listdir(os.path.join(CSRC, 'obj'))) if f.endswith('.o') and f != 'hdb_mfma_d384.o']
    procs = []
    for v in VARIANTS:
        o = os.path.join(OUT, f'mfma_{v}.o')
        procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed', '-DHDB_MFMA_CLOCK=1',
                                       f'-DHDB_MFMA_KNOCKOUT={v}', '-c', os.path.join(CSRC, 'hdb_mfma_d384.hip'), '-o', o]))
    for p in procs:
        if p.wait(): raise SystemExit('hipcc failed')
    for v in VARIANTS:
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, f'lib_{v}.so'),
                               os.path.join(OUT, f'mfma_{v}.o')] + objs)
        os.remove(os.path.join(OUT, f'mfma_{v}.o'))
    print('built', sorted(os.listdir(OUT)))

def run(dst):
    res = {"workload": "N=10M d=384 fp16 Q=256 dot_product top-100, hdb_mfma_kernel<f16,16,2,384,64,1,1,0>",
           "method": "s_memtime / s_memrealtime stamps around the tile loop, diagnostic build (HDB_MFMA_CLOCK=1), "
                     ">= 2 s of back-to-back launches first; interleaved rounds of all variants in one job", "rounds": []}
    runs = [(v, VARIANTS[v] + " (single launch, MODE 2 filter pass)", None) for v in VARIANTS]
    runs += [(0, "five-kernel pipeline's filter pass: 8 waves x 32 queries (static tiles)", "16"),
             (0, "five-kernel pipeline's filter pass: 4 waves x 64 queries, one wave per SIMD, 450 registers (measurement variant)", "64")]
    for rnd in range(2):
        for v, label, pipe in runs:
            env = dict(os.environ, HYPERDB_HIP_LIB=os.path.join(OUT, f'lib_{v}.so'))
            if pipe:
                env["HDB_Q256_PIPELINE"] = pipe
            r = subprocess.run([sys.executable, '-c', CHILD], env=env, cwd=ROOT, timeout=300, stderr=subprocess.DEVNULL, stdout=subprocess.PIPE, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith('{')]
            rec = json.loads(line[-1]) if line else {"error": r.stdout[-300:]}
            rec.update(variant=label, round=rnd)
            res["rounds"].append(rec)
            print(rec, flush=True)
    json.dump(res, open(dst, 'w'), indent=1)

if __name__ == '__main__':
    if sys.argv[1] == 'build': build()
    else: run(sys.argv[2])
