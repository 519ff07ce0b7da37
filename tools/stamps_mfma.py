"""Diagnostic: per-phase s_memtime stamps of the MFMA filter kernel (block 0, waves 0 and 4)."""
import sys, ctypes
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
n, d = 10_000_000, 384
dev = torch.device('cuda', 0)
V, lo, hi = bench.make_shard(n, d, torch.float16, 0, 1, dev)
ix = GpuIndex(V)
Q = bench.make_queries(256, d, torch.float16, dev)
mid = METRIC_IDS['dot_product']
for flags in (8, 8 | 7, 8 | 1, 8 | 4):
    ix.set_option('debug_flags', flags)
    for _ in range(3): ix.topk_device(Q, 100, mid)
    torch.cuda.synchronize()
    ptr = ix.stat('debug_buffer')
    nwords = 2 * 64 * 8 + 8
    buf = torch.empty(nwords, dtype=torch.int64, device=dev)
    ctypes.CDLL(None)
    import hyperdb._native as nat
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(nwords * 8), 3)
    torch.cuda.synchronize()
    a = buf.cpu().numpy().astype(np.int64)
    print(f"--- flags={flags}")
    for wv, name in ((0, 'wave0(A)'), (1, 'wave4(B)')):
        st = a[wv * 512:(wv + 1) * 512].reshape(64, 8)[8:56]      # skip warm-up tiles
        per_iter = np.diff(st[:, 0])
        seg = np.stack([st[:, k + 1] - st[:, k] for k in range(6)], axis=1)
        print(name, 'cycles/tile median', int(np.median(per_iter)),
              'segments [vmcnt, barrier, issue+flushchk, defer-filter, mfma, epilogue]:', [int(x) for x in np.median(seg, axis=0)])
    t0, r0, t1, r1 = a[1024], a[1025], a[1028], a[1029]
    print('clock GHz (memtime/memrealtime*0.1):', (t1 - t0) / max(r1 - r0, 1) * 0.1, 'total cycles', t1 - t0)
