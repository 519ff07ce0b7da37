"""Where a round of the float32-as-bf16-parts scan goes: per-wave shader clocks per section, summed over the rounds of one call
(diagnostic build: tools/build_round_prof_f32s.sh; HYPERDB_HIP_LIB=tools/bin/libhyperdb_hip_rp.so python tools/round_sections.py)."""
import ctypes, os, sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb import _native
from hyperdb._native import GpuIndex, METRIC_IDS
lib = _native._lib
lib.hdb_debug_read_round_prof.argtypes = [ctypes.c_void_p]
g = torch.Generator(device='cuda').manual_seed(11)
names = ["barrier", "stage", "multiply", "K barrier", "epilogue", "convert", "rounds", "wait tile"]
for n, d, nq in ((2_000_000, 384, 64), (2_000_000, 384, 128), (1_000_000, 768, 16), (1_000_000, 768, 64)):
    V = torch.randn((n, d), generator=g, device='cuda'); ix = GpuIndex(V)
    ix.set_option("f32_split_min_q", 1)
    Q = torch.randn((nq, d), generator=g, device='cuda'); mid = METRIC_IDS["dot_product"]
    for _ in range(3): ix.topk_device(Q, 100, mid)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ix.topk_device(Q, 100, mid); torch.cuda.synchronize(); us = (time.perf_counter() - t0) * 1e6
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    rc = lib.hdb_debug_read_round_prof(buf.ctypes.data)
    a = buf.reshape(256, 8, 8).astype(np.float64)
    print(f"n={n} d={d} Q={nq}: {us:.0f} us per call; mean shader clocks per round and wave (over 256 workgroups)", flush=True)
    for w in range(8):
        r = a[:, w, 6].mean()
        if r == 0: continue
        b = a[:, w, :].copy()
        if w < 4 and d <= 384: b[:, [1, 5]] = b[:, [5, 1]]           # (one-barrier kernels: waves 0-3 stage behind the epilogue, their two slots are booked crosswise)
        row = "  ".join(f"{names[k]} {b[:, k].mean() / r:7.0f}" for k in (0, 1, 7, 5, 2, 3, 4))
        print(f"  wave {w}: rounds {r:5.0f} | {row} | sum {(b[:, :6].sum(axis=1) + b[:, 7]).mean() / r:7.0f}", flush=True)
    ix.close(); del V; torch.cuda.empty_cache()
