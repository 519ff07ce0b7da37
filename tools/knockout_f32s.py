"""p50 of GpuIndex.topk_device (no host re-run: knock-out builds answer wrongly) for the float32-as-bf16-parts scan; run once per
library: HYPERDB_HIP_LIB=tools/bin/libhyperdb_hip_ko<bits>.so python tools/knockout_f32s.py (tools/build_knockouts_f32s.sh)."""
import os, sys, time
sys.path.insert(0, 'local-hyperdb_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
g = torch.Generator(device='cuda').manual_seed(11)
out = []
for n, d, nq in ((2_000_000, 384, 64), (2_000_000, 384, 128), (1_000_000, 768, 16), (1_000_000, 768, 64)):
    V = torch.randn((n, d), generator=g, device='cuda'); ix = GpuIndex(V)
    ix.set_option("f32_split_min_q", 1)
    Q = torch.randn((nq, d), generator=g, device='cuda'); mid = METRIC_IDS["dot_product"]
    for _ in range(3): ix.topk_device(Q, 100, mid)
    torch.cuda.synchronize(); ts = []
    for _ in range(15):
        t0 = time.perf_counter(); ix.topk_device(Q, 100, mid); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out.append(f"{n // 1000}k x {d} Q={nq}: {np.median(ts) * 1e6:.0f} us (split={ix.stat('f32_split')})")
    ix.close(); del V; torch.cuda.empty_cache()
print(os.environ.get("HYPERDB_HIP_LIB", "product").split("/")[-1], "|", "   ".join(out), flush=True)
