"""The 256-query filter pass (N=10M d=384 fp16 dot) with EIGHT waves x 32 queries (the product: waves 4-7 stage, two waves per
SIMD) against FOUR waves x 64 queries (one wave per SIMD, 450 registers, every wave stages and multiplies, half the LDS fragment
reads per MFMA): five-kernel pipeline, kernel time of the filter pass from the library's HIP events, interleaved rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-hyperdb_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hyperdb._native import GpuIndex, METRIC_IDS
import bench
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
V, _, _ = bench.make_shard(n, 384, torch.float16, 0, 1, dev)
Q = bench.make_queries(256, 384, torch.float16, dev).float()
ix = GpuIndex(V)
mid = METRIC_IDS["dot_product"]
ix.set_option("use_fused", 0); ix.set_option("dyn_tiles", 0)
ref = None
res = {16: [], 64: []}
for rnd in range(4):
    for variant in (16, 64):
        ix.set_option("mfma_variant", variant)
        for _ in range(3):
            out = ix.topk_device(Q, 100, mid)
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        else:
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and int(out[2].abs().sum().item()) == 0, variant
        ix.set_option("profile", 1)
        for _ in range(15):
            ix.topk_device(Q, 100, mid)
        torch.cuda.synchronize()
        res[variant].append(ix.stat("scan_time_ns") / max(1, ix.stat("scan_launches")) / 1e3)
        ix.set_option("profile", 0)
for variant, name in ((16, "8 waves x 32 queries (product)"), (64, "4 waves x 64 queries")):
    print(f"n={n} d=384 fp16 Q=256 dot filter pass, {name}: {' / '.join(f'{x:.1f}' for x in res[variant])} us per pass "
          f"= {2 * 256 * n * 384 / np.median(res[variant]) / 1e6:.0f} TFLOP/s", flush=True)
