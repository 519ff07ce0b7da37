// hdb_mfma_d384.hip -- instantiations of the MFMA row scan (hdb_mfma_kernel.h) for fp16 d = 384 (the headline width): a translation unit of its own so that
// the geometries compile in parallel (each carries 3 modes x 3 metrics x {bias, no bias} kernels).
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f16_d384(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
        const float* qsq, const float* qscl, int blocks, int variant, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    (void)variant;
    // more than 128 queries: 16 = 16x16x32 with two query tiles per wave (default: the same FLOPs and LDS traffic as the
    // 32x32x16 form, but the chip holds a higher clock on this shape: 1.78-1.85 ms against 2.07-2.25 ms for N=10M, Q=256),
    // 32 = 32x32x16 with one query tile per wave (kept for A/B measurements; hdb_set_option(ix, "mfma_variant", 16 | 32))
    // 64 = the four-wave measurement variant (64 queries per wave, 256-thread workgroups; filter pass, dot product, no bias)
    if (variant == 64 && mode == 1 && a.metric == HDB_DOT && !a.bias) return launch_four_waves<_Float16, 16, 4, 384, 64>(a, q16, qscl, nq_launch, blocks, st);
    if (nq_launch > 128 && variant == 32) return launch_mode<_Float16, 32, 1, 384, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    if (nq_launch > 128) return launch_mode<_Float16, 16, 2, 384, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    return launch_mode<_Float16, 16, 1, 384, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
}

#if HDB_MFMA_CLOCK
// Diagnostic build only: copy the stamps of the last launch of THIS translation unit's kernels.
extern "C" int hdb_debug_read_clock(unsigned long long* host_out, int wgs) {
    if (wgs > HDB_CLOCK_WGS) wgs = HDB_CLOCK_WGS;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hdb_clock_buf), (size_t)wgs * 4 * sizeof(unsigned long long));
}
#endif
#if HDB_BATCH_STAMPS
// Diagnostic build only: the phase stamps of the last single-launch batched call of THIS translation unit's kernels.
extern "C" int hdb_debug_read_batch_stamps(unsigned long long* host_out, int wgs) {
    if (wgs > 1024) wgs = 1024;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hdb_batch_stamps), (size_t)wgs * 16 * sizeof(unsigned long long));
}
#endif
