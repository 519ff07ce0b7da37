// hdb_mfma_mid.hip -- instantiations of the MFMA row scan (hdb_mfma_kernel.h) for fp16 d = 512, 640 and 768 (32-row stages), one query tile per wave: a translation unit of its own so that
// the geometries compile in parallel (each carries 3 modes x 3 metrics x {bias, no bias} kernels).
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f16_mid(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
        const float* qsq, const float* qscl, int blocks, int variant, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    (void)variant;
    if (a.d == 512) return launch_mode<_Float16, 16, 1, 512, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    if (a.d == 640) return launch_mode<_Float16, 16, 1, 640, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    if (a.d == 768) return launch_mode<_Float16, 16, 1, 768, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    return (int)hipErrorNotSupported;
}
