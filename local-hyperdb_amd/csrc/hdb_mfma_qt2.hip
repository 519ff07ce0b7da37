// hdb_mfma_qt2.hip -- fp16 MFMA row scan with TWO query tiles per wave (256 queries per pass) for the dimensions
// whose B fragments still fit the 256-register budget besides d = 384 (which lives in hdb_mfma.hip): a batch of
// 129..256 queries then reads V once instead of once per 128 queries.  d = 768 spills with two tiles and keeps one.
#include "hdb_mfma_kernel.h"

extern "C" int hdb_mfma_qt2_supported(int d) { return d == 128 || d == 256 || d == 512 || d == 640; }

extern "C" int hdb_launch_mfma_scan_f16_qt2(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                            const float* qsq, const float* qscl, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        case 128: return launch_mode<_Float16, 16, 2, 128, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        case 256: return launch_mode<_Float16, 16, 2, 256, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        case 512: return launch_mode<_Float16, 16, 2, 512, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        case 640: return launch_mode<_Float16, 16, 2, 640, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        default: return (int)hipErrorNotSupported;
    }
}
