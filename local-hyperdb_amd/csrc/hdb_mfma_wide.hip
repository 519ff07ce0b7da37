// hdb_mfma_wide.hip -- fp16 MFMA row scan for the remaining multiples of 128 between 768 and 1536 (896, 1152, 1280, 1408):
// 16-row stages of 28..44 KiB, 7..11 LDS-DMA pieces per staging wave.  Own translation unit so that the instantiations
// compile in parallel with hdb_mfma.hip.
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f16_wide(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                             const float* qsq, const float* qscl, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        case 896: return launch_mode<_Float16, 16, 1, 896, 16>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        case 1152: return launch_mode<_Float16, 16, 1, 1152, 16>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        case 1280: return launch_mode<_Float16, 16, 1, 1280, 16>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        case 1408: return launch_mode<_Float16, 16, 1, 1408, 16>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
        default: return (int)hipErrorNotSupported;
    }
}
