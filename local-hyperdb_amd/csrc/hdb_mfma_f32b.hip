// hdb_mfma_f32b.hip -- the float32 MFMA row scan (hdb_mfma_f32.hip) for d = 384, 512 and 768: a translation unit of its own
// so that the instantiations compile in parallel.
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f32_wide(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                             const float* qsq, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        case 384:
            if (nq_launch <= 64) return launch_mode<float, 16, 1, 384, 32, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
            return launch_mode<float, 16, 1, 384, 32>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 512: return launch_mode<float, 16, 1, 512, 16>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 768: return launch_mode<float, 16, 1, 768, 16>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        default: return (int)hipErrorNotSupported;
    }
}
