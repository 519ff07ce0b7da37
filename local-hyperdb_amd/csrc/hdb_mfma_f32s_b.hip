// hdb_mfma_f32s_b.hip -- float32 rows as bf16 parts (hdb_mfma_f32s.hip), d = 384, and d = 512 / 768 with the k-steps of a row
// shared between two waves (hdb_mfma_kernel.h, KP = 2: 64 queries per launch row).
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f32s_wide(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                              const float* qsq, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        case 384: return launch_mode<hdb_f32s, 16, 1, 384, 32>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 512: return launch_mode<hdb_f32s, 16, 1, 512, 16, 1, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 768: return launch_mode<hdb_f32s, 16, 1, 768, 16, 1, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        default: return (int)hipErrorNotSupported;
    }
}
