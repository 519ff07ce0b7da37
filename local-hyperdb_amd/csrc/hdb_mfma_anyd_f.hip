// hdb_mfma_anyd_f.hip -- instantiations of the any-width MFMA scan (hdb_mfma_anyd.h): float32 rows as bf16 parts with the k-steps of
// a row shared by two waves (hdb_mfma_kernel.h KP = 2), geometries 512 768
#include "hdb_mfma_anyd.h"

extern "C" int hdb_launch_mfma_anyd_f(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, const float* qscl, int blocks, void* stream) {
    const ScanArgs a = anyd_args(*args, 4);
    hipStream_t st = (hipStream_t)stream;
    switch (dpad) {
        case 512: return launch_anyd<hdb_f32s, 512, 16, 2>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 768: return launch_anyd<hdb_f32s, 768, 16, 2>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
