// hdb_mfma_anyd_d.hip -- instantiations of the any-width MFMA scan (hdb_mfma_anyd.h): float geometries 512 768 
#include "hdb_mfma_anyd.h"

extern "C" int hdb_launch_mfma_anyd_d(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, const float* qscl, int blocks, void* stream) {
    const ScanArgs a = anyd_args(*args, 4);
    hipStream_t st = (hipStream_t)stream;
    switch (dpad) {
        case 512: return launch_anyd<float, 512, 16>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 768: return launch_anyd<float, 768, 16>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
