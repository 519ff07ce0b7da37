// hdb_mfma_anyd_c.hip -- instantiations of the any-width MFMA scan (hdb_mfma_anyd.h): float geometries 128 256 384
#include "hdb_mfma_anyd.h"

extern "C" int hdb_launch_mfma_anyd_c(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, const float* qscl, int blocks, void* stream) {
    const ScanArgs a = anyd_args(*args, 4);
    hipStream_t st = (hipStream_t)stream;
    switch (dpad) {
        case 128: return launch_anyd<float, 128, 64>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 256: return launch_anyd<float, 256, 32>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 384: return launch_anyd<float, 384, 32>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
