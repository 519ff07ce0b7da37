// hdb_mfma_anyd_a.hip -- instantiations of the any-width MFMA scan (hdb_mfma_anyd.h): _Float16 geometries 128 256 384
#include "hdb_mfma_anyd.h"

extern "C" int hdb_launch_mfma_anyd_a(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, const float* qscl, int blocks, void* stream) {
    const ScanArgs a = anyd_args(*args, 2);
    hipStream_t st = (hipStream_t)stream;
    switch (dpad) {
        case 128: return launch_anyd<_Float16, 128, 64>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 256: return launch_anyd<_Float16, 256, 64>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 384: return launch_anyd<_Float16, 384, 64>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
