// hdb_mfma_anyd_b.hip -- instantiations of the any-width MFMA scan (hdb_mfma_anyd.h): _Float16 geometries 512 768 1024
#include "hdb_mfma_anyd.h"

extern "C" int hdb_launch_mfma_anyd_b(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, const float* qscl, int blocks, void* stream) {
    const ScanArgs a = anyd_args(*args, 2);
    hipStream_t st = (hipStream_t)stream;
    switch (dpad) {
        case 512: return launch_anyd<_Float16, 512, 32>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 768: return launch_anyd<_Float16, 768, 32>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 1024: return launch_anyd<_Float16, 1024, 16>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
