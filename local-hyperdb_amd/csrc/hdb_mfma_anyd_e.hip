// hdb_mfma_anyd_e.hip -- instantiations of the any-width MFMA scan (hdb_mfma_anyd.h): float32 rows as bf16 parts
// (MfmaShape<16, hdb_f32s>, hdb_mfma_f32s.hip), geometries 128 256 384
#include "hdb_mfma_anyd.h"

extern "C" int hdb_launch_mfma_anyd_e(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, const float* qscl, int blocks, void* stream) {
    const ScanArgs a = anyd_args(*args, 4);
    hipStream_t st = (hipStream_t)stream;
    switch (dpad) {
        case 128: return launch_anyd<hdb_f32s, 128, 64>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 256: return launch_anyd<hdb_f32s, 256, 32>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 384: return launch_anyd<hdb_f32s, 384, 32>(a, mode, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
