// hdb_mfma_ksplit_s.hip -- the K slices of wide float32 rows (hdb_mfma_ksplit.hip: d = 1024 / 1536 as two slices of 512 / 768) with
// the rows multiplied as bf16 parts, two waves per query group (MfmaShape<16, hdb_f32s>, KP = 2: hdb_mfma_kernel.h).
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_kslice_f32s(const ScanArgs* a, int dslice, int mode, int nq_launch, const void* q, const float* sqnorm,
                                           const float* qsq, int blocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (dslice == 512) return launch_kslice<hdb_f32s, 512, 16, 2>(*a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st);
    if (dslice == 768) return launch_kslice<hdb_f32s, 768, 16, 2>(*a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st);
    return (int)hipErrorNotSupported;
}
