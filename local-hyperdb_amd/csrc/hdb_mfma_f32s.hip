// hdb_mfma_f32s.hip -- the float32 row scan on the bf16 matrix pipe (MfmaShape<16, hdb_f32s>, hdb_mfma_kernel.h): the staging waves
// turn every tile into two bf16 parts per value in place, the queries travel as three exact parts, five part products per k-step,
// fp32 accumulation.  Same geometries and modes as hdb_mfma_f32.hip's v_mfma_f32_16x16x4_f32 (np.dot on the reference's default
// precision, hyperdb/ranking_algorithm.py:29,:41; hyperdb.py:51) at 3.2x its matrix-pipe rate: the router (hdb_mfma.hip) sends
// launches of more than 32 queries here -- up to 32 the float32 MFMAs already keep up with HBM.  Every wave stages, converts and
// multiplies.  d = 128 / 256 here, 384 in hdb_mfma_f32s_b.hip
// (translation units of their own so that the instantiations compile in parallel).
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f32s_wide(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                              const float* qsq, int blocks, void* stream, const BatchArgs* f);

extern "C" int hdb_launch_mfma_scan_f32s(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                         const float* qsq, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        // (up to 64 queries: two waves per query group, each multiplying every other 16-row tile of the stage -- all eight waves multiply)
        case 128:
            if (nq_launch <= 64) return launch_mode<hdb_f32s, 16, 1, 128, 64, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
            return launch_mode<hdb_f32s, 16, 1, 128, 64>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 256:
            if (nq_launch <= 64) return launch_mode<hdb_f32s, 16, 1, 256, 32, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
            return launch_mode<hdb_f32s, 16, 1, 256, 32>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        default: return hdb_launch_mfma_scan_f32s_wide(args, mode, nq_launch, q, sqnorm, qsq, blocks, stream, f);
    }
}
