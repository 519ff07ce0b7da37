// hdb_mfma_f32.hip -- fp32 instantiations of the MFMA row scan (hdb_mfma_kernel.h): v_mfma_f32_16x16x4_f32, exact
// fp32 products and accumulation, so batches on float32 matrices (the reference's default fp_precision,
// hyperdb.py:51) keep the 1e-5 parity contract while up to 128 queries ride on one pass over V instead of 4.
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f32_wide(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                             const float* qsq, int blocks, void* stream, const BatchArgs* f);

extern "C" int hdb_launch_mfma_scan_f32(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        // up to 64 queries: two waves per query group, each multiplying every other 16-row tile of the stage, so that
        // all four SIMDs issue fp32 MFMAs (N=4M d=384, 16 queries: 1.73 ms with one wave per group)
        case 128:
            if (nq_launch <= 64) return launch_mode<float, 16, 1, 128, 64, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
            return launch_mode<float, 16, 1, 128, 64>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 256:
            if (nq_launch <= 64) return launch_mode<float, 16, 1, 256, 32, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
            return launch_mode<float, 16, 1, 256, 32>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        default: return hdb_launch_mfma_scan_f32_wide(args, mode, nq_launch, q, sqnorm, qsq, blocks, stream, f);     // 384, 512, 768
    }
}
