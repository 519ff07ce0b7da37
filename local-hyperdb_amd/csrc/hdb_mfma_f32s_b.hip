// hdb_mfma_f32s_b.hip -- float32 rows as bf16 parts (hdb_mfma_f32s.hip), d = 384, and d = 512 / 768 with the k-steps of a row
// shared between two waves (hdb_mfma_kernel.h, KP = 2: 64 queries per launch row).
#include "hdb_mfma_kernel.h"

extern "C" int hdb_launch_mfma_scan_f32s_wide(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                              const float* qsq, int blocks, void* stream, const BatchArgs* f) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        case 384:
            if (nq_launch <= 64) return launch_mode<hdb_f32s, 16, 1, 384, 32, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
            return launch_mode<hdb_f32s, 16, 1, 384, 32>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 512: return launch_mode<hdb_f32s, 16, 1, 512, 16, 1, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        case 768: return launch_mode<hdb_f32s, 16, 1, 768, 16, 1, 2>(a, mode, q, sqnorm, qsq, nullptr, nq_launch, blocks, st, f);
        default: return (int)hipErrorNotSupported;
    }
}

#if HDB_ROUND_PROF
// Diagnostic build only: the per-wave section clocks of the last launch of THIS translation unit's kernels ([256 workgroups][8 waves][8]).
extern "C" int hdb_debug_read_round_prof(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hdb_round_prof), sizeof(unsigned long long) * 256 * 8 * 8);
}
#endif
