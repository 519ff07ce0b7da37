// hdb_mfma_fused_wide.hip -- the single-launch top-k (hdb_mfma_fused.h) for fp16 rows of 1024 .. 1536 elements: 16-row tiles,
// query fragments resident in LDS (up to 2 queries), a translation unit of its own so that it compiles beside the others.
#include "hdb_mfma_fused.h"

extern "C" int hdb_launch_mfma_fused_wide(const ScanArgs* args, const FusedArgs* fa, int blocks, void* stream) {
    const ScanArgs& a = *args;
    const FusedArgs& f = *fa;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d) {
        case 1024: return launch_fused<_Float16, 2, 1024, 16>(a, f, blocks, st);
        case 1152: return launch_fused<_Float16, 2, 1152, 16>(a, f, blocks, st);
        case 1280: return launch_fused<_Float16, 2, 1280, 16>(a, f, blocks, st);
        case 1408: return launch_fused<_Float16, 2, 1408, 16>(a, f, blocks, st);
        case 1536: return launch_fused<_Float16, 2, 1536, 16>(a, f, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
