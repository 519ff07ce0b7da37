// hdb_mfma_fused.hip -- instantiations of the single-launch top-k (hdb_mfma_fused.h) for the fp16 geometries of
// hdb_mfma.hip: a whole hdb_topk call of 1..4 dot / cosine / pearson queries (one euclidean query) in ONE kernel.
#include "hdb_mfma_fused.h"

extern "C" int hdb_mfma_tile_rows(int dtype, int d);
extern "C" int hdb_launch_mfma_fused_wide(const ScanArgs* args, const FusedArgs* f, int blocks, void* stream);

extern "C" int hdb_mfma_fused_supported(int dtype, int d, int metric, int nq, uint32_t kk) {
    // fp16: every width the batched scan takes (multiples of 128 up to 1536); beyond d = 768 the query fragments (d/8
    // registers) leave no room for the selectors' state: they stay in LDS, up to 2 queries (hdb_mfma_fused_wide.hip)
    // (d = 896: 28-KiB tiles of 28 k-steps keep the one multiplying wave busier than the stream: five kernels are 6 % faster)
    // (d = 128: 16-KiB tiles -- a round of this kernel costs ~1 us whatever the tile holds: 660 vs 400 us at 10 M rows)
    const bool shape = (dtype == HDB_F16 && hdb_mfma_tile_rows(dtype, d) > 0 && d % 128 == 0 && d != 896 && d != 128 && d <= 1536) ||
                       (dtype == HDB_F32 && (d == 128 || d == 256 || d == 384 || d == 512 || d == 768));   // float32: VALU flavour
    // float32 queries live in registers as d/4 floats per lane group: 48 registers = 2 queries up to d = 384, 1 beyond
    const int maxq = dtype == HDB_F32 ? (d <= 384 ? 2 : 1) : (d <= 768 ? HDB_FUSED_MAXQ : 2);
    // euclidean (the MFMA expansion + direct re-score of near-duplicates in the last workgroup): fp16 matrices only -- the
    // float32 VALU pipelines compute the direct difference, which this kernel's float32 flavour does not; d = 768 would spill
    // three registers (those calls take the batched single launch, hdb_mfma_kernel.h MODE 2)
    // (euclidean, 2-4 queries: wave 0 pays sqrt + rcp on all 16 MFMA columns -- 199 vs 182 us at N=1.25M d=384 with four queries; those
    // calls take the batched single launch, where eight waves share the epilogue)
    // float32 (round 3): the VALU flavour accumulates (v - q)^2 directly, as hdb_scan.hip does -- no cancellation, nothing to
    // re-score, one or two queries like dot / cosine
    const bool euclid = metric == HDB_EUCLIDEAN && ((dtype == HDB_F16 && d != 768 && nq == 1) || dtype == HDB_F32);
    return shape && (metric == HDB_DOT || metric == HDB_COSINE || metric == HDB_PEARSON || euclid) && nq >= 1 && nq <= maxq && kk <= 128;
}

// Local flavour (FusedArgs::local, hdb_mfma_fused.h): how many tiles of a workgroup fit its parking area -- 0 where the kernel
// does not park at all (fp16 d = 768; euclidean d = 640: the parking state would spill there).  Mirrors PARK / pend_max.
extern "C" int hdb_mfma_fused_local_tiles(int dtype, int d, int metric, int nq) {
    if (dtype == HDB_F32) return nq <= 1 || d > 384 ? 16 : 8;
    const bool park = (d <= 640 && !(metric == HDB_EUCLIDEAN && d > 512)) || d > 768;
    if (!park) return 0;
    return nq <= 2 ? 16 : 32 / nq;
}

// bytes of the persistent control block: 64 words of counters + the granules
extern "C" size_t hdb_mfma_fused_ctl_bytes(void) { return HDB_FUSED_HDR_BYTES + (size_t)HDB_FUSED_MAX_WG * HDB_FUSED_GRAN_PER_WG * 8; }

extern "C" int hdb_launch_mfma_fused(const ScanArgs* args, int dtype, const FusedArgs* fa, int max_blocks, void* stream) {
    const ScanArgs& a = *args;
    FusedArgs f = *fa;
    hipStream_t st = (hipStream_t)stream;
    const int cus = hdb_cu_count();
    int blocks = (int)(a.ntiles < cus ? a.ntiles : cus);
    if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
    if (blocks > HDB_FUSED_MAX_WG) blocks = HDB_FUSED_MAX_WG;
    if (blocks < 1) blocks = 1;
    f.gran = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(f.ctl) + HDB_FUSED_HDR_BYTES);
    if (dtype == HDB_F32) {
        switch (a.d) {
            case 128: return f.nq == 1 ? launch_fused<float, 1, 128, 64>(a, f, blocks, st) : launch_fused<float, 2, 128, 64>(a, f, blocks, st);
            case 256: return f.nq == 1 ? launch_fused<float, 1, 256, 32>(a, f, blocks, st) : launch_fused<float, 2, 256, 32>(a, f, blocks, st);
            case 384: return f.nq == 1 ? launch_fused<float, 1, 384, 32>(a, f, blocks, st) : launch_fused<float, 2, 384, 32>(a, f, blocks, st);
            case 512: return launch_fused<float, 1, 512, 16>(a, f, blocks, st);
            case 768: return launch_fused<float, 1, 768, 16>(a, f, blocks, st);
            default: return (int)hipErrorNotSupported;
        }
    }
    switch (a.d) {
        case 256: return launch_fused<_Float16, 2, 256, 64>(a, f, blocks, st);
        case 384: return launch_fused<_Float16, 2, 384, 64>(a, f, blocks, st);
        case 512: return launch_fused<_Float16, 2, 512, 32>(a, f, blocks, st);
        case 640: return launch_fused<_Float16, 2, 640, 32>(a, f, blocks, st);
        case 768: return launch_fused<_Float16, 2, 768, 32>(a, f, blocks, st);
        default: return hdb_launch_mfma_fused_wide(&a, &f, blocks, stream);
    }
}

#if HDB_FUSED_STAMPS
extern "C" int hdb_debug_read_fused_stamps(unsigned long long* host_out, int wgs) {
    if (wgs > HDB_CLOCK_WGS_F) wgs = HDB_CLOCK_WGS_F;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hdb_fused_stamps), (size_t)wgs * 16 * sizeof(unsigned long long));
}
#endif
