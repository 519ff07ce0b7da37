// hdb_mfma.hip -- the Q.V^T row scan of fp16 matrices on the gfx950 matrix cores (fp32 accumulate), 1..256 queries
// per pass over V.
//
// Replaces "np.dot(vectors, query.T)" (hyperdb/ranking_algorithm.py:29,:41) and, through
// ||v-q||^2 = ||v||^2 + ||q||^2 - 2 v.q, "np.linalg.norm(vectors - query, axis=1)" (:49).  The reference takes one
// query per call; here up to 8*MF*QT queries ride on ONE pass over V, and a single query takes the same kernel (one
// wave multiplies, the kernel is then a pure HBM streaming kernel: 6.9-7.1 TB/s at N=10M, d=384).
//
// Work decomposition (one workgroup = 8 waves = 512 threads, one workgroup per CU, persistent):
//   * wave w owns MF*QT queries of the batch (MF = 16: v_mfma_f32_16x16x32_f16, QT = 1 or 2 query tiles per wave;
//     MF = 32: v_mfma_f32_32x32x16_f16, kept for A/B runs); their scaled fp16 values (hdb_q16_scale) for ALL k
//     live in its registers as MFMA B fragments (QT*d/8 VGPRs for MF=16), loaded once;
//   * the workgroup streams tiles of R rows of V through a 3-deep LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction, source-side XOR swizzle so that the
//     lane-linear LDS image is bank-conflict-free for the ds_read_b128 A-fragment reads), together with
//     the per-row aux values (1/||v|| or ||v||^2, bias);
//   * every wave multiplies the whole tile by its queries, one ds_read_b128 per QT MFMAs, issued 2-3 k-steps
//     ahead from inline asm with counted lgkmcnt waits; accumulators never leave registers;
//   * epilogue in registers: scale / bias / threshold compare behind a group-max prefilter; survivors go
//     to a wave-private segment of a small LDS list (slot = running count + ballot rank, no atomic), which each wave
//     empties into the per-query candidate lists with global atomics when it fills.  The N x Q score matrix is never
//     written (10 GB at N=10M, Q=256).
// Synchronisation: one raw s_barrier per tile; tile t+2 is in flight while tile t is multiplied (counted
// s_waitcnt vmcnt, never 0 in the steady state).
// Algorithmic bytes per row: d*2 (V read exactly once per pass); FLOPs: 2*Q*d per row.
#include "hdb_mfma_kernel.h"

// fp32 -> scaled fp16 queries for the matrix pipe (hdb_q16_scale, hdb_common.h); qscl[q] = 1 / scale.  One wave per
// query.  The prep kernel does the same for plain queries; this one serves the centred copies of pearson.
__global__ __launch_bounds__(64) void hdb_q_to_f16_kernel(const float* Q, int nq, int d, _Float16* out, float* qscl) {
    const int q = blockIdx.x;
    if (q >= nq) return;
    float amax = 0.f;
    for (int e = threadIdx.x; e < d; e += 64) amax = fmaxf(amax, fabsf(Q[(int64_t)q * d + e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float scale = hdb_q16_scale(amax);
    for (int e = threadIdx.x; e < d; e += 64) out[(int64_t)q * d + e] = (_Float16)(Q[(int64_t)q * d + e] * scale);
    if (threadIdx.x == 0) qscl[q] = 1.f / scale;
}

// Euclidean scores from the MFMA path come from ||v||^2 + ||q||^2 - 2 v.q, which cancels when v ~ q (an exact
// duplicate scores 1/(1+~0.01) instead of 1).  Candidates whose squared distance is below 5 % of ||q||^2 -- where
// the rounding of the expansion, ~1e-6 (||v||^2 + ||q||^2), exceeds 4e-5 of the distance itself -- are re-scored
// from the stored row with the direct difference, like the reference (:49); beyond that the expansion is good to
// 5e-6 in the similarity.  One wave per entry.
template <typename T, bool HAS_BIAS>
__global__ __launch_bounds__(256) void hdb_rescore_euclid_kernel(unsigned long long* cand, const uint32_t* cnt, uint32_t cap,
                                                                 const T* V, int d, const float* Q, const float* qsq, int q0,
                                                                 const float* bias) {
    const int ql = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t n = cnt[ql * HDB_CNT_STRIDE] < cap ? cnt[ql * HDB_CNT_STRIDE] : cap;
    const float* qv = Q + (int64_t)(q0 + ql) * d;
    const float close2 = 0.05f * qsq[q0 + ql];
    for (uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const unsigned long long ent = cand[(int64_t)ql * cap + e];
        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(ent & 0xFFFFFFFFull);
        float s = hdb_key2f((uint32_t)(ent >> 32));
        const float b = HAS_BIAS ? bias[row] : 0.f;
        const float sim = s - b;                                   // 1 / (1 + dist) in (0, 1]; -inf for an excluded row
        const float dist = 1.f / sim - 1.f;
        if (sim > 0.f && dist * dist < close2) {                   // wave-uniform: one entry per wave
            float acc = 0.f;
            for (int k = lane; k < d; k += 64) { const float df = (float)V[(int64_t)row * d + k] - qv[k]; acc += df * df; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            s = hdb_canon(1.f / (1.f + sqrtf(acc)) + b);
            if (lane == 0) cand[(int64_t)ql * cap + e] = hdb_pack(s, row);
        }
    }
}

// Geometry: rows per LDS stage = the largest of 64 / 32 / 16 whose stage (R * row bytes) fits 48 KiB (three stages + lists
// <= 160 KiB).  Rows are multiples of 256 bytes (the XOR swizzle works on 16 chunks of 16 bytes), so every d that is a
// multiple of 128 (fp16) / 64 (fp32) works; the upper limits are the query fragments a wave holds in registers: d/8
// (fp16) or d/4 (fp32) registers for 16 queries, 192 at most.
// fp16: 16x16x32 MFMAs, 128 queries per pass (d <= 640 with more than 128 queries: two query tiles per wave, 256 per pass).
// fp32: 16x16x4 MFMAs, 128 queries per pass; the matrix pipe (157 TFLOP/s) binds from ~16 queries on, so the VALU scan
// keeps the calls of up to 4 queries (one pass at HBM speed) and this path takes the batches.
extern "C" int hdb_mfma_ksplit_slices(int dtype, int d);
extern "C" int hdb_launch_mfma_ksplit(const ScanArgs* args, int dtype, int mode, int nq_launch, const void* q, const float* sqnorm,
                                      const float* qsq, const float* qscl, int blocks, void* stream);

static int mfma_exact_tile_rows(int dtype, int d) {
    const int elem = dtype == HDB_F16 ? 2 : dtype == HDB_F32 ? 4 : 0;
    if (!elem || d <= 0) return 0;
    if (hdb_mfma_ksplit_slices(dtype, d) > 0) return 16;                  // wide rows: K slices of 16-row stages (hdb_mfma_ksplit.hip)
    const int row_bytes = d * elem;
    if (row_bytes % 256 != 0 || row_bytes > 3072) return 0;          // d <= 1536 (fp16) / 768 (fp32)
    if (dtype == HDB_F32 && d != 128 && d != 256 && d != 384 && d != 512 && d != 768) return 0;     // instantiated fp32 widths
    for (int r = 64; r >= 16; r >>= 1)
        if (r * row_bytes <= 48 * 1024) return r;
    return 0;
}

// Rows of any width that is a multiple of 16 bytes and has no geometry of its own ride the next wider one as a single K slice
// (hdb_mfma_anyd.h): -> that width, or 0.  fp16 d % 8 == 0 up to 1024, float32 d % 4 == 0 up to 768.
extern "C" int hdb_mfma_anyd_pad(int dtype, int d) {
    const int elem = dtype == HDB_F16 ? 2 : dtype == HDB_F32 ? 4 : 0;
    if (!elem || d <= 0 || (d * elem) % 16 != 0 || mfma_exact_tile_rows(dtype, d) > 0) return 0;
    static const int w16[] = {128, 256, 384, 512, 768, 1024}, w32[] = {128, 256, 384, 512, 768};
    if (dtype == HDB_F16) { for (int w : w16) if (w >= d) return w; }
    else { for (int w : w32) if (w >= d) return w; }
    return 0;
}

#define HDB_ANYD_DECL(name) extern "C" int name(const ScanArgs* args, int dpad, int mode, int nq_launch, const void* q, const float* sqnorm, \
                                              const float* qsq, const float* qscl, int blocks, void* stream)
HDB_ANYD_DECL(hdb_launch_mfma_anyd_a); HDB_ANYD_DECL(hdb_launch_mfma_anyd_b); HDB_ANYD_DECL(hdb_launch_mfma_anyd_c); HDB_ANYD_DECL(hdb_launch_mfma_anyd_d);
HDB_ANYD_DECL(hdb_launch_mfma_anyd_e); HDB_ANYD_DECL(hdb_launch_mfma_anyd_f);       // float32 rows as bf16 parts

extern "C" int hdb_mfma_tile_rows(int dtype, int d) {
    const int pad = hdb_mfma_anyd_pad(dtype, d);
    return mfma_exact_tile_rows(dtype, pad ? pad : d);
}

// queries ONE launch of the MFMA scan covers (grid.y == 1): what a single-launch (mode 2) call can take
extern "C" int hdb_mfma_batch_capacity(int dtype, int d) {
    if (hdb_mfma_tile_rows(dtype, d) <= 0 || hdb_mfma_ksplit_slices(dtype, d) > 0 || hdb_mfma_anyd_pad(dtype, d) > 0) return 0;      // (K slices, odd widths: the multi-kernel pipeline)
    if (dtype == HDB_F32) return (d == 512 || d == 768) ? 64 : 128;      // (d = 512 / 768: the bf16-part flavour pairs its waves over K, hdb_mfma_kernel.h KP)
    return (d == 384 || d == 128 || d == 256 || d == 512 || d == 640) ? 256 : 128;      // two query tiles per wave (hdb_mfma_qt2.hip)
}
// bytes of the control block of the single-launch batched call for a grid of `wgs` workgroups
extern "C" size_t hdb_mfma_batch_ctl_bytes(int wgs) { return (size_t)HDB_BATCH_GRAN_BYTE + (size_t)HDB_BATCH_MAXQ * (size_t)wgs * 8 * 8; }

extern "C" int hdb_mfma_supported(int dtype, int d, int metric) {
    return hdb_mfma_tile_rows(dtype, d) > 0 && (metric == HDB_DOT || metric == HDB_COSINE || metric == HDB_EUCLIDEAN);
}

// d=384, more than 128 queries: 16 = 16x16x32 with two query tiles per wave (default: the same FLOPs and LDS
// traffic as the 32x32x16 form, but the chip holds a higher clock on this shape: 1.78-1.85 ms against 2.07-2.25 ms
// for N=10M, Q=256), 32 = 32x32x16 with one query tile per wave (kept for A/B measurements).
// (per index: hdb_set_option(ix, "mfma_variant", 16 | 32), passed down as `variant`)

#define HDB_GEOM_DECL(name) extern "C" int name(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm, \
                                              const float* qsq, const float* qscl, int blocks, int variant, void* stream, const BatchArgs* f)
HDB_GEOM_DECL(hdb_launch_mfma_scan_f16_d384);       // hdb_mfma_d384.hip
HDB_GEOM_DECL(hdb_launch_mfma_scan_f16_narrow);     // hdb_mfma_narrow.hip
HDB_GEOM_DECL(hdb_launch_mfma_scan_f16_mid);        // hdb_mfma_mid.hip
HDB_GEOM_DECL(hdb_launch_mfma_scan_f16_1k);         // hdb_mfma_1k.hip
extern "C" int hdb_mfma_qt2_supported(int d);
extern "C" int hdb_launch_mfma_scan_f16_qt2(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                            const float* qsq, const float* qscl, int blocks, void* stream, const BatchArgs* f);
extern "C" int hdb_launch_mfma_scan_f32(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                        const float* qsq, int blocks, void* stream, const BatchArgs* f);
extern "C" int hdb_launch_mfma_scan_f32s(const ScanArgs* args, int mode, int nq_launch, const void* q, const float* sqnorm,
                                         const float* qsq, int blocks, void* stream, const BatchArgs* f);
// float32 widths whose scan also exists in bf16 parts (hdb_mfma_f32s.hip) -> the number of queries of a CALL from which that
// flavour is used (0: no such flavour; the API decides per call, ScanArgs::f32_split).  d <= 384: measured from 16 queries up, never
// slower than the float32 MFMAs and 1.5-1.7x faster from 48 (profiles/r4_f32_bf16_parts.txt); d = 512 / 768: one wave cannot hold the
// query fragments of a whole row, the float32 flavour runs 16-row tiles on one SIMD per 16 queries (2x a pass at any batch size), the
// bf16-part flavour splits K over two waves.
// Other float32 widths ride these geometries (any multiple of 4 up to 768 as one padded slice, hdb_mfma_anyd.h; 1024 / 1536 as two
// slices of 512 / 768, hdb_mfma_ksplit.hip) and follow the geometry's rule.
// ... and the largest call that flavour takes (d = 1024: the paired waves hold 64 queries per launch row, so 65-128 queries read the
// two slices twice -- 2 650 against 2 350 us at 128 queries on 1M rows, profiles/r4_f32_bf16_parts.txt)
extern "C" int hdb_mfma_f32_split_max_q(int d) { return d == 1024 ? 64 : 1 << 30; }
extern "C" int hdb_mfma_f32_split_min_q(int d) {
    if (d == 1024 || d == 1536) return 1;
    const int g = (d == 128 || d == 256 || d == 384 || d == 512 || d == 768) ? d : hdb_mfma_anyd_pad(HDB_F32, d);
    return (g == 128 || g == 256 || g == 384) ? 9 : (g == 512 || g == 768) ? 1 : 0;
}
extern "C" int hdb_launch_mfma_scan_f16_wide(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                             const float* qsq, const float* qscl, int blocks, void* stream, const BatchArgs* f);

// a.ntiles / a.tile_stride are in units of hdb_mfma_tile_rows(dtype, d) rows here.  q: the query fragments' source --
// scaled fp16 copies (+ qscl) for fp16 matrices, the float32 queries themselves (qscl = nullptr) for fp32 ones.
// mode 2 = the whole call in one launch (hdb_mfma_kernel.h): f != nullptr, nq_launch <= hdb_mfma_batch_capacity(), grid = one
// workgroup per CU at most (the in-kernel exchange needs the grid co-resident)
extern "C" int hdb_launch_mfma_scan(const ScanArgs* args, int dtype, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                    const float* qsq, const float* qscl, int max_blocks, int variant, void* stream, const BatchArgs* f) {
    const int g_mfma_variant = variant == 32 ? 32 : variant == 64 ? 64 : 16;
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    if (a.mask) return (int)hipErrorNotSupported;
    const int cus = hdb_cu_count();                      // one persistent workgroup per CU ...
    // ... or, with max_blocks < 0, -max_blocks workgroups per CU: the hardware dispatcher then hands the next workgroup
    // to whichever CU finishes first (CUs differ by up to 40 % in streaming speed, see hdb_mfma_fused.h)
    const int64_t want = max_blocks < 0 ? (int64_t)cus * (-max_blocks) : cus;
    int blocks = (int)(a.ntiles < want ? a.ntiles : want);
    if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    if (mode == 2 && blocks > cus) blocks = cus;
    if (hdb_mfma_ksplit_slices(dtype, a.d) > 0) return mode == 2 ? (int)hipErrorNotSupported : hdb_launch_mfma_ksplit(args, dtype, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, stream);
    if (const int pad = hdb_mfma_anyd_pad(dtype, a.d)) {               // a width without a geometry of its own: one slice of the next wider one
        if (mode == 2) return (int)hipErrorNotSupported;
        if (dtype == HDB_F16) return pad <= 384 ? hdb_launch_mfma_anyd_a(args, pad, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, stream)
                                                : hdb_launch_mfma_anyd_b(args, pad, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, stream);
        if (a.f32_split) return pad <= 384 ? hdb_launch_mfma_anyd_e(args, pad, mode, nq_launch, q16, sqnorm, qsq, nullptr, blocks, stream)
                                           : hdb_launch_mfma_anyd_f(args, pad, mode, nq_launch, q16, sqnorm, qsq, nullptr, blocks, stream);
        return pad <= 384 ? hdb_launch_mfma_anyd_c(args, pad, mode, nq_launch, q16, sqnorm, qsq, nullptr, blocks, stream)
                          : hdb_launch_mfma_anyd_d(args, pad, mode, nq_launch, q16, sqnorm, qsq, nullptr, blocks, stream);
    }
    if (dtype == HDB_F32) return (a.f32_split > 0 && hdb_mfma_f32_split_min_q(a.d) > 0) ? hdb_launch_mfma_scan_f32s(args, mode, nq_launch, q16, sqnorm, qsq, blocks, stream, f)
                                                                                   : hdb_launch_mfma_scan_f32(args, mode, nq_launch, q16, sqnorm, qsq, blocks, stream, f);
    if (dtype != HDB_F16) return (int)hipErrorNotSupported;
    // (mode 2 promises hdb_mfma_batch_capacity() queries in ONE launch: only the two-tile launcher holds more than 128, whatever the variant)
    if (nq_launch > 128 && hdb_mfma_qt2_supported(a.d) && (g_mfma_variant != 32 || (mode == 2 && a.d != 384)))
        return hdb_launch_mfma_scan_f16_qt2(args, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, stream, f);
    const int v = g_mfma_variant;
    switch (a.d) {
        case 128: case 256: return hdb_launch_mfma_scan_f16_narrow(args, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, v, stream, f);
        case 384: return hdb_launch_mfma_scan_f16_d384(args, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, v, stream, f);
        case 512: case 640: case 768: return hdb_launch_mfma_scan_f16_mid(args, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, v, stream, f);
        case 1024: case 1536: return hdb_launch_mfma_scan_f16_1k(args, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, v, stream, f);
        default: return hdb_launch_mfma_scan_f16_wide(args, mode, nq_launch, q16, sqnorm, qsq, qscl, blocks, stream, f);     // 896, 1152, 1280, 1408
    }
}

extern "C" int hdb_launch_q_to_f16(const float* Q, int nq, int d, void* q16, float* qscl, void* stream) {
    hipLaunchKernelGGL(hdb_q_to_f16_kernel, dim3(nq), dim3(64), 0, (hipStream_t)stream, Q, nq, d, (_Float16*)q16, qscl);
    return (int)hipGetLastError();
}

extern "C" int hdb_launch_rescore_euclid(unsigned long long* cand, const uint32_t* cnt, uint32_t cap, int nq_launch, const void* V,
                                         int dtype, int d, const float* Q, const float* qsq, int q0, const float* bias, void* stream) {
    const dim3 grid(64, nq_launch);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HDB_F16) {
        if (bias) hipLaunchKernelGGL((hdb_rescore_euclid_kernel<_Float16, true>), grid, dim3(256), 0, st, cand, cnt, cap, (const _Float16*)V, d, Q, qsq, q0, bias);
        else hipLaunchKernelGGL((hdb_rescore_euclid_kernel<_Float16, false>), grid, dim3(256), 0, st, cand, cnt, cap, (const _Float16*)V, d, Q, qsq, q0, bias);
    } else {
        if (bias) hipLaunchKernelGGL((hdb_rescore_euclid_kernel<float, true>), grid, dim3(256), 0, st, cand, cnt, cap, (const float*)V, d, Q, qsq, q0, bias);
        else hipLaunchKernelGGL((hdb_rescore_euclid_kernel<float, false>), grid, dim3(256), 0, st, cand, cnt, cap, (const float*)V, d, Q, qsq, q0, bias);
    }
    return (int)hipGetLastError();
}

