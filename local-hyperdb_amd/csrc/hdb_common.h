// hdb_common.h -- shared device helpers for the gfx950 ranking kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define HDB_WAVE 64

// ---- candidate / selection geometry -------------------------------------------------------
// Candidates that pass the per-query threshold are appended to a per-query list of CAP packed
// 64-bit entries: (orderable score key << 32) | (0xFFFFFFFF - local row).  Sorting those
// DEScending gives (score descending, row ascending) -- the canonical order of the build.
#define HDB_CAND_CAP 8192
#define HDB_RADIX_BITS 8
#define HDB_RADIX_BINS 256

struct ScanArgs {
    const void* V;          // matrix, row-major
    int64_t n;              // rows in V
    int32_t d;              // elements per row
    int32_t row_bytes;      // d * sizeof(T)
    int32_t nchunks;        // 16-byte chunks per row (row_bytes / 16) for the vector kernels
    const void* Q;          // queries [nq][d] in the accumulate type (float, or double for f64)
    int32_t q0;             // first query handled by this launch (blockIdx.y adds to it)
    int32_t metric;         // hdb_metric
    const float* inv_norm;  // [n] 1/||v|| (cosine) or nullptr
    const float* qinv;      // [nq] 1/||q|| (cosine) or nullptr
    const float* bias;      // [n] or nullptr
    const uint8_t* mask;    // [n] or nullptr
    // row tiling: tile t covers rows [t*tile_stride*16, +16)
    int64_t ntiles;         // number of 16-row tiles to process
    int64_t tile_stride;    // 1 = dense scan; >1 = strided sample
    // MODE 0 output
    float* scores;          // [nq][ld]
    int64_t ld;
    // MODE 1 output
    const float* thr;       // [nq]
    uint32_t* cnt;          // [nq]
    unsigned long long* cand;  // [nq][cap]
    uint32_t cap;
    int32_t nq;             // total number of queries behind Q
    int32_t raw;            // 1: keep NaN scores as NaN (per-metric functions); 0: NaN -> -inf (ranking)
    uint32_t* tile_ctr;     // MFMA filter pass: zeroed counter that hands out tiles dynamically (nullptr: static split)
    int64_t dyn_min_bytes;  // ... only for passes of at least this many bytes of V per workgroup
    int32_t dyn_heavy;      // ... also when all eight waves multiply (more than half of a launch's query capacity)
    int32_t f32_split;      // float32 rows on the MFMA scan: 1 = multiply as bf16 parts (hdb_mfma_f32s.hip) where that flavour exists
    // K-split MFMA scan (rows too wide for one wave's query fragments, hdb_mfma_ksplit.hip): one launch per K slice; a launch reads
    // `ks_bytes`-wide pieces of the rows at byte offset ks_off (row pitch ks_pitch, query pitch ks_dfull elements), starts its
    // accumulators from ks_partial_in (nullptr: zero) and, in MODE 3, stores the raw sums to ks_partial_out ([query][ks_ld])
    int64_t ks_pitch; int32_t ks_off; int32_t ks_dfull;
    int32_t ks_valid;       // bytes of this slice that exist in the row (0 = all of it).  Rows of ANY width that is a multiple of 16 bytes ride the
                            // geometry of the next multiple of 256 bytes (hdb_mfma_anyd_*.hip): chunks past the end of a row are not fetched
                            // (the staging re-reads chunk 0 instead) and the query fragments are zero there, so they add nothing
    const float* ks_partial_in; float* ks_partial_out; int64_t ks_ld;
};

// Extra arguments of the single-launch top-k (hdb_mfma_fused.h): sample plan, exchange block, outputs.
struct FusedArgs {
    const float* Qraw;              // [nq][d] float32 queries as the caller passed them
    int32_t nq;
    int64_t s_tiles, s_stride;      // phase A: strided sample (tiles of R rows)
    uint32_t epoch;                 // != 0, different for every launch on this control block
    uint32_t timeout_ticks;         // s_memrealtime ticks (100 MHz) a spin may last
    uint32_t* ctl;                  // [0] done counter, [1] abort word, [2..5] candidate counters, [32] tile counter (own cache line): zero between calls
    unsigned long long* gran;       // [grid][32] granules
    unsigned long long* cand;       // [nq][cap]
    uint32_t cap, k, kk;
    int64_t row_base;
    int64_t* idx_out; float* score_out; int32_t* status;
    float* thr_out;                 // [nq] thresholds (diagnostics)
    // "local" flavour (round 4, short matrices: every workgroup's tiles fit its parking area): no row sample and no exchange -- a
    // workgroup parks the scores of ALL its tiles, takes the local_m-th largest (a lower bound of it) as ITS OWN threshold and emits
    // the rows at or above it; the last workgroup checks that no workgroup's threshold reaches the k-th best of the union.
    // Its lists are SLOTTED: workgroup w writes its rows of query q to cand[q][w * local_slot ...] and leaves {count, b_w} in the
    // granule area -- no slot reservation, no atomic (fp16 euclidean excepted: its near-duplicate re-score wants a compact list).
    int32_t local; uint32_t local_m; uint32_t local_slot;
};

// Extra arguments of the single-launch BATCHED top-k (hdb_mfma_kernel.h, MODE 2): up to 256 queries, one launch does query
// preparation, the strided row sample, the per-query thresholds, the filter pass over all rows and every query's final top-k.
// Control block (BatchArgs::ctl, zero when allocated and again when a launch has finished), in 32-bit words:
//   [0] workgroups done with the filter pass, [1] workgroups that left the kernel, [32] tile counter (a line of its own),
//   [64] abort word, [128 .. 128+256) candidates appended per query.  Byte 2048: 256 threshold granules {epoch, key};
//   byte 4096: sample granules [query][workgroup][8] x {epoch, key}.
#define HDB_BATCH_MAXQ 256
#define HDB_BATCH_CTL_DONE 0
#define HDB_BATCH_CTL_EXIT 1
#define HDB_BATCH_CTL_TILE 32
#define HDB_BATCH_CTL_ABORT 64
#define HDB_BATCH_CTL_CNT 128
#define HDB_BATCH_THRW_BYTE 2048
#define HDB_BATCH_GRAN_BYTE 4096
struct BatchArgs {
    const void* Qraw;               // [nq][d] queries as the caller passed them (float32)
    int64_t s_tiles, s_stride;      // strided row sample (tiles of R rows)
    uint32_t epoch;                 // != 0, different for every launch on this control block
    uint32_t timeout_ticks;         // s_memrealtime ticks (100 MHz) a spin may last
    uint32_t* ctl;
    uint32_t k, kk;
    int64_t row_base;
    int64_t* idx_out; float* score_out; int32_t* status;
    int32_t centre;                 // pearson: the kernel centres its queries (hdb_qcentre_kernel) and multiplies by 1/sd_q; the launch is the cosine one, aux = 1/(sd_v d)
};

// Arguments of the single-launch top-k of the bit metrics (hdb_bits_fused.hip): 1-4 hamming / jaccard queries in one launch.
// Sign bits of the stored rows (hamming / jaccard): blocks of 256 rows, word-major inside a block -- [block][w][256 rows] -- so the
// W x 1 KiB of a block are one contiguous piece (a wave step of the bit scans: 64 lanes x 16 B per word) and a pass streams the
// array front to back.  (Round 2 kept whole word planes, [w][npad]: a wave step touched W places 4*npad bytes apart and the
// pass over 480 MB ran at 5.3-5.6 TB/s.)  npad is a multiple of 256.
#define HDB_BITS_BLOCK 256
__host__ __device__ __forceinline__ int64_t hdb_bits_word(int64_t row, int w, int W) {
    return (((row >> 8) * W + w) << 8) + (row & 255);
}
// the uint4 (four consecutive rows) of row quad i, word w
__device__ __forceinline__ const uint4* hdb_bits_quad(const uint32_t* bits, int64_t i, int w, int W) {
    return reinterpret_cast<const uint4*>(bits + (((i >> 6) * W + w) << 8) + 4 * (i & 63));
}
// ... loaded once per pass.  NT: non-temporal -- 3-4 % faster when the bits do not fit the 256 MB Infinity Cache (N = 10M x 384: 114 vs 119 us
// per call), 2-3 % slower when they do and a query stream keeps finding them there (5M rows: 75 vs 73 us; profiles/r3_bits_variants.txt)
typedef unsigned int hdb_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 hdb_bits_load(const uint32_t* bits, int64_t i, int w, int W) {
    const hdb_u32x4* p = reinterpret_cast<const hdb_u32x4*>(bits + (((i >> 6) * W + w) << 8) + 4 * (i & 63));
    const hdb_u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Per-query candidate counters of the multi-kernel pipeline (ScanArgs::cnt): one per 128-byte cache line -- atomics to one line
// serialize (~88 per us), and a 16-query filter pass on a small matrix flushes 512 blocks x 16 counters.  Index: cnt[q * HDB_CNT_STRIDE].
#define HDB_CNT_STRIDE 32

struct BitsArgs {
    const uint32_t* bits; int64_t npad; int32_t W;
    int64_t n; int32_t d;
    const float* Qraw; int32_t nq;
    int64_t ntiles, s_tiles, s_stride;          // 16-row tiles
    const float* bias; const uint8_t* mask;
    uint32_t epoch, timeout_ticks;
    uint32_t* ctl;                              // the control block of the batched single launch (HDB_BATCH_* layout)
    unsigned long long* cand; uint32_t cap, k, kk;
    int64_t row_base;
    int64_t* idx_out; float* score_out; int32_t* status;
    // round 4: no row sample and no exchange -- every workgroup filters with the 8-th best score of its OWN first pieces, keeps the
    // rows at or above the 8-th best of everything it collected (ties included) and the owner of a query checks those thresholds
    // against the k-th best of the union (hdb_bits_fused.hip)
    int32_t local;
};

// ---- fp16 copy of a query for the matrix pipe ----------------------------------------------------------
// Power-of-two scale that puts the largest magnitude of a query in [2^14, 2^15): an fp32 element above 65504 would
// otherwise become inf in fp16 and one below 6e-5 a subnormal.  Exact, and undone for free in the kernel epilogue
// (1/scale rides on the per-query multiplier).  Queries that already are fp16 values keep all their bits.
__device__ __forceinline__ float hdb_q16_scale(float amax) {
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.f;
    int ex = 0;
    (void)frexpf(amax, &ex);                                  // amax = m * 2^ex, m in [0.5, 1)
    int sh = 15 - ex;
    sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
    return ldexpf(1.f, sh);
}

// ---- strided row sample -------------------------------------------------------------------------
// Sample tile t of a strided sample sits at tile index t*stride + jitter(t), jitter in [0, stride): a fixed
// pseudo-random offset inside each stride window, so that a matrix with periodic structure (rows inserted
// round-robin by class, ...) cannot alias with the sampling period.  stride == 1 (dense scan) gives jitter 0.
__device__ __forceinline__ int64_t hdb_tile_index(int64_t t, int64_t stride) {
    if (stride == 1) return t;                       // dense pass: keep the division off the load path
    const uint32_t h = (uint32_t)t * 2654435761u;
    return t * stride + (int64_t)((h >> 8) % (uint32_t)stride);
}

// ---- orderable float keys -----------------------------------------------------------------
__device__ __forceinline__ uint32_t hdb_f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // ascending uint order == float order
}
__device__ __forceinline__ float hdb_key2f(uint32_t k) {
    uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long hdb_pack(float s, uint32_t row) {
    return ((unsigned long long)hdb_f2key(s) << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}
// NaN -> -inf (reference ranking_algorithm.py:174) and -0.0 -> +0.0 (so equal floats have equal keys)
__device__ __forceinline__ float hdb_canon(float s) {
    if (s != s) s = -INFINITY;
    return s + 0.0f;
}

// ---- 4 row sums over a 16-lane group with DPP ---------------------------------------------------
// Each 16-lane group holds partial sums a0..a3 of its 4 rows in every lane.  An "ownership"
// butterfly halves the number of live values at each of the first two exchanges (xor 15 =
// row_mirror, xor 7 = row_half_mirror), then two plain exchanges (quad_perm xor 2, xor 1) finish:
// 5 DPP adds instead of 16, no LDS, no bpermute.  On return every lane holds the complete sum of
// row  hdb_owned_row(l16) = 2*bit2(l16) + bit3(l16)  of its group.
template <int CTRL>
__device__ __forceinline__ float hdb_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double hdb_dpp(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#define HDB_DPP_MIRROR 0x140
#define HDB_DPP_HALF_MIRROR 0x141
#define HDB_DPP_XOR2 0x4E
#define HDB_DPP_XOR1 0xB1
template <typename A>
__device__ __forceinline__ A hdb_rows4_sum(A a0, A a1, A a2, A a3, int l16) {
    const bool b3 = (l16 & 8) != 0, b2 = (l16 & 4) != 0;
    const A v01 = (b3 ? a1 : a0) + hdb_dpp<HDB_DPP_MIRROR>(b3 ? a0 : a1);
    const A v23 = (b3 ? a3 : a2) + hdb_dpp<HDB_DPP_MIRROR>(b3 ? a2 : a3);
    A w = (b2 ? v23 : v01) + hdb_dpp<HDB_DPP_HALF_MIRROR>(b2 ? v01 : v23);
    w += hdb_dpp<HDB_DPP_XOR2>(w);
    w += hdb_dpp<HDB_DPP_XOR1>(w);
    return w;
}
// Wave-wide maximum of a 32-bit key in every lane: four DPP steps inside each 16-lane row, then the four row maxima through
// v_readlane (the shuffle form is six dependent LDS-crossbar round trips: 8 rounds of it cost the threshold exchange ~2 us).
__device__ __forceinline__ uint32_t hdb_wave_max_dpp(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, HDB_DPP_XOR1, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, HDB_DPP_XOR2, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, HDB_DPP_HALF_MIRROR, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, HDB_DPP_MIRROR, 0xF, 0xF, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ int hdb_owned_row(int l16) { return ((l16 >> 2) & 1) * 2 + ((l16 >> 3) & 1); }

// ---- launch helpers (host) -------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: `done` holds one bit per device ordinal (the
// static word lives in each kernel instantiation's launcher), so a second GPU in the process sets it again.
static inline hipError_t hdb_lds_attr_once(const void* fn, int bytes, unsigned long long* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (*done & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) *done |= bit;
    return e;
}

// Compute units of the current device (256 on MI355X), cached per device ordinal: persistent kernels launch one
// workgroup per CU.
static inline int hdb_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cached[dev & 63];
    if (c <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        c = v;
    }
    return c;
}
static inline int hdb_grid_for(int64_t work_items, int per_block, int max_blocks) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b < (int64_t)max_blocks ? b : (int64_t)max_blocks);
}
