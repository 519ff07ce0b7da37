// hdb_api.hip -- the C ABI of include/hyperdb_hip.h: handle, workspace and the top-k pipeline.
//
// Pipeline of hdb_topk for a chunk of queries (everything enqueued on the caller's stream):
//   n <= CAP            : thr = -inf -> scan(filter) -> finalize                (every row is a candidate)
//   otherwise           : scan(scores) over a strided row sample
//                         -> 4 radix-histogram passes -> thr[q] = m-th largest sample score
//                         -> scan(filter) over all rows (the only pass that touches all of V)
//                         -> finalize (sort <= 8192 candidates per query, emit k)
// hdb_topk_exact: scan(scores) over all rows -> 4 histogram passes -> collect (+ordered ties)
//                 -> finalize.  Used for hamming (integer scores, massive ties), for queries whose
//                 sampled threshold failed, and by tests as the on-device cross-check.
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"
#include <string>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <algorithm>
#include <vector>
#include <chrono>
#include <atomic>

// launchers implemented in the kernel translation units
extern "C" {
int hdb_launch_scan(const ScanArgs* args, int dtype, int mode, int nq_launch, int max_blocks, void* stream);
int hdb_launch_rownorm(const void* V, int64_t n, int d, int dtype, float* inv_norm, float* sqnorm, int* nan_flag, void* stream);
int hdb_launch_qprep(const void* Q, int nq, int d, bool f64, float* qinv, float* qsq, int* qnan, void* q16, float* qscl, void* stream);
int hdb_launch_qprep2(const void* Q, int nq, int d, bool f64, float* qinv, float* qsq, int* qnan, void* q16, float* qscl,
                      float* thr_init, uint32_t* cnt_init, uint32_t* qbits, int W, void* stream);
int hdb_launch_signpack(const void* V, int64_t n, int d, int dtype, int64_t row0, uint32_t* bits, void* stream);
int hdb_launch_qsign(const void* Q, int nq, int d, bool f64, int W, uint32_t* qbits, void* stream);
int hdb_launch_hamming(const ScanArgs* args, int mode, int nq_launch, const uint32_t* bits, int64_t npad, int W,
                       const uint32_t* qbits, void* stream);
int hdb_launch_hist(const float* scores, int64_t n, int64_t ld, int nq, uint32_t* hist, int pass, uint32_t k, void* stream);
int hdb_launch_thr(const uint32_t* hist, int nq, int npass, uint32_t m, uint32_t sample_n, float* thr, uint32_t* cnt, void* stream);
int hdb_launch_fill_thr(float* thr, uint32_t* cnt, int nq, float v, void* stream);
int hdb_launch_sample_thr(const float* scores, int64_t n, int64_t ld, int nq, uint32_t m, float* thr, uint32_t* cnt, uint32_t* tile_ctr, void* stream);
int hdb_launch_collect(const float* scores, int64_t n, int64_t ld, int nq, const uint32_t* hist, int npass, uint32_t k, uint32_t* cnt,
                       unsigned long long* cand, uint32_t cap, uint32_t* tie_info, void* stream);
int hdb_launch_finalize(const unsigned long long* cand, const uint32_t* cnt, uint32_t cap, int nq, uint32_t k, uint32_t kk,
                        int64_t row_base, int64_t* idx_out, float* score_out, int32_t* status, const int* qnan, int threads, int inf_status, void* stream);
int hdb_launch_status_nan(const int* qnan, int nq, int32_t* status, void* stream);
int hdb_launch_merge(const void* idx_base, int64_t idx_stride, const void* score_base, int64_t score_stride,
                     const void* status_base, int64_t status_stride, int parts, int nq, uint32_t k, int64_t* idx_out,
                     float* score_out, int32_t* status_out, void* stream);
int hdb_launch_rowstats(const void* V, int64_t n, int d, int dtype, float* pscale, void* stream);
int hdb_launch_qcentre(const void* Q, int nq, int d, bool f64, void* Qc, float* qscale, void* stream);
int hdb_launch_recency(const double* ts, int64_t n, double rb, double ts_max, float* out, void* stream);
int hdb_launch_recency2(const double* ts, const uint8_t* mask, int64_t n, double rb, double ts_max, double first_max, float* out, void* stream);
int hdb_launch_maskbias(const uint8_t* mask, const float* bias, int64_t n, float* out, void* stream);
int hdb_mfma_supported(int dtype, int d, int metric);
int hdb_mfma_tile_rows(int dtype, int d);
int hdb_launch_mfma_scan(const ScanArgs* args, int dtype, int mode, int nq_launch, const void* q16, const float* sqnorm,
                         const float* qsq, const float* qscl, int max_blocks, int variant, void* stream, const BatchArgs* f);
int hdb_mfma_batch_capacity(int dtype, int d);
int hdb_mfma_ksplit_slices(int dtype, int d);
int hdb_mfma_anyd_pad(int dtype, int d);
int hdb_mfma_f32_split_min_q(int d);
int hdb_mfma_f32_split_max_q(int d);
int hdb_l1_tile_supported(int dtype, int d);
int hdb_launch_l1_tile(const ScanArgs* args, int dtype, int mode, int nq_launch, int max_blocks, void* stream);
int hdb_bits_fused_supported(int metric, int nq, int W, uint32_t kk);
int hdb_launch_bits_fused(const BitsArgs* args, int jaccard, int max_blocks, void* stream);
size_t hdb_mfma_batch_ctl_bytes(int wgs);
int hdb_mfma_fused_supported(int dtype, int d, int metric, int nq, uint32_t kk);
size_t hdb_mfma_fused_ctl_bytes(void);
int hdb_launch_mfma_fused(const ScanArgs* args, int dtype, const FusedArgs* fa, int max_blocks, void* stream);
int hdb_mfma_fused_local_tiles(int dtype, int d, int metric, int nq);
int hdb_launch_q_to_f16(const float* Q, int nq, int d, void* q16, float* qscl, void* stream);
int hdb_sort_temp_bytes(int64_t n, size_t* bytes);
int hdb_launch_full_sort(const float* scores, int64_t n, int64_t k, int64_t row_base, uint32_t* work, void* temp, size_t temp_bytes,
                         int64_t* idx_out, float* score_out, void* stream);
int hdb_launch_gather_rows(const void* V, const int64_t* rows, int64_t m, int row_bytes, void* out, const float* inv_in,
                           const float* sq_in, float* inv_out, float* sq_out, int* nan_flag, void* stream);
int hdb_launch_rescore_euclid(unsigned long long* cand, const uint32_t* cnt, uint32_t cap, int nq_launch, const void* V, int dtype, int d,
                              const float* Q, const float* qsq, int q0, const float* bias, void* stream);
}

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(HDB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)
#define LAUNCH_TRY(expr)                                                                           \
    do {                                                                                           \
        int e_ = (expr);                                                                           \
        if (e_ != 0) return fail(HDB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e_)); \
    } while (0)

struct hdb_index {
    const void* V = nullptr;
    int64_t n = 0;
    int32_t d = 0;
    int dtype = HDB_F32;
    int device = 0;
    int64_t row_base = 0;
    // per-row caches (owned)
    float* inv_norm = nullptr;
    float* sqnorm = nullptr;
    int64_t cache_rows = 0;
    int* nan_flag = nullptr;          // device int
    hipStream_t build_stream = nullptr;
    // hamming sign bits (owned, lazy)
    uint32_t* bits = nullptr;
    int64_t bits_npad = 0;
    int W = 0;
    bool bits_valid = false;
    int64_t bits_done = 0;            // rows [0, bits_done) are packed for the CURRENT matrix (hdb_index_extend keeps them: only the appended rows are packed)
    // pearson per-row scale 1/(sd*d) (owned, lazy)
    float* pscale = nullptr;
    int64_t pscale_rows = 0;
    bool pscale_valid = false;
    int64_t pscale_done = 0;          // likewise
    // borrowed
    const float* bias = nullptr;
    const uint8_t* mask = nullptr;
    // mask folded into a bias vector for the MFMA scan (owned, rebuilt per call: the mask and bias are borrowed)
    float* mbias = nullptr;
    int64_t mbias_rows = 0;
    // control block of the single-launch pipeline (owned): counters + exchange granules, zero between calls
    char* fctl = nullptr;
    uint32_t fused_epoch = 0;
    // ... and of the single-launch BATCHED pipeline (hdb_mfma_kernel.h, MODE 2): counters, threshold words, sample granules
    char* bctl = nullptr;
    // device copy of the result record of hdb_topk_host (owned)
    char* rec = nullptr;
    size_t rec_bytes = 0;
    // scratch (owned)
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // options
    int64_t max_blocks = 0;           // 0 = automatic (row scan: 2-4 workgroups per CU, see hdb_launch_scan)
    int64_t force_exact = 0;
    int64_t sample_target = 0;        // 0 = automatic
    int64_t mfma_min_q = 1;
    int64_t use_mfma = 1;
    int64_t exact_bytes = (int64_t)1 << 30;
    int64_t bits_fused = 1;           // hamming / jaccard: try the sampled-threshold path first (exact path when it fails)
    int64_t bits_local = 1;           // ... its single launch without row sample and exchange: every workgroup its own threshold (hdb_bits_fused.hip, round 4)
    // knobs of the dispatch: -1 = the measured rule (tools/sweep_dispatch.py, profiles/r3_dispatch_few_queries.txt), else a fixed limit
    int64_t fused_max_q = -1;         // hdb_mfma_fused_kernel takes calls of up to this many queries
    int64_t f32_min_q = -1;           // float32 matrices: the matrix-core scan from this many queries on
    int64_t f32_split = 1;            // ... as bf16 parts (hdb_mfma_f32s.hip) where that flavour exists, the matrix is finite and the call has at least
    int64_t f32_split_min_q = -1;     //     this many queries (-1: hdb_mfma_f32_split_min_q(d), the measured crossover)
    int flags_host = -1;              // host copy of *nan_flag (1 = a NaN row, 2 = a row with an infinite sum of squares); -1 = not fetched since the last build
    int64_t bits_max_q = -1;          // hamming / jaccard: the single launch (four queries at a time) up to this many queries
    int64_t host_direct = 1;          // hdb_topk_host: kernels write a pinned host record themselves (no D2H copy)
    int64_t dyn_tiles = 1;            // MFMA filter pass: hand tiles out from a counter (0: static split)
    int64_t dyn_min_mb = 16;          // ... for passes of at least this many MiB of V per workgroup
    int64_t dyn_heavy = 0;            // ... also when all eight waves multiply (measured: 1.3-5 % slower at 256 queries, profiles/r3_q256_clock.json)
    int64_t host_poll = 1;            // hdb_topk_host + single-launch pipeline + pinned record: poll the status words instead of the stream
    int64_t use_fused = 1;            // 1-4 dot / cosine queries on an fp16 matrix: the whole call in ONE kernel (hdb_mfma_fused.h)
    int64_t use_local = 1;            // ... short matrices: its local flavour (no row sample, no exchange; every workgroup its own threshold)
    int64_t local_m = 0;              // ... rows every workgroup emits at least (0 = automatic: ~3072 / workgroups, 8 .. 32)
    int64_t local_max_tiles = 4;      // ... while a workgroup has at most this many tiles (the parking area holds 16)
    int64_t local_small = 0;          // ... 1: also for matrices of up to 8192 rows (measured slower than the three launches)
    int64_t local_max_q = 1;          // ... for calls of up to this many queries (two to four: the batched single launch is faster -- 36 vs 45 us at 20k rows, profiles/r4_latency_map.txt)
    int64_t use_l1_tile = 1;          // manhattan: dense passes through the LDS-staged tile kernel (hdb_l1_tile.hip)
    int64_t l1_packed = 1;            // ... fp16 rows and fp16-valued queries: packed fp16 differences (0: always float32, for A/B runs)
    int64_t use_batch1 = 1;           // 5+ queries (euclidean: 1+) on the matrix cores, k <= 128: the whole call in ONE launch per <= 256 queries (needs use_fused)
    int64_t fused_timeout_us = 2000;  // bound of every in-kernel spin of those kernels
    int64_t finalize_threads = 1024;  // workgroup size of hdb_finalize_kernel (256 | 512 | 1024)
    int64_t mfma_variant = 16;        // MFMA shape of the d=384 256-query pass (16 | 32)
    // stats of the last hdb_topk call
    int64_t st_sample_rows = 0, st_sample_m = 0, st_path = 0, st_chunks = 0, st_mfma = 0, st_host_direct = 0, st_fused = 0, st_local = 0, st_f32s = 0;
    // host-side timing of hdb_topk_host (always on: four clock reads per call), cumulative since "host_timing_reset":
    // entry -> launch, the launch call itself, launch -> record complete (poll / stream wait), calls
    int64_t ht_pre_ns = 0, ht_launch_ns = 0, ht_wait_ns = 0, ht_calls = 0;
    std::chrono::steady_clock::time_point ht_l0, ht_l1;     // around the launch of the single-launch pipelines (topk_impl)
    int64_t ht_attr_ns = 0;            // ... of which hipSetDevice + hipPointerGetAttributes (is the caller's record pinned?)
    // optional HIP-event timing of the dominant kernel (the pass over all of V)
    int64_t profile = 0;
    std::vector<hipEvent_t> ev_pool;      // pairs: [2i] start, [2i+1] stop
    size_t ev_used = 0;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Bump {
    char* base; size_t off = 0, cap;
    Bump(char* b, size_t c) : base(b), cap(c) {}
    template <typename T> T* take(size_t count) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += count * sizeof(T);
        return p;
    }
};

static int ensure_ws(hdb_index* ix, size_t bytes) {
    if (bytes <= ix->ws_bytes) return HDB_OK;
    if (ix->ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ix->ws)); ix->ws = nullptr; ix->ws_bytes = 0; }
    bytes = align_up(bytes + (bytes >> 2), 1 << 20);
    HIP_TRY(hipMalloc((void**)&ix->ws, bytes));
    ix->ws_bytes = bytes;
    return HDB_OK;
}

extern "C" int hdb_version(void) { return 100; }
extern "C" const char* hdb_last_error(void) { return g_err.c_str(); }

static int build_caches(hdb_index* ix, hipStream_t st) {
    if (ix->n > ix->cache_rows) {
        if (ix->inv_norm) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ix->inv_norm)); HIP_TRY(hipFree(ix->sqnorm)); }
        const int64_t rows = ix->n + ix->n / 4 + 64;
        HIP_TRY(hipMalloc((void**)&ix->inv_norm, rows * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&ix->sqnorm, rows * sizeof(float)));
        ix->cache_rows = rows;
    }
    HIP_TRY(hipMemsetAsync(ix->nan_flag, 0, sizeof(int), st));
    if (ix->n > 0) LAUNCH_TRY(hdb_launch_rownorm(ix->V, ix->n, ix->d, ix->dtype, ix->inv_norm, ix->sqnorm, ix->nan_flag, st));
    ix->flags_host = -1;
    ix->bits_valid = false; ix->bits_done = 0;         // a new matrix: nothing of the lazy caches survives
    ix->pscale_valid = false; ix->pscale_done = 0;
    ix->build_stream = st;
    return HDB_OK;
}

extern "C" int hdb_index_create(hdb_index** out, const void* dev_V, int64_t n, int32_t d, int dtype, int device,
                                int64_t row_base, void* stream) {
    if (!out) return fail(HDB_ERR_ARG, "hdb_index_create: out is null");
    if (n < 0 || d <= 0) return fail(HDB_ERR_ARG, "hdb_index_create: need n >= 0 and d > 0");
    if (n > 0 && !dev_V) return fail(HDB_ERR_ARG, "hdb_index_create: matrix pointer is null");
    if (dtype != HDB_F16 && dtype != HDB_F32 && dtype != HDB_F64) return fail(HDB_ERR_ARG, "hdb_index_create: dtype must be f16/f32/f64");
    if (n >= ((int64_t)1 << 32) - 1) return fail(HDB_ERR_ARG, "hdb_index_create: at most 2^32-2 rows per shard");
    if ((int64_t)d * 8 > 60 * 1024) return fail(HDB_ERR_ARG, "hdb_index_create: d too large for the query LDS tile");
    HIP_TRY(hipSetDevice(device));
    hdb_index* ix = new hdb_index();
    ix->V = dev_V; ix->n = n; ix->d = d; ix->dtype = dtype; ix->device = device; ix->row_base = row_base;
    hipError_t e = hipMalloc((void**)&ix->nan_flag, sizeof(int));
    if (e != hipSuccess) { delete ix; return fail(HDB_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    int rc = build_caches(ix, (hipStream_t)stream);
    if (rc != HDB_OK) { hdb_index_destroy(ix); return rc; }
    *out = ix;
    return HDB_OK;
}

extern "C" int hdb_index_update(hdb_index* ix, const void* dev_V, int64_t n, void* stream) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_update: null index");
    if (n < 0 || (n > 0 && !dev_V)) return fail(HDB_ERR_ARG, "hdb_index_update: bad matrix");
    if (n >= ((int64_t)1 << 32) - 1) return fail(HDB_ERR_ARG, "hdb_index_update: at most 2^32-2 rows per shard");
    HIP_TRY(hipSetDevice(ix->device));
    ix->V = dev_V; ix->n = n;
    ix->bias = nullptr; ix->mask = nullptr;
    return build_caches(ix, (hipStream_t)stream);
}

extern "C" int hdb_index_rebase(hdb_index* ix, const void* dev_V) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_rebase: null index");
    if (ix->n > 0 && !dev_V) return fail(HDB_ERR_ARG, "hdb_index_rebase: matrix pointer is null");
    ix->V = dev_V;
    return HDB_OK;
}

extern "C" int hdb_index_set_row_base(hdb_index* ix, int64_t row_base) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_set_row_base: null index");
    if (row_base < 0) return fail(HDB_ERR_ARG, "hdb_index_set_row_base: row_base must be >= 0");
    ix->row_base = row_base;
    return HDB_OK;
}

extern "C" int hdb_index_extend(hdb_index* ix, int64_t new_n, void* stream) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_extend: null index");
    if (new_n < ix->n) return fail(HDB_ERR_ARG, "hdb_index_extend: new_n must not shrink the matrix (use hdb_index_update)");
    if (new_n >= ((int64_t)1 << 32) - 1) return fail(HDB_ERR_ARG, "hdb_index_extend: at most 2^32-2 rows per shard");
    if (new_n == ix->n) return HDB_OK;
    if (!ix->V) return fail(HDB_ERR_ARG, "hdb_index_extend: no matrix registered");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t old_n = ix->n;
    if (new_n > ix->cache_rows) {                      // grow the per-row caches, keeping the old values
        const int64_t rows = new_n + new_n / 2 + 64;
        float *inv2 = nullptr, *sq2 = nullptr;
        HIP_TRY(hipMalloc((void**)&inv2, rows * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&sq2, rows * sizeof(float)));
        if (old_n > 0) {
            HIP_TRY(hipMemcpyAsync(inv2, ix->inv_norm, old_n * sizeof(float), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(sq2, ix->sqnorm, old_n * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        if (ix->inv_norm) { HIP_TRY(hipFree(ix->inv_norm)); HIP_TRY(hipFree(ix->sqnorm)); }
        ix->inv_norm = inv2; ix->sqnorm = sq2; ix->cache_rows = rows;
    }
    const size_t elem = ix->dtype == HDB_F16 ? 2 : ix->dtype == HDB_F32 ? 4 : 8;
    const char* tail = (const char*)ix->V + (size_t)old_n * ix->d * elem;
    LAUNCH_TRY(hdb_launch_rownorm(tail, new_n - old_n, ix->d, ix->dtype, ix->inv_norm + old_n, ix->sqnorm + old_n, ix->nan_flag, st));
    ix->n = new_n;
    ix->flags_host = -1;
    ix->bits_valid = false;                            // (bits_done / pscale_done stay: the next hamming / pearson call packs the appended rows only)
    ix->pscale_valid = false;
    ix->bias = nullptr; ix->mask = nullptr;            // per-row inputs of the old length no longer apply
    ix->build_stream = st;
    return HDB_OK;
}

extern "C" int hdb_index_gather(hdb_index* ix, const int64_t* dev_rows, int64_t m, void* dev_V_out, void* stream) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_gather: null index");
    if (m < 0 || m > ix->n) return fail(HDB_ERR_ARG, "hdb_index_gather: m must be in [0, n]");
    if (m > 0 && (!dev_rows || !dev_V_out)) return fail(HDB_ERR_ARG, "hdb_index_gather: null argument");
    if (m > 0 && dev_V_out == ix->V) return fail(HDB_ERR_ARG, "hdb_index_gather: the gather is out of place");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t elem = ix->dtype == HDB_F16 ? 2 : ix->dtype == HDB_F32 ? 4 : 8;
    float *inv2 = nullptr, *sq2 = nullptr;
    const int64_t rows = m + m / 4 + 64;
    HIP_TRY(hipMalloc((void**)&inv2, rows * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&sq2, rows * sizeof(float)));
    HIP_TRY(hipMemsetAsync(ix->nan_flag, 0, sizeof(int), st));
    int rc = hdb_launch_gather_rows(ix->V, dev_rows, m, (int)(ix->d * elem), dev_V_out, ix->inv_norm, ix->sqnorm, inv2, sq2,
                                    ix->nan_flag, st);
    if (rc == 0) rc = (int)hipStreamSynchronize(st);          // the old caches (and the caller's old matrix) are free after this
    if (rc != 0) { (void)hipFree(inv2); (void)hipFree(sq2); return fail(HDB_ERR_HIP, std::string("hdb_index_gather: ") + hipGetErrorString((hipError_t)rc)); }
    if (ix->inv_norm) { (void)hipFree(ix->inv_norm); (void)hipFree(ix->sqnorm); }
    ix->inv_norm = inv2; ix->sqnorm = sq2; ix->cache_rows = rows;
    ix->V = dev_V_out; ix->n = m;
    ix->flags_host = -1;
    ix->bits_valid = false; ix->pscale_valid = false; ix->bits_done = 0; ix->pscale_done = 0;
    ix->bias = nullptr; ix->mask = nullptr;
    ix->build_stream = st;
    return HDB_OK;
}

extern "C" void hdb_index_destroy(hdb_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    if (ix->inv_norm) (void)hipFree(ix->inv_norm);
    if (ix->sqnorm) (void)hipFree(ix->sqnorm);
    if (ix->nan_flag) (void)hipFree(ix->nan_flag);
    if (ix->bits) (void)hipFree(ix->bits);
    if (ix->pscale) (void)hipFree(ix->pscale);
    if (ix->mbias) (void)hipFree(ix->mbias);
    if (ix->fctl) (void)hipFree(ix->fctl);
    if (ix->bctl) (void)hipFree(ix->bctl);
    if (ix->ws) (void)hipFree(ix->ws);
    if (ix->rec) (void)hipFree(ix->rec);
    for (hipEvent_t e : ix->ev_pool) (void)hipEventDestroy(e);
    delete ix;
}

extern "C" int hdb_index_has_nan(hdb_index* ix, int* out_flag) {
    if (!ix || !out_flag) return fail(HDB_ERR_ARG, "hdb_index_has_nan: null argument");
    HIP_TRY(hipSetDevice(ix->device));
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, ix->nan_flag, sizeof(int), hipMemcpyDeviceToHost, ix->build_stream));
    HIP_TRY(hipStreamSynchronize(ix->build_stream));
    ix->flags_host = h;
    *out_flag = h & 1;
    return HDB_OK;
}
// Are all rows of the matrix finite with a finite sum of squares?  One 4-byte copy after each build, cached.
static int matrix_is_finite(hdb_index* ix, bool* out) {
    if (ix->flags_host < 0) { int f = 0; int rc = hdb_index_has_nan(ix, &f); if (rc != HDB_OK) return rc; }
    *out = ix->flags_host == 0;
    return HDB_OK;
}

extern "C" int hdb_index_set_bias(hdb_index* ix, const float* dev_bias) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_set_bias: null index");
    ix->bias = dev_bias;
    return HDB_OK;
}

extern "C" int hdb_index_set_row_mask(hdb_index* ix, const uint8_t* dev_mask) {
    if (!ix) return fail(HDB_ERR_ARG, "hdb_index_set_row_mask: null index");
    ix->mask = dev_mask;
    return HDB_OK;
}

extern "C" int hdb_set_option(hdb_index* ix, const char* name, int64_t value) {
    if (!ix || !name) return fail(HDB_ERR_ARG, "hdb_set_option: null argument");
    if (!strcmp(name, "max_blocks")) ix->max_blocks = value;
    else if (!strcmp(name, "force_exact")) ix->force_exact = value;
    else if (!strcmp(name, "sample_target")) ix->sample_target = value;
    else if (!strcmp(name, "mfma_min_q")) ix->mfma_min_q = value;
    else if (!strcmp(name, "use_mfma")) ix->use_mfma = value;
    else if (!strcmp(name, "exact_bytes")) ix->exact_bytes = std::max<int64_t>(1 << 20, value);
    else if (!strcmp(name, "finalize_threads")) { if (value == 256 || value == 512 || value == 1024) ix->finalize_threads = value; }
    else if (!strcmp(name, "mfma_variant")) { if (value == 16 || value == 32 || value == 64) ix->mfma_variant = value; }
    else if (!strcmp(name, "host_direct")) ix->host_direct = value;
    else if (!strcmp(name, "use_fused")) ix->use_fused = value;
    else if (!strcmp(name, "use_batch1")) ix->use_batch1 = value;
    else if (!strcmp(name, "use_local")) ix->use_local = value;
    else if (!strcmp(name, "local_max_q")) ix->local_max_q = value;
    else if (!strcmp(name, "local_max_tiles")) ix->local_max_tiles = std::max<int64_t>(1, value);
    else if (!strcmp(name, "local_small")) ix->local_small = value;
    else if (!strcmp(name, "local_m")) ix->local_m = std::max<int64_t>(0, std::min<int64_t>(value, 64));
    else if (!strcmp(name, "use_l1_tile")) ix->use_l1_tile = value;
    else if (!strcmp(name, "l1_packed")) ix->l1_packed = value;
    else if (!strcmp(name, "host_poll")) ix->host_poll = value;
    else if (!strcmp(name, "dyn_tiles")) ix->dyn_tiles = value;
    else if (!strcmp(name, "dyn_min_mb")) ix->dyn_min_mb = std::max<int64_t>(0, value);
    else if (!strcmp(name, "dyn_heavy")) ix->dyn_heavy = value;
    else if (!strcmp(name, "fused_timeout_us")) ix->fused_timeout_us = std::max<int64_t>(1, value);
    else if (!strcmp(name, "bits_fused")) ix->bits_fused = value;
    else if (!strcmp(name, "bits_local")) ix->bits_local = value;
    else if (!strcmp(name, "fused_max_q")) ix->fused_max_q = value;
    else if (!strcmp(name, "f32_min_q")) ix->f32_min_q = value;
    else if (!strcmp(name, "f32_split")) ix->f32_split = value;
    else if (!strcmp(name, "f32_split_min_q")) ix->f32_split_min_q = value;
    else if (!strcmp(name, "bits_max_q")) ix->bits_max_q = value;
    else if (!strcmp(name, "profile")) { ix->profile = value; ix->ev_used = 0; }
    else if (!strcmp(name, "host_timing_reset")) { ix->ht_pre_ns = ix->ht_launch_ns = ix->ht_wait_ns = ix->ht_calls = ix->ht_attr_ns = 0; }
    else return fail(HDB_ERR_ARG, std::string("hdb_set_option: unknown option ") + name);
    return HDB_OK;
}

extern "C" int hdb_get_stat(hdb_index* ix, const char* name, int64_t* value) {
    if (!ix || !name || !value) return fail(HDB_ERR_ARG, "hdb_get_stat: null argument");
    if (!strcmp(name, "sample_rows")) *value = ix->st_sample_rows;
    else if (!strcmp(name, "sample_m")) *value = ix->st_sample_m;
    else if (!strcmp(name, "path")) *value = ix->st_path;
    else if (!strcmp(name, "chunks")) *value = ix->st_chunks;
    else if (!strcmp(name, "mfma")) *value = ix->st_mfma;
    else if (!strcmp(name, "f32_split")) *value = ix->st_f32s;
    else if (!strcmp(name, "host_direct")) *value = ix->st_host_direct;
    else if (!strcmp(name, "fused")) *value = ix->st_fused;
    else if (!strcmp(name, "local")) *value = ix->st_local;
    else if (!strcmp(name, "cand_cap")) *value = HDB_CAND_CAP;
    else if (!strcmp(name, "n")) *value = ix->n;
    else if (!strcmp(name, "ws_bytes")) *value = (int64_t)ix->ws_bytes;
    else if (!strcmp(name, "scan_launches")) *value = (int64_t)(ix->ev_used / 2);
    else if (!strcmp(name, "host_pre_ns")) *value = ix->ht_pre_ns;
    else if (!strcmp(name, "host_launch_ns")) *value = ix->ht_launch_ns;
    else if (!strcmp(name, "host_wait_ns")) *value = ix->ht_wait_ns;
    else if (!strcmp(name, "host_calls")) *value = ix->ht_calls;
    else if (!strcmp(name, "host_attr_ns")) *value = ix->ht_attr_ns;
    else if (!strcmp(name, "scan_time_ns")) {      // sum over recorded launches; synchronises on the last event
        double total_ms = 0.0;
        for (size_t i = 0; i + 1 < ix->ev_used; i += 2) {
            if (hipEventSynchronize(ix->ev_pool[i + 1]) != hipSuccess) return fail(HDB_ERR_HIP, "hipEventSynchronize failed");
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ix->ev_pool[i], ix->ev_pool[i + 1]) != hipSuccess) return fail(HDB_ERR_HIP, "hipEventElapsedTime failed");
            total_ms += ms;
        }
        *value = (int64_t)(total_ms * 1.0e6);
    }
    else return fail(HDB_ERR_ARG, std::string("hdb_get_stat: unknown stat ") + name);
    return HDB_OK;
}

// Bracket one launch with HIP events on the launch stream (bench.py: roofline.achieved).
static void prof_begin(hdb_index* ix, hipStream_t st) {
    if (!ix->profile) return;
    if (ix->ev_used + 2 > ix->ev_pool.size()) {
        if (ix->ev_pool.size() >= 16384) return;     // bounded
        for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ix->ev_pool.push_back(e); }
    }
    (void)hipEventRecord(ix->ev_pool[ix->ev_used], st);
}
static void prof_end(hdb_index* ix, hipStream_t st) {
    if (!ix->profile || ix->ev_used + 2 > ix->ev_pool.size()) return;
    (void)hipEventRecord(ix->ev_pool[ix->ev_used + 1], st);
    ix->ev_used += 2;
}

static bool metric_ok(int metric) { return metric >= HDB_DOT && metric <= HDB_EUCLIDEAN_DIST; }
#define HDB_FUSED_MAXQ_RULE 4
static bool is_bits_metric(int metric) { return metric == HDB_HAMMING || metric == HDB_JACCARD; }

static int ensure_pscale(hdb_index* ix, hipStream_t st) {
    if (ix->pscale_valid) return HDB_OK;
    const size_t elem = ix->dtype == HDB_F16 ? 2 : ix->dtype == HDB_F32 ? 4 : 8;
    int64_t keep = ix->pscale ? std::min(ix->pscale_done, ix->n) : 0;          // rows whose scale is still good (appended matrix)
    if (ix->n > ix->pscale_rows) {
        const int64_t rows = ix->n + ix->n / 4 + 64;
        float* p2 = nullptr;
        HIP_TRY(hipMalloc((void**)&p2, rows * sizeof(float)));
        if (keep > 0) HIP_TRY(hipMemcpyAsync(p2, ix->pscale, keep * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (ix->pscale) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ix->pscale)); }
        ix->pscale = p2; ix->pscale_rows = rows;
    }
    if (ix->n > keep)
        LAUNCH_TRY(hdb_launch_rowstats((const char*)ix->V + (size_t)keep * ix->d * elem, ix->n - keep, ix->d, ix->dtype, ix->pscale + keep, st));
    ix->pscale_done = ix->n;
    ix->pscale_valid = true;
    return HDB_OK;
}

static int ensure_bits(hdb_index* ix, hipStream_t st) {
    if (ix->bits_valid) return HDB_OK;
    const int W = (ix->d + 31) / 32;
    if (W > 512) return fail(HDB_ERR_UNSUPPORTED, "hamming: d > 16384 not supported");
    const size_t elem = ix->dtype == HDB_F16 ? 2 : ix->dtype == HDB_F32 ? 4 : 8;
    const int64_t npad = align_up((size_t)std::max<int64_t>(ix->n, 4), 256);      // whole 256-row blocks (hdb_bits_word)
    // rows packed before the matrix grew stay where they are: the layout is a sequence of 256-row blocks, so a bigger buffer takes
    // the old blocks as a prefix (hdb_index_extend / HyperDB.add: the next bit-metric call packs the appended rows only)
    int64_t keep = (ix->bits && ix->W == W) ? std::min(ix->bits_done, ix->n) : 0;
    if (!ix->bits || ix->bits_npad < npad || ix->W != W) {
        const int64_t cap = align_up((size_t)(npad + npad / 4), 256);
        uint32_t* b2 = nullptr;
        HIP_TRY(hipMalloc((void**)&b2, (size_t)cap * W * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(b2, 0, (size_t)cap * W * sizeof(uint32_t), st));
        if (keep > 0) HIP_TRY(hipMemcpyAsync(b2, ix->bits, (size_t)align_up((size_t)keep, 256) * W * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        if (ix->bits) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ix->bits)); }
        ix->bits = b2; ix->bits_npad = cap; ix->W = W;
    } else if (keep == 0) {
        HIP_TRY(hipMemsetAsync(ix->bits, 0, (size_t)ix->bits_npad * W * sizeof(uint32_t), st));
    }
    if (ix->n > keep)
        LAUNCH_TRY(hdb_launch_signpack((const char*)ix->V + (size_t)keep * ix->d * elem, ix->n - keep, ix->d, ix->dtype, keep, ix->bits, st));
    ix->bits_done = ix->n;
    ix->bits_valid = true;
    return HDB_OK;
}

static void base_args(const hdb_index* ix, ScanArgs& a, const void* Q, int metric) {
    memset(&a, 0, sizeof(a));
    a.V = ix->V; a.n = ix->n; a.d = ix->d; a.Q = Q; a.metric = metric;
    a.inv_norm = ix->inv_norm; a.mask = ix->mask;
    a.tile_stride = 1; a.ntiles = (ix->n + 15) / 16;
    a.cap = HDB_CAND_CAP;
    a.dyn_min_bytes = ix->dyn_min_mb << 20; a.dyn_heavy = (int32_t)ix->dyn_heavy;
}

// One scan launch (VALU, hamming or MFMA flavour) for queries [a.q0, a.q0+cq).
struct QueryBufs { const float* qinv; const float* qsq; const uint32_t* qbits; const void* q16; const float* qscl; };
static int run_scan(hdb_index* ix, ScanArgs& a, int mode, int cq, const QueryBufs& qb, bool mfma, hipStream_t st) {
    a.qinv = qb.qinv;
    if (is_bits_metric(a.metric)) {
        LAUNCH_TRY(hdb_launch_hamming(&a, mode, cq, ix->bits, ix->bits_npad, ix->W, qb.qbits, st));
    } else if (a.metric == HDB_MANHATTAN && ix->use_l1_tile && cq >= 2 && a.tile_stride == 1 && !a.mask && !a.raw && a.n > HDB_CAND_CAP &&
               hdb_l1_tile_supported(ix->dtype, ix->d)) {
        // dense manhattan passes: tiles staged once in LDS, queries in registers, 8-16 queries per pass (hdb_l1_tile.hip)
        // (the tile kernel has no use for ScanArgs::dyn_heavy: 77 there = "keep the float32 arithmetic", set_option l1_packed 0)
        ScanArgs al = a; al.dyn_heavy = ix->l1_packed ? 0 : 77;
        LAUNCH_TRY(hdb_launch_l1_tile(&al, ix->dtype, mode, cq, (int)ix->max_blocks, st));
    } else if (mfma) {
        LAUNCH_TRY(hdb_launch_mfma_scan(&a, ix->dtype, mode, cq, qb.q16, ix->sqnorm, qb.qsq, qb.qscl, (int)ix->max_blocks, (int)ix->mfma_variant, st, nullptr));
    } else {
        LAUNCH_TRY(hdb_launch_scan(&a, ix->dtype, mode, cq, (int)ix->max_blocks, st));
    }
    return HDB_OK;
}

extern "C" int hdb_scores(hdb_index* ix, const void* dev_q, int metric, float* dev_out, void* stream) {
    if (!ix || !dev_q || !dev_out) return fail(HDB_ERR_ARG, "hdb_scores: null argument");
    if (!metric_ok(metric)) return fail(HDB_ERR_UNSUPPORTED, "hdb_scores: metric not built");
    if (ix->n == 0) return HDB_OK;
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const int W = (ix->d + 31) / 32;
    int rc = ensure_ws(ix, 8192 + (size_t)W * 4 + (size_t)ix->d * 8);
    if (rc) return rc;
    Bump b(ix->ws, ix->ws_bytes);
    float* qinv = b.take<float>(1); float* qsq = b.take<float>(1); int* qnan = b.take<int>(1);
    uint32_t* qbits = b.take<uint32_t>(W);
    void* qc = b.take<double>(ix->d);
    LAUNCH_TRY(hdb_launch_qprep(dev_q, 1, ix->d, ix->dtype == HDB_F64, qinv, qsq, qnan, nullptr, nullptr, st));
    if (is_bits_metric(metric)) {
        rc = ensure_bits(ix, st); if (rc) return rc;
        LAUNCH_TRY(hdb_launch_qsign(dev_q, 1, ix->d, ix->dtype == HDB_F64, W, qbits, st));
    }
    ScanArgs a; base_args(ix, a, dev_q, metric);
    if (metric == HDB_PEARSON) {        // cosine pipeline on the centred query with 1/(sd*d) row scales
        rc = ensure_pscale(ix, st); if (rc) return rc;
        LAUNCH_TRY(hdb_launch_qcentre(dev_q, 1, ix->d, ix->dtype == HDB_F64, qc, qinv, st));
        a.Q = qc; a.metric = HDB_COSINE; a.inv_norm = ix->pscale;
    }
    a.mask = nullptr;                   // per-metric functions score every row, no bias (reference :24-147)
    a.raw = 1;                          // ... and return NaN where the reference does (pearson, jaccard)
    a.scores = dev_out; a.ld = ix->n;
    QueryBufs qb{qinv, qsq, qbits, nullptr, nullptr};
    return run_scan(ix, a, 0, 1, qb, false, st);
}

// Row sample for the threshold estimate: `tiles` tiles of `tile_rows` rows, evenly strided over V.
// The m-th largest of the sampled scores is exceeded by about T rows of the full matrix (Gamma(m)
// spread), T >= 8k..16k and <= CAP/2, so both "fewer than k pass" and "more than CAP pass" are
// < 1e-9 events for exchangeable row orders; either one only costs the exact-path re-run.
// coarse: the scores take few distinct values (bit metrics), so the rows at the threshold's own level all survive; aim lower.
// Batches of 32+ queries aim at 1024 survivors per query instead of 2048: every survivor costs the filter's slow path
// (d=384, 64 queries: 1.28 -> 1.20 ms per call; 256 queries: -1 %), the sample doubles to 0.8 % of the rows, and
// P(fewer than k=100 pass) = P(Gamma(8) < 0.78) = 1.7e-6 per query, paid with one exact re-run of that query.
static void sample_plan(const hdb_index* ix, uint32_t kk, int nq, int tile_rows, bool coarse, int64_t& tiles, int64_t& stride, uint32_t& m) {
    int64_t T = ix->sample_target > 0 ? ix->sample_target : (kk <= 128 ? (nq >= 32 ? 1024 : 2048) : 4096);
    if (coarse && ix->sample_target <= 0) T /= 2;
    m = kk <= 128 ? 8u : (kk <= 512 ? 64u : 256u);
    int64_t rows = (int64_t)((double)m * (double)ix->n / (double)T);
    rows = std::max<int64_t>(rows, 16 * (int64_t)m);         // at least 16 m sample rows
    tiles = (rows + tile_rows - 1) / tile_rows;
    const int64_t all_tiles = ix->n / tile_rows;             // full tiles only: sample rows always exist
    tiles = std::min(tiles, all_tiles);
    stride = std::max<int64_t>(1, all_tiles / std::max<int64_t>(tiles, 1));
}

static int topk_impl(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric, int64_t* dev_idx,
                     float* dev_score, int32_t* dev_status, void* stream, bool exact) {
    if (!ix || !dev_idx || !dev_score) return fail(HDB_ERR_ARG, "hdb_topk: null argument");
    if (nq < 0 || k < 0) return fail(HDB_ERR_ARG, "hdb_topk: nq and k must be >= 0");
    if (nq == 0 || k == 0) return HDB_OK;
    if (!dev_Q) return fail(HDB_ERR_ARG, "hdb_topk: query pointer is null");
    const bool full_sort = k > HDB_MAX_K && ix->n > HDB_CAND_CAP;
    if (metric == HDB_EUCLIDEAN_DIST || !metric_ok(metric)) return fail(HDB_ERR_UNSUPPORTED, "hdb_topk: metric not built");
    if (metric == HDB_PEARSON && ix->d < 1) return fail(HDB_ERR_ARG, "hdb_topk: pearson needs d >= 1");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const bool f64 = ix->dtype == HDB_F64;
    const int64_t n = ix->n;
    const uint32_t kk = (uint32_t)std::min<int64_t>(k, n);
    const int W = (ix->d + 31) / 32;
    if (n == 0) {   // nothing stored: all -1 / -inf
        HIP_TRY(hipMemsetAsync(dev_idx, 0xFF, (size_t)nq * k * sizeof(int64_t), st));
        HIP_TRY(hipMemsetAsync(dev_score, 0xFF, (size_t)nq * k * sizeof(float), st));   // NaN pattern; no rows exist
        if (dev_status) HIP_TRY(hipMemsetAsync(dev_status, 0, (size_t)nq * sizeof(int32_t), st));
        return HDB_OK;
    }
    const bool small = n <= HDB_CAND_CAP;
    const bool is_ham = is_bits_metric(metric);
    const bool is_pearson = metric == HDB_PEARSON;
    // bit metrics tie massively by construction; the sampled threshold still works while the rows at and above its
    // level fit the candidate list (random data: yes), and the status word sends the rest through the exact path
    const bool exact_req = exact;                            // the caller asked for the exact selection (tests; the re-run of a failed call)
    if (is_ham && !small && !ix->bits_fused) exact = true;
    if (ix->force_exact && !small) exact = true;
    if (!small && (int64_t)kk * 32 > n) exact = true;        // k is a large share of the rows: a sampled threshold cannot help
    // fp32 matrices: the VALU scan serves up to 4 queries in one pass at HBM speed; the fp32 MFMA scan (matrix-pipe
    // bound at 157 TFLOP/s) takes over where a second VALU pass would start
    // (rows that need K slices -- float32 d >= 1024, fp16 d >= 2048 -- likewise: up to 4 queries are one VALU pass at HBM speed, the
    // slices pay a second launch and the partial sums)
    // float32: up to 4 queries are one VALU pass at HBM speed and the float32 matrix pipe binds early -- but three or four queries on
    // rows of up to 384 elements are faster through the batched single launch from ~300k rows on (n = 2M x 384: 595 / 640 -> 510 us;
    // d = 768: 1 030 vs 1 714, the VALU pass stays)
    const int64_t f32_min_q = ix->f32_min_q >= 0 ? ix->f32_min_q : ((ix->d <= 384 && n >= 300000) ? 3 : 5);
    // (widths without a geometry of their own ride the next wider one through the multi-kernel pipeline: like the K slices, from five queries on)
    const int64_t min_q = (hdb_mfma_ksplit_slices(ix->dtype, ix->d) > 0 || hdb_mfma_anyd_pad(ix->dtype, ix->d) > 0) ? std::max<int64_t>(ix->mfma_min_q, 5)
                        : ix->dtype == HDB_F32 ? std::max<int64_t>(ix->mfma_min_q, f32_min_q) : ix->mfma_min_q;
    // hdb_mfma_fused_kernel is built around ONE multiplying wave and two selector waves: with 2-4 fp16 queries its sample phase and
    // epilogue cost more than the batched single launch (eight multiplying waves) until the pass itself dominates -- n = 100k x 384,
    // three queries: 124 vs 57 us; 500k: 128 vs 97; 1M: 162 vs 149; 2M: 267 vs 268; 5M: 591 vs 608 (four queries never win).
    // float32 (VALU flavour, two queries): the single launch wins at every size.
    const int64_t fused_max_q = ix->fused_max_q >= 0 ? ix->fused_max_q
                              : ix->dtype == HDB_F32 ? HDB_FUSED_MAXQ_RULE : (n >= 1500000 ? 3 : 1);
    const bool mfma = ix->use_mfma && !is_ham && !small && nq >= min_q &&
                      hdb_mfma_supported(ix->dtype, ix->d, is_pearson ? (int)HDB_COSINE : metric);
    // 1-4 dot / cosine queries, k <= 128: one launch does everything (hdb_mfma_fused.h; fp16 on the matrix cores,
    // float32 in the VALU from the same staged tiles)
    // Short matrices: the single launch in its LOCAL flavour -- no row sample, no exchange; every workgroup parks the scores of all
    // its tiles and emits the rows at or above its own local_m-th best (hdb_mfma_fused.h).  Possible while a workgroup's tiles fit
    // its parking area (up to 16); used, by measurement (profiles/r4_latency_map.txt, same box, interleaved), up to local_max_tiles =
    // 4 tiles per workgroup (fp16 d = 384: 65 536 rows): 35 vs 37 us at 20k rows, 45 vs 45 at 100k, 62 vs 59 at 250k -- beyond that
    // the exchange flavour filters while it streams and the local one selects after its last tile.  Matrices of up to 8192 rows keep
    // the three launches (thr = -inf, scan, finalize): 26 us at 1000 rows against 31 for this kernel's launch ramp and last workgroup.
    const int fl_rows = hdb_mfma_tile_rows(ix->dtype, ix->d);
    const int64_t fl_tiles = fl_rows > 0 ? (n + fl_rows - 1) / fl_rows : 0;
    int64_t fl_grid = std::min<int64_t>(fl_tiles, hdb_cu_count());
    if (ix->max_blocks > 0) fl_grid = std::min<int64_t>(fl_grid, ix->max_blocks);
    const int fl_cap = fl_rows > 0 && nq >= 1 && nq <= HDB_FUSED_MAXQ_RULE && nq <= ix->local_max_q ? hdb_mfma_fused_local_tiles(ix->dtype, ix->d, metric, nq) : 0;
    const bool local_ok = fl_grid * 32 <= HDB_CAND_CAP && ix->use_fused && ix->use_local && !ix->force_exact && !exact_req && (ix->dtype == HDB_F32 || (ix->use_mfma && nq >= ix->mfma_min_q)) && fl_grid > 0 && fl_cap > 0 && (fl_tiles + fl_grid - 1) / fl_grid <= std::min<int64_t>(fl_cap, ix->local_max_tiles) && (!small || ix->local_small) &&
                          !is_ham && kk <= 128 && dev_status != nullptr && hdb_mfma_fused_supported(ix->dtype, ix->d, metric, nq, kk);
    if (local_ok) exact = false;                       // (k a large share of the rows: every workgroup then emits all its rows)
    const bool fused_shape = ix->use_fused && !exact && (!small || local_ok) && k <= HDB_MAX_K && dev_status != nullptr && !is_ham &&
                             hdb_mfma_fused_supported(ix->dtype, ix->d, metric, nq, kk) && (ix->dtype == HDB_F32 || mfma || local_ok) && (nq <= fused_max_q || local_ok) &&
                             // float32 d = 512 streams 32-KiB tiles (16 rows): below ~3 GB the five-kernel VALU pipeline is
                             // 2-5 % faster end to end (200 vs 210 us at 0.5 M rows, 376 vs 385 at 1 M; 728 vs 687 at 2 M)
                             !(ix->dtype == HDB_F32 && ix->d == 512 && n < 1500000) &&
                             // fp16 d = 1024 (32-KiB tiles, 32 k-steps in one wave): 189 vs 195 us at 0.5 M rows, 619 vs 627 at 2 M,
                             // but 1 491 vs 1 466 at 5 M -- the single launch up to 4 M rows
                             !(ix->dtype == HDB_F16 && ix->d == 1024 && n > 4000000);
    const int tile_rows = (mfma || fused_shape) ? hdb_mfma_tile_rows(ix->dtype, ix->d) : 16;
    // float32 rows on the matrix cores multiply in three bf16 parts (hdb_mfma_f32s.hip: 2.7x the rate of the float32 MFMAs, same
    // 1e-5 contract) -- on finite matrices: the parts of an infinite element would cancel to NaN where np.dot keeps the infinity
    bool f32s = false;
    const int f32s_auto = hdb_mfma_f32_split_min_q(ix->d);
    const int f32s_min = f32s_auto > 0 ? (int)(ix->f32_split_min_q > 0 ? ix->f32_split_min_q : f32s_auto) : 0;       // queries of the CALL (every launch of a call multiplies the same way)
    if (mfma && ix->dtype == HDB_F32 && ix->f32_split && f32s_min > 0 && nq >= f32s_min && nq <= hdb_mfma_f32_split_max_q(ix->d)) { const int rc = matrix_is_finite(ix, &f32s); if (rc != HDB_OK) return rc; }
    ix->st_f32s = f32s ? 1 : 0;
    // anything else the matrix-core scan takes (5-256 dot / cosine queries, 1-256 euclidean ones), k <= 128: one launch per
    // <= bcap queries does preparation, sample, thresholds, the pass and every query's final sort (hdb_mfma_kernel.h, MODE 2)
    const int bcap = mfma ? hdb_mfma_batch_capacity(ix->dtype, ix->d) : 0;
    const bool batch1 = ix->use_fused && ix->use_batch1 && mfma && !fused_shape && !exact && !small && !full_sort && kk <= 128 &&
                        dev_status != nullptr && bcap > 0;

    // ---- plan the chunking --------------------------------------------------------------------
    int64_t s_tiles = 0, s_stride = 1; uint32_t m = 0;
    if (!small && !exact) sample_plan(ix, kk, nq, tile_rows, is_ham, s_tiles, s_stride, m);
    const int64_t s_rows = s_tiles * tile_rows;
    const int64_t ld_s = align_up((size_t)std::max<int64_t>(s_rows, 4), 4);
    const int64_t ld_n = align_up((size_t)n, 4);
    int cq_max = batch1 ? bcap : 256;
    if (exact && !small) cq_max = (int)std::max<int64_t>(1, std::min<int64_t>(256, ix->exact_bytes / (ld_n * 4)));
    // wide rows on the matrix cores go through K slices with a [query][rows] buffer of partial sums (hdb_mfma_ksplit.hip)
    const bool ksplit = mfma && hdb_mfma_ksplit_slices(ix->dtype, ix->d) > 0;
    if (ksplit && !small) cq_max = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(cq_max, 128), ix->exact_bytes / (ld_n * 4)));
    cq_max = std::min(cq_max, (int)nq);

    size_t need = 0;
    need += 4 * align_up((size_t)nq * 4, 256) + 1024;                        // qinv, qsq, qnan, qscl
    need += align_up((size_t)nq * W * 4, 256);                               // qbits
    need += align_up((size_t)nq * ix->d * 2, 256);                           // fp16 queries (MFMA)
    need += align_up((size_t)nq * ix->d * 8, 256);                           // centred queries (pearson)
    need += align_up((size_t)cq_max * 4, 256) + align_up((size_t)cq_max * 4 * HDB_CNT_STRIDE, 256) + 256;     // thr, cnt (a cache line per query), tile counter
    need += align_up((size_t)cq_max * 4 * HDB_RADIX_BINS * 4, 256);          // hist
    need += align_up((size_t)cq_max * 16, 256);                              // tie_info
    need += align_up((size_t)cq_max * HDB_CAND_CAP * 8, 256);                // cand
    need += align_up((size_t)cq_max * (exact && !small ? ld_n : ld_s) * 4, 256) + 4096;
    if (ksplit) need += align_up((size_t)cq_max * ld_n * 4, 256);
    int rc = ensure_ws(ix, need);
    if (rc) return rc;
    Bump b(ix->ws, ix->ws_bytes);
    float* qinv = b.take<float>(nq); float* qsq = b.take<float>(nq); int* qnan = b.take<int>(nq); float* qscl = b.take<float>(nq);
    uint32_t* qbits = b.take<uint32_t>((size_t)nq * W);
    void* q16 = b.take<uint16_t>((size_t)nq * ix->d);
    void* qc = b.take<double>((size_t)nq * ix->d);
    float* thr = b.take<float>(cq_max); uint32_t* cnt = b.take<uint32_t>((size_t)cq_max * HDB_CNT_STRIDE);
    uint32_t* tile_ctr = b.take<uint32_t>(64);
    uint32_t* hist = b.take<uint32_t>((size_t)cq_max * 4 * HDB_RADIX_BINS);
    uint32_t* tie_info = b.take<uint32_t>((size_t)cq_max * 4);
    unsigned long long* cand = b.take<unsigned long long>((size_t)cq_max * HDB_CAND_CAP);
    float* sbuf = b.take<float>((size_t)cq_max * (exact && !small ? ld_n : ld_s));
    float* kbuf = ksplit ? b.take<float>((size_t)cq_max * ld_n) : nullptr;

    const bool fused = fused_shape && !full_sort && (m == 8 || local_ok);           // (no prep kernel either)
    // the MFMA scan multiplies with fp16 queries: written by the same kernel (pearson converts its centred copy later)
    const bool f16_queries = mfma && ix->dtype == HDB_F16;          // fp32 matrices multiply with the float32 queries as they are
    const bool q16_in_prep = f16_queries && !is_pearson && !full_sort;
    // hamming / jaccard: the single launch for every call (with the two-level hand-out of the pass: one query 120-124 vs 125-132 us
    // for the six launches at N=10M, 78 vs 77 at 5M, 60 vs 63 at 1.25M, 44 vs 52 at 250k rows; four queries 135 vs 172 at N=10M --
    // profiles/r3_bits_variants.txt; bits_fused = 3 keeps one-query calls on 1M+ rows with the six launches, for comparison)
    const bool bits1_pre = ix->use_fused && ix->bits_fused && is_ham && !exact && !small && !full_sort && !f64 && dev_status != nullptr &&
                           hdb_bits_fused_supported(metric, 1, W, kk) && (nq >= 2 || n < 1000000 || ix->bits_fused != 3) &&
                           // (more than four queries: the six launches take them all in one go, grid.y = query groups -- 16 queries on
                           // 100k rows 50 vs 148 us for four single launches in a row, 64 queries 67 vs 642; 10M rows 464 vs 524)
                           nq <= (ix->bits_max_q >= 0 ? ix->bits_max_q : 4);
                           // (the single launch prepares its queries itself)
    // matrices of up to 8192 rows, one chunk of queries: the prep kernel also sets thr = -inf / an empty list and packs the query sign
    // bits -- four (five) launches become three; the reference's own sizes live here (151 .. 10 000 documents)
    const bool fold_small = small && !fused && !batch1 && !bits1_pre && !full_sort && nq <= cq_max;
    if (!fused && !batch1 && !bits1_pre)
        LAUNCH_TRY(hdb_launch_qprep2(dev_Q, nq, ix->d, f64, qinv, qsq, qnan, q16_in_prep ? q16 : nullptr, qscl, fold_small ? thr : nullptr,
                                     fold_small ? cnt : nullptr, (fold_small && is_ham) ? qbits : nullptr, W, st));
    if (is_ham) {
        rc = ensure_bits(ix, st); if (rc) return rc;
        if (!bits1_pre && !fold_small) LAUNCH_TRY(hdb_launch_qsign(dev_Q, nq, ix->d, f64, W, qbits, st));
    }
    const void* Qeff = dev_Q;
    int metric_eff = metric;
    if (is_pearson) {
        rc = ensure_pscale(ix, st); if (rc) return rc;
        if (!fused && !batch1) LAUNCH_TRY(hdb_launch_qcentre(dev_Q, nq, ix->d, f64, qc, qinv, st));     // qinv <- 1/sd_q (the single launches centre their queries themselves)
        Qeff = qc; metric_eff = HDB_COSINE;
    }
    if (full_sort) {
        // cold path for huge k: one query at a time, all scores -> stable radix sort (hdb_sort.hip)
        size_t tb = 0;
        LAUNCH_TRY(hdb_sort_temp_bytes(n, &tb));
        const size_t extra = align_up((size_t)n * 4, 256) + align_up((size_t)n * 16, 256) + align_up(tb, 256) + 4096;
        // the workspace was sized before this branch was known: grow it now (all pointers above are re-derived)
        const size_t base_need = need;
        rc = ensure_ws(ix, base_need + extra); if (rc) return rc;
        Bump b2(ix->ws, ix->ws_bytes);
        float* qinv2 = b2.take<float>(nq); float* qsq2 = b2.take<float>(nq); int* qnan2 = b2.take<int>(nq); (void)b2.take<float>(nq);
        uint32_t* qbits2 = b2.take<uint32_t>((size_t)nq * W);
        (void)b2.take<uint16_t>((size_t)nq * ix->d);
        void* qc2 = b2.take<double>((size_t)nq * ix->d);
        float* sc1 = b2.take<float>((size_t)ld_n);
        uint32_t* work = b2.take<uint32_t>((size_t)n * 4);
        void* temp = b2.take<char>(tb);
        LAUNCH_TRY(hdb_launch_qprep(dev_Q, nq, ix->d, f64, qinv2, qsq2, qnan2, nullptr, nullptr, st));
        if (is_ham) LAUNCH_TRY(hdb_launch_qsign(dev_Q, nq, ix->d, f64, W, qbits2, st));
        const void* Q2 = dev_Q;
        if (is_pearson) { LAUNCH_TRY(hdb_launch_qcentre(dev_Q, nq, ix->d, f64, qc2, qinv2, st)); Q2 = qc2; }
        ix->st_path = 3; ix->st_mfma = 0; ix->st_chunks = nq;
        for (int q0 = 0; q0 < nq; ++q0) {
            QueryBufs qb{qinv2, qsq2, qbits2, nullptr, nullptr};
            ScanArgs s2; base_args(ix, s2, Q2, metric_eff);
            if (is_pearson) s2.inv_norm = ix->pscale;
            s2.q0 = q0; s2.bias = ix->bias; s2.scores = sc1; s2.ld = ld_n;
            rc = run_scan(ix, s2, 0, 1, qb, false, st); if (rc) return rc;
            LAUNCH_TRY(hdb_launch_full_sort(sc1, n, k, ix->row_base, work, temp, tb, dev_idx + (int64_t)q0 * k,
                                            dev_score + (int64_t)q0 * k, st));
        }
        if (dev_status) LAUNCH_TRY(hdb_launch_status_nan(qnan2, nq, dev_status, st));     // HDB_Q_NAN survives on this path too
        return HDB_OK;
    }
    // the MFMA scan has no mask input: excluded rows get a bias of -inf instead (never appended, like the VALU scan)
    const float* bias_eff = ix->bias;
    const uint8_t* mask_eff = ix->mask;
    // (run_scan hands manhattan calls of two or more queries to the tile kernel; a single query keeps the VALU scan and its mask input)
    const bool l1tile = metric == HDB_MANHATTAN && ix->use_l1_tile && !small && nq >= 2 && hdb_l1_tile_supported(ix->dtype, ix->d);
    if ((mfma || fused || l1tile) && ix->mask) {
        if (n > ix->mbias_rows) {
            if (ix->mbias) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ix->mbias)); ix->mbias = nullptr; }
            const int64_t rows = n + n / 4 + 64;
            HIP_TRY(hipMalloc((void**)&ix->mbias, rows * sizeof(float)));
            ix->mbias_rows = rows;
        }
        LAUNCH_TRY(hdb_launch_maskbias(ix->mask, ix->bias, n, ix->mbias, st));
        bias_eff = ix->mbias; mask_eff = nullptr;
    }
    ix->st_fused = 0; ix->st_local = 0;
    if (fused) {
        // ---- the whole call in ONE launch (hdb_mfma_fused.h): prep + sample + threshold + filter pass + finalize ----
        if (!ix->fctl) {
            const size_t cb = hdb_mfma_fused_ctl_bytes();
            HIP_TRY(hipMalloc((void**)&ix->fctl, cb));
            HIP_TRY(hipMemset(ix->fctl, 0, cb));
        }
        ScanArgs a; base_args(ix, a, dev_Q, metric);
        a.bias = bias_eff; a.mask = nullptr;
        if (metric == HDB_EUCLIDEAN) a.inv_norm = ix->sqnorm;          // the per-row aux value of the euclidean expansion
        if (is_pearson) a.inv_norm = ix->pscale;                       // 1/(sd_v d); the kernel centres the queries itself
        a.ntiles = (n + tile_rows - 1) / tile_rows;
        a.thr = thr; a.cnt = cnt; a.cand = cand; a.nq = nq;
        FusedArgs fa; memset(&fa, 0, sizeof(fa));
        fa.Qraw = static_cast<const float*>(dev_Q); fa.nq = nq;
        fa.s_tiles = s_tiles; fa.s_stride = s_stride;
        ix->fused_epoch = (ix->fused_epoch + 1) & 0x7FFFFFFFu;          // 31 bits: bit 31 of a tag is the "final" flag of the threshold words
        if (ix->fused_epoch == 0) ix->fused_epoch = 1;
        fa.epoch = ix->fused_epoch;
        fa.timeout_ticks = (uint32_t)std::min<int64_t>(ix->fused_timeout_us * 100, 0x7FFFFFFF);
        fa.ctl = reinterpret_cast<uint32_t*>(ix->fctl);
        fa.cand = cand; fa.cap = HDB_CAND_CAP; fa.k = (uint32_t)k; fa.kk = kk; fa.row_base = ix->row_base;
        fa.idx_out = dev_idx; fa.score_out = dev_score; fa.status = dev_status; fa.thr_out = thr;
        if (local_ok) {
            fa.local = 1;
            // rows every workgroup emits at least: ~3072 candidates in all (8 .. 32 per workgroup); a grid too small to hold 4 k rows
            // that way emits everything (64 = every lane maximum of a tile)
            // slots of a workgroup in the (slotted) lists: 64 while the grid leaves room for them, else 32; never fewer than twice local_m
            fa.local_slot = fl_grid * 64 <= HDB_CAND_CAP ? 64u : 32u;
            fa.local_m = ix->local_m > 0 ? (uint32_t)ix->local_m
                       : fl_grid * 32 >= 4 * (int64_t)kk ? (uint32_t)std::min<int64_t>(fa.local_slot / 2, std::max<int64_t>(8, (3072 + fl_grid - 1) / fl_grid)) : 64u;
        }
        ix->st_local = local_ok ? 1 : 0;
        ix->st_sample_rows = s_rows; ix->st_sample_m = m; ix->st_chunks = 1; ix->st_path = 1; ix->st_mfma = ix->dtype == HDB_F16 ? 1 : 0; ix->st_fused = 1;
        prof_begin(ix, st);
        ix->ht_l0 = std::chrono::steady_clock::now();
        LAUNCH_TRY(hdb_launch_mfma_fused(&a, ix->dtype, &fa, (int)ix->max_blocks, st));
        ix->ht_l1 = std::chrono::steady_clock::now();
        prof_end(ix, st);
        return HDB_OK;
    }
    // 1-4 hamming / jaccard queries per launch: prep, sample, threshold, the pass over the sign bits and the final sort in ONE
    // kernel (hdb_bits_fused.hip); larger batches go through it four queries at a time (as the multi-kernel scan re-reads the bits)
    const bool bits1 = bits1_pre;
    if (bits1) {
        if (!ix->bctl) {
            const size_t cb = hdb_mfma_batch_ctl_bytes(hdb_cu_count());
            HIP_TRY(hipMalloc((void**)&ix->bctl, cb));
            HIP_TRY(hipMemset(ix->bctl, 0, cb));
        }
        ix->st_sample_rows = s_rows; ix->st_sample_m = m; ix->st_chunks = 0; ix->st_path = 1; ix->st_mfma = 0; ix->st_fused = 3;
        {   // (the launcher's rule: the local flavour from 2 k workgroups on, hdb_bits_fused.hip)
            int64_t bl = hdb_cu_count();
            const int64_t items = ((n + 15) / 16) * 4;
            if (bl * 1024 > items) bl = (items + 1023) / 1024;
            if (ix->max_blocks > 0 && ix->max_blocks < bl) bl = ix->max_blocks;
            ix->st_local = (ix->bits_local && bl >= 2 * (int64_t)kk) ? 1 : 0;
        }
        for (int q0 = 0; q0 < nq; q0 += 4) {
            const int cq = std::min(4, nq - q0);
            ix->st_chunks++;
            BitsArgs ba; memset(&ba, 0, sizeof(ba));
            ba.bits = ix->bits; ba.npad = ix->bits_npad; ba.W = W; ba.n = n; ba.d = ix->d;
            ba.Qraw = static_cast<const float*>(dev_Q) + (size_t)q0 * ix->d; ba.nq = cq;
            ba.ntiles = (n + 15) / 16; ba.s_tiles = s_tiles; ba.s_stride = s_stride;
            ba.bias = ix->bias; ba.mask = ix->mask;
            ba.local = ix->bits_local ? 1 : 0;
            ix->fused_epoch = (ix->fused_epoch + 1) & 0x7FFFFFFFu;
            if (ix->fused_epoch == 0) ix->fused_epoch = 1;
            ba.epoch = ix->fused_epoch;
            ba.timeout_ticks = (uint32_t)std::min<int64_t>(ix->fused_timeout_us * 100, 0x7FFFFFFF);
            ba.ctl = reinterpret_cast<uint32_t*>(ix->bctl);
            ba.cand = cand; ba.cap = HDB_CAND_CAP; ba.k = (uint32_t)k; ba.kk = kk; ba.row_base = ix->row_base;
            ba.idx_out = dev_idx + (int64_t)q0 * k; ba.score_out = dev_score + (int64_t)q0 * k; ba.status = dev_status + q0;
            prof_begin(ix, st);
            ix->ht_l0 = std::chrono::steady_clock::now();
            LAUNCH_TRY(hdb_launch_bits_fused(&ba, metric == HDB_JACCARD ? 1 : 0, (int)ix->max_blocks, st));
            ix->ht_l1 = std::chrono::steady_clock::now();
            prof_end(ix, st);
        }
        return HDB_OK;
    }
    if (batch1) {
        const int cus = hdb_cu_count();
        if (!ix->bctl) {
            const size_t cb = hdb_mfma_batch_ctl_bytes(cus);
            HIP_TRY(hipMalloc((void**)&ix->bctl, cb));
            HIP_TRY(hipMemset(ix->bctl, 0, cb));
        }
        ix->st_sample_rows = s_rows; ix->st_sample_m = m; ix->st_chunks = 0; ix->st_path = 1; ix->st_mfma = 1; ix->st_fused = 2;
        const size_t qrow = (size_t)ix->d * 4;
        for (int q0 = 0; q0 < nq; q0 += cq_max) {
            const int cq = std::min(cq_max, nq - q0);
            ix->st_chunks++;
            ScanArgs a; base_args(ix, a, dev_Q, is_pearson ? (int)HDB_COSINE : metric);      // pearson: the cosine launch on queries the kernel centres, ...
            if (is_pearson) a.inv_norm = ix->pscale;                                          // ... row scale 1/(sd_v d)
            a.bias = bias_eff; a.mask = nullptr; a.q0 = 0; a.nq = cq; a.f32_split = f32s ? 1 : 0;
            a.ntiles = (n + tile_rows - 1) / tile_rows;
            a.cand = cand;
            a.tile_ctr = ix->dyn_tiles ? reinterpret_cast<uint32_t*>(ix->bctl) + HDB_BATCH_CTL_TILE : nullptr;
            BatchArgs fa; memset(&fa, 0, sizeof(fa));
            fa.Qraw = static_cast<const char*>(dev_Q) + (size_t)q0 * qrow;
            fa.s_tiles = s_tiles; fa.s_stride = s_stride; fa.centre = is_pearson ? 1 : 0;
            ix->fused_epoch = (ix->fused_epoch + 1) & 0x7FFFFFFFu;
            if (ix->fused_epoch == 0) ix->fused_epoch = 1;
            fa.epoch = ix->fused_epoch;
            fa.timeout_ticks = (uint32_t)std::min<int64_t>(ix->fused_timeout_us * 100, 0x7FFFFFFF);
            fa.ctl = reinterpret_cast<uint32_t*>(ix->bctl);
            fa.k = (uint32_t)k; fa.kk = kk; fa.row_base = ix->row_base;
            fa.idx_out = dev_idx + (int64_t)q0 * k; fa.score_out = dev_score + (int64_t)q0 * k; fa.status = dev_status + q0;
            prof_begin(ix, st);
            ix->ht_l0 = std::chrono::steady_clock::now();
            LAUNCH_TRY(hdb_launch_mfma_scan(&a, ix->dtype, 2, cq, nullptr, ix->sqnorm, nullptr, nullptr, (int)ix->max_blocks, (int)ix->mfma_variant, st, &fa));
            ix->ht_l1 = std::chrono::steady_clock::now();
            prof_end(ix, st);
        }
        return HDB_OK;
    }
    bool q16_ready = q16_in_prep;
    ix->st_sample_rows = s_rows; ix->st_sample_m = m; ix->st_chunks = 0;
    ix->st_path = small ? 0 : (exact ? 2 : 1);
    ix->st_mfma = mfma ? 1 : 0;

    for (int q0 = 0; q0 < nq; q0 += cq_max) {
        const int cq = std::min(cq_max, nq - q0);
        ix->st_chunks++;
        if (f16_queries && !q16_ready) { LAUNCH_TRY(hdb_launch_q_to_f16((const float*)Qeff, nq, ix->d, q16, qscl, st)); q16_ready = true; }
        QueryBufs qb{qinv, qsq, qbits, f16_queries ? q16 : Qeff, f16_queries ? qscl : nullptr};
        ScanArgs a; base_args(ix, a, Qeff, metric_eff);
        if (is_pearson) a.inv_norm = ix->pscale;
        a.q0 = q0; a.bias = bias_eff; a.mask = mask_eff; a.f32_split = f32s ? 1 : 0;
        a.thr = thr; a.cnt = cnt; a.cand = cand;
        a.ntiles = (n + tile_rows - 1) / tile_rows;
        a.ks_partial_out = kbuf; a.ks_ld = ld_n;         // (K slices only; sample and exact passes index it by their own tile sequence)

        if (small) {
            if (!fold_small) LAUNCH_TRY(hdb_launch_fill_thr(thr, cnt, cq, -INFINITY, st));
            rc = run_scan(ix, a, 1, cq, qb, mfma, st); if (rc) return rc;
        } else if (!exact) {
            // 1) strided row sample -> sample scores
            ScanArgs s = a;
            s.ntiles = s_tiles; s.tile_stride = s_stride; s.scores = sbuf; s.ld = ld_s;
            rc = run_scan(ix, s, 0, cq, qb, mfma, st); if (rc) return rc;
            // 2) m-th largest sample score per query
            if (m <= 16) {
                LAUNCH_TRY(hdb_launch_sample_thr(sbuf, s_rows, ld_s, cq, m, thr, cnt, tile_ctr, st));
                if (mfma && ix->dyn_tiles) a.tile_ctr = tile_ctr;       // zeroed just now: dynamic tile hand-out in the pass
            } else {
                HIP_TRY(hipMemsetAsync(hist, 0, (size_t)cq * 4 * HDB_RADIX_BINS * 4, st));
                for (int p = 0; p < 4; ++p) LAUNCH_TRY(hdb_launch_hist(sbuf, s_rows, ld_s, cq, hist, p, m, st));
                LAUNCH_TRY(hdb_launch_thr(hist, cq, 4, m, (uint32_t)s_rows, thr, cnt, st));
            }
            // 3) the pass over all of V
            prof_begin(ix, st);
            rc = run_scan(ix, a, 1, cq, qb, mfma, st); if (rc) return rc;
            prof_end(ix, st);
        } else {
            ScanArgs s = a;
            s.scores = sbuf; s.ld = ld_n;
            prof_begin(ix, st);
            rc = run_scan(ix, s, 0, cq, qb, mfma, st); if (rc) return rc;
            prof_end(ix, st);
            // cnt and hist are neighbours in the workspace: one memset clears both
            HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)((char*)hist - (char*)cnt) + (size_t)cq * 4 * HDB_RADIX_BINS * 4, st));
            // hamming scores are integers in [0, d]: their float keys are zero below the top 8 + bits(d) bits, so the
            // last radix pass (the last two for d < 128) would only re-read the scores to find every key in bin 0
            int npass = 4;
            if (metric == HDB_HAMMING && !ix->bias && !ix->mask) {
                int bits = 0;
                while ((ix->d >> bits) != 0) ++bits;
                npass = std::min(4, (8 + bits + 7) / 8);
            }
            for (int p = 0; p < npass; ++p) LAUNCH_TRY(hdb_launch_hist(sbuf, n, ld_n, cq, hist, p, kk, st));
            LAUNCH_TRY(hdb_launch_collect(sbuf, n, ld_n, cq, hist, npass, kk, cnt, cand, HDB_CAND_CAP, tie_info, st));
        }
        if (mfma && metric == HDB_EUCLIDEAN)     // the MFMA path scores through ||v||^2+||q||^2-2v.q: redo near-duplicates directly
            LAUNCH_TRY(hdb_launch_rescore_euclid(cand, cnt, HDB_CAND_CAP, cq, ix->V, ix->dtype, ix->d, (const float*)dev_Q, qsq, q0, ix->bias, st));
        LAUNCH_TRY(hdb_launch_finalize(cand, cnt, HDB_CAND_CAP, cq, (uint32_t)k, kk, ix->row_base,
                                       dev_idx + (int64_t)q0 * k, dev_score + (int64_t)q0 * k,
                                       dev_status ? dev_status + q0 : nullptr, qnan + q0, (int)ix->finalize_threads,
                                       a.f32_split ? HDB_Q_UNDERFLOW : 0, st));      // (parts of an infinite query element cancel to NaN: exact re-run)
    }
    return HDB_OK;
}

extern "C" int hdb_topk(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric, int64_t* dev_idx,
                        float* dev_score, int32_t* dev_status, void* stream) {
    return topk_impl(ix, dev_Q, nq, k, metric, dev_idx, dev_score, dev_status, stream, false);
}

extern "C" int64_t hdb_packed_bytes(int32_t nq, int32_t k);

// k-way merge of `parts` packed records in host memory (recs[p] = record of shard p) into out_record
static void merge_host_records(const char* const* recs, int32_t parts, int32_t nq, int32_t k, void* out_record) {
    int64_t* oi = reinterpret_cast<int64_t*>(out_record);
    float* os = reinterpret_cast<float*>(static_cast<char*>(out_record) + (int64_t)nq * k * 8);
    int32_t* ost = reinterpret_cast<int32_t*>(static_cast<char*>(out_record) + (int64_t)nq * k * 12);
    std::vector<int32_t> pos((size_t)parts);
    for (int32_t q = 0; q < nq; ++q) {
        int32_t st = 0;
        for (int32_t p = 0; p < parts; ++p) {
            st |= reinterpret_cast<const int32_t*>(recs[p] + (int64_t)nq * k * 12)[q];
            pos[p] = 0;
        }
        for (int32_t i = 0; i < k; ++i) {                // lists sorted by (score descending, row ascending)
            int32_t best = -1; int64_t bi = -1; float bs = 0.f;
            for (int32_t p = 0; p < parts; ++p) {
                if (pos[p] >= k) continue;
                const int64_t ci = reinterpret_cast<const int64_t*>(recs[p])[(int64_t)q * k + pos[p]];
                if (ci < 0) { pos[p] = k; continue; }    // padding: this shard has no more rows
                const float cs = reinterpret_cast<const float*>(recs[p] + (int64_t)nq * k * 8)[(int64_t)q * k + pos[p]];
                if (best < 0 || cs > bs || (cs == bs && ci < bi)) { best = p; bi = ci; bs = cs; }
            }
            if (best < 0) { oi[(int64_t)q * k + i] = -1; os[(int64_t)q * k + i] = -INFINITY; }
            else { oi[(int64_t)q * k + i] = bi; os[(int64_t)q * k + i] = bs; ++pos[best]; }
        }
        ost[q] = st;
    }
}

extern "C" int hdb_merge_topk_host(const void* records, int32_t parts, int32_t nq, int32_t k, void* out_record) {
    if (!records || !out_record || parts <= 0 || nq < 0 || k < 0) return fail(HDB_ERR_ARG, "hdb_merge_topk_host: bad argument");
    const int64_t nb = hdb_packed_bytes(nq, k);
    std::vector<const char*> recs((size_t)parts);
    for (int32_t p = 0; p < parts; ++p) recs[p] = static_cast<const char*>(records) + p * nb;
    merge_host_records(recs.data(), parts, nq, k, out_record);
    return HDB_OK;
}

extern "C" int hdb_host_exchange_merge(void* shm, int64_t stride, int32_t world, int32_t rank, uint64_t seq, const void* record,
                                       int32_t nq, int32_t k, void* out_record, double timeout_s) {
    if (!shm || !record || !out_record || world <= 0 || rank < 0 || rank >= world || stride < 64 || nq < 0 || k < 0)
        return fail(HDB_ERR_ARG, "hdb_host_exchange_merge: bad argument");
    const int64_t nb = hdb_packed_bytes(nq, k);
    if (nb + 64 > stride) return fail(HDB_ERR_ARG, "hdb_host_exchange_merge: record larger than a slot");
    char* base = static_cast<char*>(shm) + (int64_t)(seq & 1) * world * stride;
    char* mine = base + (int64_t)rank * stride;
    memcpy(mine + 64, record, (size_t)nb);
    __atomic_store_n(reinterpret_cast<uint64_t*>(mine), seq, __ATOMIC_RELEASE);          // publish: after the data
    std::vector<const char*> recs((size_t)world);
    const auto t0 = std::chrono::steady_clock::now();
    for (int32_t r = 0; r < world; ++r) {
        const uint64_t* sp = reinterpret_cast<const uint64_t*>(base + (int64_t)r * stride);
        uint32_t spins = 0;
        while (__atomic_load_n(sp, __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFu) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return fail(HDB_ERR_HIP, "hdb_host_exchange_merge: a rank did not publish its record in time");
        }
        recs[r] = base + (int64_t)r * stride + 64;
    }
    merge_host_records(recs.data(), world, nq, k, out_record);
    return HDB_OK;
}

extern "C" int hdb_topk_host(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric, void* host_record, void* stream) {
    if (!ix || !host_record) return fail(HDB_ERR_ARG, "hdb_topk_host: null argument");
    if (nq <= 0 || k <= 0) return fail(HDB_ERR_ARG, "hdb_topk_host: nq and k must be positive");
    const auto ht0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)hdb_packed_bytes(nq, k);
    // Pinned (device-visible) host memory: the last kernels of the pipeline store the record there themselves and the
    // D2H copy disappears from the critical path; anything else goes through a device record and one hipMemcpyAsync.
    bool direct = false;
    if (ix->host_direct) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, host_record) == hipSuccess) direct = attr.type == hipMemoryTypeHost && attr.devicePointer == host_record;
        else (void)hipGetLastError();
        ix->ht_attr_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ht0).count();
    }
    ix->st_host_direct = direct ? 1 : 0;
    char* rec = static_cast<char*>(host_record);
    if (!direct) {
        if (bytes > ix->rec_bytes) {
            if (ix->rec) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ix->rec)); ix->rec = nullptr; }
            HIP_TRY(hipMalloc((void**)&ix->rec, bytes * 2));
            ix->rec_bytes = bytes * 2;
        }
        rec = ix->rec;
    }
    int64_t* d_idx = reinterpret_cast<int64_t*>(rec);
    float* d_sc = reinterpret_cast<float*>(rec + (size_t)nq * k * 8);
    int32_t* d_st = reinterpret_cast<int32_t*>(rec + (size_t)nq * k * 12);
    // Pinned record: the status words are stored last (by the single-launch kernel's final workgroup, or by each query's
    // finalize workgroup), behind a system-scope release, so the host can poll them instead of waiting for the completion
    // signal of the last kernel (end-of-kernel drain, cache write-back, signal, wake-up: ~5-10 us).  The stream stays
    // ordered: the next launch queues behind the kernels.
    constexpr int32_t SENTINEL = 0x7FFFFFFF;
    volatile int32_t* poll = reinterpret_cast<volatile int32_t*>(static_cast<char*>(host_record) + (size_t)nq * k * 12);
    if (direct && ix->host_poll) for (int q = 0; q < nq; ++q) poll[q] = SENTINEL;
    ix->ht_l0 = ix->ht_l1 = std::chrono::steady_clock::now();      // (pipelines of several launches: everything counts as "pre")
    int rc = topk_impl(ix, dev_Q, nq, k, metric, d_idx, d_sc, d_st, stream, false);
    if (rc) return rc;
    if (!direct) HIP_TRY(hipMemcpyAsync(host_record, rec, bytes, hipMemcpyDeviceToHost, st));
    bool polled = false;
    if (direct && ix->host_poll && (ix->st_fused || ix->st_path != 3)) {      // (the k > 2048 full sort writes its status words elsewhere)
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            bool done = true;
            for (int q = 0; q < nq; ++q) done &= poll[q] != SENTINEL;
            if (done) { polled = true; break; }
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;   // fall back to the signal
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!polled) HIP_TRY(hipStreamSynchronize(st));
    {
        const auto ht3 = std::chrono::steady_clock::now();
        using ns = std::chrono::nanoseconds;
        ix->ht_pre_ns += std::chrono::duration_cast<ns>(ix->ht_l0 - ht0).count();
        ix->ht_launch_ns += std::chrono::duration_cast<ns>(ix->ht_l1 - ix->ht_l0).count();
        ix->ht_wait_ns += std::chrono::duration_cast<ns>(ht3 - ix->ht_l1).count();
        ix->ht_calls++;
    }
    const int32_t* h_st = reinterpret_cast<const int32_t*>(static_cast<const char*>(host_record) + (size_t)nq * k * 12);
    bool any_bad = false;
    for (int q = 0; q < nq; ++q) any_bad |= (h_st[q] & (HDB_Q_UNDERFLOW | HDB_Q_OVERFLOW)) != 0;
    if (!any_bad) return HDB_OK;
    // rare: re-run the failed queries one by one through the exact path, straight into their slots of the record
    const size_t qbytes = (size_t)ix->d * (ix->dtype == HDB_F64 ? 8 : 4);
    std::vector<char> bad(nq);
    for (int q = 0; q < nq; ++q) bad[q] = (h_st[q] & (HDB_Q_UNDERFLOW | HDB_Q_OVERFLOW)) != 0;     // the re-run rewrites h_st
    for (int q = 0; q < nq; ++q) {
        if (!bad[q]) continue;
        rc = topk_impl(ix, static_cast<const char*>(dev_Q) + (size_t)q * qbytes, 1, k, metric, d_idx + (size_t)q * k,
                       d_sc + (size_t)q * k, d_st + q, stream, true);
        if (rc) return rc;
    }
    if (!direct) HIP_TRY(hipMemcpyAsync(host_record, rec, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HDB_OK;
}

extern "C" int hdb_topk_exact(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric, int64_t* dev_idx,
                              float* dev_score, int32_t* dev_status, void* stream) {
    return topk_impl(ix, dev_Q, nq, k, metric, dev_idx, dev_score, dev_status, stream, true);
}

extern "C" int hdb_merge_topk(const int64_t* dev_idx_parts, const float* dev_score_parts, int32_t parts, int32_t nq,
                              int32_t k, int64_t* dev_idx, float* dev_score, int device, void* stream) {
    if (!dev_idx_parts || !dev_score_parts || !dev_idx || !dev_score) return fail(HDB_ERR_ARG, "hdb_merge_topk: null argument");
    if (parts <= 0 || nq < 0 || k < 0) return fail(HDB_ERR_ARG, "hdb_merge_topk: bad sizes");
    if (nq == 0 || k == 0) return HDB_OK;
    if ((int64_t)parts * k > HDB_CAND_CAP) return fail(HDB_ERR_UNSUPPORTED, "hdb_merge_topk: parts*k exceeds 8192");
    HIP_TRY(hipSetDevice(device));
    LAUNCH_TRY(hdb_launch_merge(dev_idx_parts, (int64_t)nq * k * 8, dev_score_parts, (int64_t)nq * k * 4, nullptr, 0, parts, nq,
                                (uint32_t)k, dev_idx, dev_score, nullptr, stream));
    return HDB_OK;
}

extern "C" int64_t hdb_packed_bytes(int32_t nq, int32_t k) {
    const int64_t raw = (int64_t)nq * k * 12 + (int64_t)nq * 4;
    return (raw + 15) / 16 * 16;
}

extern "C" int hdb_merge_topk_packed(const void* dev_gathered, int32_t parts, int32_t nq, int32_t k, int64_t* dev_idx,
                                     float* dev_score, int32_t* dev_status, int device, void* stream) {
    if (!dev_gathered || !dev_idx || !dev_score) return fail(HDB_ERR_ARG, "hdb_merge_topk_packed: null argument");
    if (parts <= 0 || nq < 0 || k < 0) return fail(HDB_ERR_ARG, "hdb_merge_topk_packed: bad sizes");
    if (nq == 0 || k == 0) return HDB_OK;
    if ((int64_t)parts * k > HDB_CAND_CAP) return fail(HDB_ERR_UNSUPPORTED, "hdb_merge_topk_packed: parts*k exceeds 8192");
    HIP_TRY(hipSetDevice(device));
    const int64_t stride = hdb_packed_bytes(nq, k);
    const char* base = (const char*)dev_gathered;
    LAUNCH_TRY(hdb_launch_merge(base, stride, base + (int64_t)nq * k * 8, stride, base + (int64_t)nq * k * 12, stride, parts, nq,
                                (uint32_t)k, dev_idx, dev_score, dev_status, stream));
    return HDB_OK;
}

extern "C" int hdb_recency_bias(const double* dev_ts, int64_t n, double recency_bias, double ts_max, float* dev_out,
                                int device, void* stream) {
    if (n < 0 || (n > 0 && (!dev_ts || !dev_out))) return fail(HDB_ERR_ARG, "hdb_recency_bias: null argument");
    if (n == 0) return HDB_OK;
    HIP_TRY(hipSetDevice(device));
    LAUNCH_TRY(hdb_launch_recency(dev_ts, n, recency_bias, ts_max, dev_out, stream));
    return HDB_OK;
}

extern "C" int hdb_recency_bias_twice(const double* dev_ts, const uint8_t* dev_mask, int64_t n, double recency_bias, double ts_max,
                                      double ts_min, float* dev_out, int device, void* stream) {
    if (n < 0 || (n > 0 && (!dev_ts || !dev_out))) return fail(HDB_ERR_ARG, "hdb_recency_bias_twice: null argument");
    if (n == 0) return HDB_OK;
    HIP_TRY(hipSetDevice(device));
    // max over the kept rows of first_i = rb * exp(-ts_max + ts_i): at the newest row for rb > 0 (exp(0) = 1), at the oldest for rb < 0
    const double first_max = recency_bias >= 0.0 ? recency_bias * std::exp(-ts_max + ts_max) : recency_bias * std::exp(-ts_max + ts_min);
    LAUNCH_TRY(hdb_launch_recency2(dev_ts, dev_mask, n, recency_bias, ts_max, first_max, dev_out, stream));
    return HDB_OK;
}

// ================================================================================================
// Single-process multi-GPU group: one row shard per entry (its own hdb_index, device and stream), queried together
// behind ONE call -- what HyperDB.query() (hyperdb/hyperdb.py:1584, a single-process call) needs to reach several
// GPUs without torchrun.  Per call:
//   calling thread: copies the queries into a pinned, portable staging buffer (every device can read it), publishes the job
//       (an atomic sequence number) and waits for the shards' arrival counter -- spinning first, a condition variable only
//       when a call takes longer than the spin window;
//   worker thread p (one per shard; between calls it spins on the sequence number for a short window, then parks on the
//       condition variable: a query stream never pays a futex wake-up): hdb_topk_host on the shard's own stream -- up to four
//       queries are read by the kernels straight from the staging buffer, larger batches go through one asynchronous copy --
//       which lets the shard's last kernel store its packed record STRAIGHT into slice p of a pinned, portable host buffer,
//       status words last, polls those words instead of waiting for the stream, and re-runs a failed threshold locally
//       through the exact selection (per-shard exactness is all the merge needs);
//   calling thread: k-way merge of the P records on the host (merge_host_records): no merge launch, no collective.
// The exchange unit is the same packed record as the multi-process path (hdb_packed_bytes); it is 1.2 KB per shard at
// nq=1, k=100, so the step is latency-bound and needs no collective library in-process.
// Shards that share a device (a test layout) must not run two single-launch pipelines at once -- each wants every CU for its
// in-kernel exchange -- so their calls take the multi-kernel pipeline; the shards' own handles keep their options.
// ================================================================================================
#include <thread>
#include <mutex>
#include <condition_variable>

struct hdb_group {
    int parts = 0;
    std::vector<hdb_index*> ix;
    std::vector<hipStream_t> st;
    std::vector<void*> qdev;
    std::vector<size_t> qcap;
    std::vector<char> shared_dev;                         // shard p shares its device with another shard of the group
    char* gather = nullptr; size_t gather_bytes = 0;     // pinned + portable host memory: parts records
    char* qpin = nullptr; size_t qpin_bytes = 0;         // pinned + portable host memory: the queries of the current call
    // workers
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::atomic<uint64_t> job_seq{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};
    // current job (written before job_seq moves)
    size_t q_bytes = 0;
    int nq = 0, k = 0, metric = 0; size_t stride = 0;
    std::vector<int> rc;
    std::vector<std::string> err;
};

static constexpr long HDB_GROUP_SPIN_US = 200;           // how long a worker / the caller spins before it parks

static void group_worker(hdb_group* g, int p) {
    (void)hipSetDevice(g->ix[p]->device);
    uint64_t seen = 0;
    for (;;) {
        // wait for a job: spin for a short window (a query stream keeps the workers hot), then park
        bool got = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (g->stop.load(std::memory_order_acquire)) return;
            if (g->job_seq.load(std::memory_order_acquire) != seen) { got = true; break; }
            __builtin_ia32_pause();
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(HDB_GROUP_SPIN_US)) break;
        }
        if (!got) {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->stop.load(std::memory_order_acquire) || g->job_seq.load(std::memory_order_acquire) != seen; });
            if (g->stop.load(std::memory_order_acquire)) return;
        }
        seen = g->job_seq.load(std::memory_order_acquire);
        int rc = HDB_OK;
        std::string msg;
        hdb_index* ix = g->ix[p];
        hipStream_t st = g->st[p];
        if (ix->n > 0) {
            const void* dq = g->qpin;                                // up to four queries: the kernels read the pinned staging buffer
            if (g->nq > 4) {
                hipError_t e = hipSuccess;
                if (g->q_bytes > g->qcap[p]) {
                    if (g->qdev[p]) { (void)hipStreamSynchronize(st); (void)hipFree(g->qdev[p]); g->qdev[p] = nullptr; g->qcap[p] = 0; }
                    e = hipMalloc(&g->qdev[p], g->q_bytes * 2);
                    if (e == hipSuccess) g->qcap[p] = g->q_bytes * 2;
                }
                if (e == hipSuccess) e = hipMemcpyAsync(g->qdev[p], g->qpin, g->q_bytes, hipMemcpyHostToDevice, st);
                if (e != hipSuccess) { rc = HDB_ERR_HIP; msg = std::string("hdb_group: query upload: ") + hipGetErrorString(e); }
                dq = g->qdev[p];
            }
            if (rc == HDB_OK) {
                const int64_t saved = ix->use_fused;
                if (g->shared_dev[p]) ix->use_fused = 0;             // for this call only (one call in flight per handle)
                rc = hdb_topk_host(ix, dq, g->nq, g->k, g->metric, g->gather + (size_t)p * g->stride, st);
                ix->use_fused = saved;
                if (rc != HDB_OK) msg = hdb_last_error();
            }
        } else {                                                     // an empty shard contributes padding
            char* rec = g->gather + (size_t)p * g->stride;
            int64_t* ri = reinterpret_cast<int64_t*>(rec);
            float* rs = reinterpret_cast<float*>(rec + (size_t)g->nq * g->k * 8);
            int32_t* rst = reinterpret_cast<int32_t*>(rec + (size_t)g->nq * g->k * 12);
            for (size_t i = 0; i < (size_t)g->nq * g->k; ++i) { ri[i] = -1; rs[i] = -INFINITY; }
            for (int q = 0; q < g->nq; ++q) rst[q] = 0;
        }
        g->rc[p] = rc; g->err[p] = msg;
        if (g->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            { std::lock_guard<std::mutex> lk(g->mu); }
            g->cv_done.notify_all();
        }
    }
}

extern "C" void hdb_group_destroy(hdb_group* g) {
    if (!g) return;
    g->stop.store(true, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(g->mu); }
    g->cv_job.notify_all();
    for (auto& t : g->th) if (t.joinable()) t.join();
    for (int p = 0; p < (int)g->st.size(); ++p) {
        (void)hipSetDevice(g->ix[p]->device);
        if (g->st[p]) { (void)hipStreamSynchronize(g->st[p]); (void)hipStreamDestroy(g->st[p]); }
        if (p < (int)g->qdev.size() && g->qdev[p]) (void)hipFree(g->qdev[p]);
    }
    if (g->gather) (void)hipHostFree(g->gather);
    if (g->qpin) (void)hipHostFree(g->qpin);
    delete g;
}

extern "C" int hdb_group_create(hdb_group** out, hdb_index* const* shards, int32_t parts) {
    if (!out || !shards) return fail(HDB_ERR_ARG, "hdb_group_create: null argument");
    if (parts < 1 || parts > 64) return fail(HDB_ERR_ARG, "hdb_group_create: 1..64 shards");
    for (int p = 0; p < parts; ++p) {
        if (!shards[p]) return fail(HDB_ERR_ARG, "hdb_group_create: null shard");
        if (shards[p]->d != shards[0]->d || shards[p]->dtype != shards[0]->dtype)
            return fail(HDB_ERR_ARG, "hdb_group_create: shards must share d and dtype");
    }
    hdb_group* g = new hdb_group();
    g->parts = parts;
    g->ix.assign(shards, shards + parts);
    g->st.assign(parts, nullptr); g->qdev.assign(parts, nullptr); g->qcap.assign(parts, 0);
    g->rc.assign(parts, 0); g->err.assign(parts, std::string());
    g->shared_dev.assign(parts, 0);
    for (int p = 0; p < parts; ++p)
        for (int r = 0; r < parts; ++r)
            if (r != p && g->ix[r]->device == g->ix[p]->device) g->shared_dev[p] = 1;
    hipError_t e = hipSuccess;
    for (int p = 0; p < parts && e == hipSuccess; ++p) {
        e = hipSetDevice(g->ix[p]->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->st[p], hipStreamNonBlocking);
    }
    if (e != hipSuccess) { hdb_group_destroy(g); return fail(HDB_ERR_HIP, std::string("hdb_group_create: ") + hipGetErrorString(e)); }
    for (int p = 0; p < parts; ++p) g->th.emplace_back(group_worker, g, p);
    *out = g;
    return HDB_OK;
}

static int group_pinned(char** buf, size_t* have, size_t need) {
    if (need <= *have) return HDB_OK;
    if (*buf) { HIP_TRY(hipHostFree(*buf)); *buf = nullptr; *have = 0; }
    need = align_up(need * 2, 4096);
    HIP_TRY(hipHostMalloc((void**)buf, need, hipHostMallocPortable | hipHostMallocMapped));
    *have = need;
    return HDB_OK;
}

extern "C" int hdb_group_topk_host(hdb_group* g, const void* host_Q, int32_t nq, int32_t k, int metric, void* host_record) {
    if (!g || !host_Q || !host_record) return fail(HDB_ERR_ARG, "hdb_group_topk_host: null argument");
    if (nq <= 0 || k <= 0) return fail(HDB_ERR_ARG, "hdb_group_topk_host: nq and k must be positive");
    const size_t bytes = (size_t)hdb_packed_bytes(nq, k);
    const size_t qelem = g->ix[0]->dtype == HDB_F64 ? 8 : 4;
    const size_t q_bytes = (size_t)g->ix[0]->d * qelem * nq;
    if (bytes * g->parts > g->gather_bytes || q_bytes > g->qpin_bytes) {      // (no call is in flight: the previous one returned)
        HIP_TRY(hipSetDevice(g->ix[0]->device));
        int rc = group_pinned(&g->gather, &g->gather_bytes, bytes * g->parts); if (rc) return rc;
        rc = group_pinned(&g->qpin, &g->qpin_bytes, q_bytes); if (rc) return rc;
    }
    memcpy(g->qpin, host_Q, q_bytes);
    g->q_bytes = q_bytes; g->nq = nq; g->k = k; g->metric = metric; g->stride = bytes;
    g->pending.store(g->parts, std::memory_order_release);
    g->job_seq.fetch_add(1, std::memory_order_acq_rel);
    { std::lock_guard<std::mutex> lk(g->mu); }
    g->cv_job.notify_all();
    // wait for the shards: spin (a shard-sized call takes ~0.2 ms), then park
    {
        const auto t0 = std::chrono::steady_clock::now();
        bool done = false;
        for (unsigned spins = 0;; ++spins) {
            if (g->pending.load(std::memory_order_acquire) == 0) { done = true; break; }
            __builtin_ia32_pause();
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
        if (!done) {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_done.wait(lk, [&] { return g->pending.load(std::memory_order_acquire) == 0; });
        }
    }
    for (int p = 0; p < g->parts; ++p)
        if (g->rc[p] != HDB_OK) return fail(g->rc[p], "shard " + std::to_string(p) + ": " + g->err[p]);
    std::vector<const char*> recs((size_t)g->parts);
    for (int p = 0; p < g->parts; ++p) recs[p] = g->gather + (size_t)p * g->stride;
    merge_host_records(recs.data(), g->parts, nq, k, host_record);
    return HDB_OK;
}
