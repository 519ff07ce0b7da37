// hdb_select.hip -- top-k selection kernels for gfx950 (the np.argpartition + np.argsort of
// hyperdb/ranking_algorithm.py:194-200, restated for a GPU).
//
// Ordering rule everywhere: (score descending, row index ascending) -- a total order, so the
// top-k is unique and independent of launch geometry and of the number of GPUs.
//
//  * radix histogram passes over a dense float score array (8 bits per pass on the orderable
//    key).  After 2 passes the lower edge of the bin holding the m-th largest SAMPLE score is a
//    safe per-query threshold for the fused scan; after 4 passes the k-th largest key is exact.
//  * collect: everything above the k-th key plus its ties -> candidate list (exact path).
//  * ordered tie scan: when the k-th key has more ties than the candidate list can hold, one
//    workgroup walks the array in index order and keeps the first `need` of them.
//  * finalize: one workgroup per query bitonic-sorts <= 8192 packed candidates in LDS and writes
//    the first k as (int64 index + row_base, float score).
//  * merge: the same sort over the all-gathered per-shard lists (multi-GPU exchange step).
//
// No state survives between kernels except the histograms: each kernel re-derives the radix
// prefix from hist[q][0..pass) in its prologue, so kernel boundaries provide all the ordering.
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"
#include "hdb_finalize.h"

// hist layout: [nq][4][256] uint32
struct Prefix { uint32_t key; uint32_t need; uint32_t bin_count; };

// Executed by wave 0 of a block.  Walks passes [0, npass): in each, finds (from the top) the bin
// holding the `need`-th largest remaining key.  Returns the key prefix (low bits zero), how many
// are still needed from inside that bin, and the population of the final bin.
__device__ Prefix hdb_derive_prefix(const uint32_t* hist_q, int npass, uint32_t k, int lane) {
    uint32_t key = 0, need = k, bin_count = 0;
    for (int p = 0; p < npass; ++p) {
        const uint32_t* h = hist_q + p * HDB_RADIX_BINS;
        // lane owns descending bins 255-4*lane .. 252-4*lane
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = h[255 - (4 * lane + j)]; tot += c[j]; }
        uint32_t incl = tot;                       // inclusive scan over lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        uint32_t before = incl - tot;              // keys in bins above this lane's bins
        int found_bin = -1; uint32_t found_before = 0, found_cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (found_bin < 0 && before < need && need <= before + c[j]) {
                found_bin = 255 - (4 * lane + j); found_before = before; found_cnt = c[j];
            }
            before += c[j];
        }
        const unsigned long long who = __ballot(found_bin >= 0);
        int src = who ? (int)__ffsll((long long)who) - 1 : 63;
        int bin = __shfl(found_bin, src, 64);
        uint32_t fb = __shfl(found_before, src, 64);
        uint32_t fc = __shfl(found_cnt, src, 64);
        if (!who) { bin = 0; fb = 0; fc = 0; }     // fewer keys than needed: take everything
        key |= (uint32_t)bin << (24 - 8 * p);
        need = who ? need - fb : need;
        bin_count = fc;
    }
    Prefix r; r.key = key; r.need = need; r.bin_count = bin_count;
    return r;
}

// ------------------------------------------------------------------------------------------------
// Histogram pass.  grid = (blocks, nq).  Keys not matching the prefix of earlier passes are skipped.
// LDS atomics, with ONE round of wave aggregation first: the lanes that share the first active lane's bin
// elect one adder (in the leading passes nearly all keys of a wave sit in one exponent bin, which would
// serialise 64 plain atomics), the others add for themselves (a few-way conflict at most; looping the
// election over every distinct bin cost 144 us per pass on hamming scores, ~30 distinct bins per wave).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hdb_hist_kernel(const float* scores, int64_t n, int64_t ld, uint32_t* hist,
                                                       int pass, uint32_t k) {
    __shared__ uint32_t lh[HDB_RADIX_BINS];
    __shared__ uint32_t s_prefix;
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    lh[threadIdx.x] = 0;
    if (threadIdx.x < 64) {
        const Prefix pf = hdb_derive_prefix(hist + (int64_t)q * 4 * HDB_RADIX_BINS, pass, k, lane);
        if (lane == 0) s_prefix = pf.key;
    }
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t mask = pass == 0 ? 0u : (0xFFFFFFFFu << (32 - 8 * pass));
    const int shift = 24 - 8 * pass;
    const float* sq = scores + (int64_t)q * ld;
    const int64_t nround = (n + 255) / 256 * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nround; i += (int64_t)gridDim.x * 256) {
        bool active = i < n;
        uint32_t bin = 0;
        if (active) {
            const uint32_t key = hdb_f2key(sq[i]);
            active = (key & mask) == prefix;
            bin = (key >> shift) & 255u;
        }
        const unsigned long long todo = __ballot(active);
        if (todo) {
            const int leader = (int)__ffsll((long long)todo) - 1;
            const uint32_t lb = __shfl(bin, leader, 64);
            const bool with_leader = active && bin == lb;
            const unsigned long long same = __ballot(with_leader);
            if (lane == leader) atomicAdd(&lh[lb], (uint32_t)__popcll(same));
            else if (active && !with_leader) atomicAdd(&lh[bin], 1u);
        }
    }
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    if (c) atomicAdd(&hist[((int64_t)q * 4 + pass) * HDB_RADIX_BINS + threadIdx.x], c);
}

// Threshold from the sample histograms: lower edge of the bin holding the m-th largest sample score
// after `npass` passes (so at least m sample scores are >= thr).  Also resets the candidate counter.
// One wave per query.
__global__ __launch_bounds__(64) void hdb_thr_kernel(const uint32_t* hist, int npass, uint32_t m, uint32_t sample_n,
                                                     float* thr, uint32_t* cnt) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const Prefix pf = hdb_derive_prefix(hist + (int64_t)q * 4 * HDB_RADIX_BINS, npass, m, lane);
    if (lane == 0) {
        thr[q] = (sample_n < m) ? -INFINITY : hdb_key2f(pf.key);
        cnt[q * HDB_CNT_STRIDE] = 0;
    }
}

// Sample threshold in ONE launch (m <= 16): one workgroup per query.  Every thread keeps only the maximum of
// its strided share of the sample; the m-th largest of those 1024 maxima is a lower bound of the m-th largest
// sample score (order statistics of a subset), and equals it unless two of the top m fell to one thread
// (a few % of cases, then it is the (m+1)-th or so).  A lower bound is all the filter needs: the threshold only
// has to keep >= k rows and not too many.  No histograms, no LDS atomics: wave-level max extraction only.
// thr[q] = that bound, cnt[q] = 0.  Replaces memset + 4 x hdb_hist_kernel + hdb_thr_kernel per query.
__device__ __forceinline__ uint32_t hdb_wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
__global__ __launch_bounds__(1024) void hdb_sample_thr_kernel(const float* scores, int64_t n, int64_t ld, uint32_t m,
                                                              float* thr, uint32_t* cnt, uint32_t* tile_ctr) {
    __shared__ uint32_t top[16 * 16];
    const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tile_ctr && q == 0 && threadIdx.x == 0) *tile_ctr = 0u;       // the filter pass's tile hand-out counter
    const float* sq = scores + (int64_t)q * ld;
    uint32_t best = 0u;                                  // key 0 is below every real score, even -inf
    const int64_t n4 = n / 4;
    const float4* sq4 = reinterpret_cast<const float4*>(sq);   // ld and base are multiples of 4 floats / 16 bytes
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = sq4[i];
        best = max(max(best, hdb_f2key(v.x)), max(max(hdb_f2key(v.y), hdb_f2key(v.z)), hdb_f2key(v.w)));
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) best = max(best, hdb_f2key(sq[i]));
    // wave: extract its m largest maxima
    uint32_t v = best;
    for (uint32_t r = 0; r < m; ++r) {
        const uint32_t wm = hdb_wave_max_u32(v);
        const unsigned long long who = __ballot(v == wm);
        if (lane == (int)__ffsll((long long)who) - 1) v = 0u;
        if (lane == 0) top[wave * 16 + r] = wm;
    }
    __syncthreads();
    if (wave == 0) {
        // 16 waves x m values: lane holds up to 4 of them
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int e = lane + 64 * j; c[j] = (uint32_t)(e & 15) < m ? top[e] : 0u; }
        uint32_t kth = 0u;
        for (uint32_t r = 0; r < m; ++r) {
            const uint32_t lm = max(max(c[0], c[1]), max(c[2], c[3]));
            const uint32_t wm = hdb_wave_max_u32(lm);
            const unsigned long long who = __ballot(lm == wm);
            if (lane == (int)__ffsll((long long)who) - 1) {
                if (c[0] == wm) c[0] = 0u; else if (c[1] == wm) c[1] = 0u; else if (c[2] == wm) c[2] = 0u; else c[3] = 0u;
            }
            kth = wm;
        }
        if (lane == 0) {
            thr[q] = (n < (int64_t)m || kth == 0u) ? -INFINITY : hdb_key2f(kth);
            cnt[q * HDB_CNT_STRIDE] = 0;
        }
    }
}

__global__ void hdb_fill_thr_kernel(float* thr, uint32_t* cnt, int nq, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) { thr[i] = v; cnt[i * HDB_CNT_STRIDE] = 0; }
}

// ------------------------------------------------------------------------------------------------
// Exact path: collect keys above the k-th key, and its ties when they fit.
// tie_info[q] = {kth key, need (ties still wanted), ties_all flag, count_gt}
// ------------------------------------------------------------------------------------------------
// npass < 4 is for scores whose keys are known to be zero below the bits the passes covered (hamming counts).
__global__ __launch_bounds__(256) void hdb_collect_kernel(const float* scores, int64_t n, int64_t ld, const uint32_t* hist, int npass,
                                                          uint32_t k, uint32_t* cnt, unsigned long long* cand,
                                                          uint32_t cap, uint32_t* tie_info) {
    __shared__ uint32_t s_key, s_need, s_all;
    const int q = blockIdx.y, lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        const Prefix pf = hdb_derive_prefix(hist + (int64_t)q * 4 * HDB_RADIX_BINS, npass, k, lane);
        if (lane == 0) {
            const uint32_t count_gt = k - pf.need;
            const uint32_t all = (count_gt + pf.bin_count <= cap) ? 1u : 0u;
            s_key = pf.key; s_need = pf.need; s_all = all;
            if (blockIdx.x == 0) {
                tie_info[q * 4 + 0] = pf.key; tie_info[q * 4 + 1] = pf.need;
                tie_info[q * 4 + 2] = all;    tie_info[q * 4 + 3] = count_gt;
            }
        }
    }
    __syncthreads();
    const uint32_t kth = s_key;
    const bool ties_all = s_all != 0;
    const float* sq = scores + (int64_t)q * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float s = sq[i];
        const uint32_t key = hdb_f2key(s);
        if (key > kth || (ties_all && key == kth)) {
            const uint32_t pos = atomicAdd(&cnt[q * HDB_CNT_STRIDE], 1u);
            if (pos < cap) cand[(int64_t)q * cap + pos] = hdb_pack(s, (uint32_t)i);
        }
    }
}

// Ordered tie scan (only when ties did not fit): first `need` rows, in index order, whose key
// equals the k-th key.  One workgroup of 1024 threads per query, stops as soon as it has enough.
__global__ __launch_bounds__(1024) void hdb_ties_seq_kernel(const float* scores, int64_t n, int64_t ld, uint32_t* cnt,
                                                            unsigned long long* cand, uint32_t cap, const uint32_t* tie_info) {
    const int q = blockIdx.x;
    if (tie_info[q * 4 + 2]) return;                      // ties were all collected already
    const uint32_t kth = tie_info[q * 4 + 0], need = tie_info[q * 4 + 1], count_gt = tie_info[q * 4 + 3];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_taken;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_taken = 0;
    __syncthreads();
    const float* sq = scores + (int64_t)q * ld;
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const bool hit = (i < n) && hdb_f2key(sq[i]) == kth;
        const unsigned long long b = __ballot(hit);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = s_taken;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const uint32_t rank = before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (hit && rank < need) {
            // slots [count_gt, count_gt+need) are reserved for ties: deterministic placement
            cand[(int64_t)q * cap + count_gt + rank] = hdb_pack(sq[i], (uint32_t)i);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            s_taken += tot;
        }
        __syncthreads();
        if (s_taken >= need) break;
    }
    if (threadIdx.x == 0) cnt[q * HDB_CNT_STRIDE] = count_gt + need;       // > kth entries occupy [0, count_gt)
}

// ------------------------------------------------------------------------------------------------
// Finalize kernel: one workgroup per query (body in hdb_finalize.h).  1024 threads, 136 KiB of LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hdb_finalize_kernel(const unsigned long long* cand, const uint32_t* cnt,
                                                            uint32_t cap, uint32_t k, uint32_t kk /* min(k, n) */,
                                                            int64_t row_base, int64_t* idx_out, float* score_out,
                                                            int32_t* status, const int* qnan, int inf_status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long buf[];
    const int q = blockIdx.x;
    // (round 4: the register-resident flavour where the list fits 16 entries per thread and k <= 256, else the general body)
    hdb_finalize_fast(buf, cand + (int64_t)q * cap, cnt[q * HDB_CNT_STRIDE], q, cap, k, kk, row_base, idx_out, score_out, status,
                      qnan ? (qnan[q] & 1) : 0, (qnan && (qnan[q] & 2)) ? inf_status : 0);       // (qnan: 1 = the query holds a NaN, 2 = an infinity)
}

// ------------------------------------------------------------------------------------------------
// Merge of per-shard lists: parts x k candidates per query.  Shard p's rows precede shard p+1's
// (contiguous row sharding) and each list is already in canonical order, so position p*k+i is a
// valid tie-break for equal scores.
// ------------------------------------------------------------------------------------------------
// Rank-by-binary-search merge: every part list is already sorted (score descending, row ascending), so the global
// rank of entry i of part p is  i + sum over the other parts of #{entries that sort before it}: parts-1 binary
// searches over keys in LDS (row ids are read from global memory only on key ties), no sort and no
// synchronisation after the keys are staged.  One workgroup per query, one thread per entry.
__global__ __launch_bounds__(1024) void hdb_merge_kernel(const char* idx_base, int64_t idx_stride, const char* score_base,
                                                         int64_t score_stride, const char* status_base, int64_t status_stride,
                                                         int parts, int nq, uint32_t k, int64_t* idx_out, float* score_out,
                                                         int32_t* status_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long buf[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(buf);            // [parts][k], 0 for empty slots (below every real key)
    const int q = blockIdx.x;
    const uint32_t total = (uint32_t)parts * k;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        const uint32_t p = i / k, j = i - p * k;
        const int64_t off = (int64_t)q * k + j;
        const int64_t id = reinterpret_cast<const int64_t*>(idx_base + (int64_t)p * idx_stride)[off];
        const float sc = reinterpret_cast<const float*>(score_base + (int64_t)p * score_stride)[off];
        keys[i] = id >= 0 ? hdb_f2key(hdb_canon(sc)) : 0u;
    }
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) { idx_out[(int64_t)q * k + i] = -1; score_out[(int64_t)q * k + i] = -INFINITY; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        const uint32_t key = keys[i];
        if (key == 0u) continue;
        const uint32_t p = i / k, j = i - p * k;
        const int64_t off = (int64_t)q * k + j;
        const int64_t my_id = reinterpret_cast<const int64_t*>(idx_base + (int64_t)p * idx_stride)[off];
        uint32_t rank = j;
        for (uint32_t pp = 0; pp < (uint32_t)parts && rank < k; ++pp) {
            if (pp == p) continue;
            const uint32_t* lst = keys + pp * k;
            const int64_t* ids = reinterpret_cast<const int64_t*>(idx_base + (int64_t)pp * idx_stride) + (int64_t)q * k;
            // lists are (key descending, row ascending): count the entries that sort before (key, my_id)
            uint32_t lo = 0, hi = k;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t v = lst[mid];
                const bool before = v > key || (v == key && ids[mid] < my_id);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            idx_out[(int64_t)q * k + rank] = my_id;
            score_out[(int64_t)q * k + rank] = reinterpret_cast<const float*>(score_base + (int64_t)p * score_stride)[off];
        }
    }
    if (threadIdx.x == 0 && status_out) {
        int32_t st = 0;
        if (status_base)
            for (int p = 0; p < parts; ++p) st |= reinterpret_cast<const int32_t*>(status_base + (int64_t)p * status_stride)[q];
        status_out[q] = st;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
extern "C" int hdb_launch_hist(const float* scores, int64_t n, int64_t ld, int nq, uint32_t* hist, int pass, uint32_t k, void* stream) {
    const dim3 grid(hdb_grid_for(n, 256 * 8, 1024), nq);
    hipLaunchKernelGGL(hdb_hist_kernel, grid, dim3(256), 0, (hipStream_t)stream, scores, n, ld, hist, pass, k);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_thr(const uint32_t* hist, int nq, int npass, uint32_t m, uint32_t sample_n, float* thr, uint32_t* cnt, void* stream) {
    hipLaunchKernelGGL(hdb_thr_kernel, dim3(nq), dim3(64), 0, (hipStream_t)stream, hist, npass, m, sample_n, thr, cnt);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_sample_thr(const float* scores, int64_t n, int64_t ld, int nq, uint32_t m, float* thr, uint32_t* cnt, uint32_t* tile_ctr, void* stream) {
    hipLaunchKernelGGL(hdb_sample_thr_kernel, dim3(nq), dim3(1024), 0, (hipStream_t)stream, scores, n, ld, m, thr, cnt, tile_ctr);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_fill_thr(float* thr, uint32_t* cnt, int nq, float v, void* stream) {
    hipLaunchKernelGGL(hdb_fill_thr_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, thr, cnt, nq, v);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_collect(const float* scores, int64_t n, int64_t ld, int nq, const uint32_t* hist, int npass, uint32_t k,
                                  uint32_t* cnt, unsigned long long* cand, uint32_t cap, uint32_t* tie_info, void* stream) {
    const dim3 grid(hdb_grid_for(n, 256 * 8, 1024), nq);
    hipLaunchKernelGGL(hdb_collect_kernel, grid, dim3(256), 0, (hipStream_t)stream, scores, n, ld, hist, npass, k, cnt, cand, cap, tie_info);
    hipLaunchKernelGGL(hdb_ties_seq_kernel, dim3(nq), dim3(1024), 0, (hipStream_t)stream, scores, n, ld, cnt, cand, cap, tie_info);
    return (int)hipGetLastError();
}
// status[q] = HDB_Q_NAN if the query held a NaN, else 0 (the full-sort path has no finalize kernel to write it)
__global__ void hdb_status_nan_kernel(const int* qnan, int nq, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) status[i] = (qnan[i] & 1) ? HDB_Q_NAN : 0;
}
extern "C" int hdb_launch_status_nan(const int* qnan, int nq, int32_t* status, void* stream) {
    hipLaunchKernelGGL(hdb_status_nan_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, qnan, nq, status);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_finalize(const unsigned long long* cand, const uint32_t* cnt, uint32_t cap, int nq, uint32_t k,
                                   uint32_t kk, int64_t row_base, int64_t* idx_out, float* score_out, int32_t* status,
                                   const int* qnan, int threads, int inf_status, void* stream) {
    const size_t lds = (size_t)cap * 16 + 2048 * 4 + 64;
    static unsigned long long attr_done = 0;
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(hdb_finalize_kernel), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    if (threads != 256 && threads != 512) threads = 1024;
    hipLaunchKernelGGL(hdb_finalize_kernel, dim3(nq), dim3(threads), lds, (hipStream_t)stream, cand, cnt, cap, k, kk,
                       row_base, idx_out, score_out, status, qnan, inf_status);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_merge(const void* idx_base, int64_t idx_stride, const void* score_base, int64_t score_stride,
                                const void* status_base, int64_t status_stride, int parts, int nq, uint32_t k,
                                int64_t* idx_out, float* score_out, int32_t* status_out, void* stream) {
    const size_t lds = ((size_t)parts * k * 4 + 15) / 16 * 16;
    const unsigned threads = (size_t)parts * k >= 1024 ? 1024 : 256;
    hipLaunchKernelGGL(hdb_merge_kernel, dim3(nq), dim3(threads), lds, (hipStream_t)stream, (const char*)idx_base, idx_stride,
                       (const char*)score_base, score_stride, (const char*)status_base, status_stride, parts, nq, k, idx_out,
                       score_out, status_out);
    return (int)hipGetLastError();
}
