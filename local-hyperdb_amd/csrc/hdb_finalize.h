// hdb_finalize.h -- the last step of the top-k pipeline as device functions: pre-selection + sort of one query's
// candidate list and the write-out of its k results (the np.argsort of hyperdb/ranking_algorithm.py:200 over the
// survivors of the threshold filter).  Shared by hdb_select.hip (hdb_finalize_kernel, hdb_merge_kernel) and by the fused
// single-launch scan (hdb_mfma_fused.h), whose last workgroup finalizes in place.
#pragma once
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"

#ifndef HDB_FIN_STAMP
#define HDB_FIN_STAMP(slot) do { } while (0)      // diagnostic builds of the fused kernel stamp the phases (tools/stamps_fused.py)
#endif

// ------------------------------------------------------------------------------------------------
// Finalize: sort the candidate list of one query (<= HDB_CAND_CAP packed entries) descending and
// emit the first k.  1024 threads, 64 KiB of LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hdb_bitonic_desc(unsigned long long* buf, int P) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int i = ((t / stride) * (stride << 1)) + (t % stride);
                const int j = i + stride;
                const bool desc = (i & size) == 0;
                const unsigned long long x = buf[i], y = buf[j];
                if (desc ? (x < y) : (x > y)) { buf[i] = y; buf[j] = x; }
            }
            __syncthreads();
        }
    }
}

// Pre-selection for finalize/merge: keeps (in buf[0..ns)) a superset of the kk largest of buf[0..nc) that is
// usually only a little larger than kk, so that the O(log^2) bitonic network runs on ~256 instead of ~4096
// entries.  Monotone linear binning of the 32-bit score key into up to 2048 bins between the smallest and the
// largest key present; everything in or above the bin holding the kk-th largest survives.  Exactness is
// untouched: the survivors always contain the true top-kk, ties included.  Returns ns (>= min(kk, nc)).
// hist: 2048 words of LDS; scratch: nc u64 of LDS (may alias nothing else).
static __device__ uint32_t hdb_preselect(unsigned long long* buf, uint32_t nc, uint32_t kk, uint32_t* hist,
                                  unsigned long long* scratch, uint32_t* ctl /* 8 words */) {
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
    constexpr uint32_t NB = 2048;
    // min / max key
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (uint32_t i = tid; i < nc; i += nth) { const uint32_t k = (uint32_t)(buf[i] >> 32); kmin = min(kmin, k); kmax = max(kmax, k); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64)); }
    if (tid < 8) ctl[tid] = tid == 0 ? 0xFFFFFFFFu : 0u;
    for (uint32_t i = tid; i < NB; i += nth) hist[i] = 0;
    __syncthreads();
    if (lane == 0) { atomicMin(&ctl[0], kmin); atomicMax(&ctl[1], kmax); }
    __syncthreads();
    kmin = ctl[0]; kmax = ctl[1];
    // monotone binning by a shift (no 64-bit division: that alone was ~3 us of an 8 us pre-selection): the key range is
    // cut into 1024..2047 equal bins
    const uint32_t range = kmax - kmin;
    const int bits = range ? 32 - __clz((int)range) : 0;
    const int sh = bits > 11 ? bits - 11 : 0;
    auto bin_of = [&](uint32_t k) { return (k - kmin) >> sh; };
    for (uint32_t i = tid; i < nc; i += nth) atomicAdd(&hist[bin_of((uint32_t)(buf[i] >> 32))], 1u);
    __syncthreads();
    // suffix scan from the top bin: thread t owns bins NB-1-2t, NB-2-2t (1024 threads) -- generic stride
    const uint32_t per = (NB + nth - 1) / nth;
    uint32_t loc[8]; uint32_t tot = 0;              // per <= 8 (blockDim >= 256)
    for (uint32_t j = 0; j < per; ++j) { const uint32_t b = tid * per + j; loc[j] = b < NB ? hist[NB - 1 - b] : 0; tot += loc[j]; }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    uint32_t* wsum = hist;      // reuse after everyone has read its bins
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - tot;
    for (int w2 = 0; w2 < wave; ++w2) before += wsum[w2];
    for (uint32_t j = 0; j < per; ++j) {
        if (before < kk && kk <= before + loc[j]) ctl[2] = NB - 1 - (tid * per + j);     // the bin holding the kk-th largest
        before += loc[j];
    }
    __syncthreads();
    const uint32_t bsel = ctl[2];
    // compact survivors
    for (uint32_t i = tid; i < nc; i += nth) {
        const unsigned long long e = buf[i];
        if (bin_of((uint32_t)(e >> 32)) >= bsel) scratch[atomicAdd(&ctl[3], 1u)] = e;
    }
    __syncthreads();
    const uint32_t ns = ctl[3];
    for (uint32_t i = tid; i < ns; i += nth) buf[i] = scratch[i];
    __syncthreads();
    (void)nwaves;
    return ns;
}

// Finalize of ONE query by the calling workgroup (any size that is a multiple of 64, >= 256 threads).  `buf` is LDS:
// buf (cap u64) | scratch (cap u64) | hist (2048 u32) | ctl (8 u32) = cap*16 + 8224 bytes.  `total` = number of
// candidates that were appended (may exceed cap: overflow); extra_status is OR-ed into the status word.
// Called by hdb_finalize_kernel (one workgroup per query) and by the last workgroup of the fused scan kernel.
// With `floor_ptr` the function returns, to every thread, whether the kk-th best candidate scores ABOVE
// canon(*floor_ptr * floor_mul), i.e. whether at least kk candidates do (the list may hold entries that only some
// workgroups collected, see hdb_mfma_fused.h; ctl word 8).  The load of *floor_ptr overlaps the candidate loads.
// `fix(buf, nc)` runs on the candidates once they are in LDS (all threads call it; it ends with a barrier of its own if it
// writes): the single-launch batched scan re-scores euclidean near-duplicates there (hdb_mfma_kernel.h).
struct HdbNoFix { __device__ __forceinline__ void operator()(unsigned long long*, uint32_t) const {} };
template <typename Fix = HdbNoFix>
__device__ __forceinline__ uint32_t hdb_finalize_body(unsigned long long* buf, const unsigned long long* cand, uint32_t total, int q,
                                                      uint32_t cap, uint32_t k, uint32_t kk /* min(k, n) */, int64_t row_base,
                                                      int64_t* idx_out, float* score_out, int32_t* status, int qnan_flag,
                                                      int32_t extra_status, const float* floor_ptr = nullptr, float floor_mul = 1.f,
                                                      const Fix& fix = Fix()) {
    unsigned long long* scratch = buf + cap;
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch + cap);
    uint32_t* ctl = hist + 2048;                    // 16 words
    const uint32_t nc = total < cap ? total : cap;
    HDB_FIN_STAMP(8);
    float floor_v = 0.f;
    if (floor_ptr) {
        floor_v = __hip_atomic_load(floor_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) ctl[8] = 0;
    }
    for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) buf[i] = cand[i];      // `cand` = this query's list
    __syncthreads();
    fix(buf, nc);
    const uint32_t floor_key = floor_ptr ? hdb_f2key(hdb_canon(floor_v * floor_mul)) : 0u;
    HDB_FIN_STAMP(9);
    uint32_t ns = nc;
    if (nc > 512 && kk < nc / 2) ns = hdb_preselect(buf, nc, kk, hist, scratch, ctl);
    HDB_FIN_STAMP(10);
    const uint32_t nout = nc < kk ? nc : kk;             // entries that exist
    if (ns <= 256) {
        // Few survivors (the usual case: ~kk plus one histogram bin): rank sort.  Packed entries are distinct, so the
        // number of larger entries is the output position; one pass of LDS broadcast reads, no barrier.
        for (uint32_t i = nout + threadIdx.x; i < k; i += blockDim.x) { idx_out[(int64_t)q * k + i] = -1; score_out[(int64_t)q * k + i] = -INFINITY; }
        if (threadIdx.x < ns) {
            const unsigned long long mine = buf[threadIdx.x];
            uint32_t rank = 0;
            uint32_t j = 0;
            for (; j + 2 <= ns; j += 2) {
                const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(buf + j);
                rank += (pr.x > mine) + (pr.y > mine);
            }
            if (j < ns) rank += buf[j] > mine;
            if (rank < nout) {
                idx_out[(int64_t)q * k + rank] = row_base + (int64_t)(0xFFFFFFFFu - (uint32_t)(mine & 0xFFFFFFFFull));
                score_out[(int64_t)q * k + rank] = hdb_key2f((uint32_t)(mine >> 32));
            }
            if (floor_ptr && rank == kk - 1) ctl[8] = (uint32_t)(mine >> 32) > floor_key ? 1u : 0u;      // the kk-th best
        }
    } else {
    int P = 64;
    while ((uint32_t)P < ns) P <<= 1;
    for (int i = ns + threadIdx.x; i < P; i += blockDim.x) buf[i] = 0ull;
    __syncthreads();
    hdb_bitonic_desc(buf, P);
    if (floor_ptr && threadIdx.x == 0 && nc >= kk && kk > 0) ctl[8] = (uint32_t)(buf[kk - 1] >> 32) > floor_key ? 1u : 0u;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        if (i < nc && i < kk) {
            const unsigned long long e = buf[i];
            idx_out[(int64_t)q * k + i] = row_base + (int64_t)(0xFFFFFFFFu - (uint32_t)(e & 0xFFFFFFFFull));
            score_out[(int64_t)q * k + i] = hdb_key2f((uint32_t)(e >> 32));
        } else {
            idx_out[(int64_t)q * k + i] = -1;
            score_out[(int64_t)q * k + i] = -INFINITY;
        }
    }
    }
    if (status) {
        // The status word goes out LAST, behind every thread's result stores and a system-scope release: a host that polls it
        // in a pinned record (hdb_topk_host) may read this query's results as soon as the word has left its sentinel value.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t st = extra_status;
            if (total > cap) st |= HDB_Q_OVERFLOW;
            if (nc < kk) st |= HDB_Q_UNDERFLOW;
            if (qnan_flag) st |= HDB_Q_NAN;
            __hip_atomic_store(status + q, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t kth_above = 0;
    if (floor_ptr) { __syncthreads(); kth_above = ctl[8]; }
    return kth_above;
}

