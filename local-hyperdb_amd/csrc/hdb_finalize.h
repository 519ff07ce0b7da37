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
            if (floor_ptr && kk > 0 && ctl[8] == 0u) st |= HDB_Q_UNDERFLOW;      // the k-th best does not clear the floor: the list may be incomplete
            if (qnan_flag) st |= HDB_Q_NAN;
            __hip_atomic_store(status + q, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t kth_above = 0;
    if (floor_ptr) { __syncthreads(); kth_above = ctl[8]; }
    return kth_above;
}

// ------------------------------------------------------------------------------------------------
// The same step with the candidates in REGISTERS (round 4): the usual case -- a few hundred to a few thousand candidates, k <= 256 --
// needs no LDS copy of the list, no separate min/max pass and no copy-back: four barriers instead of eleven
// (profiles/r4_finalize_fast.txt).  Loads -> per-wave min/max (DPP) -> 1024 shift bins between the smallest and the largest key ->
// wave 0 alone scans the bins from the top -> entries in or above the bin of the kk-th largest go to a short LDS list -> rank sort.
// Anything else (more than 8 entries per thread, a fix-up functor, very short or very tie-heavy lists) takes hdb_finalize_body;
// both produce identical results (the survivors always contain the true top-kk; the order is the total order of the packed entries).
// `have_floor_key`: the caller passes the floor as an orderable key of the SCORE domain instead of floor_ptr / floor_mul.
// LDS: hist (1024 u32) | wave minima / maxima (32 u32) | ctl (16 u32) | survivors (1024 u64) = 12.5 KiB at `buf`.
// ------------------------------------------------------------------------------------------------
template <typename A, typename B> struct HdbSame { static constexpr bool value = false; };
template <typename A> struct HdbSame<A, A> { static constexpr bool value = true; };

template <typename Fix = HdbNoFix>
__device__ __forceinline__ uint32_t hdb_finalize_fast(unsigned long long* buf, const unsigned long long* cand, uint32_t total, int q,
                                                      uint32_t cap, uint32_t k, uint32_t kk /* min(k, n) */, int64_t row_base,
                                                      int64_t* idx_out, float* score_out, int32_t* status, int qnan_flag,
                                                      int32_t extra_status, const float* floor_ptr = nullptr, float floor_mul = 1.f,
                                                      const Fix& fix = Fix(), bool have_floor_key = false, uint32_t floor_key_in = 0u,
                                                      const uint32_t* slot_cnt = nullptr, uint32_t slot_size = 0u, uint32_t slot_wgs = 0u,
                                                      int max_passes = 4 /* 1: coarse scores (bit metrics) -- a crowded bin is a tie, narrowing the window cannot split it */) {
    // Slotted lists (the local flavour of hdb_mfma_fused.h): workgroup w left slot_cnt[w] (LDS) entries at cand[w * slot_size ...];
    // `total` is their sum.  Such a list is only ever read here (it is not compact: hdb_finalize_body cannot take it), so this path
    // serves every size of it; ties too massive for the survivor list come back as "k-th best not above the floor" (0).
    constexpr int NE = 16;
    constexpr uint32_t NB = 1024, SCAP = 1024;
    const bool slotted = slot_cnt != nullptr;
    const uint32_t nc = total < cap ? total : cap;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    const uint32_t nload = slotted ? slot_wgs * slot_size : nc;      // list positions to look at
    const int slot_sh = slotted ? 31 - __clz((int)slot_size) : 0;
    const bool fast = slotted || (HdbSame<Fix, HdbNoFix>::value && nc > 256u && nc <= (uint32_t)(NE * nth) && kk <= 256u && kk > 0u && nw <= 16);
    if (!fast) {
        if (have_floor_key) {                        // (hdb_finalize_body takes its floor through memory: park the key in its ctl area)
            float* fk = reinterpret_cast<float*>(reinterpret_cast<uint32_t*>(buf + 2 * (size_t)cap) + 2048 + 12);
            if (tid == 0) *fk = hdb_key2f(floor_key_in);
            __syncthreads();
            return hdb_finalize_body(buf, cand, total, q, cap, k, kk, row_base, idx_out, score_out, status, qnan_flag, extra_status, fk, 1.f, fix);
        }
        return hdb_finalize_body(buf, cand, total, q, cap, k, kk, row_base, idx_out, score_out, status, qnan_flag, extra_status, floor_ptr, floor_mul, fix);
    }
    uint32_t* hist = reinterpret_cast<uint32_t*>(buf);
    uint32_t* wred = hist + NB;
    uint32_t* ctl = wred + 32;
    unsigned long long* sbuf = reinterpret_cast<unsigned long long*>(ctl + 16);
    HDB_FIN_STAMP(8);
    float floor_v = 0.f;
    if (floor_ptr && !have_floor_key) floor_v = __hip_atomic_load(floor_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long e[NE];
    bool ev[NE];                                     // position j of this thread holds an entry
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const uint32_t i = (uint32_t)(tid + j * nth);
        ev[j] = i < nload && (!slotted || (i & (slot_size - 1u)) < slot_cnt[i >> slot_sh]);      // (slot_size is a power of two)
        e[j] = ev[j] ? cand[i] : 0ull;
    }
    for (uint32_t i = tid; i < NB; i += nth) hist[i] = 0u;
    if (tid < 16) ctl[tid] = 0u;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < NE; ++j)
        if (ev[j]) { const uint32_t kx = (uint32_t)(e[j] >> 32); kmin = min(kmin, kx); kmax = max(kmax, kx); }
    kmin = ~hdb_wave_max_dpp(~kmin);
    kmax = hdb_wave_max_dpp(kmax);
    if (lane == 0) { wred[wave] = kmin; wred[16 + wave] = kmax; }
    __syncthreads();
    HDB_FIN_STAMP(9);
    for (int w2 = 0; w2 < nw; ++w2) { kmin = min(kmin, wred[w2]); kmax = max(kmax, wred[16 + w2]); }
    if (kmin > kmax) { kmin = 0u; kmax = 0u; }       // (an empty list)
    // Shift bins over the key window [lo, hi]; when the bin of the kk-th largest is crowded the window shrinks to that bin and the
    // pass repeats (at most three times: 10 bits per pass).  Orderable keys are far apart around zero -- scores of -0.5 and +0.5 sit
    // 2^31 keys apart -- so one pass over a list that straddles zero puts all the large scores into a dozen bins (found with the
    // local flavour at local_m = 24 of 64 rows: thresholds below zero, 300+ entries in the bin of the k-th, round 4).
    uint32_t lo = kmin, hi = kmax, need = kk, cutoff = kmin;
#pragma unroll 1
    for (int pass = 0; pass < max_passes; ++pass) {
        const uint32_t range = hi - lo;
        const int bits = range ? 32 - __clz((int)range) : 0;
        const int sh = bits > 10 ? bits - 10 : 0;
        if (pass > 0) {
            for (uint32_t i = tid; i < NB; i += nth) hist[i] = 0u;
            if (tid == 0) { ctl[2] = 0u; ctl[4] = 0u; ctl[5] = 0u; }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const uint32_t kx = (uint32_t)(e[j] >> 32);
            if (ev[j] && kx >= lo && kx <= hi) atomicAdd(&hist[(kx - lo) >> sh], 1u);
        }
        __syncthreads();
        if (wave == 0) {                             // bins from the top: lane l owns bins NB-1-16l .. NB-16-16l
            uint32_t loc[16], tot = 0u;
#pragma unroll
            for (int j = 0; j < 16; ++j) { loc[j] = hist[NB - 1 - (16 * lane + j)]; tot += loc[j]; }
            uint32_t incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
            uint32_t before = incl - tot;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (before < need && need <= before + loc[j]) {      // the bin holding the wanted entry (none: fewer entries than that, bin 0 = everything)
                    ctl[2] = NB - 1 - (uint32_t)(16 * lane + j); ctl[4] = before; ctl[5] = loc[j];
                }
                before += loc[j];
            }
        }
        __syncthreads();
        const uint32_t bsel = ctl[2], above = ctl[4], inbin = ctl[5];
        cutoff = lo + (bsel << sh);
        if (sh == 0 || inbin <= 48u || pass + 1 >= max_passes) break;          // few enough around the kk-th (or single keys: only ties are left)
        need -= above;                               // the wanted entry is the need-th largest inside the bin
        const uint32_t top = cutoff + ((1u << sh) - 1u);
        lo = cutoff; hi = top < hi ? top : hi;
        __syncthreads();                             // (everybody has read ctl before the next pass clears it)
    }
#pragma unroll
    for (int j = 0; j < NE; ++j)
        if (ev[j] && (uint32_t)(e[j] >> 32) >= cutoff) {
            const uint32_t pos = atomicAdd(&ctl[3], 1u);
            if (pos < SCAP) sbuf[pos] = e[j];
        }
    __syncthreads();
    HDB_FIN_STAMP(10);
    const uint32_t ns = ctl[3];
    if (ns > SCAP || ns > (uint32_t)nth) {           // massive ties around the kk-th score: the general path sorts them (workgroup-uniform branch)
        __syncthreads();
        if (slotted) return 0u;                      // (no compact list to hand over: reported as a failed floor check, the caller's exact re-run sorts it out)
        if (have_floor_key) {
            float* fk = reinterpret_cast<float*>(reinterpret_cast<uint32_t*>(buf + 2 * (size_t)cap) + 2048 + 12);
            if (tid == 0) *fk = hdb_key2f(floor_key_in);
            __syncthreads();
            return hdb_finalize_body(buf, cand, total, q, cap, k, kk, row_base, idx_out, score_out, status, qnan_flag, extra_status, fk, 1.f, fix);
        }
        return hdb_finalize_body(buf, cand, total, q, cap, k, kk, row_base, idx_out, score_out, status, qnan_flag, extra_status, floor_ptr, floor_mul, fix);
    }
    const bool want_floor = floor_ptr != nullptr || have_floor_key;
    const uint32_t floor_key = have_floor_key ? floor_key_in : (floor_ptr ? hdb_f2key(hdb_canon(floor_v * floor_mul)) : 0u);
    const uint32_t nout = nc < kk ? nc : kk;
    for (uint32_t i = nout + tid; i < k; i += nth) { idx_out[(int64_t)q * k + i] = -1; score_out[(int64_t)q * k + i] = -INFINITY; }
    if ((uint32_t)tid < ns) {
        const unsigned long long mine = sbuf[tid];
        uint32_t rank = 0, j = 0;
        for (; j + 2 <= ns; j += 2) {
            const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(sbuf + j);
            rank += (pr.x > mine) + (pr.y > mine);
        }
        if (j < ns) rank += sbuf[j] > mine;
        if (rank < nout) {
            idx_out[(int64_t)q * k + rank] = row_base + (int64_t)(0xFFFFFFFFu - (uint32_t)(mine & 0xFFFFFFFFull));
            score_out[(int64_t)q * k + rank] = hdb_key2f((uint32_t)(mine >> 32));
        }
        if (want_floor && rank == kk - 1) ctl[8] = (uint32_t)(mine >> 32) > floor_key ? 1u : 0u;      // the kk-th best
    }
    if (status) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int32_t st = extra_status;
            if (total > cap) st |= HDB_Q_OVERFLOW;
            if (nc < kk) st |= HDB_Q_UNDERFLOW;
            if (want_floor && kk > 0 && ctl[8] == 0u) st |= HDB_Q_UNDERFLOW;      // (see hdb_finalize_body)
            if (qnan_flag) st |= HDB_Q_NAN;
            __hip_atomic_store(status + q, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t kth_above = 0;
    if (want_floor) { __syncthreads(); kth_above = ctl[8]; }
    return kth_above;
}

