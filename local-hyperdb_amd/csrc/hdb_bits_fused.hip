// hdb_bits_fused.hip -- ONE launch for a whole hdb_topk call of 1..4 hamming / jaccard queries.
//
// Replaces, for the bit metrics (reference hyperdb/ranking_algorithm.py:63-75, :116-147, then :194-200), the six dependent
// launches of the multi-kernel pipeline (query prep, query sign bits, sample scan, sample threshold, filter scan, finalize:
// ~45 us of fixed cost around a ~70 us scan of the 480 MB of sign bits at N = 10M, d = 384).
//
// One persistent workgroup of 1024 threads per CU (16 waves: every thread keeps W 16-byte loads in flight); the scan itself is
// hdb_hamming_kernel's: sign bits in 256-row blocks, word-major inside (hdb_bits_word), a thread owns four consecutive rows and
// all QH queries of the call; a wave step (64 quads) reads one contiguous block of W KiB.
//   prologue  every workgroup packs the sign bits of the queries (x > 0, hdb_qsign_kernel) into LDS and notes a NaN;
//   sample    the strided, jittered row sample of the multi-kernel pipeline, split over the grid; every thread keeps the two
//             largest scores it has seen per query; the workgroup reduces them to its eight largest per query (DPP wave
//             maxima, then wave 0) and publishes them as {epoch, key} granules [query][workgroup][8];
//   exchange  the owner of query q (workgroup q) sweeps that query's G x 8 granules until all carry this call's epoch, takes
//             the 8-th largest -- a lower bound of the 8-th largest sample score, which is all a threshold needs -- and
//             publishes it as one {epoch, key} word; every workgroup polls the nq words;
//   filter    the pass over all sign bits, handed out in two levels (1024-quad chunks from a global counter, 64-quad pieces from
//             an LDS word) and begun before the threshold is there; rows at or above it wait in LDS (the final sort's buffers are
//             idle until then) and go to the per-query candidate lists with one atomic per query at the end;
//   finish    drain, agent-scope release, arrive; when every workgroup has arrived, the owner of each query sorts its list
//             (hdb_finalize_body) and writes the k results and the status word; the last workgroup out zeroes the counters.
// Round 4 (BitsArgs::local, the default): no row sample and no exchange.  Every wave scores its first piece; the 8-th best of the
// workgroup's first 4096 rows is that workgroup's filter threshold b0; at the end of the pass the 8-th best b_w of everything it
// collected (>= b0) is taken and ONLY the rows at or above b_w -- all of them, ties included -- go to the global lists, with b_w in
// the granule area.  A row that was not emitted scores below its workgroup's b_w, so the union holds the global top-k (under the
// build's order: score descending, row ascending) as soon as every b_w is <= the k-th best score of the union; the owner of a
// query checks that and reports UNDERFLOW otherwise (host: exact re-run).  This drops the sample pass (its tiles were read twice:
// +6 % HBM traffic at N = 10M), the publish / sweep / poll round trips (~8 us) and every cross-workgroup dependency before the
// arrival counter.  P(a workgroup holds 8 of the 99 best rows of random data) = 2e-6 per call at 256 workgroups.
// Scores are small integers (hamming) with massive ties: everything at the threshold's own level survives, the sample plan aims
// lower for that (sample_plan's `coarse`), and an overflowing list comes back as HDB_Q_OVERFLOW for the exact path.
// Every spin is bounded (s_memrealtime); a workgroup that gives up raises the abort word: statuses become HDB_Q_UNDERFLOW and
// the host re-runs the call through the exact selection.
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"
#include "hdb_finalize.h"


// Diagnostic build only (tools/stamps_bits.py; product: 0): s_memrealtime stamps of the phases, per workgroup, in a buffer nothing reads.
#ifndef HDB_BITS_STAMPS
#define HDB_BITS_STAMPS 0
#endif
#if HDB_BITS_STAMPS
static __device__ unsigned long long hdb_bits_stamps[16 * 1024];
#define HDB_XSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024) hdb_bits_stamps[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int hdb_debug_read_bits_stamps(unsigned long long* host_out, int wgs) {
    if (wgs > 1024) wgs = 1024;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hdb_bits_stamps), (size_t)wgs * 16 * sizeof(unsigned long long));
}
#else
#define HDB_XSTAMP(slot) do { } while (0)
#endif
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long hdb_bgu64;

#ifndef HDB_BITS_THREADS
#define HDB_BITS_THREADS 1024
#endif
#ifndef HDB_BITS_TAIL
#define HDB_BITS_TAIL 1                  // the last ~15 % of the pass is handed out in quarter chunks (tools/exp_bits_variants.py: 5M rows 83 -> 78 us,
                                         // four queries -2..3 %; 8 waves per workgroup instead of 16: +5..20 %)
#endif
#ifndef HDB_BITS_TAILDIV
#define HDB_BITS_TAILDIV 4               // tail chunks = a chunk / this
#endif
#ifndef HDB_BITS_TAILPCT
#define HDB_BITS_TAILPCT 15              // share of the pass handed out in tail chunks
#endif
#ifndef HDB_BITS_PREF
#define HDB_BITS_PREF 1                  // the next chunk is asked for when draw number 1 (1) or np / 2 (2) of the current one is taken
#endif
#ifndef HDB_BITS_POLL
#define HDB_BITS_POLL 0                  // how the threshold words and the arrival counter are polled: 0 = sc1 loads, 1 = returning atomics (or 0)
#endif
#define HDB_BITS_WAVES (HDB_BITS_THREADS / 64)
#define HDB_BITS_MAXW 512

// The eight largest of the (up to two, a >= b) keys every thread brings: each wave extracts its eight largest (DPP maxima),
// wave 0 the eight largest of those.  Returns, in lane r < 8 of wave 0, the r-th largest (0 elsewhere).  scratch: waves x 8 words.
// PW < 8: every wave brings only its PW largest -- the result is then the 8-th largest of those, a LOWER BOUND of the true 8-th
// largest (all a threshold needs) at a fraction of the extraction rounds.
template <int PW = 8>
__device__ __forceinline__ uint32_t hdb_wg_top8(uint32_t a, uint32_t b, uint32_t* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (PW < 8 && lane < 8) scratch[w * 8 + lane] = 0u;
#pragma unroll
    for (int r = 0; r < PW; ++r) {
        const uint32_t v = hdb_wave_max_dpp(a);
        const unsigned long long who = __ballot(a == v);
        if (lane == (int)__ffsll((long long)who) - 1) { a = b; b = 0u; }
        if (lane == r) scratch[w * 8 + r] = v;
    }
    __syncthreads();
    uint32_t out = 0u;
    if (w == 0) {
        uint32_t x = scratch[lane], y = HDB_BITS_WAVES > 8 ? scratch[64 + lane] : 0u;          // waves x 8 keys
        uint32_t hi = max(x, y), lo = min(x, y);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t v = hdb_wave_max_dpp(hi);
            const unsigned long long who = __ballot(hi == v);
            if (lane == (int)__ffsll((long long)who) - 1) { hi = lo; lo = 0u; }
            if (lane == r) out = v;
        }
    }
    __syncthreads();
    return out;
}

template <bool JACCARD, int QH, bool NT>
__global__ __launch_bounds__(HDB_BITS_THREADS) void hdb_bits_fused_kernel(BitsArgs a) {
    static_assert(HDB_BITS_WAVES == 16 || HDB_BITS_WAVES == 8, "hdb_wg_top8 reads 64 or 128 keys");
    extern __shared__ __attribute__((aligned(16))) unsigned long long fbuf[];       // hdb_finalize_body's buffers first ...
    char* xbase = reinterpret_cast<char*>(fbuf) + (size_t)HDB_CAND_CAP * 16 + 2048 * 4 + 64;
    uint32_t* qb = reinterpret_cast<uint32_t*>(xbase);                                // ... then [QH][HDB_BITS_MAXW] query sign bits,
    uint32_t* scratch = qb + QH * HDB_BITS_MAXW;                                      // [128] reduction scratch,
    float* xthr = reinterpret_cast<float*>(scratch + 128);                            // [QH] thresholds,
    uint32_t* xflag = reinterpret_cast<uint32_t*>(xthr + QH);                         // [0..3] query holds a NaN, [4] vote, [5] time up, [6] aborted, [7] last out, [8..9] chunk hand-over, [10..13] survivors kept in LDS per query, [14] flush base

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t G = gridDim.x, b = blockIdx.x;
    const int nq = a.nq, W = a.W;
    hdb_bgu64* const thrw = (hdb_bgu64*)(reinterpret_cast<char*>(a.ctl) + HDB_BATCH_THRW_BYTE);
    uint32_t* const gcnt = a.ctl + HDB_BATCH_CTL_CNT;
    auto expired = [&](unsigned long long t0) {
        return (unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)a.timeout_ticks;
    };

    HDB_XSTAMP(0);
    // ---- prologue: sign bits of the queries (hdb_qsign_kernel), NaN flags
    if (tid < 16) xflag[tid] = tid == 8 ? (uint32_t)b << 6 : 0u;          // [8]: the first chunk of the filter pass is this workgroup's own; [10..13]: survivors kept in LDS per query
    for (int i = tid; i < QH * HDB_BITS_MAXW; i += HDB_BITS_THREADS) qb[i] = 0u;
    __syncthreads();
    {
        const int chunks = (a.d + 63) / 64;
        for (int job = w; job < nq * chunks; job += HDB_BITS_WAVES) {
            const int qq = job / chunks, e0 = (job - qq * chunks) * 64;
            const int e = e0 + lane;
            const float x = e < a.d ? a.Qraw[(int64_t)qq * a.d + e] : 0.f;
            const unsigned long long m = __ballot(e < a.d && x > 0.f);
            const unsigned long long nn = __ballot(x != x);
            if (lane == 0) {
                qb[qq * HDB_BITS_MAXW + (e0 >> 5)] = (uint32_t)m;
                if (e0 + 32 < a.d) qb[qq * HDB_BITS_MAXW + (e0 >> 5) + 1] = (uint32_t)(m >> 32);
                if (nn) atomicOr(&xflag[qq], 1u);
            }
        }
    }
    __syncthreads();

    // score of the four rows of quad i for every query of the call -> s[qq][u] (final: bias added, masked rows -inf, NaN -> -inf)
    // score of the four rows of quad i for every query of the call -> s[qq][u] (bias added, NaN -> -inf); returns the mask of
    // rows that exist and are not filtered out.  Twelve 16-byte loads are in flight per thread and step (six with four queries:
    // 16 waves x 12 KiB per CU -- with six the pass ran at 4.8 TB/s, latency-bound).
    auto score_quad = [&](int64_t i, float (&s)[QH][4]) -> unsigned int {
        uint32_t mism[QH][4], uni[QH][4];
#pragma unroll
        for (int qq = 0; qq < QH; ++qq)
#pragma unroll
            for (int u = 0; u < 4; ++u) { mism[qq][u] = 0; uni[qq][u] = 0; }
        constexpr int CW = QH <= 2 ? 12 : 6;              // 16-byte loads in flight per thread and step (registers: 4 per load)
        for (int w0 = 0; w0 < W; w0 += CW) {
            uint4 v[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c)
                v[c] = (w0 + c < W) ? hdb_bits_load<NT>(a.bits, i, w0 + c, W) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const uint32_t vv[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
                for (int qq = 0; qq < QH; ++qq) {
                    const uint32_t qw = (w0 + c < W) ? qb[qq * HDB_BITS_MAXW + w0 + c] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (JACCARD) { mism[qq][u] += __popc(vv[u] & qw); uni[qq][u] += __popc(vv[u] | qw); }
                        else mism[qq][u] += __popc(vv[u] ^ qw);
                    }
                }
            }
        }
        unsigned int livebits = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = 4 * i + u;
            const bool exists = row < a.n;
            const bool live = exists && !(a.mask && !a.mask[exists ? row : 0]);
            const float bb = (a.bias && exists) ? a.bias[row] : 0.f;
            livebits |= live ? (1u << u) : 0u;
#pragma unroll
            for (int qq = 0; qq < QH; ++qq) {
                float sc = JACCARD ? (float)mism[qq][u] / (float)uni[qq][u] : (float)(a.d - (int)mism[qq][u]);
                if (a.bias) sc += bb;
                s[qq][u] = live ? hdb_canon(sc) : -INFINITY;
            }
        }
        return livebits;
    };

    HDB_XSTAMP(1);
    // ---- sample: the two largest scores per query this thread has seen
    uint32_t top0[QH], top1[QH];
#pragma unroll
    for (int qq = 0; qq < QH; ++qq) { top0[qq] = 0u; top1[qq] = 0u; }
    const bool local = a.local != 0;
    for (int64_t j = (int64_t)tid * G + b; !local && j < a.s_tiles * 4; j += G * HDB_BITS_THREADS) {      // (every workgroup takes a share: the sample is ~20 k quads)
        const int64_t i = hdb_tile_index(j >> 2, a.s_stride) * 4 + (j & 3);
        if (4 * i >= a.npad) continue;
        float s[QH][4];
        const unsigned int livebits = score_quad(i, s);
#pragma unroll
        for (int qq = 0; qq < QH; ++qq)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t key = ((livebits >> u) & 1u) ? hdb_f2key(s[qq][u]) : 0u;
                const uint32_t lo = min(top0[qq], key);
                top0[qq] = max(top0[qq], key);
                top1[qq] = max(top1[qq], lo);
            }
    }
    HDB_XSTAMP(2);
    // ---- publish this workgroup's eight largest per query: granules [query][workgroup][8] x {epoch, key}
#pragma unroll
    for (int qq = 0; qq < QH; ++qq) {
        if (qq < nq && !local) {
            const uint32_t r8 = hdb_wg_top8(top0[qq], top1[qq], scratch);
            if (w == 0) {
                const uint32_t k0 = (uint32_t)__shfl((int)r8, 2 * (lane & 3), 64), k1 = (uint32_t)__shfl((int)r8, 2 * (lane & 3) + 1, 64);
                if (lane < 4) {
                    const char* dst = reinterpret_cast<const char*>(a.ctl) + HDB_BATCH_GRAN_BYTE + ((int64_t)qq * G + b) * 64 + lane * 16;
                    const u32x4b v = {k0, a.epoch, k1, a.epoch};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
                }
            }
        }
    }
    const unsigned long long x_t0 = __builtin_amdgcn_s_memrealtime();
    HDB_XSTAMP(3);
    // ---- owners: workgroup q sweeps the G x 8 granules of query q (NG / 2 pairs of granules, one 16-byte load per thread and step)
    for (int q = (int)b; !local && q < nq; q += (int)G) {
        const char* srcb = reinterpret_cast<const char*>(a.ctl) + HDB_BATCH_GRAN_BYTE + (int64_t)q * G * 64;
        const int NP = (int)G * 4;
        bool gave_up = false;
        uint32_t ka = 0u, kb = 0u;
        for (;;) {
            bool ok = true;
            ka = 0u; kb = 0u;
            for (int i0 = tid; i0 < NP; i0 += HDB_BITS_THREADS) {
                u32x4b x;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(srcb + (int64_t)i0 * 16) : "memory");
                ok &= x.y == a.epoch && x.w == a.epoch;
                const uint32_t k0 = x.y == a.epoch ? x.x : 0u, k1 = x.w == a.epoch ? x.z : 0u;
                // keep the two largest of {ka, kb, k0, k1}
                const uint32_t m0 = max(k0, k1), m1 = min(k0, k1);
                const uint32_t hi = max(ka, m0), mid = max(min(ka, m0), max(kb, m1));
                ka = hi; kb = mid;
            }
            if (tid == 0) { xflag[4] = 0u; xflag[5] = 0u; }
            __syncthreads();
            if (!ok) xflag[4] = 1u;
            if (!ok && expired(x_t0)) xflag[5] = 1u;
            __syncthreads();
            if (xflag[4] == 0u) break;
            if (xflag[5] != 0u) { gave_up = true; break; }
            __syncthreads();
            __builtin_amdgcn_s_sleep(4);
        }
        // (a thread's third-largest is lost when G > 256 gives it more than one pair: a lower bound still)
        const uint32_t r8 = hdb_wg_top8(ka, kb, scratch);
        if (w == 0) {
            uint32_t kth = (uint32_t)__shfl((int)r8, 7, 64);
            if (kth <= 1u) kth = hdb_f2key(-INFINITY);                 // fewer than 8 sample values exist: no threshold
            if (gave_up) {
                kth = hdb_f2key(INFINITY);
                if (lane == 0) atomicOr(a.ctl + HDB_BATCH_CTL_ABORT, 1u);
            }
            if (lane == 0) __hip_atomic_store(thrw + q, ((unsigned long long)a.epoch << 32) | kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // ---- filter, first step: every wave draws its first piece of the pass and scores it BEFORE it looks for the threshold (the
    // exchange takes ~5 us; the scores wait in registers), owners after they have published theirs.
    // Hand-out in two levels: a chunk = 1024 quads (16 pieces of 64 quads = one wave step of 12 KiB at d = 384) comes from the global
    // counter, one returning atomic per chunk and workgroup, asked for while the chunk before is being worked on; the waves of the
    // workgroup draw pieces from an LDS word {chunk << 6 | draws: up to 32 per chunk, 16 of them pieces}: no barrier, and a wave that the memory system serves late
    // simply draws fewer pieces (static thread-interleaved split: wave 0 of the median workgroup was done 33 us before its last
    // wave, workgroups differed by 19 us, profiles/r3_bits_timeline.txt).
    uint32_t* const cq = xflag + 8;                              // [0] {chunk << 6 | draws}, [1] prefetched chunk + 1
    constexpr uint32_t ENDC = 0x03FFFFFFu;
    constexpr uint32_t NPC = HDB_BITS_WAVES, TP = NPC / HDB_BITS_TAILDIV;          // pieces per chunk, per quarter chunk of the tail
    const uint32_t npieces = (uint32_t)((a.ntiles * 4 + 63) / 64);
    uint32_t nbig = (npieces + NPC - 1) / NPC, nsmall = 0u;
    if (HDB_BITS_TAIL && nbig >= 4u * (uint32_t)G) {
        nbig = (uint32_t)((uint64_t)npieces * (100 - HDB_BITS_TAILPCT) / 100 / NPC);
        nsmall = (npieces - nbig * NPC + TP - 1) / TP;
    }
    const uint32_t nchunks = nbig + nsmall;
    auto draw = [&]() __attribute__((always_inline)) -> int64_t {        // wave-uniform: first quad of the piece, or -1 at the end
        for (;;) {
            uint32_t t = 0u;
            if (lane == 0) t = __hip_atomic_fetch_add(&cq[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            const uint32_t sub = t & 63u, chunk = t >> 6;
            if (chunk == ENDC) return -1;
            const uint32_t np = chunk < nbig ? NPC : TP;
            const int64_t p0 = chunk < nbig ? (int64_t)chunk * NPC : (int64_t)nbig * NPC + (int64_t)(chunk - nbig) * TP;
            if (sub == (HDB_BITS_PREF == 2 ? np / 2 : 1u) && lane == 0) {   // ask for the chunk after this one
                const uint32_t nx = (uint32_t)G + __hip_atomic_fetch_add(a.ctl + HDB_BATCH_CTL_TILE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&cq[1], nx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (sub < np) return (p0 + sub) * 64;
            if (sub == np) {                                               // this wave swaps the next chunk in and takes its piece 0
                uint32_t nx = 0u;
                if (lane == 0) {
                    while ((nx = __hip_atomic_load(&cq[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) __builtin_amdgcn_s_sleep(1);
                    __hip_atomic_store(&cq[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&cq[0], nx - 1u >= nchunks ? ENDC << 6 : (((nx - 1u) << 6) | 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx) - 1u;
                if (nx >= nchunks) return -1;
                return (nx < nbig ? (int64_t)nx * NPC : (int64_t)nbig * NPC + (int64_t)(nx - nbig) * TP) * 64;
            }
            if (lane == 0)                                                 // pieces all taken: wait for the swap
                while ((__hip_atomic_load(&cq[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 6) == chunk) __builtin_amdgcn_s_sleep(1);
        }
    };
    float s0[QH][4];
    unsigned int live0 = 0u;
    int64_t i0 = draw();
    if (i0 >= 0) {
        i0 += lane;
        if (4 * i0 < a.npad) live0 = score_quad(i0, s0);
    }
    // ---- local flavour: the threshold is the 8-th best of this workgroup's first pieces (fewer rows than that: none)
    if (local) {
#pragma unroll
        for (int qq = 0; qq < QH; ++qq) {
            if (qq < nq) {                                     // (workgroup-uniform)
                uint32_t t0 = 0u, t1 = 0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t key = ((live0 >> u) & 1u) ? hdb_f2key(s0[qq][u]) : 0u;
                    const uint32_t lo = min(t0, key);
                    t0 = max(t0, key); t1 = max(t1, lo);
                }
                const uint32_t r8 = hdb_wg_top8<3>(t0, t1, scratch);
                if (w == 0 && lane == 7) xthr[qq] = r8 ? hdb_key2f(r8) : -INFINITY;
            }
        }
    }
    // ---- everybody: thread t fetches the threshold word of query t
    if (!local && tid < nq) {
        unsigned long long v;
        for (;;) {
            if (HDB_BITS_POLL) v = __hip_atomic_fetch_or(thrw + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __hip_atomic_load(thrw + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(v >> 32) == a.epoch) break;
            if (expired(x_t0)) { v = hdb_f2key(INFINITY); atomicOr(a.ctl + HDB_BATCH_CTL_ABORT, 1u); break; }
            __builtin_amdgcn_s_sleep(4);
        }
        xthr[tid] = hdb_key2f((uint32_t)v);
    }
    __syncthreads();
    float thr[QH];
#pragma unroll
    for (int qq = 0; qq < QH; ++qq) thr[qq] = qq < nq ? xthr[qq] : INFINITY;

    HDB_XSTAMP(4);
    // ---- filter: rows at or above the threshold go to the per-query candidate lists
    uint32_t* const lcnt = xflag + 10;                          // [QH] survivors of this workgroup kept in LDS
    unsigned long long* const lbuf = fbuf;                      // [QH][LCAP], over the (still idle) buffers of the final sort
    constexpr uint32_t LCAP = (uint32_t)HDB_CAND_CAP * 2u / QH;
    auto keep = [&](int64_t i, const float (&s)[QH][4], unsigned int livebits) __attribute__((always_inline)) {
#pragma unroll
        for (int qq = 0; qq < QH; ++qq)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = 4 * i + u;
                if (qq < nq && ((livebits >> u) & 1u) && s[qq][u] >= thr[qq]) {
                    // survivors wait in LDS (the final sort's buffers are idle until everybody has arrived) and reach the global list
                    // with ONE atomic per query and workgroup: a returning atomic per survivor on the nq counters of one cache line
                    // is what a four-query call on 1.25M rows spent most of its time in (125 us for ~16 k survivors, 89 us for ~8 k)
                    const unsigned long long ent = hdb_pack(s[qq][u], (uint32_t)row);
                    const uint32_t lp = __hip_atomic_fetch_add(&lcnt[qq], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (lp < LCAP) lbuf[(uint32_t)qq * LCAP + lp] = ent;
                    else {
                        const uint32_t pos = atomicAdd(&gcnt[qq], 1u);
                        if (pos < a.cap) a.cand[(int64_t)qq * a.cap + pos] = ent;
                    }
                }
            }
    };
    if (live0) keep(i0, s0, live0);
    if (i0 >= 0)
        for (;;) {
            const int64_t p = draw();
            if (p < 0) break;
            const int64_t i = p + lane;
            if (4 * i < a.npad) {                         // (no `continue`: draw() is a wave-level step)
                float s[QH][4];
                const unsigned int livebits = score_quad(i, s);
                keep(i, s, livebits);
            }
        }

    HDB_XSTAMP(5);
    // ---- finish: flush the survivors kept in LDS (one atomic per query), drain, release, arrive; wait for everybody; owners sort
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < QH; ++qq) {
        if (qq < nq) {
            const uint32_t have = min(lcnt[qq], LCAP);
            if (local && have <= 24u) {
                // a short list goes out as it is: everything at or above the FIRST threshold b0 was collected, so b0 is this workgroup's b_w
                if (tid == 0) {
                    xflag[14] = have ? atomicAdd(&gcnt[qq], have) : 0u;
                    __hip_atomic_store((hdb_bgu64*)(reinterpret_cast<char*>(a.ctl) + HDB_BATCH_GRAN_BYTE + ((int64_t)qq * G + b) * 64),
                                       (unsigned long long)(thr[qq] == -INFINITY ? 0u : hdb_f2key(thr[qq])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                const uint32_t base = xflag[14];
                for (uint32_t e = tid; e < have; e += HDB_BITS_THREADS)
                    if (base + e < a.cap) a.cand[(int64_t)qq * a.cap + base + e] = lbuf[(uint32_t)qq * LCAP + e];
                __syncthreads();
            } else if (local) {
                // b_w = the 8-th best of what this workgroup collected (every thread brings the two best of its share of the list;
                // a list longer than two entries per thread makes it a lower bound of the 8-th best, which is all the check needs)
                // (entry e goes to wave e mod 16: a list of a hundred entries must reach every wave, each of which brings three)
                uint32_t t0 = 0u, t1 = 0u;
                for (uint32_t e = (uint32_t)w + (uint32_t)HDB_BITS_WAVES * (uint32_t)lane; e < have; e += HDB_BITS_THREADS) {
                    const uint32_t key = (uint32_t)(lbuf[(uint32_t)qq * LCAP + e] >> 32);
                    const uint32_t lo = min(t0, key);
                    t0 = max(t0, key); t1 = max(t1, lo);
                }
                const uint32_t r8 = hdb_wg_top8<3>(t0, t1, scratch);
                if (tid == 7) { xflag[14] = r8; xflag[15] = 0u; }       // (lane 7 of wave 0: the 8-th largest; 0 = fewer than eight: everything goes)
                __syncthreads();
                const uint32_t bkey = xflag[14];
                uint32_t mine = 0u;
                for (uint32_t e = tid; e < have; e += HDB_BITS_THREADS) mine += (uint32_t)(lbuf[(uint32_t)qq * LCAP + e] >> 32) >= bkey ? 1u : 0u;
                if (mine) atomicAdd(&xflag[15], mine);
                __syncthreads();
                const uint32_t npass = xflag[15];
                if (tid == 0) {
                    xflag[14] = npass ? atomicAdd(&gcnt[qq], npass) : 0u;
                    xflag[15] = 0u;
                    // b_w for the owner's check (scores here are final scores: no domain to convert)
                    __hip_atomic_store((hdb_bgu64*)(reinterpret_cast<char*>(a.ctl) + HDB_BATCH_GRAN_BYTE + ((int64_t)qq * G + b) * 64),
                                       (unsigned long long)bkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                const uint32_t base = xflag[14];
                for (uint32_t e = tid; e < have; e += HDB_BITS_THREADS) {
                    const unsigned long long ent = lbuf[(uint32_t)qq * LCAP + e];
                    if ((uint32_t)(ent >> 32) >= bkey) {
                        const uint32_t pos = base + atomicAdd(&xflag[15], 1u);
                        if (pos < a.cap) a.cand[(int64_t)qq * a.cap + pos] = ent;
                    }
                }
                __syncthreads();
            } else if (have > 0u) {                              // (workgroup-uniform)
                if (tid == 0) xflag[14] = atomicAdd(&gcnt[qq], have);
                __syncthreads();
                const uint32_t base = xflag[14];
                for (uint32_t e = tid; e < have; e += HDB_BITS_THREADS)
                    if (base + e < a.cap) a.cand[(int64_t)qq * a.cap + base + e] = lbuf[(uint32_t)qq * LCAP + e];
                __syncthreads();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HDB_XSTAMP(6);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HDB_XSTAMP(9);
        __hip_atomic_fetch_add(a.ctl + HDB_BATCH_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        HDB_XSTAMP(10);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool all_in = true;
        // Only the owners of a query (workgroups 0 .. nq-1) wait for everybody: the others have nothing left to do, and 256 pollers
        // on the line of the arrival counter delayed the arrivals themselves by ~6 us (profiles/r4_bits_timeline.txt)
        while ((int64_t)b < (int64_t)nq && ((HDB_BITS_POLL ? __hip_atomic_fetch_or(a.ctl + HDB_BATCH_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : __hip_atomic_load(a.ctl + HDB_BATCH_CTL_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned int)G)) {
            if (expired(t0)) { all_in = false; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!all_in) atomicOr(a.ctl + HDB_BATCH_CTL_ABORT, 1u);
        const unsigned int ab = __hip_atomic_load(a.ctl + HDB_BATCH_CTL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xflag[6] = (ab != 0u || !all_in) ? 1u : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const bool aborted = xflag[6] != 0u;
    HDB_XSTAMP(7);
    for (int q = (int)b; q < nq; q += (int)G) {
        const uint32_t tot0 = __hip_atomic_load(gcnt + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int qn = (int)xflag[q];
        uint32_t floor_key = 0u;
        if (local) {
            // the highest b_w of any workgroup: the list is complete for every score >= it, so the top-k is exact iff the k-th best of
            // the union is >= it (the floor test is strict: one key below)
            uint32_t bk = 0u;
            for (int64_t wg = tid; wg < G; wg += HDB_BITS_THREADS)
                bk = max(bk, (uint32_t)__hip_atomic_load((hdb_bgu64*)(reinterpret_cast<char*>(a.ctl) + HDB_BATCH_GRAN_BYTE + ((int64_t)q * G + wg) * 64),
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            bk = hdb_wave_max_dpp(bk);
            if (lane == 0) scratch[w] = bk;
            __syncthreads();
            for (int w2 = 0; w2 < HDB_BITS_WAVES; ++w2) floor_key = max(floor_key, scratch[w2]);
            floor_key = floor_key ? floor_key - 1u : 0u;
            __syncthreads();
        }
        hdb_finalize_fast(fbuf, a.cand + (int64_t)q * a.cap, aborted ? 0u : tot0, q, a.cap, a.k, a.kk, a.row_base, a.idx_out, a.score_out,
                          a.status, qn, 0, nullptr, 1.f, HdbNoFix(), local, floor_key, nullptr, 0u, 0u, JACCARD ? 2 : 1);
        __syncthreads();
    }
    __syncthreads();
    HDB_XSTAMP(8);
    if (tid == 0) {
        const unsigned int left = __hip_atomic_fetch_add(a.ctl + HDB_BATCH_CTL_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xflag[7] = left == (unsigned int)G - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (xflag[7]) {
        if (tid < nq) __hip_atomic_store(a.ctl + HDB_BATCH_CTL_CNT + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(a.ctl + HDB_BATCH_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl + HDB_BATCH_CTL_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl + HDB_BATCH_CTL_TILE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl + HDB_BATCH_CTL_ABORT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int hdb_bits_fused_supported(int metric, int nq, int W, uint32_t kk) {
    return (metric == HDB_HAMMING || metric == HDB_JACCARD) && nq >= 1 && nq <= 4 && W <= HDB_BITS_MAXW && kk <= 128;
}

extern "C" int hdb_launch_bits_fused(const BitsArgs* args, int jaccard, int max_blocks, void* stream) {
    const BitsArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)HDB_CAND_CAP * 16 + 2048 * 4 + 64 + (size_t)4 * HDB_BITS_MAXW * 4 + 128 * 4 + 4 * 4 + 16 * 4 + 64;
    int blocks = hdb_cu_count();
    const int64_t items = a.ntiles * 4;
    if ((int64_t)blocks * HDB_BITS_THREADS > items) blocks = (int)((items + HDB_BITS_THREADS - 1) / HDB_BITS_THREADS);
    if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    // the local flavour needs every workgroup's share of the k best rows far below the 8 it emits at least: grids of 2 k workgroups
    // and more (k = 100: matrices of 820k rows and more; P(Poisson(0.5) >= 8) = 2e-7 per workgroup), else the exchange flavour
    BitsArgs a_eff = a;
    if ((int64_t)blocks < 2 * (int64_t)a.kk) a_eff.local = 0;
    const bool nt = (size_t)a.npad * a.W * 4 > ((size_t)256 << 20);      // the sign bits do not fit the Infinity Cache: stream them past it (5M x 384 = 240 MB still gain from it)
#define HDB_BITS_LAUNCH(JAC_, QH_)                                                                                          \
    do {                                                                                                                    \
        auto kern = nt ? hdb_bits_fused_kernel<JAC_, QH_, true> : hdb_bits_fused_kernel<JAC_, QH_, false>;                  \
        static unsigned long long attr_done[2] = {0, 0};                                                                    \
        hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done[nt ? 1 : 0]);            \
        if (e != hipSuccess) return (int)e;                                                                                 \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(HDB_BITS_THREADS), lds, st, a_eff);                                     \
    } while (0)
    if (a.nq == 1) { if (jaccard) HDB_BITS_LAUNCH(true, 1); else HDB_BITS_LAUNCH(false, 1); }
    else if (a.nq == 2) { if (jaccard) HDB_BITS_LAUNCH(true, 2); else HDB_BITS_LAUNCH(false, 2); }
    else { if (jaccard) HDB_BITS_LAUNCH(true, 4); else HDB_BITS_LAUNCH(false, 4); }
#undef HDB_BITS_LAUNCH
    return (int)hipGetLastError();
}
