// hdb_mfma_fused.h -- ONE launch for a whole hdb_topk call of 1..4 queries on an fp16 matrix (dot / cosine):
// query preparation, strided row sample, threshold, the filter pass over all rows and the final top-k, which the
// multi-kernel pipeline (hdb_api.hip) does in five dependent launches (qprep -> sample scan -> sample threshold ->
// filter scan -> finalize, ~45 us of fixed cost of which ~17 us are queue gaps between the launches).  The reference
// does all of it in one Python call per query (hyperdb/ranking_algorithm.py:149-204).
//
// Structure (persistent, one 512-thread workgroup per CU, the LDS ring of hdb_mfma_kernel.h):
//   prologue  every workgroup converts the float32 queries itself (fp16 copies scaled by a power of two into LDS,
//             1/||q||, NaN flag): 4 x d elements, cheaper than a launch.
//   phase A   workgroup b multiplies sample tiles b, b+G, ... of the strided, jittered row sample (same plan as the
//             multi-kernel path); wave 0 holds the queries as MFMA B fragments, its scores of each tile go to a small
//             LDS buffer; waves 1..nq ("selectors", otherwise only staging) keep the m largest sample scores of their
//             query in registers (m rounds of wave-max extraction per tile).
//   exchange  each selector publishes its m values as 8-byte {epoch, value} granules (one sc1 store per lane; the
//             data is its own flag) and sweeps the G x m granules of its query until every tag carries this call's
//             epoch: an all-gather without a grid barrier, ~3 us, during which the staging of the first tiles of
//             phase B is already in flight.  The m-th largest of the gathered values is the EXACT m-th largest
//             sample score: the per-query threshold, computed redundantly by every workgroup.
//   phase B   the filter pass over all rows (the only pass over V); wave 0 defers the epilogue of a tile by one
//             tile, so the threshold is first needed one tile after the exchange started.
//   finish    survivors go to the global candidate lists (atomics); each workgroup drains, releases and takes a
//             ticket; the LAST workgroup re-uses the ring LDS to pre-select, sort and write the k results (and the
//             status words) of every query, straight into the caller's (pinned host or device) buffers.
// Every spin is bounded by a wall-clock timeout (s_memrealtime): a workgroup that gives up publishes nothing harmful,
// filters with threshold +inf, and raises the abort word, which turns every status into HDB_Q_UNDERFLOW so that the
// host re-runs the call through the exact path (and resets the control block).  This only happens when the grid is
// not co-resident (another kernel holds CUs), never on an idle GPU: grid <= CU count, one workgroup fits per CU.
#pragma once
#include "hdb_mfma_kernel.h"
#include "hdb_finalize.h"

#define HDB_FUSED_MAXQ 4            // queries per fused call
#define HDB_FUSED_M 8               // sample order statistic (k <= 128)
#define HDB_FUSED_GRAN_PER_WG 32    // HDB_FUSED_MAXQ * HDB_FUSED_M granules per workgroup
#define HDB_FUSED_MAX_WG 1024


// LDS accesses in inline asm: hipcc cannot prove them disjoint from the ring that LDS-DMA writes and would drain the
// wave's in-flight staging (s_waitcnt vmcnt(0)) in front of every one of them.
__device__ __forceinline__ void hdb_lds_st32(unsigned int addr, float v) {
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void hdb_lds_st16(unsigned int addr, unsigned int v) {
    asm volatile("ds_write_b16 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void hdb_lds_st128(unsigned int addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ float hdb_lds_ld32(unsigned int addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

typedef __attribute__((address_space(1))) unsigned long long hdb_gu64;
typedef __attribute__((address_space(1))) unsigned int hdb_gu32;

template <int D, int R, int METRIC, bool HAS_BIAS>
__global__ __launch_bounds__(512) void hdb_mfma_fused_kernel(ScanArgs a, FusedArgs f, const float* __restrict__ aux0g) {
    using Shape = MfmaShape<16, _Float16>;
    using Vec = typename Shape::Vec;
    using Acc = typename Shape::Acc;
    constexpr int MF = 16;
    constexpr int ROWB = D * 2;
    constexpr int CPR = ROWB / 16;
    constexpr int CPS = Shape::CPS;
    constexpr int KS = CPR / CPS;
    constexpr int RT = R / MF;
    constexpr int STAGE = R * ROWB;
    constexpr int NG = R * CPR / 64 / 8;
    constexpr bool AUX0 = METRIC != 0;
    constexpr int NLOADA = NG;
    constexpr int NLOADB = NG + (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);
    constexpr int M = HDB_FUSED_M;
    static_assert(R % MF == 0 && R <= 64 && (R * CPR) % 512 == 0 && ROWB % 256 == 0, "tile geometry");
    static_assert(METRIC == 0 || METRIC == 1, "dot / cosine");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* auxbuf = reinterpret_cast<float*>(smem + 3 * STAGE);                    // [3 stages][2][64]
    unsigned long long* cb = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 3 * 2 * 64 * 4);
    unsigned short* cbq = reinterpret_cast<unsigned short*>(cb + HDB_MFMA_CB);
    unsigned int* ctl = reinterpret_cast<unsigned int*>(cbq + HDB_MFMA_CB);       // [0] count, [1..2] flush flags, [3] last-workgroup flag
    float* tsc = reinterpret_cast<float*>(ctl + 16);                               // [2][MAXQ][64] scores of the latest sample tiles
    float* qpar = tsc + 2 * HDB_FUSED_MAXQ * 64;                                   // [MAXQ] multiplier, [MAXQ] NaN flag, [MAXQ] threshold
    _Float16* qlds = reinterpret_cast<_Float16*>(smem + 2 * STAGE);               // prologue scratch: [nq][D] fp16 (ring slot 2, not yet in use)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & (MF - 1);
    const int h = lane / MF;
    const int nq = f.nq;
    const bool grpB = w >= 4;
    const bool mfma_wave = w == 0;                  // nq <= 16: one wave multiplies, the kernel is a streaming kernel
    const bool selector = w >= 1 && w <= nq;        // wave q+1 keeps the top-M sample scores of query q
    const int64_t G = gridDim.x;
    const int64_t b = blockIdx.x;

    if (tid < 16) ctl[tid] = 0;

    // ---- tile sequence: phase A (sample) then phase B (all rows) through ONE ring ------------------
    const char* const Vb = reinterpret_cast<const char*>(a.V);
    const int64_t n_rows = a.n;
    const int64_t nA = f.s_tiles > b ? (f.s_tiles - b + G - 1) / G : 0;
    const int64_t nB = a.ntiles > b ? (a.ntiles - b + G - 1) / G : 0;
    const int64_t total = nA + nB;
    auto row0_of = [&](int64_t i) -> int64_t {
        return i < nA ? hdb_tile_index(b + i * G, f.s_stride) * R : (b + (i - nA) * G) * (int64_t)R;
    };

    int g_off[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int slot = (w + 8 * j) * 64 + lane;
        const int r = slot / CPR, cpos = slot - r * CPR;
        g_off[j] = r * ROWB + (cpos ^ (r & 15)) * 16;
    }
    auto issue = [&](int64_t i, int st) {
        const int64_t row0 = row0_of(i);
        const int64_t last = n_rows - 1 - row0;
        char* sdst = smem + st * STAGE;
        const char* tile_base = Vb + row0 * (int64_t)ROWB;
        if (last >= R - 1) {
#pragma unroll
            for (int j = 0; j < NG; ++j)
                __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + (unsigned int)g_off[j]),
                                                 HDB_LDS_PTR(sdst + (w + 8 * j) * 1024), 16, 0, 2);
        } else {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int r = g_off[j] / ROWB;
                const int rr = r <= (int)last ? r : (int)last;
                const unsigned int off = (unsigned int)(g_off[j] + (rr - r) * ROWB);
                __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + (w + 8 * j) * 1024), 16, 0, 2);
            }
        }
        if ((AUX0 || HAS_BIAS) && grpB) {
            const int64_t rr = lane <= last ? lane : last;
            if (AUX0) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(aux0g + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 0) * 64), 4, 0, 0);
            if (HAS_BIAS) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 1) * 64), 4, 0, 0);
        }
    };
    // ---- query preparation (hdb_qprep_kernel, per workgroup): wave q prepares query q.  The query loads are issued
    // BEFORE the first staging so that waiting for them does not wait for the tiles (vmcnt retires in order).
    constexpr int QPL = (D + 63) / 64;
    float qreg[QPL];
    if (w < nq) {
        const float* qv = f.Qraw + (int64_t)w * D;
#pragma unroll
        for (int u = 0; u < QPL; ++u) { const int e = lane + 64 * u; qreg[u] = e < D ? qv[e] : 0.f; }
    }
    if (total > 0) issue(0, 0);
    if (total > 1) issue(1, 1);
    const unsigned int tsc_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(tsc);
    const unsigned int qpar_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(qpar);
    const unsigned int qlds_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(qlds);
    if (w < nq) {
        float ss = 0.f, amax = 0.f;
#pragma unroll
        for (int u = 0; u < QPL; ++u) { ss += qreg[u] * qreg[u]; amax = fmaxf(amax, fabsf(qreg[u])); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); amax = fmaxf(amax, __shfl_xor(amax, o, 64)); }
        const float scale = hdb_q16_scale(amax);
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            const int e = lane + 64 * u;
            const _Float16 hv = (_Float16)(qreg[u] * scale);
            if (e < D) hdb_lds_st16(qlds_addr + (unsigned int)(w * D + e) * 2u, (unsigned int)__builtin_bit_cast(unsigned short, hv));
        }
        if (lane == 0) {
            const float qinv = (ss == 0.f) ? 1.0f : 1.0f / sqrtf(ss);
            hdb_lds_st32(qpar_addr + (unsigned int)w * 4u, (METRIC == 1 ? qinv : 1.0f) * (1.f / scale));
            hdb_lds_st32(qpar_addr + (unsigned int)(HDB_FUSED_MAXQ + w) * 4u, (ss != ss) ? 1.f : 0.f);
        }
    }
    hdb_lds_barrier();

    // ---- wave 0: B fragments and per-query constants ---------------------------------------------
    const bool q_ok = rl < nq;
    Vec Bq[KS];
    float qinv_l = 1.f;
    if (mfma_wave) {
        const unsigned int src = qlds_addr + (unsigned int)((q_ok ? rl : 0) * D) * 2u + (unsigned int)h * 16u;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Vec v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(src), "i"(CPS * 16 * s));
            Bq[s] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("" : "+v"(Bq[s]));                  // values are final only behind the wait above
            if (!q_ok) Bq[s] = Vec{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (q_ok) qinv_l = hdb_lds_ld32(qpar_addr + (unsigned int)rl * 4u);
    }

    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    const unsigned int ctl_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(ctl);
    const unsigned int cb_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb);
    const unsigned int cbq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq);
    const unsigned int rd_base = (unsigned int)(rl * CPR * 16);
    const unsigned int hx = (unsigned int)((h ^ (rl & 15)) << 4);

    auto flush = [&]() {
        hdb_lds_barrier();
        const unsigned int ne = ctl[0] < HDB_MFMA_CB ? ctl[0] : HDB_MFMA_CB;
        for (unsigned int e = tid; e < ne; e += 512) {
            const unsigned long long ent = cb[e];
            const unsigned int qe = cbq[e];
            const unsigned int pos = atomicAdd(&f.ctl[2 + qe], 1u);
            if (pos < f.cap) f.cand[(int64_t)qe * f.cap + pos] = ent;
        }
        hdb_lds_barrier();
        if (tid == 0) ctl[0] = 0;
        hdb_lds_barrier();
    };

    // Comparison domain of the filter == domain of the sample scores (identical arithmetic in both phases): the raw
    // dot (x 1/||v|| for cosine) without bias, the finished score with bias.
    float thr_cmp = INFINITY;
    auto filter = [&](const Acc (&tv)[RT], int64_t row0) {
        float gm[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) gm[rt] = fmaxf(fmaxf(tv[rt][0], tv[rt][1]), fmaxf(tv[rt][2], tv[rt][3]));
        float m = gm[0];
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) m = fmaxf(m, gm[rt]);
        if (m >= thr_cmp) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (gm[rt] >= thr_cmp) {
                    const int64_t rowg = row0 + rt * 16 + 4 * h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = tv[rt][j];
                        if (x >= thr_cmp && rowg + j < n_rows && !(HAS_BIAS && x == -INFINITY)) {
                            const float sc = hdb_canon(HAS_BIAS ? x : x * qinv_l);
                            unsigned int pos;
                            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&v"(pos) : "v"(ctl_addr), "v"(1u) : "memory");
                            if (pos < HDB_MFMA_CB) {
                                const unsigned long long ent = hdb_pack(sc, (uint32_t)(rowg + j));
                                asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                             :: "v"(cb_addr + pos * 8u), "v"(ent), "v"(cbq_addr + pos * 2u), "v"((unsigned int)rl) : "memory");
                            } else {
                                const unsigned int gpos = atomicAdd(&f.ctl[2 + rl], 1u);
                                if (gpos < f.cap) f.cand[(int64_t)rl * f.cap + gpos] = hdb_pack(sc, (uint32_t)(rowg + j));
                            }
                        }
                    }
                }
            }
        }
    };

    // ---- selector state: the M largest sample scores of this wave's query, as orderable keys, lane r < M holds one
    uint32_t keep = 0u;                              // key 0 sorts below every float, -inf included
    auto merge_tile = [&](unsigned int src64_addr) {      // 64 new scores, one per lane
        uint32_t cur = hdb_f2key(hdb_canon(hdb_lds_ld32(src64_addr + (unsigned int)lane * 4u)));
        uint32_t out = 0u;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            uint32_t v = max(cur, keep);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
            const unsigned long long who = __ballot(max(cur, keep) == v);
            const int first = (int)__ffsll((long long)who) - 1;
            if (lane == first) { if (cur >= keep) cur = 0u; else keep = 0u; }
            if (lane == r) out = v;
        }
        keep = out;
    };
    auto spin_expired = [&](unsigned long long t0) {
        return (unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)f.timeout_ticks;
    };

    const int chk_shift = a.ntiles >= 65536 ? 4 : 0;
    const int64_t chk_mask = (1 << chk_shift) - 1;
    Acc acc[RT];
    int64_t row0_prev = 0;
    bool have_prev = false;                          // wave 0: acc holds an unfiltered phase-B tile
    int st_cur = 0;
    for (int64_t i = 0; i <= total; ++i) {           // one extra round (i == total) drains the deferred work
        const bool tile = i < total;
        if (tile) {
            if (i + 1 >= total) hdb_wait_vmcnt<0>();
            else if (grpB) hdb_wait_vmcnt<NLOADB>();
            else hdb_wait_vmcnt<NLOADA>();
        }
        const bool chk = tile && i >= nA && ((i - nA) & chk_mask) == chk_mask;
        const int chk_slot = 1 + (int)(((i - nA) >> chk_shift) & 1);
        if (chk && tid == 0) ctl[chk_slot] = (ctl[0] >= HDB_MFMA_CB / 4) ? 1u : 0u;
        hdb_lds_barrier();                           // tile i is in LDS; everyone is done with tile i-1; tsc/qpar hand-offs
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;
        if (i + 2 < total) issue(i + 2, st_next2);
        if (chk && ctl[chk_slot]) flush();

        // ---- selectors: sample bookkeeping, one tile behind wave 0 -----------------------------------
        if (selector) {
            const int q = w - 1;
            if (i >= 1 && i <= nA) merge_tile(tsc_addr + (unsigned int)(((int)((i - 1) & 1) * HDB_FUSED_MAXQ + q) * 64) * 4u);
            if (i == nA) {
                // publish M granules {epoch, key}, then gather everybody's: the data is the flag (no barrier, no fence)
                hdb_gu64* mine = (hdb_gu64*)(f.gran + (b * HDB_FUSED_GRAN_PER_WG + q * M));
                if (lane < M) __hip_atomic_store(mine + lane, ((unsigned long long)f.epoch << 32) | keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // lane l sweeps granule (l % M) of workgroups l / M, l / M + 8, ...
                constexpr int WPL = 64 / M;                      // workgroups covered per sweep instruction
                const int gi = lane % M, wg0 = lane / M;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool failed = false;
                uint32_t top = 0u;
                for (;;) {
                    bool ok = true;
                    uint32_t lmax[M];                            // per lane: the M largest of ITS granules (sorted desc)
#pragma unroll
                    for (int r = 0; r < M; ++r) lmax[r] = 0u;
                    for (int64_t wg = wg0; wg < G; wg += WPL) {
                        const unsigned long long x = __hip_atomic_load((hdb_gu64*)(f.gran + (wg * HDB_FUSED_GRAN_PER_WG + q * M + gi)),
                                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (uint32_t)(x >> 32) == f.epoch;
                        uint32_t v = (uint32_t)x;
#pragma unroll
                        for (int r = 0; r < M; ++r) { const uint32_t hi = max(lmax[r], v); v = min(lmax[r], v); lmax[r] = hi; }
                    }
                    if (__all(ok)) {
                        // M rounds: the wave-wide maximum of the lane heads; its owner pops its head
                        uint32_t kth = 0u;
#pragma unroll
                        for (int r = 0; r < M; ++r) {
                            uint32_t v = lmax[0];
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
                            const unsigned long long who = __ballot(lmax[0] == v);
                            if (lane == (int)__ffsll((long long)who) - 1) {
#pragma unroll
                                for (int t = 0; t + 1 < M; ++t) lmax[t] = lmax[t + 1];
                                lmax[M - 1] = 0u;
                            }
                            kth = v;
                        }
                        top = kth;
                        break;
                    }
                    if (spin_expired(t0)) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (lane == 0) {
                    const float thr = failed ? INFINITY : (top == 0u ? -INFINITY : hdb_key2f(top));
                    hdb_lds_st32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + q) * 4u, thr);
                    if (b == 0) f.thr_out[q] = thr;
                    if (failed) atomicOr(&f.ctl[1], 1u);
                }
            }
        }

        // ---- wave 0: multiply tile i; epilogue of a sample tile at once, of a filter tile one round later ----
        if (mfma_wave) {
            if (have_prev) {                         // deferred epilogue of tile i-1 (phase B)
                if (i == nA + 1) {                   // published by the selectors before this round's barrier
                    const float t = hdb_lds_ld32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + (q_ok ? rl : 0)) * 4u);
                    thr_cmp = q_ok ? t : INFINITY;
                }
                filter(acc, row0_prev);
                have_prev = false;
            }
            if (tile) {
                const int64_t row0 = row0_of(i);
                const unsigned int sb_addr = smem_addr + (unsigned int)(st_cur * STAGE) + rd_base;
                constexpr int PF = 3;
                Vec abuf[PF + 1][RT];
                auto fetch = [&](int s, Vec (&dst)[RT]) {
                    const unsigned int ad = sb_addr + ((unsigned int)(16 * CPS * s) ^ hx);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                    if constexpr (RT > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "i"(MF * CPR * 16));
                    if constexpr (RT > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(ad), "i"(2 * MF * CPR * 16));
                    if constexpr (RT > 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[3]) : "v"(ad), "i"(3 * MF * CPR * 16));
                };
                auto wait_frag = [&](int pend, Vec (&fr)[RT]) {
                    static_assert(RT == 1 || RT == 2 || RT == 4, "RT");
#define HDB_WAITF(N)                                                                                                  \
                    do {                                                                                               \
                        if constexpr (RT == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fr[0]));                 \
                        else if constexpr (RT == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fr[0]), "+v"(fr[1])); \
                        else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3])); \
                    } while (0)
                    const int cnt = pend * RT;
                    if (cnt >= 12) HDB_WAITF(12); else if (cnt == 8) HDB_WAITF(8); else if (cnt == 6) HDB_WAITF(6);
                    else if (cnt == 4) HDB_WAITF(4); else if (cnt == 3) HDB_WAITF(3); else if (cnt == 2) HDB_WAITF(2);
                    else if (cnt == 1) HDB_WAITF(1); else HDB_WAITF(0);
#undef HDB_WAITF
                };
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
#pragma unroll
                for (int s = 0; s < PF && s < KS; ++s) fetch(s, abuf[s % (PF + 1)]);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + PF < KS) fetch(s + PF, abuf[(s + PF) % (PF + 1)]);
                    const int pend = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;
                    wait_frag(pend, abuf[s % (PF + 1)]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = Shape::mma(abuf[s % (PF + 1)][rt], Bq[s], acc[rt]);
                }
                // comparable values, in place
                if (METRIC != 0 || HAS_BIAS) {
                    const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
                    const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int rl0 = rt * 16 + 4 * h;
                        float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (AUX0) av = *reinterpret_cast<const float4*>(ax0 + rl0);
                        if (HAS_BIAS) bv = *reinterpret_cast<const float4*>(ax1 + rl0);
                        const float aj[4] = {av.x, av.y, av.z, av.w};
                        const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float dot = acc[rt][j];
                            const float raw = METRIC == 1 ? dot * aj[j] : dot;
                            acc[rt][j] = HAS_BIAS ? fmaf(raw, qinv_l, bj[j]) : raw;
                        }
                    }
                }
                if (i < nA) {                        // sample tile: scores to the selectors (read after the next barrier)
                    if (q_ok) {
                        const unsigned int dst = tsc_addr + (unsigned int)(((int)(i & 1) * HDB_FUSED_MAXQ + rl) * 64) * 4u;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) hdb_lds_st128(dst + (unsigned int)(rt * 16 + 4 * h) * 4u, acc[rt]);
                        if (R < 64)                  // shorter tiles: the rest of the 64-entry row is -inf
                            for (int e = R + 4 * h; e < 64; e += 16) hdb_lds_st128(dst + (unsigned int)e * 4u, f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY});
                    }
                } else {
                    row0_prev = row0;
                    have_prev = true;
                }
            }
        }
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
    }
    flush();

    // ---- finish: drain, release, ticket; the last workgroup finalizes every query -------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket = __hip_atomic_fetch_add(f.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == (unsigned int)G - 1u;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ctl[3] = last ? 1u : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (ctl[3]) {
        unsigned long long* fbuf = reinterpret_cast<unsigned long long*>(smem);     // the ring is free now
        const unsigned int aborted = __hip_atomic_load(f.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int qnan_bits = 0u;
#pragma unroll
        for (int q = 0; q < HDB_FUSED_MAXQ; ++q) if (q < nq && qpar[HDB_FUSED_MAXQ + q] != 0.f) qnan_bits |= 1u << q;
        __syncthreads();                             // everyone has its copy: fbuf may now cover ctl / qpar
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            const uint32_t tot = __hip_atomic_load(f.ctl + 2 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hdb_finalize_body(fbuf, f.cand + (int64_t)q * f.cap, aborted ? 0u : tot, q, f.cap, f.k, f.kk, f.row_base, f.idx_out, f.score_out,
                              f.status, (int)((qnan_bits >> q) & 1u), 0);
            __syncthreads();
        }
        if (tid < 2 + HDB_FUSED_MAXQ) __hip_atomic_store(f.ctl + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next call
    }
}

static size_t fused_lds_bytes(int stage_bytes) {
    const size_t scan = mfma_lds_bytes(stage_bytes) + 2 * HDB_FUSED_MAXQ * 64 * 4 + 3 * HDB_FUSED_MAXQ * 4 + 64;
    const size_t fin = (size_t)HDB_CAND_CAP * 16 + 2048 * 4 + 64;           // hdb_finalize_body in the last workgroup
    return scan > fin ? scan : fin;
}

template <int D, int R, int METRIC, bool HAS_BIAS>
static int launch_fused_one(const ScanArgs& a, const FusedArgs& f, const float* aux0, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_fused_kernel<D, R, METRIC, HAS_BIAS>;
    const size_t lds = fused_lds_bytes(R * D * 2);
    static unsigned long long attr_done = 0;
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, a, f, aux0);
    return (int)hipGetLastError();
}

template <int D, int R>
static int launch_fused(const ScanArgs& a, const FusedArgs& f, int blocks, hipStream_t st) {
    const bool bias = a.bias != nullptr;
    if (a.metric == HDB_DOT) return bias ? launch_fused_one<D, R, 0, true>(a, f, nullptr, blocks, st) : launch_fused_one<D, R, 0, false>(a, f, nullptr, blocks, st);
    if (a.metric == HDB_COSINE) return bias ? launch_fused_one<D, R, 1, true>(a, f, a.inv_norm, blocks, st) : launch_fused_one<D, R, 1, false>(a, f, a.inv_norm, blocks, st);
    return (int)hipErrorNotSupported;
}
