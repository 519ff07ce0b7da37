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
//     to a small LDS list that is flushed to the per-query candidate lists with global atomics every few
//     hundred tiles.  The N x Q score matrix is never written (10 GB at N=10M, Q=256).
// Synchronisation: one raw s_barrier per tile; tile t+2 is in flight while tile t is multiplied (counted
// s_waitcnt vmcnt, never 0 in the steady state).
// Algorithmic bytes per row: d*2 (V read exactly once per pass); FLOPs: 2*Q*d per row.
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HDB_MFMA_CB 1024            // LDS candidate list entries per workgroup
#ifndef HDB_PF_QT2
#define HDB_PF_QT2 2
#endif

#define HDB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HDB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void hdb_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N < 0, "add this vmcnt immediate");
}
__device__ __forceinline__ void hdb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int MF> struct MfmaShape;
template <> struct MfmaShape<32> {
    using Acc = f32x16;
    static constexpr int KSTEP = 16, NGRP = 4;      // k per MFMA; groups of 4 consecutive rows per lane per tile
    __device__ static __forceinline__ Acc mma(half8 a, half8 b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaShape<16> {
    using Acc = f32x4;
    static constexpr int KSTEP = 32, NGRP = 1;
    __device__ static __forceinline__ Acc mma(half8 a, half8 b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// METRIC: 0 dot, 1 cosine (aux0 = 1/||v||), 2 euclidean similarity (aux0 = ||v||^2)
// Fragment maps (lane l):  MF=32: row/query l&31, k = 16s + 8(l>>5) + j, C reg e -> row (e&3) + 8(e>>2) + 4(l>>5)
//                          MF=16: row/query l&15, k = 32s + 8(l>>4) + j, C reg e -> row 4(l>>4) + e
template <int MF, int QT, int D, int R, int MODE, int METRIC, bool HAS_BIAS>
__global__ __launch_bounds__(512) void hdb_mfma_kernel(ScanArgs a, const _Float16* __restrict__ q16,
                                                       const float* __restrict__ aux0g, const float* __restrict__ qsq, const float* __restrict__ qscl,
                                                       int nq_end) {
    using Shape = MfmaShape<MF>;
    using Acc = typename Shape::Acc;
    constexpr int KSTEP = Shape::KSTEP, NGRP = Shape::NGRP;
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int KS = D / KSTEP;               // MFMA k-steps
    constexpr int CPS = KSTEP / 8;              // chunks per k-step (2 or 4)
    constexpr int RT = R / MF;                  // MFMA row tiles per stage
    constexpr int STAGE = R * D * 2;            // bytes of V per stage
    constexpr int NG = R * CPR / 64 / 8;        // LDS-DMA instructions per wave per tile
    constexpr bool AUX0 = METRIC != 0;
    constexpr int NLOADA = NG;
    constexpr int NLOADB = NG + (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);     // B waves also stage the per-row aux values
    constexpr int QPW = MF * QT;                // queries per wave (QT query tiles share every A fragment)
    static_assert(R % MF == 0 && R <= 64 && (R * CPR) % 512 == 0 && D % 128 == 0 && KS % NG == 0, "tile geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* auxbuf = reinterpret_cast<float*>(smem + 3 * STAGE);                    // [3 stages][2][64]
    unsigned long long* cb = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 3 * 2 * 64 * 4);
    unsigned short* cbq = reinterpret_cast<unsigned short*>(cb + HDB_MFMA_CB);
    unsigned int* ctl = reinterpret_cast<unsigned int*>(cbq + HDB_MFMA_CB);       // [0] count, [1..2] flush flags

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & (MF - 1);             // row of the A fragment == query of the B fragment
    const int h = lane / MF;                    // which 8-element k-chunk of the step (0..CPS-1)

    // ---- this wave's queries --------------------------------------------------------------------
    const int qw0 = a.q0 + blockIdx.y * (8 * QPW) + w * QPW;
    const bool wave_active = qw0 < nq_end;
    bool q_ok[QT];
    int ql[QT];
    half8 Bq[QT][KS];
    float thr_l[QT], qinv_l[QT], qsq_l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = qw0 + qt * MF + rl;
        q_ok[qt] = q < nq_end;
        ql[qt] = q - a.q0;
        const int qq = q_ok[qt] ? q : (nq_end - 1);
        const uint4* src = reinterpret_cast<const uint4*>(q16 + (int64_t)qq * D + 8 * h);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 v = src[CPS * s];
            if (!q_ok[qt]) v = make_uint4(0, 0, 0, 0);
            Bq[qt][s] = *reinterpret_cast<half8*>(&v);
        }
        thr_l[qt] = 0.f; qinv_l[qt] = 1.f; qsq_l[qt] = 0.f;
        if (q_ok[qt]) {
            if (MODE == 1) thr_l[qt] = a.thr[ql[qt]];
            // qscl = 2^-e: the fp16 queries were scaled by 2^e into [2^14, 2^15) (hdb_q16_scaled), undone here for free:
            // it rides on the per-query multiplier of dot / cosine and on the -2 of the euclidean expansion
            const float qs = qscl[q];
            qinv_l[qt] = METRIC == 1 ? a.qinv[q] * qs : qs;
            if (METRIC == 2) qsq_l[qt] = qsq[q];
        }
    }
    if (tid < 4) ctl[tid] = 0;

    // ---- roles ----------------------------------------------------------------------------------------
    // Waves 0-3 ("A") and 4-7 ("B") are the two waves of each SIMD.  B runs its threshold epilogue one tile
    // late so that the two waves of a SIMD do not reach MFMA phase, epilogue and barrier in lock-step.
    const bool grpB = w >= 4;
    // "heavy": all eight waves multiply (more than 4*MF queries in this pass), the matrix pipe is the bottleneck.
    // Then A stages its half of tile i+2 right after the barrier while B already multiplies, and B stages its
    // half after its MFMA phase while A finishes -- the two waves of a SIMD never issue LDS-DMA (~90 cycles of
    // blocked issue per 1 KiB piece) at the same time.  Otherwise (HBM-bound) everyone stages right away.
    const bool heavy = (nq_end - (a.q0 + (int)blockIdx.y * (8 * QPW))) > 4 * QPW;

    // ---- staging geometry (per lane, fixed for the whole kernel) -----------------------------------
    int g_off[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int slot = (w + 8 * j) * 64 + lane;
        const int r = slot / CPR, cpos = slot - r * CPR;
        g_off[j] = r * (D * 2) + (cpos ^ (r & 15)) * 16;   // source byte for this LDS slot (XOR swizzle of the chunk)
    }
    // loop-invariant scalars, read once (keeps kernel-argument loads out of the tile loop)
    const char* const Vb = reinterpret_cast<const char*>(a.V);
    const int64_t n_rows = a.n;
    const int64_t ntiles = a.ntiles;
    const int64_t tstride = a.tile_stride;                      // 1 = dense pass, > 1 = strided row sample
    const int64_t gstep = gridDim.x;
    const int64_t my_tiles = (ntiles > blockIdx.x) ? (ntiles - blockIdx.x + gstep - 1) / gstep : 0;

    // Stage tile number t (global tile index) into ring slot st: NG LDS-DMA pieces of 1 KiB per wave,
    // non-temporal (V is read once per pass by exactly one CU: +2-3 % on the HBM-bound shapes), plus the
    // per-row aux values (B waves only).  Only the last tile of the matrix can be ragged.
    auto issue = [&](int64_t t, int st) {
        const int64_t row0 = hdb_tile_index(t, tstride) * R;
        const int64_t last = n_rows - 1 - row0;          // >= 0
        char* sdst = smem + st * STAGE;
        const char* tile_base = Vb + row0 * (int64_t)(D * 2);          // wave-uniform
        if (last >= R - 1) {
#pragma unroll
            for (int j = 0; j < NG; ++j)
                __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + (unsigned int)g_off[j]),
                                                 HDB_LDS_PTR(sdst + (w + 8 * j) * 1024), 16, 0, 2);
        } else {                                                        // clamp rows past the end to the last row
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int r = g_off[j] / (D * 2);
                const int rr = r <= (int)last ? r : (int)last;
                const unsigned int off = (unsigned int)(g_off[j] + (rr - r) * (D * 2));
                __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + (w + 8 * j) * 1024), 16, 0, 2);
            }
        }
        if ((AUX0 || HAS_BIAS) && grpB) {
            const int64_t rr = lane <= last ? lane : last;
            if (AUX0) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(aux0g + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 0) * 64), 4, 0, 0);
            if (HAS_BIAS) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 1) * 64), 4, 0, 0);
        }
    };

    auto flush = [&]() {
        hdb_lds_barrier();
        const unsigned int ne = ctl[0] < HDB_MFMA_CB ? ctl[0] : HDB_MFMA_CB;
        for (unsigned int e = tid; e < ne; e += 512) {
            const unsigned long long ent = cb[e];
            const unsigned int qe = cbq[e];
            const unsigned int pos = atomicAdd(&a.cnt[qe], 1u);
            if (pos < a.cap) a.cand[(int64_t)qe * a.cap + pos] = ent;
        }
        hdb_lds_barrier();
        if (tid == 0) ctl[0] = 0;
        hdb_lds_barrier();
    };

    int64_t t_cur = blockIdx.x;                      // global index of the tile being multiplied
    if (my_tiles > 0) issue(t_cur, 0);
    if (my_tiles > 1) issue(t_cur + gstep, 1);

    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    const unsigned int ctl_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(ctl);
    const unsigned int cb_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb);
    const unsigned int cbq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq);
    // per-lane LDS read address: row rl of a row tile; chunk (CPS*s + h) ^ rx of k-step s is at byte
    // ((16*CPS*s) ^ hx) of the row image, hx = (h ^ rx) << 4  (CPS*s and h occupy disjoint bits)
    const unsigned int rd_base = (unsigned int)(rl * CPR * 16);
    const unsigned int hx = (unsigned int)((h ^ (rl & 15)) << 4);
    // first of the 4 consecutive tile rows this lane's accumulator group g holds
    auto grp_row = [&](int rt, int g) { return MF == 32 ? rt * 32 + 8 * g + 4 * h : rt * 16 + 4 * h; };

    // threshold in the domain the epilogue compares in (see below); +inf for padding lanes
    float thr_cmp[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        thr_cmp[qt] = INFINITY;
        if (MODE == 1 && q_ok[qt]) {
            if (METRIC != 2 && !HAS_BIAS) { const float tc = thr_l[qt] / qinv_l[qt]; thr_cmp[qt] = tc - fabsf(tc) * 1e-6f; }
            else thr_cmp[qt] = thr_l[qt];
        }
    }

    // Filter, second half: group maxima (v_max3) let the common no-hit case finish in ~25 VALU
    // instructions; survivors go to the workgroup's LDS list.  `tv` holds comparable values (below).
    auto filter1 = [&](const Acc (&tv)[RT], int64_t row0, const float thr_cmp, const float qinv_l, const int ql) {
        float gm[RT][NGRP];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < NGRP; ++g)
                gm[rt][g] = fmaxf(fmaxf(tv[rt][4 * g], tv[rt][4 * g + 1]), fmaxf(tv[rt][4 * g + 2], tv[rt][4 * g + 3]));
        float m = gm[0][0];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < NGRP; ++g) m = fmaxf(m, gm[rt][g]);
        if (m >= thr_cmp) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    if (gm[rt][g] >= thr_cmp) {
                        const int64_t rowg = row0 + grp_row(rt, g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = tv[rt][4 * g + j];
                            if (x >= thr_cmp && rowg + j < n_rows && !(HAS_BIAS && x == -INFINITY)) {   // bias -inf = masked row
                                const float sc = hdb_canon((METRIC != 2 && !HAS_BIAS) ? x * qinv_l : x);
                                // LDS ops in inline asm: hipcc would otherwise drain every in-flight LDS-DMA
                                // (s_waitcnt vmcnt(0)) before touching LDS it cannot prove disjoint from the ring.
                                unsigned int pos;
                                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                             : "=&v"(pos) : "v"(ctl_addr), "v"(1u) : "memory");
                                if (pos < HDB_MFMA_CB) {
                                    const unsigned long long ent = hdb_pack(sc, (uint32_t)(rowg + j));
                                    asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                                 :: "v"(cb_addr + pos * 8u), "v"(ent), "v"(cbq_addr + pos * 2u), "v"((unsigned int)ql) : "memory");
                                } else {
                                    // LDS list full (dense hits on a small matrix): append straight to the global list.
                                    // The returning atomic makes hipcc drain this wave's LDS-DMA here -- rare and only slow.
                                    const unsigned int gpos = atomicAdd(&a.cnt[ql], 1u);
                                    if (gpos < a.cap) a.cand[(int64_t)ql * a.cap + gpos] = hdb_pack(sc, (uint32_t)(rowg + j));
                                }
                            }
                        }
                    }
                }
            }
        }
    };

    auto filter = [&](const Acc (&tv)[QT][RT], int64_t row0) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) filter1(tv[qt], row0, thr_cmp[qt], qinv_l[qt], ql[qt]);
    };

    const int chk_shift = ntiles >= 65536 ? 4 : 0;
    const int64_t chk_mask = (1 << chk_shift) - 1;
    Acc acc[QT][RT];
    int64_t row0_prev = 0;
    int st_cur = 0;
    for (int64_t i = 0; i < my_tiles; ++i, t_cur += gstep) {
        if (i + 1 >= my_tiles) hdb_wait_vmcnt<0>();
        else if (grpB) hdb_wait_vmcnt<NLOADB>();
        else hdb_wait_vmcnt<NLOADA>();
        // the LDS candidate list is checked for a flush every 16 tiles on large matrices (a few hits per tile,
        // 1024 slots), every tile on small ones (hits per tile ~ T*Q/ntiles): the check costs two LDS round trips
        // on every wave's critical path
        // The decision word alternates between ctl[1] and ctl[2] from one check to the next: waves are at most one
        // barrier apart, so wave 0 cannot overwrite a decision that a slower wave has not read yet (a torn decision
        // would send only part of the workgroup into flush()'s barriers).
        const bool chk = MODE == 1 && (i & chk_mask) == chk_mask;
        const int chk_slot = 1 + (int)((i >> chk_shift) & 1);
        if (chk && tid == 0) ctl[chk_slot] = (ctl[0] >= HDB_MFMA_CB / 4) ? 1u : 0u;
        hdb_lds_barrier();                                   // tile i is in LDS; everyone is done with tile i-1
        // Stage the whole next-but-one tile right after the barrier, into the buffer tile i-1 used (measured:
        // 5.4-5.7 TB/s on the HBM-bound shapes vs 4.7-5.0 with the pieces spread between the MFMAs).
        const bool more = i + 2 < my_tiles;
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;
        if (more && !(heavy && grpB)) issue(t_cur + 2 * gstep, st_next2);
        if (chk && ctl[chk_slot]) flush();

        if (wave_active) {
            const int64_t row0 = hdb_tile_index(t_cur, tstride) * R;
            if (MODE == 1 && grpB && i > 0) filter(acc, row0_prev);        // deferred epilogue of tile i-1
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4 * NGRP; ++e) acc[qt][rt][e] = 0.f;

            // A fragments: LDS reads issued two k-steps ahead of the MFMAs that consume them.  The reads
            // and their counted waits are inline asm so that hipcc cannot sink a read next to its use
            // (it otherwise emits read, lgkmcnt(0), MFMA per step and exposes the LDS latency every step).
            // lgkmcnt(n*RT) = "all but the n*RT newest LDS ops are back" = the oldest pending step's fragments;
            // stray scalar loads can only make that wait longer, never shorter.
            const unsigned int sb_addr = smem_addr + (unsigned int)(st_cur * STAGE) + rd_base;
            constexpr int PF = HDB_PF_QT2 > 0 && QT == 2 ? HDB_PF_QT2 : 3;   // k-steps of LDS prefetch (PF+1 fragment sets)
            half8 abuf[PF + 1][RT];
            auto fetch = [&](int s, half8 (&dst)[RT]) {
                const unsigned int ad = sb_addr + ((unsigned int)(16 * CPS * s) ^ hx);
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                if constexpr (RT > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "i"(MF * CPR * 16));
                if constexpr (RT > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(ad), "i"(2 * MF * CPR * 16));
                if constexpr (RT > 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[3]) : "v"(ad), "i"(3 * MF * CPR * 16));
            };
            // wait until at most `pend` k-steps of fragment reads are outstanding: lgkmcnt(pend*RT)
            auto wait_frag = [&](int pend, half8 (&f)[RT]) {
                static_assert(RT == 1 || RT == 2 || RT == 4, "RT");
#define HDB_WAITF(N)                                                                                               \
                do {                                                                                               \
                    if constexpr (RT == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]));                  \
                    else if constexpr (RT == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1])); \
                    else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); \
                } while (0)
                const int cnt = pend * RT;
                if (cnt >= 12) HDB_WAITF(12); else if (cnt == 8) HDB_WAITF(8); else if (cnt == 6) HDB_WAITF(6);
                else if (cnt == 4) HDB_WAITF(4); else if (cnt == 3) HDB_WAITF(3); else if (cnt == 2) HDB_WAITF(2);
                else if (cnt == 1) HDB_WAITF(1); else HDB_WAITF(0);
#undef HDB_WAITF
            };
#pragma unroll
            for (int s = 0; s < PF && s < KS; ++s) fetch(s, abuf[s % (PF + 1)]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + PF < KS) fetch(s + PF, abuf[(s + PF) % (PF + 1)]);
                const int pend = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;       // steps still in flight behind step s
                wait_frag(pend, abuf[s % (PF + 1)]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) acc[qt][rt] = Shape::mma(abuf[s % (PF + 1)][rt], Bq[qt][s], acc[qt][rt]);
            }

            // ---- epilogue, first half: turn the dot products into the values that are stored (MODE 0) or
            // compared (MODE 1), in place.  Filter mode compares the score itself, except dot / cosine without
            // bias: the raw dot (dot/||v||) against thr divided by the per-query multiplier.
            const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
            const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
            if (METRIC != 0 || HAS_BIAS || MODE == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g = 0; g < NGRP; ++g) {
                        const int rl0 = grp_row(rt, g);
                        float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (AUX0) av = *reinterpret_cast<const float4*>(ax0 + rl0);
                        if (HAS_BIAS) bv = *reinterpret_cast<const float4*>(ax1 + rl0);
                        const float aj[4] = {av.x, av.y, av.z, av.w};
                        const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float dot = acc[qt][rt][4 * g + j];
                                float x;
                                if (METRIC == 0) x = HAS_BIAS ? fmaf(dot, qinv_l[qt], bj[j]) : dot * qinv_l[qt];   // MODE 0 only without bias
                                else if (METRIC == 1) {
                                    if (MODE == 1 && !HAS_BIAS) x = dot * aj[j];
                                    else x = HAS_BIAS ? fmaf(dot * aj[j], qinv_l[qt], bj[j]) : dot * aj[j] * qinv_l[qt];
                                } else {
                                    const float d2 = fmaxf(fmaf(-2.f * qinv_l[qt], dot, aj[j] + qsq_l[qt]), 0.f);
                                    // v_sqrt_f32 / v_rcp_f32 (1 ulp each): the IEEE expansions of sqrtf and the division
                                    // cost ~25 VALU per score, on every row x query, for a result needed to 1e-3
                                    x = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_sqrtf(d2)) + (HAS_BIAS ? bj[j] : 0.f);
                                }
                                acc[qt][rt][4 * g + j] = x;
                            }
                        }
                    }
                }
            }
            if (MODE == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g = 0; g < NGRP; ++g) {
                        const int rl0 = grp_row(rt, g);
                        const int64_t rowg = row0 + rl0;
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            float sj[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) sj[j] = hdb_canon(acc[qt][rt][4 * g + j]);
                            if (q_ok[qt]) {
                                float* dst = a.scores + (int64_t)ql[qt] * a.ld + (t_cur * R + rl0);
                                if (rowg + 3 < n_rows) *reinterpret_cast<float4*>(dst) = make_float4(sj[0], sj[1], sj[2], sj[3]);
                                else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) if (rowg + j < n_rows) dst[j] = sj[j];
                                }
                            }
                        }
                    }
                }
            } else {
                if (!grpB) filter(acc, row0);
                else row0_prev = row0;
            }
        }
        if (more && heavy && grpB) issue(t_cur + 2 * gstep, st_next2);      // B's half of the staging, after its MFMA phase
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
    }
    if (MODE == 1) {
        if (wave_active && grpB && my_tiles > 0) filter(acc, row0_prev);
        flush();
    }
}

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

// Euclidean scores from the MFMA path come from ||v||^2 + ||q||^2 - 2 v.q, which cancels when v ~ q (an
// exact duplicate scores 1/(1+~0.01) instead of 1).  Candidates whose similarity exceeds 0.5 (distance < 1)
// are re-scored from the stored row with the direct difference, like the reference (:49).  One wave per entry.
template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void hdb_rescore_euclid_kernel(unsigned long long* cand, const uint32_t* cnt, uint32_t cap,
                                                                 const _Float16* V, int d, const float* Q, int q0,
                                                                 const float* bias) {
    const int ql = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t n = cnt[ql] < cap ? cnt[ql] : cap;
    const float* qv = Q + (int64_t)(q0 + ql) * d;
    for (uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const unsigned long long ent = cand[(int64_t)ql * cap + e];
        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(ent & 0xFFFFFFFFull);
        float s = hdb_key2f((uint32_t)(ent >> 32));
        const float b = HAS_BIAS ? bias[row] : 0.f;
        if (s - b > 0.5f) {                                        // wave-uniform: one entry per wave
            float acc = 0.f;
            for (int k = lane; k < d; k += 64) { const float df = (float)V[(int64_t)row * d + k] - qv[k]; acc += df * df; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            s = hdb_canon(1.f / (1.f + sqrtf(acc)) + b);
            if (lane == 0) cand[(int64_t)ql * cap + e] = hdb_pack(s, row);
        }
    }
}

// ------------------------------------------------------------------------------------------------
static size_t mfma_lds_bytes(int stage_bytes) {
    return (size_t)3 * stage_bytes + 3 * 2 * 64 * 4 + (size_t)HDB_MFMA_CB * 8 + (size_t)HDB_MFMA_CB * 2 + 64;
}

template <int MF, int QT, int D, int R, int MODE, int METRIC, bool HAS_BIAS>
static int launch_one(const ScanArgs& a, const void* q16, const float* aux0, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_kernel<MF, QT, D, R, MODE, METRIC, HAS_BIAS>;
    const size_t lds = mfma_lds_bytes(R * D * 2);
    static bool attr_done = false;          // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const dim3 grid(blocks, (nq_launch + 8 * MF * QT - 1) / (8 * MF * QT));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, (const _Float16*)q16, aux0, qsq, qscl, a.q0 + nq_launch);
    return (int)hipGetLastError();
}

template <int MF, int QT, int D, int R, int MODE>
static int launch_metric(const ScanArgs& a, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    const bool b = a.bias != nullptr;
    if (a.metric == HDB_DOT) return b ? launch_one<MF, QT, D, R, MODE, 0, true>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st)
                                      : launch_one<MF, QT, D, R, MODE, 0, false>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st);
    if (a.metric == HDB_COSINE) return b ? launch_one<MF, QT, D, R, MODE, 1, true>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st)
                                         : launch_one<MF, QT, D, R, MODE, 1, false>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st);
    if (a.metric == HDB_EUCLIDEAN) return b ? launch_one<MF, QT, D, R, MODE, 2, true>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st)
                                            : launch_one<MF, QT, D, R, MODE, 2, false>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
    return (int)hipErrorNotSupported;
}

template <int MF, int QT, int D, int R>
static int launch_mode(const ScanArgs& a, int mode, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    if (mode == 0) return launch_metric<MF, QT, D, R, 0>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
    return launch_metric<MF, QT, D, R, 1>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
}

// Geometry table: rows per LDS stage (stage = R*d*2 bytes <= 48 KiB, three stages + lists <= 160 KiB).
// d=384 with more than 128 queries uses the 32x32x16 shape (256 queries per pass); everything else the
// 16x16x32 shape (128 queries per pass, B fragments d/8 VGPRs).
extern "C" int hdb_mfma_tile_rows(int d) {
    switch (d) {
        case 128: case 256: case 384: return 64;
        case 512: case 640: case 768: return 32;
        case 1024: case 1536: return 16;
        default: return 0;
    }
}

extern "C" int hdb_mfma_supported(int dtype, int d, int metric) {
    return dtype == HDB_F16 && hdb_mfma_tile_rows(d) > 0 &&
           (metric == HDB_DOT || metric == HDB_COSINE || metric == HDB_EUCLIDEAN);
}

// d=384, more than 128 queries: 16 = 16x16x32 with two query tiles per wave (default: the same FLOPs and LDS
// traffic as the 32x32x16 form, but the chip holds a higher clock on this shape: 1.78-1.85 ms against 2.07-2.25 ms
// for N=10M, Q=256), 32 = 32x32x16 with one query tile per wave (kept for A/B measurements).
static int g_mfma_variant = 16;
extern "C" void hdb_set_mfma_variant(int v) { if (v == 16 || v == 32) g_mfma_variant = v; }

extern "C" int hdb_mfma_queries_per_pass(int d, int nq) { return (d == 384 && nq > 128) ? 256 : 128; }

// a.ntiles / a.tile_stride are in units of hdb_mfma_tile_rows(d) rows here.
extern "C" int hdb_launch_mfma_scan(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                    const float* qsq, const float* qscl, int max_blocks, void* stream) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    if (a.mask) return (int)hipErrorNotSupported;
    int blocks = (int)(a.ntiles < 256 ? a.ntiles : 256);
    if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    switch (a.d) {
        case 128: return launch_mode<16, 1, 128, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 256: return launch_mode<16, 1, 256, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 384:
            if (nq_launch > 128 && g_mfma_variant == 32) return launch_mode<32, 1, 384, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
            if (nq_launch > 128) return launch_mode<16, 2, 384, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
            return launch_mode<16, 1, 384, 64>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 512: return launch_mode<16, 1, 512, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 640: return launch_mode<16, 1, 640, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 768: return launch_mode<16, 1, 768, 32>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 1024: return launch_mode<16, 1, 1024, 16>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        case 1536: return launch_mode<16, 1, 1536, 16>(a, mode, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}

extern "C" int hdb_launch_q_to_f16(const float* Q, int nq, int d, void* q16, float* qscl, void* stream) {
    hipLaunchKernelGGL(hdb_q_to_f16_kernel, dim3(nq), dim3(64), 0, (hipStream_t)stream, Q, nq, d, (_Float16*)q16, qscl);
    return (int)hipGetLastError();
}

extern "C" int hdb_launch_rescore_euclid(unsigned long long* cand, const uint32_t* cnt, uint32_t cap, int nq_launch, const void* V,
                                         int d, const float* Q, int q0, const float* bias, void* stream) {
    const dim3 grid(64, nq_launch);
    if (bias) hipLaunchKernelGGL(hdb_rescore_euclid_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, cand, cnt, cap, (const _Float16*)V, d, Q, q0, bias);
    else hipLaunchKernelGGL(hdb_rescore_euclid_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, cand, cnt, cap, (const _Float16*)V, d, Q, q0, bias);
    return (int)hipGetLastError();
}
