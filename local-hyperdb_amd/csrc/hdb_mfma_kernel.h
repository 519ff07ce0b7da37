// hdb_mfma_kernel.h -- the MFMA row-scan kernel template and its launch helpers, shared by the fp16 (hdb_mfma.hip)
// and fp32 (hdb_mfma_f32.hip) translation units.  See hdb_mfma.hip for the description of the kernel.
#pragma once
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HDB_MFMA_CB 1024            // LDS candidate list entries per workgroup
#define HDB_MFMA_SEG (HDB_MFMA_CB / 8)   // ... per wave
// Measurement builds only (tools/knockout_q256.py; the product is built with 0): 1 = survivors are never appended,
// 2 = no LDS-DMA once the ring is primed (stale but random tiles), 4 = no per-tile barrier, 8 = half of the fragment reads (row tiles 1 and 3 reuse 0 and 2).  Results are wrong by design.
#ifndef HDB_MFMA_KNOCKOUT
#define HDB_MFMA_KNOCKOUT 0
#endif

// Diagnostic build only (tools/clock_q256.py; the product is built with 0): stamp s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) around the tile loop; lane 0 of wave 0 of every workgroup stores the four stamps
// into a buffer of their own that no kernel reads (MI355X_MICROARCH.md, DVFS give-back item 6).  The in-kernel clock
// is d(memtime) / d(memrealtime) x 100 MHz.
#ifndef HDB_MFMA_CLOCK
#define HDB_MFMA_CLOCK 0
#endif
#if HDB_MFMA_CLOCK
#define HDB_CLOCK_WGS 1024
static __device__ unsigned long long hdb_clock_buf[4 * HDB_CLOCK_WGS];
#endif

// Staging roles (measured, profiles/r2_q256_experiments.json, N=10M d=384 Q=256, interleaved rounds per variant on one
// MI355X each): with every wave staging its eighth of a tile (round 1) the two waves of a SIMD both stalled on their own
// LDS-DMA issue (~100-185 blocked cycles per 1-KiB piece): 1.906 ms.  Variants: pieces spread between the MFMAs 1.935,
// fragment reads before the deferred epilogue 1.938, s_setprio 1 for waves 4-7 1.890 on one box and worse on another,
// for waves 0-3 1.906, waves 0-3 staging everything 1.803, only waves 6-7 staging 1.912, and the one that is shipped:
// WAVES 4-7 STAGE EVERY TILE (PPL pieces each) right after the barrier, waves 0-3 never issue LDS-DMA: 1.768 ms.
// The two waves of a SIMD then take turns on the matrix pipe -- A multiplies while B stages and filters, then B
// multiplies while A filters.  In the HBM-bound passes (up to 64 queries waves 4-7 do not multiply at all) the same
// split gives d=384 Q=8 1.18 -> 1.12 ms, Q=64 1.30 -> 1.21, d=768 Q=64 euclidean + bias 2.54 -> 2.41, d=128 Q=48
// 0.586 -> 0.489, N=1.25M Q=16 251 -> 214 us.
// Later A/B on the shipped roles (256 queries, one box, 1.805-1.81 ms that day): waves 0-3 taking 3, 4, 6 or 8 twelfths of
// the pieces and issuing them at the END of their round (after multiply and filter, while waves 4-7 multiply): 1.812,
// 1.806, 1.807, 1.810; all twelve twelfths: 1.795 (-0.8 %).  The schedule of the staging no longer moves the pass: with
// the staging knocked out altogether it takes 1.53 ms, with staging from L2-resident tiles 1.70 -- what is left is the
// energy of the stream itself (HBM read + LDS write) under the power limit, see DESIGN.md section 6.

#define HDB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HDB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void hdb_wait_vmcnt() {      // s_waitcnt takes an immediate: one asm per value
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (N == 23) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    else if constexpr (N == 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
    else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if constexpr (N == 29) asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
    else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    else if constexpr (N == 31) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
    else static_assert(N < 0, "vmcnt immediate out of range");
}
__device__ __forceinline__ void hdb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// MF: rows / queries per MFMA tile; E: element type of V and of the query fragments.  CPS = 16-byte chunks per k-step
// (one ds_read_b128 per lane and k-step: lane group h = lane / MF holds chunk CPS*s + h of its row).
template <int MF, typename E> struct MfmaShape;
template <> struct MfmaShape<32, _Float16> {
    using Acc = f32x16; using Vec = half8;
    static constexpr int CPS = 2, NGRP = 4;          // groups of 4 consecutive rows per lane per tile
    __device__ static __forceinline__ Acc mma(Vec a, Vec b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaShape<16, _Float16> {
    using Acc = f32x4; using Vec = half8;
    static constexpr int CPS = 4, NGRP = 1;
    __device__ static __forceinline__ Acc mma(Vec a, Vec b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
// fp32 data: v_mfma_f32_16x16x4_f32 takes ONE float of A and of B per lane (k slot = lane / 16).  The 16-byte
// fragment a lane reads holds 4 consecutive floats of its chunk, used as the k slots of four chained MFMAs: the
// assignment of actual k indices to (MFMA, slot) pairs is a permutation that A and B share, which is all a sum needs.
template <> struct MfmaShape<16, float> {
    using Acc = f32x4; using Vec = f32x4;
    static constexpr int CPS = 4, NGRP = 1;
    __device__ static __forceinline__ Acc mma(Vec a, Vec b, Acc c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    }
};

// METRIC: 0 dot, 1 cosine (aux0 = 1/||v||), 2 euclidean similarity (aux0 = ||v||^2)
// Fragment maps (lane l):  MF=32: row/query l&31, k = 16s + 8(l>>5) + j, C reg e -> row (e&3) + 8(e>>2) + 4(l>>5)
//                          MF=16: row/query l&15, k = 32s + 8(l>>4) + j, C reg e -> row 4(l>>4) + e
template <typename E, int MF, int QT, int D, int R, int RS, int MODE, int METRIC, bool HAS_BIAS>
__global__ __launch_bounds__(512) void hdb_mfma_kernel(ScanArgs a, const E* __restrict__ q16,
                                                       const float* __restrict__ aux0g, const float* __restrict__ qsq, const float* __restrict__ qscl,
                                                       int nq_end) {
    using Shape = MfmaShape<MF, E>;
    using Vec = typename Shape::Vec;
    constexpr int ES = (int)sizeof(E);          // bytes per element
    constexpr int ROWB = D * ES;                // bytes per row
    using Acc = typename Shape::Acc;
    constexpr int NGRP = Shape::NGRP;
    constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
    constexpr int CPS = Shape::CPS;             // chunks per k-step (2 or 4)
    constexpr int KS = CPR / CPS;               // k-steps (one fragment read each)
    // RS > 1 ("row split"): RS consecutive waves share one query group and take every RS-th row tile of the stage each,
    // so that few queries still spread their MFMAs over all four SIMDs (fp32 MFMAs bind long before HBM does)
    constexpr int RT = R / MF / RS;             // MFMA row tiles per stage and wave
    constexpr int STAGE = R * ROWB;             // bytes of V per stage
    constexpr int PPL = R * CPR / 64 / 4;       // LDS-DMA pieces (1 KiB) per staging wave and tile: waves 4-7 stage
    constexpr bool AUX0 = METRIC != 0;
    constexpr int NAUX = (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);            // per-row aux values, staged by every B wave
    constexpr int QPW = MF * QT;                // queries per wave (QT query tiles share every A fragment)
    static_assert(R % (MF * RS) == 0 && 8 % RS == 0 && R <= 64 && (R * CPR) % 256 == 0 && ROWB % 256 == 0 && PPL + NAUX <= 31, "tile geometry");

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
    const int part = RS > 1 ? w % RS : 0;       // which row tiles of a stage this wave multiplies: part, part + RS, ...
    const int qw0 = a.q0 + blockIdx.y * ((8 / RS) * QPW) + (w / RS) * QPW;
    const bool wave_active = qw0 < nq_end;
    bool q_ok[QT];
    int ql[QT];
    Vec Bq[QT][KS];
    float thr_l[QT], qinv_l[QT], qsq_l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = qw0 + qt * MF + rl;
        q_ok[qt] = q < nq_end;
        ql[qt] = q - a.q0;
        const int qq = q_ok[qt] ? q : (nq_end - 1);
        const uint4* src = reinterpret_cast<const uint4*>(q16 + (int64_t)qq * D) + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 v = src[CPS * s];
            if (!q_ok[qt]) v = make_uint4(0, 0, 0, 0);
            Bq[qt][s] = *reinterpret_cast<Vec*>(&v);
        }
        thr_l[qt] = 0.f; qinv_l[qt] = 1.f; qsq_l[qt] = 0.f;
        if (q_ok[qt]) {
            if (MODE == 1) thr_l[qt] = a.thr[ql[qt]];
            // qscl = 2^-e: the fp16 queries were scaled by 2^e into [2^14, 2^15) (hdb_q16_scaled), undone here for free:
            // it rides on the per-query multiplier of dot / cosine and on the -2 of the euclidean expansion
            const float qs = qscl ? qscl[q] : 1.f;             // fp32 data: the queries are used as they are
            qinv_l[qt] = METRIC == 1 ? a.qinv[q] * qs : qs;
            if (METRIC == 2) qsq_l[qt] = qsq[q];
        }
    }
    if (tid < 4) ctl[tid] = 0;

    // ---- roles ----------------------------------------------------------------------------------------
    // Waves 0-3 ("A") and 4-7 ("B") are the two waves of each SIMD.  B runs its threshold epilogue one tile
    // late so that the two waves of a SIMD do not reach MFMA phase, epilogue and barrier in lock-step.
    const bool grpB = w >= 4;
    // B also stages every tile (see the note on staging roles at the top of this file); with up to 4*MF*QT queries in a
    // pass the B waves have no queries and do nothing else.

    // loop-invariant scalars, read once (keeps kernel-argument loads out of the tile loop)
    const char* const Vb = reinterpret_cast<const char*>(a.V);
    const int64_t n_rows = a.n;
    const int64_t ntiles = a.ntiles;
    const int64_t tstride = a.tile_stride;                      // 1 = dense pass, > 1 = strided row sample
    const int64_t gstep = gridDim.x;

    // Stage tile number t (global tile index) into ring slot st: waves 4-7 issue PPL LDS-DMA pieces of 1 KiB each
    // (piece p = (w & 3) + 4 j; its source offset carries the XOR swizzle of the chunk), non-temporal (V is read once per
    // pass by exactly one CU), plus the per-row aux values.  Only the last tile of the matrix can be ragged.
    auto issue_aux = [&](int64_t row0, int st) {
        if ((AUX0 || HAS_BIAS) && grpB) {
            const int64_t last = n_rows - 1 - row0;
            const int64_t rr = lane <= last ? lane : last;
            if (AUX0) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(aux0g + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 0) * 64), 4, 0, 0);
            if (HAS_BIAS) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 1) * 64), 4, 0, 0);
        }
    };
    auto issue_rows = [&](int64_t row0, int st) {
        const int64_t last = n_rows - 1 - row0;
        char* sdst = smem + st * STAGE;
        const char* tile_base = Vb + row0 * (int64_t)ROWB;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int pc = (w & 3) + 4 * j;
            const int slot = pc * 64 + lane;
            const int r = slot / CPR, cpos = slot - r * CPR;
            const int rr = r <= (int)last ? r : (int)last;
            const unsigned int off = (unsigned int)(rr * ROWB + (cpos ^ (r & 15)) * 16);
            __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + pc * 1024), 16, 0, 2);
        }
    };

    // Survivors of the filter go to a WAVE-PRIVATE segment of the LDS candidate list (HDB_MFMA_SEG entries per wave): the
    // slot of a survivor is the wave's running count plus its rank among the lanes that hit in the same step (one ballot,
    // no LDS atomic and no wait on its return in the filter's slow path), and a wave empties its own segment into the
    // global per-query lists whenever it is three quarters full -- no workgroup barrier, no flush decision to agree on.
    // (Survivors cost the batched passes 5 % with the shared list: 1.77 -> 1.68 ms at 256 queries with the append knocked out.)
    int wcnt = 0;                                    // wave-uniform: entries in this wave's segment
    const unsigned int seg_cb = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb) + (unsigned int)w * HDB_MFMA_SEG * 8u;
    const unsigned int seg_cbq = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq) + (unsigned int)w * HDB_MFMA_SEG * 2u;
    auto flush = [&]() {                             // this wave's segment -> a.cand, then empty
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int e = lane; e < wcnt; e += 64) {
            unsigned long long ent; unsigned int qe;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_u16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ent), "=&v"(qe) : "v"(seg_cb + (unsigned int)e * 8u), "v"(seg_cbq + (unsigned int)e * 2u) : "memory");
            const unsigned int pos = atomicAdd(&a.cnt[qe], 1u);
            if (pos < a.cap) a.cand[(int64_t)qe * a.cap + pos] = ent;
        }
        wcnt = 0;
    };

    // ---- tile sequence ---------------------------------------------------------------------------------
    // Static: workgroup b takes tiles b, b+G, ...  Dynamic (filter pass of up to 4*MF*QT queries, one query block, a zeroed
    // counter in a.tile_ctr, >= 32 tiles per workgroup): the first chunk of CH tiles is fixed, later chunks come from the
    // counter (CH tiles per request, CH/2 near the end) -- CUs differ in streaming speed (hdb_mfma_fused.h) and the slowest
    // workgroup of a static split finishes 3-5 % of the pass after the median one.  Wave 3 owns the counter: it is the last
    // of the multiplying waves to get queries, so up to 3*MF*QT queries it has nothing else to do, and beyond that it
    // multiplies for a quarter of a round; waves 4-7, which stage, never wait for the counter.  The request goes out LOOK
    // rounds (~3 us of streaming, the counter answers in ~1.5 us) before the chunk is needed; the answer is handed over
    // through LDS.
    const bool heavy = (nq_end - (a.q0 + (int)blockIdx.y * ((8 / RS) * QPW))) > (4 / RS) * QPW;    // all eight waves multiply
    const int64_t G = gstep, bidx = blockIdx.x;
    // Measured (10 M rows, 8-64 queries, static -> dynamic): d=768 2.29 -> 2.18 ms, d=1536 (2.5 M rows) 1.149 -> 1.109,
    // d=512 (5 M) 0.775 -> 0.764, d=384 1.120 -> 1.106; but d=128 401 -> 438 us, d=256 (5 M) 394 -> 404, d=384 at 2.5 M rows
    // 298 -> 303: short rows (rounds shorter than the hand-over) and short passes (the end of a pass is decided in chunks)
    // lose, so dynamic hand-out needs rows of >= 768 bytes and >= 16 MiB of V per workgroup.
    const bool dyn = MODE == 1 && a.tile_ctr != nullptr && tstride == 1 && gridDim.y == 1 && !heavy && ROWB >= 768 &&
                     ntiles * STAGE >= G * (16ll << 20) && !(HDB_MFMA_KNOCKOUT & 2);
    constexpr int LOOK = STAGE >= 48 * 1024 ? 2 : STAGE >= 32 * 1024 ? 3 : STAGE >= 24 * 1024 ? 4 : STAGE >= 16 * 1024 ? 6 : 8;
    const int64_t CH = 2 * LOOK, dyn0 = G * CH;
    unsigned int* dq = ctl + 4;                      // [2] {first tile - dyn0, length} handed over by wave 3
    const unsigned int dq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(dq);
    int64_t gp = 0, cidx = 0, coff = 0, cbase = 0, clen = CH, seen = 0;
    auto gen = [&](int64_t& t, int64_t& row0, bool& valid) {     // -> tile and row0 of sequence position gp (and whether it exists)
        if (!dyn) {
            t = bidx + gp * G;
            valid = bidx + gp * G < ntiles;
        } else {
            if (coff == 0) {
                if (cidx == 0) { cbase = bidx * CH; clen = CH; }
                else {
                    unsigned long long pr;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pr) : "v"(dq_addr + (unsigned int)(cidx & 1) * 8u) : "memory");
                    cbase = dyn0 + (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)pr);
                    clen = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(pr >> 32));
                }
            }
            if (w == 3 && coff == clen - LOOK) {                     // request the next chunk LOOK rounds before it is needed
                const unsigned int want = ntiles - dyn0 - seen > (CH + LOOK) * G ? (unsigned int)CH : (unsigned int)LOOK;
                unsigned int got = 0u;
                if (lane == 0) got = __hip_atomic_fetch_add(a.tile_ctr, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = (unsigned int)__builtin_amdgcn_readfirstlane((int)got);
                seen = (int64_t)got + want;
                const unsigned long long pr = ((unsigned long long)want << 32) | got;
                if (lane == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(dq_addr + (unsigned int)((cidx + 1) & 1) * 8u), "v"(pr) : "memory");
            }
            t = cbase + coff;
            valid = t < ntiles;
            if (++coff == clen) { coff = 0; ++cidx; }
        }
        row0 = hdb_tile_index(t, tstride) * R;
        ++gp;
    };
    auto issue = [&](int64_t row0, int st) {
        if ((HDB_MFMA_KNOCKOUT & 2) && gp > 3) return;           // knock-out: the ring keeps its first three tiles
        if (grpB) { issue_rows(row0, st); issue_aux(row0, st); }
    };
    int64_t tA, tB, tC = 0, rA, rB, rC = 0; bool vA, vB, vC = false;       // tile / first row / existence of sequence positions i, i+1, i+2
    gen(tA, rA, vA);
    gen(tB, rB, vB);
    const bool had_tiles = vA;
    if (vA) issue(rA, 0);
    if (vB) issue(rB, 1);

    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    // per-lane LDS read address: row rl of a row tile; chunk (CPS*s + h) ^ rx of k-step s is at byte
    // ((16*CPS*s) ^ hx) of the row image, hx = (h ^ rx) << 4  (CPS*s and h occupy disjoint bits)
    const unsigned int rd_base = (unsigned int)((rl + part * MF) * CPR * 16);
    const unsigned int hx = (unsigned int)((h ^ (rl & 15)) << 4);
    // first of the 4 consecutive tile rows this lane's accumulator group g holds
    auto grp_row = [&](int rt, int g) { const int rg = part + RS * rt; return MF == 32 ? rg * 32 + 8 * g + 4 * h : rg * 16 + 4 * h; };

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
        // every branch below is wave-uniform (ballots): the running count stays a scalar
        if (__ballot((HDB_MFMA_KNOCKOUT & 1) ? (m == 1.2345e30f) : (m >= thr_cmp)) != 0ull) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    if (__ballot(gm[rt][g] >= thr_cmp) != 0ull) {
                        const int64_t rowg = row0 + grp_row(rt, g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = tv[rt][4 * g + j];
                            const bool hit = x >= thr_cmp && rowg + j < n_rows && !(HAS_BIAS && x == -INFINITY);   // bias -inf = masked row
                            const unsigned long long act = __ballot(hit);
                            if (act != 0ull) {
                                const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u));
                                if (hit) {
                                    const float sc = hdb_canon((METRIC != 2 && !HAS_BIAS) ? x * qinv_l : x);
                                    const unsigned long long ent = hdb_pack(sc, (uint32_t)(rowg + j));
                                    if (pos < HDB_MFMA_SEG) {
                                        // LDS ops in inline asm: hipcc would otherwise drain every in-flight LDS-DMA
                                        // (s_waitcnt vmcnt(0)) before touching LDS it cannot prove disjoint from the ring.
                                        asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                                     :: "v"(seg_cb + (unsigned int)pos * 8u), "v"(ent), "v"(seg_cbq + (unsigned int)pos * 2u), "v"((unsigned int)ql) : "memory");
                                    } else {         // cannot happen while flushes keep 64 slots free; kept as a safety net
                                        const unsigned int gpos = atomicAdd(&a.cnt[ql], 1u);
                                        if (gpos < a.cap) a.cand[(int64_t)ql * a.cap + gpos] = ent;
                                    }
                                }
                                const int added = (int)__popcll(act);
                                wcnt = wcnt + added < HDB_MFMA_SEG ? wcnt + added : HDB_MFMA_SEG;
                                if (wcnt > HDB_MFMA_SEG - 64) flush();
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

#if HDB_MFMA_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0) alone: keeps the loop's counted LDS waits as written
#endif
    Acc acc[QT][RT];
    int64_t row0_prev = 0;
    int st_cur = 0;
    for (int64_t i = 0; vA; ++i) {
        if (!vB) hdb_wait_vmcnt<0>();
        else if (grpB) hdb_wait_vmcnt<PPL + NAUX>();         // all but the newest tile's pieces are in
        if (!(HDB_MFMA_KNOCKOUT & 4)) hdb_lds_barrier();     // tile i is in LDS; everyone is done with tile i-1; dq hand-over
        // Stage the whole next-but-one tile right after the barrier, into the buffer tile i-1 used.
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;
        if (vB) gen(tC, rC, vC); else vC = false;
        if (vC) issue(rC, st_next2);

        if (wave_active) {
            const int64_t row0 = rA;
            if (MODE == 1 && grpB && i > 0) filter(acc, row0_prev);        // deferred epilogue of tile i-1
            // A fragments: LDS reads issued two k-steps ahead of the MFMAs that consume them.  The reads
            // and their counted waits are inline asm so that hipcc cannot sink a read next to its use
            // (it otherwise emits read, lgkmcnt(0), MFMA per step and exposes the LDS latency every step).
            // lgkmcnt(n*RT) = "all but the n*RT newest LDS ops are back" = the oldest pending step's fragments;
            // stray scalar loads can only make that wait longer, never shorter.
            const unsigned int sb_addr = smem_addr + (unsigned int)(st_cur * STAGE) + rd_base;
            constexpr int PF = QT == 2 ? 2 : 3;                // k-steps of LDS prefetch (PF+1 fragment sets; two query tiles: registers)
            Vec abuf[PF + 1][RT];
            auto fetch = [&](int s, Vec (&dst)[RT]) {
                const unsigned int ad = sb_addr + ((unsigned int)(16 * CPS * s) ^ hx);
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                if constexpr (RT > 1) { if ((HDB_MFMA_KNOCKOUT & 8) && RT == 4) dst[1] = dst[0]; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "i"(RS * MF * CPR * 16)); }
                if constexpr (RT > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(ad), "i"(2 * RS * MF * CPR * 16));
                if constexpr (RT > 3) { if (HDB_MFMA_KNOCKOUT & 8) dst[3] = dst[2]; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[3]) : "v"(ad), "i"(3 * RS * MF * CPR * 16)); }
            };
            // wait until at most `pend` k-steps of fragment reads are outstanding: lgkmcnt(pend*RT)
            auto wait_frag = [&](int pend, Vec (&f)[RT]) {
                static_assert(RT == 1 || RT == 2 || RT == 4, "RT");
#define HDB_WAITF(N)                                                                                               \
                do {                                                                                               \
                    if constexpr (RT == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]));                  \
                    else if constexpr (RT == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1])); \
                    else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); \
                } while (0)
                const int cnt = pend * (((HDB_MFMA_KNOCKOUT & 8) && RT == 4) ? 2 : RT);
                if (cnt >= 12) HDB_WAITF(12); else if (cnt == 8) HDB_WAITF(8); else if (cnt == 6) HDB_WAITF(6);
                else if (cnt == 4) HDB_WAITF(4); else if (cnt == 3) HDB_WAITF(3); else if (cnt == 2) HDB_WAITF(2);
                else if (cnt == 1) HDB_WAITF(1); else HDB_WAITF(0);
#undef HDB_WAITF
            };
#pragma unroll
            for (int s = 0; s < PF && s < KS; ++s) fetch(s, abuf[s % (PF + 1)]);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4 * NGRP; ++e) acc[qt][rt][e] = 0.f;
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
                                float* dst = a.scores + (int64_t)ql[qt] * a.ld + (tA * R + rl0);     // sample passes store compactly
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
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
        tA = tB; rA = rB; vA = vB; tB = tC; rB = rC; vB = vC;
    }
#if HDB_MFMA_CLOCK
    {
        const unsigned long long clk_c1 = __builtin_amdgcn_s_memtime(), clk_r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (tid == 0 && blockIdx.y == 0 && blockIdx.x < HDB_CLOCK_WGS) {
            unsigned long long* o = hdb_clock_buf + 4 * blockIdx.x;
            o[0] = clk_c0; o[1] = clk_c1; o[2] = clk_r0; o[3] = clk_r1;
        }
    }
#endif
    if (MODE == 1) {
        if (wave_active && grpB && had_tiles) filter(acc, row0_prev);
        flush();
    }
}

// ------------------------------------------------------------------------------------------------
static size_t mfma_lds_bytes(int stage_bytes) {
    return (size_t)3 * stage_bytes + 3 * 2 * 64 * 4 + (size_t)HDB_MFMA_CB * 8 + (size_t)HDB_MFMA_CB * 2 + 64;
}

template <typename E, int MF, int QT, int D, int R, int RS, int MODE, int METRIC, bool HAS_BIAS>
static int launch_one(const ScanArgs& a, const void* q16, const float* aux0, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_kernel<E, MF, QT, D, R, RS, MODE, METRIC, HAS_BIAS>;
    const size_t lds = mfma_lds_bytes(R * D * (int)sizeof(E));
    static unsigned long long attr_done = 0;          // per instantiation, one bit per device
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(blocks, (nq_launch + (8 / RS) * MF * QT - 1) / ((8 / RS) * MF * QT));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, (const E*)q16, aux0, qsq, qscl, a.q0 + nq_launch);
    return (int)hipGetLastError();
}

template <typename E, int MF, int QT, int D, int R, int RS, int MODE>
static int launch_metric(const ScanArgs& a, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    const bool b = a.bias != nullptr;
    if (a.metric == HDB_DOT) return b ? launch_one<E, MF, QT, D, R, RS, MODE, 0, true>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st)
                                      : launch_one<E, MF, QT, D, R, RS, MODE, 0, false>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st);
    if (a.metric == HDB_COSINE) return b ? launch_one<E, MF, QT, D, R, RS, MODE, 1, true>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st)
                                         : launch_one<E, MF, QT, D, R, RS, MODE, 1, false>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st);
    if (a.metric == HDB_EUCLIDEAN) return b ? launch_one<E, MF, QT, D, R, RS, MODE, 2, true>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st)
                                            : launch_one<E, MF, QT, D, R, RS, MODE, 2, false>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
    return (int)hipErrorNotSupported;
}

template <typename E, int MF, int QT, int D, int R, int RS = 1>
static int launch_mode(const ScanArgs& a, int mode, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    if (mode == 0) return launch_metric<E, MF, QT, D, R, RS, 0>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
    return launch_metric<E, MF, QT, D, R, RS, 1>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
}

