// hdb_mfma.hip -- batched Q.V^T scan on the gfx950 matrix cores (fp16 data, fp32 accumulate).
//
// Replaces "np.dot(vectors, query.T)" (hyperdb/ranking_algorithm.py:29,:41) for a batch of queries:
// the reference takes one query per call; here up to 256 queries ride on ONE pass over V.
//
// Work decomposition (one workgroup = 8 waves = 512 threads, one workgroup per CU, persistent):
//   * wave w owns queries [32w, 32w+32) of the batch; their fp16 values for ALL k live in its
//     registers as MFMA B fragments (D/16 fragments x 4 VGPRs = 96 VGPRs at d=384), loaded once;
//   * the workgroup streams tiles of R rows of V through a 3-deep LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction, source-side XOR swizzle so that the
//     lane-linear LDS image is bank-conflict-free for the ds_read_b128 A-fragment reads);
//   * every wave multiplies the whole tile by its 32 queries: v_mfma_f32_32x32x16_f16, one
//     ds_read_b128 per MFMA, accumulators never leave registers;
//   * epilogue in registers: scale / bias / threshold compare; survivors go to a small LDS list
//     that is flushed to the per-query candidate lists with global atomics every few hundred tiles.
//     The N x Q score matrix is never written (10 GB at N=10M, Q=256).
// Synchronisation: one raw s_barrier per tile; tile t+2 is in flight while tile t is multiplied
// (counted s_waitcnt vmcnt, never 0 in the steady state).
// Algorithmic bytes per row: d*2 (V read exactly once per 256 queries); FLOPs: 2*Q*d per row.
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HDB_MFMA_CB 1024            // LDS candidate list entries per workgroup

#define HDB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HDB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void hdb_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else static_assert(N < 0, "add this vmcnt immediate");
}
__device__ __forceinline__ unsigned long long hdb_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ __forceinline__ unsigned long long hdb_stamp_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ __forceinline__ void hdb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// METRIC: 0 dot, 1 cosine (aux0 = 1/||v||), 2 euclidean similarity (aux0 = ||v||^2)
template <int D, int R, int MODE, int METRIC, bool HAS_BIAS>
__global__ __launch_bounds__(512) void hdb_mfma_kernel(ScanArgs a, const _Float16* __restrict__ q16,
                                                       const float* __restrict__ aux0g, const float* __restrict__ qsq,
                                                       int nq_end) {
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int KS = D / 16;                  // k-steps of v_mfma_f32_32x32x16_f16
    constexpr int RT = R / 32;                  // 32-row MFMA tiles per stage
    constexpr int STAGE = R * D * 2;            // bytes of V per stage
    constexpr int NG = R * CPR / 64 / 8;        // LDS-DMA instructions per wave per tile
    constexpr bool AUX0 = METRIC != 0;
    constexpr int NLOAD = NG + (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);
    static_assert(R % 32 == 0 && (R * CPR) % 512 == 0 && D % 128 == 0, "tile geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* auxbuf = reinterpret_cast<float*>(smem + 3 * STAGE);                    // [3 stages][2][64]
    unsigned long long* cb = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 3 * 2 * 64 * 4);
    unsigned short* cbq = reinterpret_cast<unsigned short*>(cb + HDB_MFMA_CB);
    unsigned int* ctl = reinterpret_cast<unsigned int*>(cbq + HDB_MFMA_CB);       // [0] count, [1..2] flush flags

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;

    // ---- this wave's queries --------------------------------------------------------------------
    const int qw0 = a.q0 + blockIdx.y * 256 + w * 32;
    const bool wave_active = qw0 < nq_end;
    const int q = qw0 + r31;
    const bool q_ok = q < nq_end;
    const int ql = q - a.q0;
    half8 Bq[KS];
    {
        const int qq = q_ok ? q : (nq_end - 1);
        const uint4* src = reinterpret_cast<const uint4*>(q16 + (int64_t)qq * D + 8 * h);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 v = src[2 * s];
            if (!q_ok) v = make_uint4(0, 0, 0, 0);
            Bq[s] = *reinterpret_cast<half8*>(&v);
        }
    }
    float thr_l = 0.f, qinv_l = 1.f, qsq_l = 0.f;
    if (q_ok) {
        if (MODE == 1) thr_l = a.thr[ql];
        if (METRIC == 1) qinv_l = a.qinv[q];
        if (METRIC == 2) qsq_l = qsq[q];
    }
    if (tid < 4) ctl[tid] = 0;

    // ---- roles ----------------------------------------------------------------------------------------
    // Waves 0-3 ("A") and 4-7 ("B") are the two waves of each SIMD.  An LDS-DMA instruction costs its wave
    // ~90 issue cycles (48 of them per tile), so the staging of tile i+2 is split: B issues its half at the
    // START of interval i and runs its threshold epilogue one tile late, A (raised priority) owns the
    // matrix pipe first and issues its half at the END, while B multiplies.  Without this split both waves
    // of a SIMD reach DMA issue, MFMA, epilogue and barrier together and the matrix pipe idles ~40 %.
    const bool grpB = w >= 4;
    constexpr int NLOADA = NG;
    constexpr int NLOADB = NG + (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);     // B also stages the per-row aux values

    // ---- staging geometry (per lane, fixed for the whole kernel) -----------------------------------
    int g_off[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int slot = (w + 8 * j) * 64 + lane;
        const int r = slot / CPR, cpos = slot - r * CPR;
        g_off[j] = r * (D * 2) + (cpos ^ (r & 15)) * 16;   // source byte for this LDS slot (XOR swizzle of the chunk)
    }
    const char* Vb = reinterpret_cast<const char*>(a.V);
    const int64_t my_tiles = (a.ntiles > blockIdx.x) ? (a.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

    // one LDS-DMA piece (1 KiB) of tile i into stage st
    auto issue_piece = [&](int64_t i, int st, int j) {
        const int64_t t = blockIdx.x + i * gridDim.x;
        const int64_t row0 = t * a.tile_stride * R;
        const int64_t last = a.n - 1 - row0;          // >= 0
        char* sdst = smem + st * STAGE;
        const char* tile_base = Vb + row0 * (int64_t)(D * 2);          // wave-uniform
        unsigned int off = (unsigned int)g_off[j];
        if (last < R - 1) {                                             // ragged last tile: clamp rows to the last one
            const int r = g_off[j] / (D * 2);
            const int rr = r <= (int)last ? r : (int)last;
            off = (unsigned int)(g_off[j] + (rr - r) * (D * 2));
        }
        __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + (w + 8 * j) * 1024), 16, 0, 0);
    };
    auto issue_aux = [&](int64_t i, int st) {
        if ((AUX0 || HAS_BIAS) && grpB) {
            const int64_t t = blockIdx.x + i * gridDim.x;
            const int64_t row0 = t * a.tile_stride * R;
            const int64_t last = a.n - 1 - row0;
            const int64_t rr = lane <= last ? lane : last;
            if (AUX0) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(aux0g + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 0) * 64), 4, 0, 0);
            if (HAS_BIAS) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 1) * 64), 4, 0, 0);
        }
    };
    auto issue = [&](int64_t i, int st) {
#pragma unroll
        for (int j = 0; j < NG; ++j) issue_piece(i, st, j);
        issue_aux(i, st);
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

    if (my_tiles > 0) issue(0, 0);
    if (my_tiles > 1) issue(1, 1);

    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    const unsigned int ctl_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(ctl);
    const unsigned int cb_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb);
    const unsigned int cbq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq);
    // per-lane LDS read address: row r31 of a 32-row tile; chunk (2s+h)^rx of k-step s is at
    // byte (32s ^ hx) of the row image, hx = (h ^ rx) << 4  (2s and h occupy disjoint bits)
    const unsigned int rd_base = (unsigned int)(r31 * CPR * 16);
    const unsigned int hx = (unsigned int)((h ^ (r31 & 15)) << 4);

    // threshold in the domain the epilogue compares in (see below); +inf for padding lanes
    float thr_cmp = INFINITY;
    if (MODE == 1 && q_ok) {
        if (METRIC == 1 && !HAS_BIAS) { const float tc = thr_l / qinv_l; thr_cmp = tc - fabsf(tc) * 1e-6f; }
        else thr_cmp = thr_l;
    }

    // Filter, second half: group maxima (v_max3) let the common no-hit case finish in ~25 VALU
    // instructions; survivors go to the workgroup's LDS list.  `tv` holds comparable values (below).
    auto filter = [&](const f32x16 (&tv)[RT], int64_t row0) {
        float gm[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gm[rt][g] = fmaxf(fmaxf(tv[rt][4 * g], tv[rt][4 * g + 1]), fmaxf(tv[rt][4 * g + 2], tv[rt][4 * g + 3]));
        float m = gm[0][0];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) m = fmaxf(m, gm[rt][g]);
        if (m >= thr_cmp) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (gm[rt][g] >= thr_cmp) {
                        const int64_t rowg = row0 + rt * 32 + 8 * g + 4 * h;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = tv[rt][4 * g + j];
                            if (x >= thr_cmp && rowg + j < a.n) {
                                const float sc = hdb_canon((METRIC == 1 && !HAS_BIAS) ? x * qinv_l : x);
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
                                    atomicAdd(&a.cnt[ql], a.cap + 1u);   // LDS list overflowed: force the exact-path fallback
                                }
                            }
                        }
                    }
                }
            }
        }
    };

    f32x16 acc[RT];
    int64_t row0_prev = 0;
    int st_cur = 0;
    // diagnostic stamps (dbg & 8, timing study only): block 0, lane 0 of waves 0 and 4, first 64 tiles,
    // 8 stamps per tile, written to the (otherwise unused in filter mode) scores pointer
    const bool stamping = MODE == 1 && (a.dbg & 8) && blockIdx.x == 0 && lane == 0 && (w == 0 || w == 4);
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.scores) + (w == 4 ? 64 * 8 : 0);
#define HDB_STAMP(k) do { if (stamping && i < 64) stamps[i * 8 + (k)] = hdb_stamp(); } while (0)
    if (stamping) { stamps[2 * 64 * 8 + (w == 4 ? 2 : 0)] = hdb_stamp(); stamps[2 * 64 * 8 + (w == 4 ? 3 : 1)] = hdb_stamp_real(); }
    for (int64_t i = 0; i < my_tiles; ++i) {
        HDB_STAMP(0);
        if (i + 1 >= my_tiles) hdb_wait_vmcnt<0>();
        else if (grpB) hdb_wait_vmcnt<NLOADB>();
        else hdb_wait_vmcnt<NLOADA>();
        if (MODE == 1 && tid == 0) ctl[1 + (i & 1)] = (ctl[0] >= HDB_MFMA_CB / 2) ? 1u : 0u;
        HDB_STAMP(1);
        hdb_lds_barrier();                                   // tile i is in LDS; everyone is done with tile i-1
        HDB_STAMP(2);
        const bool more = i + 2 < my_tiles && !(a.dbg & 1);
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;      // buffer of tile i+2 == the one tile i-1 used
        if (more) issue_aux(i + 2, st_next2);
        if (more && !wave_active) {
#pragma unroll
            for (int j = 0; j < NG; ++j) issue_piece(i + 2, st_next2, j);
        }
        if (MODE == 1 && ctl[1 + (i & 1)]) flush();
        HDB_STAMP(3);

        if (wave_active) {
            const int64_t t = blockIdx.x + i * gridDim.x;
            const int64_t row0 = t * a.tile_stride * R;
            if (MODE == 1 && grpB && i > 0 && !(a.dbg & 4)) filter(acc, row0_prev);        // deferred epilogue of tile i-1
            HDB_STAMP(4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;

            // A fragments: LDS reads issued one k-step ahead of the MFMAs that consume them.  The reads
            // and their counted waits are inline asm so that hipcc cannot sink a read next to its use
            // (it otherwise emits read, lgkmcnt(0), MFMA per step and exposes the LDS latency 24 times).
            // lgkmcnt(RT) = "all but the RT newest LDS ops are back" = the previous step's fragments;
            // stray scalar loads can only make that wait longer, never shorter.
            const unsigned int sb_addr = smem_addr + (unsigned int)(st_cur * STAGE) + rd_base;
            half8 abuf[3][RT];
            auto fetch = [&](int s, half8 (&dst)[RT]) {
                const unsigned int ad = sb_addr + ((unsigned int)(32 * s) ^ hx);
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                if constexpr (RT > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "i"(32 * CPR * 16));
            };
            auto wait_frag = [&](int pending_steps, half8 (&f)[RT]) {     // fragments of the oldest step are back
                if constexpr (RT > 1) {
                    if (pending_steps == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]));
                    else if (pending_steps == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
                } else {
                    if (pending_steps == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]));
                    else if (pending_steps == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[0]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]));
                }
            };
            if (!grpB && !(a.dbg & 16)) __builtin_amdgcn_s_setprio(2);
            if (!(a.dbg & 2)) {
            fetch(0, abuf[0]);
            if (KS > 1) fetch(1, abuf[1]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 2 < KS) fetch(s + 2, abuf[(s + 2) % 3]);
                wait_frag(s + 2 < KS ? 2 : (s + 1 < KS ? 1 : 0), abuf[s % 3]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abuf[s % 3][rt], Bq[s], acc[rt], 0, 0, 0);
                // LDS-DMA pieces of tile i+2 ride between the MFMAs: one every KS/NG k-steps
                constexpr int EVERY = KS / NG;
                if (s % EVERY == EVERY / 2 && s / EVERY < NG) { if (more) issue_piece(i + 2, st_next2, s / EVERY); }
            }
            } else if (more) {
#pragma unroll
                for (int j = 0; j < NG; ++j) issue_piece(i + 2, st_next2, j);
            }
            if (!grpB) __builtin_amdgcn_s_setprio(0);
            HDB_STAMP(5);

            // ---- epilogue, first half: lane holds query q and rows rt*32 + 8g + 4h + j (register 4g+j).
            // Turn the dot products into the values that are stored (MODE 0) or compared (MODE 1), in place.
            // Filter mode compares the score itself, except cosine without bias: dot/||v|| vs thr/qinv.
            const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
            const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
            if (METRIC != 0 || HAS_BIAS) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int rl0 = rt * 32 + 8 * g + 4 * h;
                        float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (AUX0) av = *reinterpret_cast<const float4*>(ax0 + rl0);
                        if (HAS_BIAS) bv = *reinterpret_cast<const float4*>(ax1 + rl0);
                        const float aj[4] = {av.x, av.y, av.z, av.w};
                        const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float dot = acc[rt][4 * g + j];
                            float x;
                            if (METRIC == 0) x = dot + bj[j];
                            else if (METRIC == 1) {
                                if (MODE == 1 && !HAS_BIAS) x = dot * aj[j];
                                else x = HAS_BIAS ? fmaf(dot * aj[j], qinv_l, bj[j]) : dot * aj[j] * qinv_l;
                            } else {
                                const float d2 = fmaxf(aj[j] + qsq_l - 2.f * dot, 0.f);
                                x = 1.f / (1.f + sqrtf(d2)) + (HAS_BIAS ? bj[j] : 0.f);
                            }
                            acc[rt][4 * g + j] = x;
                        }
                    }
                }
            }
            if (MODE == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int rl0 = rt * 32 + 8 * g + 4 * h;
                        const int64_t rowg = row0 + rl0;
                        float sj[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) sj[j] = hdb_canon(acc[rt][4 * g + j]);
                        if (q_ok) {
                            float* dst = a.scores + (int64_t)ql * a.ld + (t * R + rl0);
                            if (rowg + 3 < a.n) *reinterpret_cast<float4*>(dst) = make_float4(sj[0], sj[1], sj[2], sj[3]);
                            else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) if (rowg + j < a.n) dst[j] = sj[j];
                            }
                        }
                    }
                }
            } else {
                if (!grpB) { if (!(a.dbg & 4)) filter(acc, row0); }
                else row0_prev = row0;
            }
            HDB_STAMP(6);
        }
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
    }
    if (stamping) { stamps[2 * 64 * 8 + (w == 4 ? 6 : 4)] = hdb_stamp(); stamps[2 * 64 * 8 + (w == 4 ? 7 : 5)] = hdb_stamp_real(); }
#undef HDB_STAMP
    if (MODE == 1) {
        if (wave_active && grpB && my_tiles > 0) filter(acc, row0_prev);
        flush();
    }
}

// fp32 -> fp16 queries (round to nearest even, like numpy's astype(float16))
__global__ void hdb_q_to_f16_kernel(const float* Q, int64_t count, _Float16* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (_Float16)Q[i];
}

// ------------------------------------------------------------------------------------------------
static size_t mfma_lds_bytes(int stage_bytes) {
    return (size_t)3 * stage_bytes + 3 * 2 * 64 * 4 + (size_t)HDB_MFMA_CB * 8 + (size_t)HDB_MFMA_CB * 2 + 64;
}

template <int D, int R, int MODE, int METRIC, bool HAS_BIAS>
static int launch_one(const ScanArgs& a, const void* q16, const float* aux0, const float* qsq, int nq_launch, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_kernel<D, R, MODE, METRIC, HAS_BIAS>;
    const size_t lds = mfma_lds_bytes(R * D * 2);
    static bool attr_done = false;          // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const dim3 grid(blocks, (nq_launch + 255) / 256);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, (const _Float16*)q16, aux0, qsq, a.q0 + nq_launch);
    return (int)hipGetLastError();
}

template <int D, int R, int MODE, int METRIC>
static int launch_bias(const ScanArgs& a, const void* q16, const float* aux0, const float* qsq, int nq_launch, int blocks, hipStream_t st) {
    if (a.bias) return launch_one<D, R, MODE, METRIC, true>(a, q16, aux0, qsq, nq_launch, blocks, st);
    return launch_one<D, R, MODE, METRIC, false>(a, q16, aux0, qsq, nq_launch, blocks, st);
}

template <int D, int R, int MODE>
static int launch_metric(const ScanArgs& a, const void* q16, const float* sqnorm, const float* qsq, int nq_launch, int blocks, hipStream_t st) {
    if (a.metric == HDB_DOT) return launch_bias<D, R, MODE, 0>(a, q16, nullptr, qsq, nq_launch, blocks, st);
    if (a.metric == HDB_COSINE) return launch_bias<D, R, MODE, 1>(a, q16, a.inv_norm, qsq, nq_launch, blocks, st);
    return (int)hipErrorNotSupported;
}

extern "C" int hdb_mfma_tile_rows(int d) { return d == 384 ? 64 : 0; }

extern "C" int hdb_mfma_supported(int dtype, int d, int metric) {
    return dtype == HDB_F16 && hdb_mfma_tile_rows(d) > 0 && (metric == HDB_DOT || metric == HDB_COSINE);
}

// a.ntiles / a.tile_stride are in units of hdb_mfma_tile_rows(d) rows here.
extern "C" int hdb_launch_mfma_scan(const ScanArgs* args, int mode, int nq_launch, const void* q16, const float* sqnorm,
                                    const float* qsq, int max_blocks, void* stream) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    if (a.mask) return (int)hipErrorNotSupported;
    int blocks = (int)(a.ntiles < 256 ? a.ntiles : 256);
    if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    if (a.d == 384) {
        if (mode == 0) return launch_metric<384, 64, 0>(a, q16, sqnorm, qsq, nq_launch, blocks, st);
        return launch_metric<384, 64, 1>(a, q16, sqnorm, qsq, nq_launch, blocks, st);
    }
    return (int)hipErrorNotSupported;
}

extern "C" int hdb_launch_q_to_f16(const float* Q, int nq, int d, void* q16, void* stream) {
    const int64_t count = (int64_t)nq * d;
    hipLaunchKernelGGL(hdb_q_to_f16_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Q, count, (_Float16*)q16);
    return (int)hipGetLastError();
}
