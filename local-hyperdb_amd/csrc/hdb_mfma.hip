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

    // ---- staging geometry (per lane, fixed for the whole kernel) -----------------------------------
    int g_row[NG], g_col[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int slot = (w + 8 * j) * 64 + lane;
        const int r = slot / CPR, cpos = slot - r * CPR;
        g_row[j] = r;
        g_col[j] = (cpos ^ (r & 15)) * 16;          // source chunk for this LDS slot (XOR swizzle)
    }
    const char* Vb = reinterpret_cast<const char*>(a.V);
    const int64_t my_tiles = (a.ntiles > blockIdx.x) ? (a.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

    auto issue = [&](int64_t i, int st) {
        const int64_t t = blockIdx.x + i * gridDim.x;
        const int64_t row0 = t * a.tile_stride * R;
        const int64_t last = a.n - 1 - row0;          // >= 0
        char* sdst = smem + st * STAGE;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int64_t rr = g_row[j] <= last ? g_row[j] : last;
            const char* gp = Vb + (row0 + rr) * (int64_t)(D * 2) + g_col[j];
            __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(gp), HDB_LDS_PTR(sdst + (w + 8 * j) * 1024), 16, 0, 0);
        }
        if (AUX0 || HAS_BIAS) {
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

    if (my_tiles > 0) issue(0, 0);
    if (my_tiles > 1) issue(1, 1);

    const unsigned int ctl_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(ctl);
    const unsigned int cb_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb);
    const unsigned int cbq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq);
    // per-lane LDS read base: row r31 of a 32-row tile, chunk selected per k-step
    const int rd_base = r31 * CPR * 16;
    const int rx = r31 & 15;

    int st_cur = 0;
    for (int64_t i = 0; i < my_tiles; ++i) {
        if (i + 1 < my_tiles) hdb_wait_vmcnt<NLOAD>(); else hdb_wait_vmcnt<0>();
        if (MODE == 1 && tid == 0) ctl[1 + (i & 1)] = (ctl[0] >= HDB_MFMA_CB / 2) ? 1u : 0u;
        hdb_lds_barrier();                                   // tile i is in LDS; everyone is done with tile i-1
        if (i + 2 < my_tiles) issue(i + 2, st_cur == 0 ? 2 : st_cur - 1);   // into the buffer tile i-1 used
        if (MODE == 1 && ctl[1 + (i & 1)]) flush();

        if (wave_active) {
            const int64_t t = blockIdx.x + i * gridDim.x;
            const int64_t row0 = t * a.tile_stride * R;
            const char* sb = smem + st_cur * STAGE;
            f32x16 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
            // A fragments are fetched one k-step ahead (two register sets) so that the LDS latency of
            // step s+1 hides under the MFMAs of step s.
            half8 abuf[2][RT];
            auto fetch = [&](int s, half8 (&dst)[RT]) {
                const int cx = ((2 * s + h) ^ rx) << 4;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    dst[rt] = *reinterpret_cast<const half8*>(sb + rt * 32 * CPR * 16 + rd_base + cx);
            };
            fetch(0, abuf[0]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) fetch(s + 1, abuf[(s + 1) & 1]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abuf[s & 1][rt], Bq[s], acc[rt], 0, 0, 0);
                // pin the interleave: the reads of step s+1 go out BEFORE the MFMAs of step s
                __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);   // DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);   // MFMAs
            }
            // ---- epilogue: lane holds query q and rows rt*32 + 8g + 4h + j ----------------------------
            const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
            const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
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
                    float sj[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dot = acc[rt][4 * g + j];
                        float s;
                        if (METRIC == 0) s = dot;
                        else if (METRIC == 1) s = dot * aj[j] * qinv_l;
                        else { const float d2 = fmaxf(aj[j] + qsq_l - 2.f * dot, 0.f); s = 1.f / (1.f + sqrtf(d2)); }
                        if (HAS_BIAS) s += bj[j];
                        sj[j] = hdb_canon(s);
                    }
                    const int64_t rowg = row0 + rl0;
                    if (MODE == 0) {
                        if (q_ok) {
                            float* dst = a.scores + (int64_t)ql * a.ld + (t * R + rl0);
                            if (rowg + 3 < a.n) *reinterpret_cast<float4*>(dst) = make_float4(sj[0], sj[1], sj[2], sj[3]);
                            else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) if (rowg + j < a.n) dst[j] = sj[j];
                            }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (q_ok && rowg + j < a.n && sj[j] >= thr_l) {
                                // LDS ops in inline asm: hipcc would otherwise drain every in-flight LDS-DMA
                                // (s_waitcnt vmcnt(0)) before touching LDS it cannot prove disjoint from the ring.
                                unsigned int pos;
                                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                             : "=&v"(pos) : "v"(ctl_addr), "v"(1u) : "memory");
                                if (pos < HDB_MFMA_CB) {
                                    const unsigned long long ent = hdb_pack(sj[j], (uint32_t)(rowg + j));
                                    asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                                 :: "v"(cb_addr + pos * 8u), "v"(ent), "v"(cbq_addr + pos * 2u), "v"((unsigned int)ql) : "memory");
                                } else {
                                    atomicAdd(&a.cnt[ql], a.cap + 1u);     // LDS list overflowed: force the exact-path fallback
                                }
                            }
                        }
                    }
                }
            }
        }
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
    }
    if (MODE == 1) flush();
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
